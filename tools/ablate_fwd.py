"""Diagnostic: time k_render_fwd on S-1M-640 (native loop, bin-by-tile path) with parts switched off
(gsr_debug_ablate bits 12-15: 1 = no compositing loop, 2 = no LDS sort, 4 = no n_touched atomics, 8 = no gathers)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S, pipelines as PL
lib = _lib.load(); dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
proj = PL.getProjectionMatrix2(0.01, 100.0, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, W=W, H=H).transpose(0, 1).to(dev)
vp = PL.Camera(0, None, None, torch.eye(4, device=dev), proj, sc.fx, sc.fy, sc.cx, sc.cy, PL.focal2fov(sc.fx, W), PL.focal2fov(sc.fy, H), H, W, device=dev)
with torch.no_grad():
    pkg = PL.render(vp, model, PL.PipelineParams(), bg)
vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
init = torch.tensor(S.se3_exp([0.01, 0.01, 0.01, 0.01, 0.0, 0.0]), dtype=torch.float32, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
for ab in [0, 0x1000, 0x2000, 0x4000, 0x9000, 0xB000, 0]:
    lib.gsr_debug_ablate(ab)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=3, stop_on_converged=False, speculative=True)
    lib.gsr_profile_enable((1 << nk) - 1)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=40, stop_on_converged=False, speculative=True)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    d = {names[i]: round(ms[i] / max(cnt[i], 1), 4) for i in range(nk)}
    print("ablate 0x%x" % ab, "render_fwd %.4f (x%d)" % (d["render_fwd"], cnt[names.index("render_fwd")]), "render_bwd", d["render_bwd"], flush=True)
lib.gsr_debug_ablate(0)
