"""EXPERIMENT: one speculative iteration (the second of a two-iteration call) with and without split tiles against the CPU oracle at
the same pose: per-pixel image errors, gradient errors.  usage: python tools/dbg/split_vs_oracle.py [room|object] [P] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL, util as U
from tests.test_gpu_lean import _setup, _run, _camera_of_the_pose_state
from oracle import oracle as O
O.set_threads(min(64, os.cpu_count() or 1))
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 300_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sc = S.VARIANTS[kind](P=P)
model, bg, view, init = _setup(sc, seed=9)
for flags, tag in ((0, "split"), (_lib.REFINE_NO_SPLIT, "nosplit")):
    fr = PL.FusedRefiner(model, sc.H, sc.W, device="cuda:0")
    vp = view()
    gt_image, gt_depth = vp.original_image.clone(), vp.depth.clone()
    run = _run(fr, vp, init, bg, K, flags=flags, lean_min_P=1)
    info = run["info"]
    st = fr.seg_stats() if flags == 0 else None
    vm, pm, cp = _camera_of_the_pose_state(info["R_last_forward_host"], info["T_last_forward_host"], S.camera_matrices(sc)[2])
    f = O.forward(sc.means3D, sc.opacities, vm, pm, cp, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree, shs=sc.shs,
                  scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
    out = {}
    for k, b in (("color", f.color), ("depth", f.depth), ("alpha", f.alpha)):
        a = run[k].cpu().numpy()
        d = np.abs(a - b)
        out[k] = ("rel %.2e" % U.rel_l1(a, b), "max %.2e" % d.max(), "n>1e-3 %d" % int((d > 1e-3).sum()), "at", np.unravel_index(d.argmax(), d.shape))
    nt = run["n_touched"].cpu().numpy()
    ex = info["exposure_last_forward_host"]
    class _V: pass
    v = _V()
    v.exposure_a, v.exposure_b = torch.tensor([float(ex[0])], device="cuda:0"), torch.tensor([float(ex[1])], device="cuda:0")
    v.original_image, v.depth, v.grad_mask = gt_image, gt_depth, torch.ones((1, sc.H, sc.W), dtype=torch.bool, device="cuda:0")
    ti, td = run["color"].clone().requires_grad_(True), run["depth"].clone().requires_grad_(True)
    PL.tracking_loss(PL.TRACKING_CONFIG, ti, td, run["alpha"], v).backward()
    go = O.backward(f, ti.grad.cpu().numpy(), td.grad.cpu().numpy(), np.zeros((1, sc.H, sc.W), np.float32), pose_mode=True)
    g = {k: "%.2e" % U.rel_l1(getattr(fr, "g_" + k).cpu().numpy().reshape(go[ok].shape), go[ok]) for k, ok in (("m3d", "means3D"), ("sh", "sh"), ("opac", "opacities"), ("scale", "scales"), ("rot", "rotations"))}
    print(kind, tag, "seg", st, {k: info[k] for k in ("iters", "fallbacks", "lean_iters")}, out, "radii diff", int((run["radii"].cpu().numpy() != f.radii).sum()),
          "n_touched diff", int(np.abs(nt - f.n_touched).sum()), "of", int(f.n_touched.sum()), "tau %.2e" % U.rel_l1(fr.g_tau.cpu().numpy(), go["tau"]), g, flush=True)
