"""Runs the native refinement loop on every named synthetic scene (BASELINE.json configs) and prints it/s,
the number of redone (failed speculative) forwards and the pose error after 50 iterations."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
lib = _lib.load(); dev = torch.device("cuda:0")
for make in (S.s_50k_fern, S.s_800k_chess, S.s_1m_640, S.s_3m_cam):
    sc = make(); H, W = sc.H, sc.W
    model = PL.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
    with torch.no_grad():
        pkg = PL.render(vp, model, bg)
    vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
    rng = np.random.default_rng(7)
    d_t = rng.normal(size=3); d_t *= 0.02 / np.linalg.norm(d_t)
    d_r = rng.normal(size=3); d_r *= math.radians(1.0) / np.linalg.norm(d_r)
    init = torch.tensor(S.se3_exp(np.concatenate([d_t, d_r])), dtype=torch.float32, device=dev)
    fr = PL.FusedRefiner(model, H, W, device=dev)
    out = {}
    for spec in (False, True):
        fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False, speculative=spec, count_instances=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=100, stop_on_converged=False, speculative=spec, count_instances=True)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        out[spec] = (100 / el, dict(fr.last_info))
    Rr, Tt, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=50, stop_on_converged=True, count_instances=True)
    te, re = PL.pose_errors(np.eye(3), np.zeros(3), Rr.detach().cpu().numpy(), Tt.detach().cpu().numpy())
    print(f"{sc.name:14s} P={sc.P:8d} {W}x{H}: plain {out[False][0]:7.1f} it/s (R'={out[False][1]['num_rendered']}), speculative {out[True][0]:7.1f} it/s "
          f"(R'={out[True][1]['num_rendered']}, redone {out[True][1]['fallbacks']}); 50-iter refine: {100*te:.2f} cm {re:.3f} deg, iters {info['iters']}, converged {info['converged']}", flush=True)
    del model, fr, vp, pkg
    torch.cuda.empty_cache()
