"""A/B of K7's dead-pixel shortcut (round 6): per-kernel times of the speculative loop under the reference's mask, with the product
library and with a build that walks dead pixels too (GSR_DEFS=-DGSR_NO_DEAD_PIXEL_SKIP, GSR_LIB_PATH).  Run twice, once per library:
  python tools/dbg/k7_skip_ab.py ; GSR_LIB_PATH=build_ab/libgsr_noskip.so python tools/dbg/k7_skip_ab.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
lib = _lib.load(); dev = torch.device("cuda:0"); bg = torch.zeros(3, device=dev)
out = {}
for name, make in (("S-1M-640", S.s_1m_640), ("object", S.s_1m_640_object), ("walls", S.s_1m_640_walls), ("room", S.s_room_640)):
    sc = make(); model = PL.GaussianMap.from_scene(sc, device=dev)
    frames = [PL.make_frame(sc, model, dev, bg, uid=u) for u in (0, 1)]
    inits = [PL.perturbed_start(1000 + u, device=dev) for u in (0, 1)]
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
    call = lambda g, n: fr.refine(frames[g], PL.TRACKING_CONFIG, inits[g][:3, :3].clone(), inits[g][:3, 3].clone(), bg, iters=n, stop_on_converged=False)
    call(1, 5); call(0, 20)
    kms, _ = bench._profile_ms(lib, lambda: call(0, 40), 3)
    live = float((frames[0].grad_mask[0] & (fr.alpha[0] > 0.99)).float().mean())
    out[name] = {"live_pixel_share": round(live, 3), "render_bwd_us": round(1e3 * kms["render_bwd"] / 40, 1), "preprocess_bwd_us": round(1e3 * kms["preprocess_bwd"] / 40, 1)}
    del fr, model, frames; torch.cuda.empty_cache()
print(os.environ.get("GSR_LIB_PATH", "product"), json.dumps(out))
