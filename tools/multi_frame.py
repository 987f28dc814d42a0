"""Experiment: several query frames refined concurrently on ONE GPU (one host thread + one stream per frame)."""
import sys, os, time, threading, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
def make_view():
    vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
    with torch.no_grad():
        pkg = PL.render(vp, model, bg)
    vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
    return vp
K = 200
for nf, spec in ((4, True), (6, True)) * 3:
    frs = [PL.FusedRefiner(model, H, W, device=dev) for _ in range(nf)]
    vps = [make_view() for _ in range(nf)]
    inits = [torch.tensor(S.se3_exp(np.r_[0.01 * (i + 1), 0.01, -0.01, 0.01, 0.0, 0.005 * i]), dtype=torch.float32, device=dev) for i in range(nf)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nf)]
    def work(i, iters):
        with torch.cuda.stream(streams[i]):
            frs[i].refine(vps[i], PL.TRACKING_CONFIG, inits[i][:3, :3].clone(), inits[i][:3, 3].clone(), bg, iters=iters, stop_on_converged=False, speculative=spec)
    def run(iters):
        ts = [threading.Thread(target=work, args=(i, iters)) for i in range(nf)]
        [t.start() for t in ts]; [t.join() for t in ts]
        torch.cuda.synchronize()
    run(5)
    t0 = time.perf_counter(); run(K); el = time.perf_counter() - t0
    print(f"spec={spec} frames in flight {nf}: {nf * K / el:8.1f} it/s total, {1e3 * el / K:.3f} ms per iteration-round", flush=True)
    del frs, vps
