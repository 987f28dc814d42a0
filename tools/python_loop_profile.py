"""Where the reference-style Python loop (tests/replay.py loop_iteration on the drop-in package (B)) spends its time on
S-1M-640: per-phase wall time with a device sync after each phase, next to the unsynchronised loop rate."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
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
cfg = PL.TRACKING_CONFIG

def loop(K, synced):
    vp.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
    opt = PL.pose_adam(vp)
    acc = np.zeros(5)
    sync = torch.cuda.synchronize if synced else (lambda: None)
    torch.cuda.synchronize(); t_all = time.perf_counter()
    for _ in range(K):
        t0 = time.perf_counter()
        rp = PL.render(vp, model, bg); sync(); t1 = time.perf_counter()
        opt.zero_grad()
        loss = PL.tracking_loss(cfg, rp["render"], rp["depth"], rp["opacity"], vp); sync(); t2 = time.perf_counter()
        loss.backward(); sync(); t3 = time.perf_counter()
        with torch.no_grad():
            opt.step(); sync(); t4 = time.perf_counter()
            conv = PL.apply_pose_delta(vp, converged_threshold=1e-4); bool(conv); sync(); t5 = time.perf_counter()      # `if converged: break` reads it back
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
    torch.cuda.synchronize()
    return (time.perf_counter() - t_all) / K, acc / K

print("GSR_SPECULATION =", os.environ.get("GSR_SPECULATION", "1 (default)"))
loop(10, False)
tot, _ = loop(300, False)
tot_s, a = loop(100, True)
print(f"python loop S-1M-640: {tot * 1e3:.3f} ms/iter unsynchronised ({1 / tot:.0f} it/s); with a sync after every phase {tot_s * 1e3:.3f} ms: "
      f"render {a[0] * 1e3:.3f}, loss {a[1] * 1e3:.3f}, backward {a[2] * 1e3:.3f}, Adam {a[3] * 1e3:.3f}, update_pose {a[4] * 1e3:.3f}")
_, h = loop(100, False)
print(f"host-side time per phase without syncs (enqueue cost): render {h[0] * 1e3:.3f} (includes the forward's own read-back), loss {h[1] * 1e3:.3f}, "
      f"backward {h[2] * 1e3:.3f}, Adam {h[3] * 1e3:.3f}, update_pose {h[4] * 1e3:.3f} (includes the converged read-back)")
# A/B in one process (host timings drift by several percent between runs): alternate the two settings
if os.environ.get("AB", "0") != "0":
    import statistics
    r = {"0": [], "1": []}
    for rep in range(8):
        for s in ("0", "1"):
            os.environ["GSR_SPECULATION"] = s
            loop(10, False)
            r[s].append(loop(150, False)[0] * 1e3)
    print("A/B ms per iteration, median of 8 x 150 iterations: plain gsr_forward %.3f, gsr_forward_speculative %.3f" %
          (statistics.median(r["0"]), statistics.median(r["1"])), [round(x, 3) for x in r["0"]], [round(x, 3) for x in r["1"]])
