"""Property test of the exact tile culling and of the per-wave quadrant lists: a culled (splat, tile) or (splat, quadrant)
pair must be one that no pixel would have blended, so a build WITHOUT any culling (-DGSR_NO_CULL) has to produce
bit-identical images, depth, opacity and n_touched, and gradients equal up to summation order.
    python tools/cull_check.py dump gpurun_out/cull        # with the product library
    (swap in the library built with GSR_DEFS=-DGSR_NO_CULL)
    python tools/cull_check.py compare gpurun_out/cull
Random scenes as in tools/fuzz_parity.py (needles, threshold opacities, near-plane splats, partial tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gs_localization_amd import scenes as S
from tests import util as U
mode, d = sys.argv[1], sys.argv[2]
os.makedirs(d, exist_ok=True)
os.environ["GSR_SPECULATION"] = "0"
N = int(os.environ.get("CASES", 150))
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
bad = 0; worst = 0.0
for case in range(N):
    W = int(rng.integers(17, 200)); H = int(rng.integers(17, 160))
    P = int(rng.choice([50, 400, 2500, 20000]))
    deg = int(rng.integers(0, 4))
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=int(rng.integers(1 << 30)), scale_med=float(rng.choice([0.005, 0.03, 0.15, 0.6])))
    kind = rng.integers(4)
    if kind == 0:
        sc.scales[:, 0] *= 12.0; sc.scales[:, 1] *= 0.05
    elif kind == 1:
        sc.opacities[:] = rng.choice([0.003, 0.0039, 0.004, 0.01, 0.5, 0.999, 1.0], size=sc.opacities.shape).astype(np.float32)
    elif kind == 2:
        sc.means3D[:, 2] = rng.uniform(-0.5, 1.0, sc.P).astype(np.float32)
    w2c = S.se3_exp(rng.normal(size=6) * np.array([0.2, 0.2, 0.2, 0.1, 0.1, 0.1]))
    grads = U.random_grads(sc, seed=case)
    o, g = U.hip_run(sc, U.scene_inputs(sc, w2c), grads, pose=True)
    f = os.path.join(d, f"case{case}.npz")
    if mode == "dump":
        np.savez(f, color=o["color"], depth=o["depth"], alpha=o["alpha"], radii=o["radii"], n_touched=o["n_touched"], **{"g_" + k: v for k, v in g.items() if v is not None})
    else:
        r = np.load(f)
        for k in ("color", "depth", "alpha", "radii", "n_touched"):
            if not np.array_equal(r[k], o[k]):
                bad += 1
                print("DIFFERENT", case, k, W, H, P, "kind", kind, "max abs", float(np.abs(r[k].astype(np.float64) - o[k]).max()), "pixels", int((r[k] != o[k]).sum()))
        for k, v in g.items():
            if v is not None and np.abs(v).sum() > 0:
                worst = max(worst, U.rel_l1(r["g_" + k], v))
if mode == "compare":
    print(f"{N} cases: {bad} output arrays differ between the culling and the no-culling build; worst gradient rel-L1 between them {worst:.2e}")
