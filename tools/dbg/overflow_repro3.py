import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
seed = 8
rng = np.random.default_rng(100 + seed)
W, H = int(rng.integers(150, 260)), int(rng.integers(100, 200))
sc = S.small(P=60000, W=W, H=H, sh_degree=3, seed=int(rng.integers(1 << 30)), scale_med=0.2)
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
init = torch.tensor(S.se3_exp(rng.normal(size=6) * 0.01), dtype=torch.float32, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
K = int(os.environ.get("K", 2))
outs = {}
for name, kw in (("plain", dict(speculative=False)), ("spec", dict(speculative=True)), ("spec_margin", dict(speculative=True, bound_margin=(0.5, 0.5))),
                 ("spec_nobal", dict(speculative=True, flags=_lib.REFINE_DETERMINISTIC | _lib.REFINE_NO_BALANCE))):
    vp = PL.make_frame(sc, model, dev, bg)
    kw.setdefault("flags", _lib.REFINE_DETERMINISTIC)
    try:
        R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=K, stop_on_converged=False,
                               warm_start=False, lean_min_P=1, count_instances=True, **kw)
    except TypeError as e:
        print(name, "skipped:", e); continue
    torch.cuda.synchronize()
    o = {"R": R.clone(), "T": T.clone(), "color": fr.color.clone(), "alpha": fr.alpha.clone(), "depth": fr.depth.clone(), "radii": fr.radii.clone(),
         "n_touched": fr.n_touched.clone(), "loss": fr.loss_out.clone()}
    for k in ("m2d", "conic", "opac", "col", "m3d", "tau", "img", "depth"):
        o["g_" + k] = getattr(fr, "g_" + k).clone()
    outs[name] = o
    print(name, {k: v for k, v in info.items() if k in ("fallbacks", "host_redos", "lean_iters", "num_rendered")}, "loss", fr.loss_out.tolist(), flush=True)
ref = outs["plain"]
for name, o in outs.items():
    if name == "plain": continue
    d = {k: float((o[k].double() - ref[k].double()).abs().max()) for k in o}
    print(name, "vs plain:", {k: "%.2e" % v for k, v in d.items()})
    dc = (o["color"] - ref["color"]).abs().amax(0)
    ys, xs = torch.nonzero(dc > 0, as_tuple=True)
    if len(ys):
        tiles = sorted(set((int(y) // 16, int(x) // 16) for y, x in zip(ys.tolist(), xs.tolist())))
        print("  pixels differing:", len(ys), "in tiles (ty, tx):", tiles[:40])
    bad = torch.nonzero(o["radii"] != ref["radii"]).flatten()
    print("  radii differing:", len(bad), [(int(i), int(o["radii"][i]), int(ref["radii"][i])) for i in bad[:8]])
