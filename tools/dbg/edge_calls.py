"""Diagnostic: gsr_refine on degenerate inputs -- ten Gaussians, a small map, nothing in front of the camera -- through sequences of calls with and
without the early exit on the same workspaces (the final pass that writes the gradient rows must cope with empty lists and one-iteration calls)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
for P, tag in ((10, "tiny"), (3000, "small"), (3000, "behind")):
    sc = S.small(P=P, W=96, H=64, sh_degree=1, seed=3)
    if tag == "behind":
        sc.means3D[:, 2] = -np.abs(sc.means3D[:, 2])      # everything behind the camera: no survivors at all
    model = PL.GaussianMap.from_scene(sc, device=dev); bg = torch.zeros(3, device=dev)
    vp = PL.make_frame(sc, model, dev, bg); init = PL.perturbed_start(1, device=dev)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
    for it, stop in ((3, False), (1, False), (4, True), (3, False)):
        R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=it, stop_on_converged=stop)
        torch.cuda.synchronize()
        g = fr.g_m3d
        print(tag, "iters", it, "stop", stop, "->", info["iters"], info["converged"], "finite", bool(torch.isfinite(R).all() and torch.isfinite(g).all()), "|g|", float(g.abs().sum()))
print("ok")
