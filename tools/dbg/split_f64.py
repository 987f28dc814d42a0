"""EXPERIMENT: is the split-tile backward further from the TRUE gradients than the unsplit one, or only further from the fp32 oracle?
A small scene with a few very heavy tiles (faint splats crowded into a third of the image); the native loop with and without split
tiles, the CPU oracle, and float64 autograd (oracle/autograd_ref.py, frozen decisions) at the pose of each loop's last forward.
usage: python tools/dbg/split_f64.py [P] [W] [H]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL, util as U
from tests.test_gpu_lean import _camera_of_the_pose_state
from oracle import oracle as O, autograd_ref as AG

P = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 96
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")


from tests.test_gpu_split import _heavy_left, _float64_gradients


sc = _heavy_left(P, W, H, opac=float(os.environ.get("OPAC", "0.2")))
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
init = PL.perturbed_start(3, device=dev)
names = (("m3d", "means3D"), ("sh", "sh"), ("opac", "opacities"), ("scale", "scales"), ("rot", "rotations"), ("tau", "tau"))
for flags, tag in ((0, "split"), (_lib.REFINE_NO_SPLIT, "unsplit")):
    vp = PL.make_frame(sc, model, dev, bg)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False, flags=flags, lean_min_P=1, warm_start=False)
    torch.cuda.synchronize()
    st = fr.seg_stats() if flags == 0 else None
    gi, gd = fr.g_img.cpu().numpy(), fr.g_depth.cpu().numpy()
    t0 = time.time()
    f, go, g64 = _float64_gradients(sc, info, gi, gd)
    nc = f.state()["n_contrib"]
    print("pixel gradients: |gi| %.3e |gd| %.3e" % (np.abs(gi).sum(), np.abs(gd).sum()))
    print("%-8s seg %s  fallbacks %d  mean / max n_contrib %.0f / %d  (float64 pass %.0f s)" % (tag, st, info["fallbacks"], nc.mean(), nc.max(), time.time() - t0))
    for k, ok in names:
        a = getattr(fr, "g_" + k).cpu().numpy().reshape(g64[ok].shape)
        b = go[ok].reshape(g64[ok].shape)
        print("   %-6s loop vs float64 %.2e   oracle vs float64 %.2e   loop vs oracle %.2e" % (k, U.rel_l1(a, g64[ok]), U.rel_l1(b, g64[ok]), U.rel_l1(a, b)))
