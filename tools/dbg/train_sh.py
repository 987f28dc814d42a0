import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util as U
from tests.train_replay import TrainReplay
from oracle import oracle as O
O.set_threads(64)
tr = TrainReplay(P0=200_000, P1=1_500_000, densify_from=1, densification_interval=2, densify_until=11)
tr.step(1, keep=True)
L = tr.last
sc = tr.as_scene(L["act"])
cam = U.scene_inputs(sc, L["view"]["w2c"])
gc, gd, ga = (g.cpu().numpy() for g in L["pix_grads"])
O.set_accumulate_double(True)
f, go = U.oracle_run(sc, cam, (gc, gd, ga), pose=False)
a_own = f.alpha.copy()
f.alpha = L["alpha"].detach().cpu().numpy().copy()
g_alt = O.backward(f, gc, gd, ga, pose_mode=False)
f.alpha = a_own
for k in ("means3D", "means2D", "opacities", "sh", "scales", "rotations"):
    got = L["grads"][k].cpu().numpy().reshape(go[k].shape)
    print(k, "hip vs oracle %.2e | hip vs oracle(with hip alpha) %.2e | oracle vs oracle(with hip alpha) %.2e" % (U.rel_l1(got, go[k]), U.rel_l1(got, g_alt[k]), U.rel_l1(g_alt[k], go[k])))
rng = np.random.default_rng(0)
up = rng.uniform(size=a_own.shape) < 0.5
f.alpha = np.where(up, np.nextafter(a_own, np.float32(2)), np.nextafter(a_own, np.float32(0))).astype(np.float32)
g_n = O.backward(f, gc, gd, ga, pose_mode=False)
f.alpha = a_own
for k in ("means3D", "means2D", "opacities", "sh", "scales", "rotations"):
    print(k, "oracle vs oracle(alpha +-1ulp) %.2e" % U.rel_l1(g_n[k], go[k]))
print("fraction of pixels with alpha > 0.999:", float((a_own > 0.999).mean()))
