import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util as U
from tests.train_replay import TrainReplay
from oracle import oracle as O
from gs_localization_amd import scenes as S
O.set_threads(64)
tr = TrainReplay(P0=200_000, P1=1_500_000, densify_from=1, densification_interval=2, densify_until=11)
tr.step(1, keep=True)
L = tr.last
sc = tr.as_scene(L["act"])
cam = U.scene_inputs(sc, L["view"]["w2c"])
f, _ = U.oracle_run(sc, cam, None, pose=False)
r = L["radii"].cpu().numpy()
bad = np.nonzero(r != f.radii)[0]
print("mismatches", len(bad), "of", sc.P)
for i in bad[:10]:
    print(i, r[i], f.radii[i], sc.means3D[i], sc.scales[i], sc.opacities[i], sc.rotations[i], np.linalg.norm(sc.rotations[i]))
# same tensors through the plain test path
o, _ = U.hip_run(sc, cam, None, pose=False)
print("hip_run mismatches", int((o["radii"] != f.radii).sum()))
