"""Which pixel separates the split loop from the unsplit loop on S-room (300 k), and what does the oracle say there?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
from tests.test_gpu_lean import _setup, _run, _camera_of_the_pose_state
from oracle import oracle as O
O.set_threads(64)
sc = S.s_room_640(P=300_000)
model, bg, view, init = _setup(sc, seed=7)
for rep in range(6):
    fr = PL.FusedRefiner(model, sc.H, sc.W, device="cuda:0"); fr2 = PL.FusedRefiner(model, sc.H, sc.W, device="cuda:0")
    a = _run(fr, view(), init, bg, 10, flags=0, lean_min_P=1)
    b = _run(fr2, view(), init, bg, 10, flags=_lib.REFINE_NO_SPLIT, lean_min_P=1)
    d = (a["color"] - b["color"]).abs().amax(0)
    print("rep", rep, "max colour diff", float(d.max()), "pose diff", float((a["R"] - b["R"]).abs().max()), float((a["T"] - b["T"]).abs().max()), "n > 5e-3:", int((d > 5e-3).sum()))
    if float(d.max()) > 0.05:
        y, x = np.unravel_index(int(d.argmax()), d.shape)
        print("  pixel", x, y, "tile", (y // 16) * 40 + x // 16, "split", a["color"][:, y, x].tolist(), "plain", b["color"][:, y, x].tolist(),
              "alpha", float(a["alpha"][0, y, x]), float(b["alpha"][0, y, x]), "depth", float(a["depth"][0, y, x]), float(b["depth"][0, y, x]))
        for nm, run in (("split", a), ("plain", b)):
            info = run["info"]
            vm, pm, cp = _camera_of_the_pose_state(info["R_last_forward_host"], info["T_last_forward_host"], S.camera_matrices(sc)[2])
            f = O.forward(sc.means3D, sc.opacities, vm, pm, cp, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
            dd = np.abs(run["color"].cpu().numpy() - f.color).max(0)
            yy, xx = np.unravel_index(int(dd.argmax()), dd.shape)
            print("  ", nm, "oracle at its pose: colour there", f.color[:, y, x].tolist(), "alpha", float(f.alpha[0, y, x]), "n_contrib", int(f.state()["n_contrib"][y, x]),
                  "| worst pixel vs oracle", float(dd.max()), "at", xx, yy, "count > 5e-3:", int((dd > 5e-3).sum()))
        print("  seg stats", fr.seg_stats())
        nb = d[max(0, y - 2):y + 3, max(0, x - 2):x + 3]
        print("  neighbourhood diff\n", np.round(nb.cpu().numpy(), 4))
        break
