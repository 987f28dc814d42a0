#!/usr/bin/env python3
"""The reference's pipeline chained once: train a map of a synthetic room (train.py's loop, 7 000 steps by default), write
point_cloud.ply, load it, localise query frames of the WORLD against it under the reference's masks with the early exit
(tests/trained_map.py).  One JSON line: training report, median pose error, iterations used, iterations/s single-frame and in flight,
and one direct-oracle parity check at the pose of the last forward on that map.
usage: python tools/trained_map.py [--steps 7000] [--frames 32] [--in-flight 16] [--no-oracle]"""
import argparse, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--in-flight", type=int, default=16)
    ap.add_argument("--world", type=int, default=300_000)
    ap.add_argument("--p0", type=int, default=60_000)
    ap.add_argument("--p1", type=int, default=250_000)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--ply", default=None, help="where to write the map (default: a temporary directory)")
    a = ap.parse_args()
    from tests import trained_map as TM
    path = a.ply or os.path.join(tempfile.mkdtemp(prefix="gsr_map_"), "point_cloud", "iteration_%d" % a.steps, "point_cloud.ply")
    world, train = TM.train_room_map(path, steps=a.steps, world_P=a.world, P0=a.p0, P1=a.p1, sh_degree=a.sh_degree, log=lambda m: print(m, file=sys.stderr))
    rep, (gmap, fr, frames, inits, bg) = TM.localise_against(path, world, n_frames=a.frames, in_flight=a.in_flight)
    # (Adam moves every pose component by ~lr = 1e-3 per iteration whatever the gradient's size: fifty iterations cover 5 cm / 2.9 deg per
    # axis at best, so a start 5 cm / 3 deg off -- VERDICT r5's figure -- cannot be closed inside the reference's 50 iterations on ANY map;
    # its own starts come from the feature-matching stage.  The closer starts are reported next to it.)
    others = {}
    for st in ((0.02, 1.0), (0.01, 0.5)):
        r2, _ = TM.localise_against(path, world, n_frames=a.frames, in_flight=a.in_flight, start=st)
        others["%g cm / %g deg" % (100 * st[0], st[1])] = {k: r2[k] for k in ("pose_err_cm_median", "pose_err_deg_median", "iterations_used_median", "single_frame_iters_per_s", "in_flight_iters_per_s")}
    rep["other_start_offsets"] = others
    out = {"workload": "train (tests/trained_map.py: S-room world, create_from_pcd-style start, train.py cadence) -> point_cloud.ply -> GaussianMap.from_ply -> "
                       "gsr_grad_mask + FusedRefiner.refine with the early exit, query frames = renders of the WORLD", "train": train, "localise": rep}
    if not a.no_oracle:
        from oracle import oracle as O
        from tests.test_gpu_lean import oracle_check_at_the_last_forward, _run
        O.set_threads(min(64, os.cpu_count() or 1))
        sc = TM.scene_of_map(gmap, world)
        f = 0
        run = _run(fr, frames[f], inits[f], bg, 12, flags=0)
        summary, report = oracle_check_at_the_last_forward(sc, fr, run, frames[f], frames[f].original_image, frames[f].depth)
        out["oracle_parity_at_the_last_forward"] = {"summary": summary, "per_tensor": report, "tiles_split": fr.seg_stats()[1]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
