"""Randomised campaign for the split-tile path (gsr_kernels.h, SegCtl): structured scenes of random kind, size and seed; a short
native loop with heavy tiles split across workgroups; the call's last forward / backward against the CPU oracle at the pose it ran
with (radii exact, images 1e-4, dL/dtau 1e-5, gradients 2e-5 and per row with flip accounting, n_touched exact with causes -- round 6: one set of bars whatever the share of split tiles), and the same loop with GSR_REFINE_NO_SPLIT (poses 2e-5: two runs of a loop whose loss is made of sign functions end that far apart with or without splitting).
usage: CASES=40 SEED=1 python tools/fuzz_split.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL, util as U
from tests.test_gpu_lean import oracle_check_at_the_last_forward
from oracle import oracle as O
O.set_threads(min(64, os.cpu_count() or 1))
dev = "cuda:0"
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
CASES = int(os.environ.get("CASES", "30"))


def make(kind, P, seed):
    if kind == "room":
        return S.s_room_640(P=P, seed=seed)
    if kind == "object":
        return S.s_1m_640_object(P=P, seed=seed, vseed=seed + 3)
    if kind == "walls":
        return S.s_1m_640_walls(P=P, seed=seed, vseed=seed + 3)
    if kind == "plates":          # thousands of splats at EXACTLY the same depths in a few tiles: pivots on equal keys, ranges beyond the in-LDS sort
        sc = S.s_1m_640(P=P, seed=seed)
        m = sc.means3D.copy()
        r = np.random.default_rng(seed + 9)
        sel = r.random(P) < 0.25
        z = r.choice([1.5, 1.5, 2.0], size=int(sel.sum())).astype(np.float32)
        m[sel, 0] = (r.normal(0, 0.05, sel.sum()) * z).astype(np.float32)
        m[sel, 1] = (r.normal(0, 0.05, sel.sum()) * z).astype(np.float32)
        m[sel, 2] = z
        sc.means3D = np.ascontiguousarray(m)
        sc.opacities = np.ascontiguousarray(np.where(sel[:, None], 0.05, sc.opacities).astype(np.float32))
        sc.name = "plates"
        return sc
    raise ValueError(kind)


worst = {}
n_split_cases = 0
n_fail = 0
t_start = time.time()
for case in range(CASES):
    kind = ["room", "object", "walls", "plates", "room", "object"][int(rng.integers(6))]
    P = int(rng.integers(150_000, 700_000))
    seed = int(rng.integers(1, 10_000))
    K = int(rng.integers(2, 7))
    sc = make(kind, P, seed)
    model = PL.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    init = PL.perturbed_start(seed, float(rng.uniform(0.005, 0.04)), float(rng.uniform(0.2, 2.0)), device=dev)
    runs = {}
    for flags, tag in ((0, "split"), (_lib.REFINE_NO_SPLIT, "nosplit")):
        vp = PL.make_frame(sc, model, dev, bg)
        gt_image, gt_depth = vp.original_image.clone(), vp.depth.clone()
        fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
        R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=K, stop_on_converged=False, flags=flags, lean_min_P=1, warm_start=False)
        torch.cuda.synchronize()
        runs[tag] = dict(R=R.clone(), T=T.clone(), info=info, fr=fr, vp=vp, color=fr.color.clone(), depth=fr.depth.clone(), alpha=fr.alpha.clone(), gt=(gt_image, gt_depth))
    a = runs["split"]
    fr, info = a["fr"], a["info"]
    st = fr.seg_stats()
    n_split_cases += int(st[1] > 0)
    dT = float((a["T"] - runs["nosplit"]["T"]).abs().max()); dR = float((a["R"] - runs["nosplit"]["R"]).abs().max())
    # round 6: ONE set of bars whatever the share of split tiles -- tests/test_gpu_lean.py::oracle_check_at_the_last_forward (images 1e-4,
    # radii exact, dL/dtau 1e-5, every gradient tensor 2e-5 in aggregate, per-row bars over the rows without a cause, n_touched exact with
    # causes: tests/util.py::flip_accounted_parity), under the frame's own mask (the reference's, tests/replay.py::make_frame)
    bad, msg = False, ""
    for tag in ("split", "nosplit") if os.environ.get("BOTH") else ("split",):
        r = runs[tag]
        run = dict(info=r["info"], color=r["color"], depth=r["depth"], alpha=r["alpha"], n_touched=r["fr"].n_touched.clone(), radii=r["fr"].radii.clone())
        try:
            summary, report = oracle_check_at_the_last_forward(sc, r["fr"], run, r["vp"], r["gt"][0], r["gt"][1])
            msg += " [%s: %s]" % (tag, summary[:260])
        except AssertionError as ex:
            bad = True
            msg += " [%s: FAILED %s]" % (tag, str(ex)[:600])
    if max(dT, dR) > 2e-5:
        bad = True
    worst["dT"] = max(worst.get("dT", 0), dT)
    n_fail += int(bad)
    print("%s case %d %s P=%d seed=%d K=%d seg=%s fallbacks=%d/%d  dT %.1e %s" % ("FAIL" if bad else "ok  ", case, kind, P, seed, K, st, info["fallbacks"], runs["nosplit"]["info"]["fallbacks"], dT, msg), flush=True)
    del model, runs, fr, a
    torch.cuda.empty_cache()
print("cases", CASES, "with split tiles", n_split_cases, "FAILED", n_fail, "worst", {k: float("%.2e" % v) for k, v in worst.items()}, "in %.0f s" % (time.time() - t_start))
