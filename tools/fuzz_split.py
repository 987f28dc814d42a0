"""Randomised campaign for the split-tile path (gsr_kernels.h, SegCtl): structured scenes of random kind, size and seed; a short
native loop with heavy tiles split across workgroups; the call's last forward / backward against the CPU oracle at the pose it ran
with (radii exact, images 1e-4, gradients 2e-5, dL/dtau 1e-5), and the same loop with GSR_REFINE_NO_SPLIT (poses 2e-5: two runs of a loop whose loss is made of sign functions end that far apart with or without splitting).
usage: CASES=40 SEED=1 python tools/fuzz_split.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL, util as U
from tests.test_gpu_lean import _camera_of_the_pose_state
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
        runs[tag] = dict(R=R.clone(), T=T.clone(), info=info, fr=fr, color=fr.color.clone(), depth=fr.depth.clone(), alpha=fr.alpha.clone(), gt=(gt_image, gt_depth))
    a = runs["split"]
    fr, info = a["fr"], a["info"]
    st = fr.seg_stats()
    n_split_cases += int(st[1] > 0)
    dT = float((a["T"] - runs["nosplit"]["T"]).abs().max()); dR = float((a["R"] - runs["nosplit"]["R"]).abs().max())
    vm, pm, cp = _camera_of_the_pose_state(info["R_last_forward_host"], info["T_last_forward_host"], S.camera_matrices(sc)[2])
    f = O.forward(sc.means3D, sc.opacities, vm, pm, cp, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
    errs = {"radii": int((fr.radii.cpu().numpy() != f.radii).sum())}
    for k, b in (("color", f.color), ("depth", f.depth), ("alpha", f.alpha)):
        errs[k] = U.rel_l1(a[k].cpu().numpy(), b)
    errs["n_touched"] = float(np.abs(fr.n_touched.cpu().numpy() - f.n_touched).sum() / max(1, f.n_touched.sum()))
    ex = info["exposure_last_forward_host"]
    class _V: pass
    v = _V()
    v.exposure_a, v.exposure_b = torch.tensor([float(ex[0])], device=dev), torch.tensor([float(ex[1])], device=dev)
    v.original_image, v.depth, v.grad_mask = a["gt"][0], a["gt"][1], torch.ones((1, sc.H, sc.W), dtype=torch.bool, device=dev)
    ti, td = a["color"].clone().requires_grad_(True), a["depth"].clone().requires_grad_(True)
    PL.tracking_loss(PL.TRACKING_CONFIG, ti, td, a["alpha"], v).backward()
    go = O.backward(f, ti.grad.cpu().numpy(), td.grad.cpu().numpy(), np.zeros((1, sc.H, sc.W), np.float32), pose_mode=True)
    errs["tau"] = U.rel_l1(fr.g_tau.cpu().numpy(), go["tau"])
    for k, ok in (("m3d", "means3D"), ("sh", "sh"), ("opac", "opacities"), ("scale", "scales"), ("rot", "rotations")):
        errs[k] = U.rel_l1(getattr(fr, "g_" + k).cpu().numpy().reshape(go[ok].shape), go[ok])
    bad = (errs["radii"] != 0 or max(errs["color"], errs["depth"], errs["alpha"]) > 1e-4 or errs["n_touched"] > 1e-4 or errs["tau"] > 1e-5 or
           max(errs[k] for k in ("m3d", "sh", "opac", "scale", "rot")) > (5e-5 if st[1] * 4 > st[3] // 3 else 2e-5) or max(dT, dR) > 2e-5)
    # (a case with more than a quarter of its tiles split is held to 5e-5: on long lists the fp32 oracle itself is 2e-5 from float64 -- its
    # T by repeated division, backward.cu:516 -- and the split path, which restarts each depth range from double-precision sums, is not:
    # tests/test_gpu_split.py::test_split_backward_against_float64_autograd; BOTH=1 prints the unsplit loop's distance next to it)
    for k, e in errs.items():
        worst[k] = max(worst.get(k, 0), e)
    worst["dT"] = max(worst.get("dT", 0), dT)
    print("%s case %d %s P=%d seed=%d K=%d seg=%s fallbacks=%d/%d  dT %.1e  " % ("FAIL" if bad else "ok  ", case, kind, P, seed, K, st, info["fallbacks"], runs["nosplit"]["info"]["fallbacks"], dT) +
          " ".join("%s %.1e" % (k, e) for k, e in errs.items()), flush=True)
    if os.environ.get("BOTH"):          # the unsplit loop against the oracle at ITS last pose: is the split path further from the oracle than the unsplit one?
        b = runs["nosplit"]
        frb, infb = b["fr"], b["info"]
        vm, pm, cp = _camera_of_the_pose_state(infb["R_last_forward_host"], infb["T_last_forward_host"], S.camera_matrices(sc)[2])
        fb = O.forward(sc.means3D, sc.opacities, vm, pm, cp, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree, shs=sc.shs, scales=sc.scales, rotations=sc.rotations)
        exb = infb["exposure_last_forward_host"]
        v.exposure_a, v.exposure_b = torch.tensor([float(exb[0])], device=dev), torch.tensor([float(exb[1])], device=dev)
        ti, td = b["color"].clone().requires_grad_(True), b["depth"].clone().requires_grad_(True)
        PL.tracking_loss(PL.TRACKING_CONFIG, ti, td, b["alpha"], v).backward()
        gb = O.backward(fb, ti.grad.cpu().numpy(), td.grad.cpu().numpy(), np.zeros((1, sc.H, sc.W), np.float32), pose_mode=True)
        print("      unsplit loop against the oracle:", " ".join("%s %.1e" % (k, U.rel_l1(getattr(frb, "g_" + k).cpu().numpy().reshape(gb[ok].shape), gb[ok]))
                                                                 for k, ok in (("m3d", "means3D"), ("sh", "sh"), ("opac", "opacities"), ("scale", "scales"), ("rot", "rotations"))), flush=True)
    del model, runs, fr, a
    torch.cuda.empty_cache()
print("cases", CASES, "with split tiles", n_split_cases, "worst", {k: float("%.2e" % v) for k, v in worst.items()}, "in %.0f s" % (time.time() - t_start))
