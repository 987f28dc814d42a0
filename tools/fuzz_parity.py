"""Randomised parity against the CPU oracle: small random scenes with extreme shapes (needle-like and huge splats, opacities
near 1/255 and near 1, splats straddling the image border and the near plane, partial tiles), both packages.  Checks
radii exactly, images per PIXEL (max abs difference: a wrongly culled tile instance would show up as ~1/255 on single
pixels, which a relative-L1 test over the image would not notice) and gradients against float64 autograd, next to the
oracle's own error."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S
from tests import util as U
from oracle import oracle as O, autograd_ref as AG
O.set_threads(min(16, os.cpu_count() or 1))
# Needle-like splats, near-plane splats and five-Gaussian scenes make the gradients ill-conditioned in fp32: the oracle (the
# reference's algorithm, fp32 atomics) and the HIP path can then be 1e-2 apart while both are equally far from the exact
# value.  So both are measured against a float64 autograd evaluation, and the HIP path must not be worse than the oracle
# by more than a factor (or 2e-4).
def run(N=60, seed=1, only=None, verbose=True, needle=(12.0, 0.05)):
    """Runs N random cases; returns a dict: failures (list of strings; empty = pass), nbad, flipped, worst."""
    rng = np.random.default_rng(seed)
    failures = []
    os.environ["GSR_SPECULATION"] = os.environ.get("GSR_SPECULATION", "1")
    worst = dict(color=0.0, depth=0.0, alpha=0.0, grad=0.0, ratio=0.0)
    flipped = 0; nbad = 0; nflipgrad = 0
    for case in range(N):
        W = int(rng.integers(17, 150)); H = int(rng.integers(17, 120))
        P = int(rng.choice([50, 400, 2500]))
        deg = int(rng.integers(0, 4))
        sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=int(rng.integers(1 << 30)), scale_med=float(rng.choice([0.005, 0.03, 0.15, 0.6])))
        kind = rng.integers(4)
        if kind == 0:      # needles
            sc.scales[:, 0] *= needle[0]; sc.scales[:, 1] *= needle[1]
        elif kind == 1:    # opacities around the 1/255 threshold and near 1
            sc.opacities[:] = rng.choice([0.003, 0.0039, 0.004, 0.01, 0.5, 0.999, 1.0], size=sc.opacities.shape).astype(np.float32)
        elif kind == 2:    # close to the near plane / behind the camera
            sc.means3D[:, 2] = rng.uniform(-0.5, 1.0, sc.P).astype(np.float32)
        w2c = S.se3_exp(rng.normal(size=6) * np.array([0.2, 0.2, 0.2, 0.1, 0.1, 0.1]))
        cam = U.scene_inputs(sc, w2c)
        grads = U.random_grads(sc, seed=case)
        pose = bool(rng.integers(2))
        if rng.random() < 0.5:
            sc.bg[:] = rng.random(3).astype(np.float32)
        grads = (grads[0], grads[1], np.zeros_like(grads[2]))      # (the float64 reference below has no alpha-gradient input)
        if only is not None and case != only:
            continue
        f, go = U.oracle_run(sc, cam, grads, pose=pose)
        # float64 autograd of the same composition with the oracle's threshold decisions frozen: the yardstick for BOTH fp32 paths
        t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
        m, op, sh, scl, rot = t(sc.means3D), t(sc.opacities), t(sc.shs), t(sc.scales), t(sc.rotations)
        tau = torch.zeros(6, dtype=torch.float64, requires_grad=True) if pose else None
        col, dep, alp, aux = AG.render_autograd(f.state(), f.radii, m, op, torch.tensor(np.asarray(w2c, np.float64)),
                                                torch.tensor(cam["proj_raw"].T.astype(np.float64)), sc.W, sc.H, sc.tanfovx, sc.tanfovy,
                                                torch.tensor(sc.bg.astype(np.float64)), sh_degree=sc.sh_degree, tau=tau, depth_to_mean=pose,
                                                shs=sh, scales=scl, rotations=rot)
        truth = {}
        Lsum = (col * torch.tensor(grads[0].astype(np.float64))).sum() + (dep * torch.tensor(grads[1][0].astype(np.float64))).sum()
        if Lsum.requires_grad:          # (nothing visible: no gradient path)
            Lsum.backward()
            truth = {k: v.grad.numpy() for k, v in dict(means3D=m, opacities=op, sh=sh, scales=scl, rotations=rot).items() if v.grad is not None}
            if pose and tau.grad is not None:
                truth["tau"] = tau.grad.numpy()
        case_flips = 0
        for rep in range(2):      # second render: speculative
            o, g = U.hip_run(sc, cam, grads, pose=pose)
            if not np.array_equal(o["radii"], f.radii):
                failures.append(f"RADII MISMATCH case {case} {W}x{H} P={P} kind={kind}"); break
            for k, ref in (("color", f.color), ("depth", f.depth), ("alpha", f.alpha)):
                scale = max(1.0, float(np.abs(ref).max()))
                dd = np.abs(o[k].reshape(ref.shape) - ref) / scale
                # v_exp_f32 (2 ulp) against expf: a pair whose alpha lies within an ulp of 1/255 (or whose T lies at 1e-4) is
                # blended by one side and skipped by the other -- up to alpha * T * colour on that pixel.  A handful of such
                # pixels per image is rounding; more, or a larger jump, is a bug.  (That the tile / quadrant culling itself
                # never drops a blended pair is checked bit-exactly by tools/cull_check.py.)
                flips = int((dd > 3e-4).sum()); flipped += flips; case_flips += flips
                d = float(dd[dd <= 3e-4].max()) if (dd <= 3e-4).any() else 0.0
                worst[k] = max(worst[k], d)
                if flips > 4 or float(dd.max()) > 2e-2:
                    failures.append(f"IMAGE MISMATCH case {case} rep {rep} {k} max {float(dd.max()):.3e} flips {flips} {W}x{H} P={P} kind={kind} deg={deg}")
            for k, tr in truth.items():
                if np.abs(tr).sum() == 0:
                    continue
                e_hip = U.rel_l1(np.asarray(g[k]).reshape(tr.shape), tr); e_orc = U.rel_l1(np.asarray(go[k]).reshape(tr.shape), tr)
                worst["grad"] = max(worst["grad"], e_hip); worst["ratio"] = max(worst["ratio"], e_hip / max(e_orc, 1e-6))
                if e_hip > max(5.0 * e_orc, 2e-4):
                    # (the oracle's own fp32 atomics add in another order every run: on an ill-conditioned case its error moves by a
                    # factor of five from run to run -- seed 103, case 28: a 0.6 m x 1 mm needle 0.32 m from the camera, oracle error
                    # 0.0015 ... 0.0073 against float64, HIP 0.0081.  So a flag counts only if it survives two more oracle runs.)
                    for _ in range(2):
                        _f2, go2 = U.oracle_run(sc, cam, grads, pose=pose)
                        e_orc = max(e_orc, U.rel_l1(np.asarray(go2[k]).reshape(tr.shape), tr))
                # (The float64 yardstick freezes the ORACLE's threshold decisions.  A pixel whose alpha lies within an ulp of 1/255 is blended
                # by one fp32 path and skipped by the other -- the image check above counts those -- and under white-noise pixel gradients
                # that one pixel is a visible share of a thin splat's gradient: seed 5001 case 269, needles, four flipped pixels, dL/dtau
                # 3e-4 off with every other tensor within 2.3e-5, same numbers with all culling switched off.  A case WITH flipped pixels
                # is therefore held to 1e-3 and reported separately.)
                flip_case = case_flips > 0 and e_hip <= 1e-3
                if flip_case and e_hip > max(5.0 * e_orc, 2e-4) and only is None:
                    nflipgrad += 1
                    if verbose:
                        print("note: case", case, "rep", rep, k, "HIP", e_hip, "oracle", e_orc, "with", case_flips, "threshold-flipped pixel values")
                    continue
                if e_hip > max(5.0 * e_orc, 2e-4) or (only is not None and k in ("tau", "means3D")):
                    if verbose:
                        print("GRAD", case, rep, k, "HIP", e_hip, "oracle", e_orc, W, H, P, kind, deg)
                    if only is None:
                        nbad += 1; failures.append(f"GRAD case {case} rep {rep} {k}: HIP {e_hip:.3e} vs oracle {e_orc:.3e} against float64 ({W}x{H} P={P} kind={kind} deg={deg})")
                    if only is not None:          # where does the difference sit?
                        gh, gor = np.asarray(g[k]).reshape(tr.shape), np.asarray(go[k]).reshape(tr.shape)
                        rows = np.abs(gh - tr).reshape(tr.shape[0], -1).sum(1) if tr.ndim > 1 else np.abs(gh - tr)
                        top = np.argsort(rows)[::-1][:6]
                        for i in top:
                            extra = (" z_view %.4f scale %s opacity %.3f" % (float((np.asarray(w2c)[:3, :3] @ sc.means3D[i] + np.asarray(w2c)[:3, 3])[2]), np.round(sc.scales[i], 4), float(sc.opacities[i]))) if tr.ndim > 1 and tr.shape[0] == sc.P else ""
                            print("   row", int(i), "HIP", np.round(gh[i], 5), "oracle", np.round(gor[i], 5), "float64", np.round(tr[i], 5), extra)
                            if k == "means3D":      # (which input of the chain rule differs: the other per-Gaussian gradients of the same row)
                                for k2 in ("means2D", "scales", "rotations", "opacities"):
                                    if g.get(k2) is not None and go.get(k2) is not None:
                                        a2, b2 = np.asarray(g[k2]).reshape(sc.P, -1)[i], np.asarray(go[k2]).reshape(sc.P, -1)[i]
                                        t2 = truth[k2].reshape(sc.P, -1)[i] if k2 in truth else None
                                        print("        ", k2, "HIP", np.round(a2, 5), "oracle", np.round(b2, 5), "" if t2 is None else ("float64 %s" % np.round(t2, 5)))
                        print("   share of the total |HIP - truth| in these rows: %.3f" % (rows[top].sum() / max(rows.sum(), 1e-30)))
    summary = (f"{N} cases (seed {seed}), {nbad} gradient tensors more than 5x (and 2e-4) further from float64 than the oracle ({nflipgrad} more in cases with threshold-flipped pixels, within 1e-3); worst per-pixel image "
               f"differences {worst['color']:.2e} / {worst['depth']:.2e} / {worst['alpha']:.2e} (colour / depth / alpha, relative to max(1, |image|max); "
               f"{flipped} pixel values beyond that from threshold flips); gradients against float64 autograd: worst HIP error {worst['grad']:.2e}, "
               f"worst HIP error / oracle error {worst['ratio']:.1f}")
    if verbose:
        print(summary)
    return dict(failures=failures, nbad=nbad, flipped=flipped, worst=worst, summary=summary)


if __name__ == "__main__":
    r = run(N=int(os.environ.get("CASES", 60)), seed=int(os.environ.get("SEED", 1)), only=(int(os.environ["ONLY"]) if "ONLY" in os.environ else None),
            needle=(float(os.environ.get("NEEDLE_LONG", 12.0)), float(os.environ.get("NEEDLE_THIN", 0.05))))
    for f_ in r["failures"]:
        print(f_)
    sys.exit(1 if r["failures"] else 0)
