"""EXPERIMENT: the native loop on variants of S-1M-640 that look more like a trained map than the uniform random scene does --
a dense object, large background splats, two thin depth layers ("walls"), Morton order -- it/s, failed speculations, host redos and
per-kernel HIP-event times, speculative and plain.  argv: variant names (default: all)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
lib = _lib.load()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]

def variant(kind):
    sc = S.s_1m_640()
    rng = np.random.default_rng(11)
    P = sc.P
    m = sc.means3D.copy()
    if kind == "object":            # half of the map inside a cone around the optical axis: a tenth of the image width
        sel = rng.random(P) < 0.5
        z = m[sel, 2]
        m[sel, 0] = rng.normal(0, 0.03, sel.sum()) * z; m[sel, 1] = rng.normal(0, 0.03, sel.sum()) * z
    elif kind == "background":      # 1 % large splats far away
        sel = rng.random(P) < 0.01
        sc.scales = sc.scales.copy(); sc.scales[sel] *= 30.0
        m[sel, 2] = rng.uniform(5.0, 6.0, sel.sum())
    elif kind == "walls":           # two thin depth layers
        lay = rng.random(P) < 0.5
        zz = np.where(lay, 3.0, 5.0) + rng.normal(0, 0.02, P)
        m[:, 0] *= zz / m[:, 2]; m[:, 1] *= zz / m[:, 2]; m[:, 2] = zz
    elif kind == "opaque":          # every splat nearly opaque: short walks, very short needed lists
        sc.opacities = np.full_like(sc.opacities, 0.95)
    elif kind == "faint":           # every splat faint: nothing saturates, no depth bound ever holds
        sc.opacities = np.full_like(sc.opacities, 0.02)
    sc.means3D = np.ascontiguousarray(m.astype(np.float32))
    return sc

kinds = sys.argv[1:] or ["base", "object", "background", "walls", "opaque", "faint"]
for kind in kinds:
    sc = variant(kind); H, W = sc.H, sc.W
    model = PL.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    vp = PL.make_frame(sc, model, dev, bg)
    init = PL.perturbed_start(1000, device=dev)
    fr = PL.FusedRefiner(model, H, W, device=dev)
    for spec in (True, False):
        fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False, speculative=spec)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 50
        fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=N, stop_on_converged=False, speculative=spec, count_instances=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        info = dict(fr.last_info)
        lib.gsr_profile_enable((1 << nk) - 1)
        fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=20, stop_on_converged=False, speculative=spec)
        torch.cuda.synchronize()
        ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
        print("%-10s %-5s it/s %6.0f" % (kind, "spec" if spec else "plain", N / dt), info,
              {names[i]: round(1e3 * ms[i] / 20, 1) for i in range(nk) if ms[i] > 0}, flush=True)
    del model, fr, vp
    torch.cuda.empty_cache()
