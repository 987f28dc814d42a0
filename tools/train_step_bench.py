"""Times one train.py-style step (gaussian_splatting/train.py:80-160) on the drop-in package (A):
activations -> diff_gaussian_rasterization forward -> L1 + SSIM + Pearson depth loss -> backward with gradients
for every Gaussian parameter -> densification statistics -> Adam on the six parameter groups.
Scene: S-train-garden-like: SH degree 1, white background, P = 0.2 M ... 1.5 M, grad_depth != 0.
Like train.py:71-75 every step renders another, randomly picked training camera (16 views, up to 0.4 m / 12 deg apart), so
the drop-in's depth speculation (gsr_forward_speculative) mostly misses and backs off; the counters are printed.
Prints ms per step and where it goes (rasterizer forward / loss / backward / optimizer)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S, train_epilogue as TE, rasterizer as RZ
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
W, H = int(os.environ.get("TW", 1296)), int(os.environ.get("TH", 840))
FX = 0.9 * W


def build(P):
    sc = S._draw("S-train-garden", P, W, H, FX, FX, 0.5, 6.0, 0.012, 0.6, 1, 0)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    rng = np.random.default_rng(5)
    rasts = []
    for v in range(16):
        tau = np.concatenate([rng.uniform(-0.4, 0.4, 3), np.radians(rng.uniform(-12, 12, 3))]) if v else np.zeros(6)
        view, proj, _, campos = S.camera_matrices(sc, S.se3_exp(tau))
        rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy,
                                           bg=torch.ones(3, device=dev), scale_modifier=1.0, viewmatrix=t(view), projmatrix=t(proj),
                                           sh_degree=1, campos=t(campos), prefiltered=False, debug=False)
        rasts.append(GaussianRasterizer(rs))
    par = dict(xyz=t(sc.means3D), f_dc=t(sc.shs[:, :1]), f_rest=t(sc.shs[:, 1:]),
               scaling=torch.log(t(sc.scales)), rotation=t(sc.rotations), opacity=torch.logit(t(sc.opacities).clamp(1e-4, 1 - 1e-4)))
    for v in par.values():
        v.requires_grad_(True)
    return sc, par, rasts


def render(par, rast):
    P = par["xyz"].shape[0]
    screenspace = torch.zeros_like(par["xyz"], requires_grad=True)
    screenspace.retain_grad()
    shs = torch.cat((par["f_dc"], par["f_rest"]), dim=1)
    color, radii, depth, alpha = rast(means3D=par["xyz"], means2D=screenspace, shs=shs, colors_precomp=None,
                                      opacities=torch.sigmoid(par["opacity"]), scales=torch.exp(par["scaling"]),
                                      rotations=torch.nn.functional.normalize(par["rotation"]), cov3D_precomp=None)
    return color, radii, depth, screenspace


def main():
    for P in (200_000, 800_000, 1_500_000):
        sc, par, rasts = build(P)
        gts = []
        with torch.no_grad():
            for rast in rasts:
                gt, _, d0, _ = render(par, rast)
                gts.append(((gt + 0.03 * torch.randn_like(gt)).clamp(0, 1), 100.0 / (d0[0] + 0.5) + torch.randn_like(d0[0])))
        pick = np.random.default_rng(9)
        RZ._spec_cache.states.clear()
        opt = torch.optim.Adam([{"params": [v], "lr": lr} for v, lr in zip(par.values(), (1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 5e-2))], eps=1e-15)
        max_radii = torch.zeros(P, device=dev); accum = torch.zeros(P, 1, device=dev); denom = torch.zeros(P, 1, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        acc = np.zeros(4); n = 0

        def step(timed):
            nonlocal n
            if timed: ev[0].record()
            v = int(pick.integers(len(rasts)))
            gt, pseudo = gts[v]
            img, radii, depth, ss = render(par, rasts[v])
            if timed: ev[1].record()
            loss = TE.training_loss(img, gt, 0.2, depth[0], pseudo, 0.1)
            if timed: ev[2].record()
            loss.backward()
            if timed: ev[3].record()
            with torch.no_grad():
                TE.add_densification_stats(radii, ss.grad, max_radii, accum, denom)
                opt.step(); opt.zero_grad(set_to_none=True)
            if timed:
                ev[4].record(); torch.cuda.synchronize()
                for i in range(4): acc[i] += ev[i].elapsed_time(ev[i + 1])
                n += 1
            return loss

        for _ in range(5): step(False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 50
        for _ in range(K): l = step(False)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
        for _ in range(10): step(True)
        a = acc / n
        print(f"train step {W}x{H} P={P:8d} SH1: {el * 1e3:6.2f} ms/step ({1 / el:6.1f} it/s); rasterizer fwd (+activations) {a[0]:.2f} ms, "
              f"loss epilogue {a[1]:.2f} ms, backward {a[2]:.2f} ms, densification stats + Adam {a[3]:.2f} ms; loss {float(l.detach()):.4f}; "
              f"speculative forwards verified/missed {RZ.speculation_counters()}", flush=True)
        del par, rasts, opt, gts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
