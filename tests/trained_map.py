"""TEST INFRASTRUCTURE: the reference's whole pipeline, chained once -- train a map, save it, load it, localise against it.

  gs/*_gs*.py -> gaussian_splatting/train.py        training on posed images        -> tests/train_replay.py (package (A), fused loss)
  scene.save -> GaussianModel.save_ply              point_cloud.ply                 -> gs_localization_amd/map_io.write_ply
  Model.load_ply (7scenes_localize_full_dslam.py:301-302)                           -> GaussianMap.from_ply (gsr_map_from_ply_rows)
  compute_grad_mask | create_mask -> gradient_decent (:352-389)                     -> gsr_grad_mask, FusedRefiner.refine (early exit)

No dataset exists here, so the WORLD is synthetic: S-room (a box room with furniture, flattened splats on its surfaces,
gs_localization_amd/scenes.py).  What makes this different from every other test: the map that is localised against is a TRAINED one --
initialised like create_from_pcd (scene/gaussian_model.py:124-146: a subsample of surface points, isotropic scales from distCUDA2,
opacity 0.1, identity rotations, colours in the DC coefficient), optimised by the replay of train.py with its densification cadence,
written to disk in the reference's PLY layout and read back -- and the query frames are renders of the WORLD, not of the map: the
residual the tracking loss sees is real (the map is not the world), its anisotropy, opacities and ORDER are what training and
densification left (parents first, children appended), not a seeded generator's.
"""
import math
import os
import time

import numpy as np
import torch

from gs_localization_amd import scenes as S
from tests import replay as RP
from tests.train_replay import TrainReplay, LRS

C0 = 0.28209479177387814          # utils/sh_utils.py:26


class RoomTrainer(TrainReplay):
    """tests/train_replay.py with another world: observations are renders of `world` (a scenes.Scene) at `n_views` poses around its
    reference view, the model starts from a point cloud as `create_from_pcd` builds it, SH degree `sh_degree`, black background."""

    def __init__(self, world, P0, P1, n_views=24, sh_degree=3, device="cuda:0", seed=0, densify_from=100, densification_interval=50,
                 densify_until=10**9, opacity_reset_interval=3000, spread=(0.35, 14.0), lambda_dssim=0.2, depth_weight=0.1):
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        from simple_knn._C import distCUDA2
        self.dev = torch.device(device)
        self.W, self.H = world.W, world.H
        self.P0, self.P1 = P0, P1
        self.sh_degree = sh_degree
        self.densify_from, self.interval, self.densify_until = densify_from, densification_interval, densify_until
        self.lambda_dssim, self.depth_weight = lambda_dssim, depth_weight
        self.opacity_reset_interval = opacity_reset_interval
        self.scene = world
        self.rng = np.random.default_rng(seed + 5)
        self.gen = torch.Generator(device=self.dev); self.gen.manual_seed(seed + 7)
        self.bg = torch.zeros(3, device=self.dev)
        t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.dev)
        # ---- observations: the world through package (A)
        wmap = dict(means3D=t(world.means3D), shs=t(world.shs), opacities=t(world.opacities), scales=t(world.scales), rotations=t(world.rotations))
        self.views = []
        for v in range(n_views):
            tau = np.concatenate([self.rng.uniform(-spread[0], spread[0], 3), np.radians(self.rng.uniform(-spread[1], spread[1], 3))]) if v else np.zeros(6)
            w2c = S.se3_exp(tau)
            view, proj, _, campos = S.camera_matrices(world, w2c)
            vw = dict(w2c=w2c, view=t(view), proj=t(proj), campos=t(campos))
            rast = GaussianRasterizer(GaussianRasterizationSettings(
                image_height=self.H, image_width=self.W, tanfovx=world.tanfovx, tanfovy=world.tanfovy, bg=self.bg, scale_modifier=1.0,
                viewmatrix=vw["view"], projmatrix=vw["proj"], sh_degree=world.sh_degree, campos=vw["campos"], prefiltered=False, debug=False))
            with torch.no_grad():
                image, radii, depth, alpha = rast(means2D=torch.zeros_like(wmap["means3D"]), colors_precomp=None, cov3D_precomp=None, **wmap)
            vw["gt"] = image.clamp(0, 1)
            vw["pseudo"] = 100.0 / (depth[0] + 0.5)          # (a monocular estimator's output: inverse-depth-like, as in tests/train_replay.py)
            self.views.append(vw)
        # ---- create_from_pcd: surface points + their colour, isotropic scales from the three nearest neighbours, opacity 0.1
        pick = self.rng.choice(world.P, size=P0, replace=False)
        xyz = t(world.means3D[pick] + self.rng.normal(0, 0.004, (P0, 3)))
        rgb = np.clip(0.5 + C0 * world.shs[pick, 0, :], 0.0, 1.0)
        M = (sh_degree + 1) ** 2
        dist2 = torch.clamp_min(distCUDA2(xyz), 0.0000001)
        self.par = dict(xyz=xyz, f_dc=t(((rgb - 0.5) / C0)[:, None, :]), f_rest=torch.zeros((P0, M - 1, 3), device=self.dev),
                        opacity=torch.logit(torch.full((P0, 1), 0.1, device=self.dev)), scaling=torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3),
                        rotation=torch.tensor([1.0, 0, 0, 0], device=self.dev).repeat(P0, 1))
        for k in self.par:
            self.par[k] = self.par[k].contiguous().requires_grad_(True)
        self.opt = torch.optim.Adam([{"params": [self.par[k]], "lr": LRS[k], "name": k} for k in LRS], lr=0.0, eps=1e-15)
        self._reset_stats()
        n_events = len([i for i in range(1, min(densify_until, 10**6)) if i > densify_from and i % densification_interval == 0]) if densify_until < 10**8 else 1
        self.n_events = n_events
        self.growth = (P1 / P0) ** (1.0 / max(n_events, 1))
        self.events = 0
        self.last = None

    def rasterizer(self, vw):
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        sc = self.scene
        return GaussianRasterizer(GaussianRasterizationSettings(
            image_height=self.H, image_width=self.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=self.bg, scale_modifier=1.0,
            viewmatrix=vw["view"], projmatrix=vw["proj"], sh_degree=self.sh_degree, campos=vw["campos"], prefiltered=False, debug=False))

    def save_ply(self, path):
        """GaussianModel.save_ply (scene/gaussian_model.py:197-213): the RAW parameters"""
        from gs_localization_amd import map_io
        n = lambda x: x.detach().cpu().numpy()
        p = self.par
        return map_io.write_ply(path, n(p["xyz"]), n(p["f_dc"]), n(p["f_rest"]), n(p["opacity"]), n(p["scaling"]), n(p["rotation"]))


def train_room_map(path, steps=7000, world_P=300_000, P0=60_000, P1=250_000, sh_degree=3, seed=0, n_views=24, log=None):
    """world -> trained map on disk.  Returns (world scene, training report)."""
    world = S.s_room_640(P=world_P, seed=seed)
    until = steps
    tr = RoomTrainer(world, P0, P1, n_views=n_views, sh_degree=sh_degree, seed=seed, densify_from=min(500, steps // 6),
                     densification_interval=max(20, min(100, steps // 40)), densify_until=until, opacity_reset_interval=3000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = last = None
    for it in range(1, steps + 1):
        loss = tr.step(it)
        if it == 1:
            first = float(loss)
        if log and it % max(1, steps // 10) == 0:
            log(f"  train step {it}: P = {tr.P}, loss {float(loss):.4f}")
    last = float(loss)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tr.save_ply(path)
    with torch.no_grad():          # how well does the map explain a training view?  (PSNR of view 0)
        out = tr.render(tr.views[0])
        mse = float(((out["image"].clamp(0, 1) - tr.views[0]["gt"]) ** 2).mean())
    rep = dict(steps=steps, P_first=P0, P_last=tr.P, loss_first_last=[first, last], train_wall_s=wall, ms_per_step=1e3 * wall / steps,
               psnr_view0_db=-10.0 * math.log10(max(mse, 1e-12)), ply_bytes=os.path.getsize(path), sh_degree=sh_degree)
    return world, rep


def world_frame(world, wmap, gt_w2c, uid, device, bg):
    """a query frame: the WORLD seen from gt_w2c (image, depth), its mask as the reference's scripts build it"""
    fr = RP.QueryFrame(uid, RP.intrinsics_projection(world, device), world, device, gt_w2c=torch.tensor(gt_w2c, dtype=torch.float32, device=device))
    g = torch.tensor(gt_w2c, dtype=torch.float32, device=device)
    fr.update_RT(g[:3, :3].clone(), g[:3, 3].clone())
    with torch.no_grad():
        obs = RP.render(fr, wmap, bg)
    fr.original_image, fr.depth = obs["render"].detach().clamp(0, 1).clone(), obs["depth"].detach()[0].clone()
    fr.grad_mask = RP.reference_mask(fr.original_image, uid)
    return fr


def localise_against(path, world, n_frames=16, iters=50, start=(0.05, 3.0), spread=(0.25, 10.0), device="cuda:0", in_flight=16, seed=0, max_sh_degree=None):
    """point_cloud.ply -> GaussianMap.from_ply -> per frame: mask, FusedRefiner.refine with the early exit
    (7scenes_localize_full_dslam.py:301-302,352-389).  Start poses `start` (m, deg) off the ground truth.  Returns a report and the
    last frame's (scene of the map, refiner, run dict, frame) for the oracle check."""
    import threading
    dev = torch.device(device)
    bg = torch.zeros(3, device=dev)
    gmap = RP.GaussianMap.from_ply(path, device=dev, max_sh_degree=max_sh_degree)
    wmap = RP.GaussianMap.from_scene(world, device=dev, requires_grad=False)
    rng = np.random.default_rng(seed + 99)
    frames, inits, gts = [], [], []
    for f in range(n_frames):
        gt = S.se3_exp(np.concatenate([rng.uniform(-spread[0], spread[0], 3), np.radians(rng.uniform(-spread[1], spread[1], 3))]))
        dt = rng.normal(size=3); dt *= start[0] / np.linalg.norm(dt)
        dr = rng.normal(size=3); dr *= math.radians(start[1]) / np.linalg.norm(dr)
        gts.append(gt)
        inits.append(torch.tensor(S.se3_exp(np.concatenate([dt, dr])) @ gt, dtype=torch.float32, device=dev))
        frames.append(world_frame(world, wmap, gt, f, dev, bg))
    H, W = world.H, world.W
    fr0 = RP.FusedRefiner(gmap, H, W, device=dev)

    def one(refiner, f, stop=True, n=iters):
        frames[f].grad_mask = RP.reference_mask(frames[f].original_image, f)          # (per frame, inside the timed call, as the scripts do)
        return refiner.refine(frames[f], RP.TRACKING_CONFIG, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=n, stop_on_converged=stop)
    one(fr0, 0)          # warm-up (allocations)
    torch.cuda.synchronize()
    errs, used, t_single, it_single = [], [], 0.0, 0
    for f in range(n_frames):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R, T, info = one(fr0, f)
        torch.cuda.synchronize()
        t_single += time.perf_counter() - t0
        it_single += info["iters"]
        used.append(info["iters"])
        errs.append(RP.pose_errors(gts[f][:3, :3], gts[f][:3, 3], info["R_host"], info["T_host"]))
    e0 = [RP.pose_errors(gts[f][:3, :3], gts[f][:3, 3], inits[f][:3, :3].cpu().numpy(), inits[f][:3, 3].cpu().numpy()) for f in range(n_frames)]
    # F frames in flight: one refiner, stream and host thread each
    F = max(1, min(in_flight, n_frames))
    refs = [fr0] + [RP.FusedRefiner(gmap, H, W, device=dev) for _ in range(F - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    done = [0] * F

    def worker(s):
        with torch.cuda.stream(streams[s]):
            for f in range(s, n_frames, F):
                _, _, inf = one(refs[s], f)
                done[s] += inf["iters"]
            streams[s].synchronize()
    for s in range(F):          # warm every refiner
        with torch.cuda.stream(streams[s]):
            one(refs[s], s % n_frames)
    torch.cuda.synchronize()
    done = [0] * F
    th = [threading.Thread(target=worker, args=(s,)) for s in range(F)]
    t0 = time.perf_counter()
    [x.start() for x in th]; [x.join() for x in th]
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    errs = np.array(errs)
    rep = dict(frames=n_frames, map_gaussians=int(gmap.get_xyz.shape[0]), start_cm_deg=[100 * start[0], start[1]],
               start_err_cm_deg_median=[100 * float(np.median([e[0] for e in e0])), float(np.median([e[1] for e in e0]))],
               pose_err_cm_median=100 * float(np.median(errs[:, 0])), pose_err_deg_median=float(np.median(errs[:, 1])),
               pose_err_cm_max=100 * float(errs[:, 0].max()), iterations_used_median=float(np.median(used)), iterations_used=[int(u) for u in used],
               single_frame_iters_per_s=it_single / t_single, in_flight=F, in_flight_iters_per_s=sum(done) / t_all, frames_per_s_in_flight=n_frames / t_all,
               mask_share=float(np.mean([float(fr.grad_mask.float().mean()) for fr in frames])))
    return rep, (gmap, fr0, frames, inits, bg)


def scene_of_map(gmap, world):
    """the loaded map at the rasterizer boundary (post-activation, numpy) as a scenes.Scene: what the CPU oracle renders"""
    n = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), np.float32)
    shs = n(gmap.get_features)
    return S.Scene(name=f"trained-room@{shs.shape[0]}", W=world.W, H=world.H, fx=world.fx, fy=world.fy, cx=world.cx, cy=world.cy, znear=world.znear,
                   zfar=world.zfar, sh_degree=int(gmap.active_sh_degree), means3D=n(gmap.get_xyz), scales=n(gmap.get_scaling), rotations=n(gmap.get_rotation),
                   opacities=n(gmap.get_opacity).reshape(-1, 1), shs=shs, bg=np.zeros(3, np.float32))
