"""The BASELINE.json workloads as bench.py times them, in one place so that the GPU parity tests run on EXACTLY the
scene the headline number is quoted on: seeded synthetic Gaussians (synthetic.py), the
shape-matched deformation network, and the uniform log-scale offset that calibrates the DEFORMED scene to ~8 tiles
per Gaussian.  Host plumbing over the product operators only (needs a HIP device; nothing here touches oracle/).
"""
from __future__ import annotations

import types

import numpy as np
import torch

from . import synthetic as syn
from .renderer import Camera, GaussianParams
from .scene.deformation import deform_network


def deform_args(name: str):
    """Shape-matched deformation configs: mutant-shaped for C1-C4 (arguments/dnerf/others/mutant.py
    + dnerf_default.py), force_blend-shaped for C5 (arguments/dnerf/force_blend.py)."""
    a = types.SimpleNamespace(
        net_width=64, defor_depth=0, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2, grid_pe=0, force_pe=2,
        time_pe=2, bounds=1.6, multires=[1, 2], no_dx=False, no_ds=False, no_dr=False, no_do=True, no_dshs=False,
        empty_voxel=False, static_mlp=False, apply_rotation=True, use_time=True, use_force=False,
        blend_time_force=True, prev_frames=0, no_grid=False,
        kplanes_config={"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                        "resolution": [64, 64, 64, 75]})
    if name in ("C3", "C4"):
        a.kplanes_config["resolution"] = [64, 64, 64, 150]
    if name == "C5":
        a.kplanes_config.update(output_coordinate_dim=128, resolution=[64, 64, 64, 256])
        a.defor_depth, a.force_pe, a.time_pe, a.use_force = 5, 4, 4, True
    return a


def rasterize_untimed(pc, deform, cam, cam_np, bg):
    """One forward through the C-level API on the deformed scene -> rasterize_gaussians' 7-tuple."""
    from .diff_gaussian_rasterization import _C as Craw
    P, dev = pc._xyz.shape[0], pc._xyz.device
    with torch.no_grad():
        m3, sc, rot, op, sh = deform.forward_dynamic(pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features,
                                                     torch.full((P, 1), cam.time, device=dev),
                                                     torch.full((P, 1), cam.force, device=dev), None)
        e = torch.Tensor([])
        return Craw.rasterize_gaussians(
            bg, m3, e, torch.sigmoid(op), torch.exp(sc), torch.nn.functional.normalize(rot), 1.0, e,
            cam.world_view_transform, cam.full_proj_transform, cam_np.tanfovx, cam_np.tanfovy, cam_np.image_height,
            cam_np.image_width, sh, 3, cam.camera_center, False, False)


def calibrate_log_scale_offset(pc, deform, cam, cam_np, bg, target_tiles: float = 8.0) -> float:
    """s0 is chosen per config so that the RENDERED scene has ~8 tiles per Gaussian.  The random-init
    deformation heads shift the log-scales, so a uniform log-scale offset is found by a few fixed-point steps
    (tiles ~ scale^2) on this view and ADDED to pc._scaling in place.  Returns the offset."""
    P = pc._xyz.shape[0]
    offset = 0.0
    for _ in range(8):
        ratio = rasterize_untimed(pc, deform, cam, cam_np, bg)[0] / P
        if abs(ratio - target_tiles) < 0.2 or ratio == 0:
            break
        d = 0.5 * float(np.log(target_tiles / ratio))
        offset += d
        with torch.no_grad():
            pc._scaling.add_(d)
    return offset


def build_views(name: str, dev, rank: int = 0, n: int = 8):
    """The n distinct (camera, timestamp, target image) views rank `rank` cycles through, one per training step, as
    train.py:168-189 draws a random view per iteration.  View 0 is build_scene's view (the one the parity tests and the
    CPU baseline use); the others move the camera around and above the scene and draw their own timestamp / target."""
    cfg = syn.CONFIGS[name]
    W, H = cfg["W"], cfg["H"]
    rng = np.random.default_rng(1234 + rank)
    views = []
    for j in range(n):
        cam_np = syn.look_at_camera(W, H, azimuth=0.7 + 0.6 * rank + 0.785 * j, elevation=0.35 + 0.12 * ((j * 3) % 5 - 2) * (j > 0))
        cam = Camera(cam_np, dev, time=float(rng.uniform(0, 1)), force=0.0)
        target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(99 + rank + 1000 * j))
        views.append(types.SimpleNamespace(cam=cam, cam_np=cam_np, target=target))
    return views


def morton_permutation(xyz: np.ndarray, bits: int = 10) -> np.ndarray:
    """Row order that sorts the points by the Morton code of their positions (stable): GaussianModel.reorder_spatially
    on the host."""
    lo, hi = xyz.min(axis=0), xyz.max(axis=0)
    q = np.clip(((xyz - lo) / np.maximum(hi - lo, 1e-12) * ((1 << bits) - 1)).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(xyz.shape[0], dtype=np.int64)
    for b in range(bits):
        for axis in range(3):
            code |= ((q[:, axis] >> b) & 1) << (3 * b + axis)
    return np.argsort(code, kind="stable")


def build_scene(name: str, dev, P: int | None = None, rank: int = 0, calibrate: bool = True, spatial_order: bool = False):
    """-> dict(pc, deform, cam, cam_np, raw, bg, target, log_s0_offset, W, H, P) for a BASELINE.json config: what
    bench.py's rank `rank` renders each step (its own azimuth / timestamp / target image)."""
    cfg = syn.CONFIGS[name]
    _, _, raw = syn.make_workload(name, P=P)
    if spatial_order:
        # the same Gaussians stored in the Morton order of their positions (the generator's order is i.i.d., any order is
        # the same scene): neighbouring rows then read neighbouring HexPlane texels
        perm = morton_permutation(raw["xyz"])
        raw = {k: np.ascontiguousarray(v[perm]) for k, v in raw.items()}
    W, H = cfg["W"], cfg["H"]
    torch.manual_seed(syn.seed_for(name))
    pc = GaussianParams(raw, dev)
    deform = deform_network(deform_args(name))
    deform.deformation_net.set_aabb([1.3, 1.3, 1.3], [-1.3, -1.3, -1.3])
    deform = deform.to(dev)
    rng = np.random.default_rng(1234 + rank)
    cam_np = syn.look_at_camera(W, H, azimuth=0.7 + 0.6 * rank)
    cam = Camera(cam_np, dev, time=float(rng.uniform(0, 1)), force=0.0)
    bg = torch.ones(3, device=dev)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(99 + rank))
    offset = calibrate_log_scale_offset(pc, deform, cam, cam_np, bg) if calibrate else 0.0
    raw = dict(raw, log_scales=pc._scaling.detach().cpu().numpy().copy())      # exactly what the device holds
    return dict(pc=pc, deform=deform, cam=cam, cam_np=cam_np, raw=raw, bg=bg, target=target, log_s0_offset=offset,
                W=W, H=H, P=raw["xyz"].shape[0])
