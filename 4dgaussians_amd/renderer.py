"""Render orchestration for the hot path: the counterpart of gaussian_renderer/renderer.py's
render() (:82-221) and render_for_image() (:258-329) restricted to what SURVEY.md section 8 puts
in scope -- settings, one deformation query, the three activations
(scene/gaussian_model.py:38-43: exp / L2-normalise / sigmoid) and the rasterizer call.  The
fork-specific regularisers (two extra deformation queries + the k=100 kNN, :173-203) are a "next" row:
`render(..., regularise=True)` adds them (rigidity.py) and returns the reference's 7-tuple.

Host-side plumbing only: all compute happens in libgs4d.so through the two drop-in modules.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _shared_zeros
from .fused import activate
from .render_call import RenderCall, render_call, render_step


class GaussianParams(nn.Module):
    """The six per-Gaussian parameter tensors of scene/gaussian_model.py:176-183 (59 floats per Gaussian)."""

    def __init__(self, raw: dict, device):
        super().__init__()
        t = lambda a: nn.Parameter(torch.from_numpy(np.ascontiguousarray(a)).to(device))
        self._xyz = t(raw["xyz"])
        self._features = t(raw["sh"])            # (P,16,3) as returned by get_features
        self._opacity = t(raw["opacity_logit"])
        self._scaling = t(raw["log_scales"])
        self._rotation = t(raw["rot"])
        self.active_sh_degree = 3


class Camera:
    def __init__(self, cam, device, time: float = 0.0, force: float = 0.0):
        self.image_width, self.image_height = cam.image_width, cam.image_height
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        self.world_view_transform = torch.from_numpy(cam.viewmatrix).to(device)
        self.full_proj_transform = torch.from_numpy(cam.projmatrix).to(device)
        self.camera_center = torch.from_numpy(cam.campos).to(device)
        self.time, self.force = float(time), float(force)
        self._columns = {}

    def time_force_columns(self, n: int, device):
        """(n, 1) columns filled with this view's time and force: constant per camera, so built once per point count
        instead of two fill kernels per step (the reference rebuilds them every call, renderer.py:131-135)."""
        key = (n, str(device), self.time, self.force)
        cols = self._columns.get(key)
        if cols is None:
            self._columns.clear()
            cols = self._columns[key] = (torch.full((n, 1), self.time, dtype=torch.float32, device=device),
                                         torch.full((n, 1), self.force, dtype=torch.float32, device=device))
        return cols


BATCHED_QUERY_MAX_POINTS = 2_000_000             # above this the regularised render queries the field three times instead of once
MAX_FRAME = 25                                   # scene/dataset_readers.py:30: the regulariser's time step is 1 / (MAX_FRAME - 1)


STAGED_ONLY = bool(os.environ.get("GS4D_RENDER_STAGED"))      # tests / experiments: never take the one-call path


def _one_call(cam, pc, deform, bg_color, scaling_modifier, stage, shs_grad_hook, screenspace_points, l1_target,
              step_scale=None):
    """The whole view through gs4d_render_forward (render_call.py) -> (image, radii, depth, loss), or None when the
    one-call path does not apply (non-float32 / non-contiguous / CPU tensors, an empty scene, a deformation module
    without a cached plan): the caller then takes the staged path, which also owns every error message."""
    shs = pc._features if hasattr(pc, "_features") else pc.get_features
    xyz, opacity, scaling, rotation = pc._xyz, pc._opacity, pc._scaling, pc._rotation
    n = xyz.shape[0]
    if n == 0 or shs.dim() != 3 or shs.shape[1] != 16 or not isinstance(bg_color, torch.Tensor):
        return None
    net, time, force = None, None, None
    if "coarse" not in stage and deform is not None:
        net = getattr(deform, "deformation_net", deform)
        if not hasattr(net, "cached_plan") or not hasattr(cam, "time_force_columns"):
            return None
        time, force = cam.time_force_columns(n, xyz.device)
    try:
        call = RenderCall(cam, bg_color, scaling_modifier, pc.active_sh_degree, False, net, time, force, l1_target,
                          shs_grad_hook)
    except NotImplementedError:
        return None                       # unsupported deformation configuration: the staged path raises it
    if not call.applicable((xyz, scaling, rotation, opacity, shs)):
        return None
    if step_scale is not None:
        return render_step(call, xyz, scaling, rotation, opacity, shs, screenspace_points, step_scale)
    return render_call(call, xyz, scaling, rotation, opacity, shs, screenspace_points)


def render_with_l1(cam: Camera, pc: GaussianParams, deform: Optional[nn.Module], bg_color: torch.Tensor,
                   target: torch.Tensor, scaling_modifier: float = 1.0, stage: str = "fine", shs_grad_hook=None):
    """render() followed by l1_loss(image, target) (train.py:203, utils/loss_utils.py:20-21) with the loss computed
    inside the same library call -> (loss, image, screenspace_points, visibility_filter, radii, depth)."""
    if not STAGED_ONLY and torch.is_grad_enabled():
        screenspace_points = _shared_zeros(tuple(pc._xyz.shape), pc._xyz.device).detach().requires_grad_(True)
        out = _one_call(cam, pc, deform, bg_color, scaling_modifier, stage, shs_grad_hook, screenspace_points, target)
        if out is not None:
            image, radii, depth, loss = out
            return loss, image, screenspace_points, radii > 0, radii, depth
    from .fused import l1_loss
    image, screenspace_points, vis, radii, depth = render(cam, pc, deform, bg_color, scaling_modifier, stage,
                                                          shs_grad_hook=shs_grad_hook)
    return l1_loss(image, target), image, screenspace_points, vis, radii, depth


def render_l1_step(cam: Camera, pc: GaussianParams, deform: Optional[nn.Module], bg_color: torch.Tensor,
                   target: torch.Tensor, loss_scale: float = 1.0, scaling_modifier: float = 1.0, stage: str = "fine",
                   shs_grad_hook=None):
    """render_with_l1() AND the backward pass of loss_scale * loss in one go: what the reference's iteration does with
    `render(...)`, `l1_loss(image, gt)` and `loss.backward()` (train.py:190-203,230) when the L1 term is the image's only
    consumer.  The gradients are accumulated into `.grad` of the Gaussian parameters, the deformation parameters and the
    returned screenspace_points; further loss terms (regularisers) run their own backward and accumulate as usual.
    Forward and backward are issued back to back (render_call.render_step): the host has no autograd work between the
    two library calls, which is where the GPU idled 20-90 us per step.
    -> (loss, image, screenspace_points, visibility_filter, radii, depth), all detached; loss is the unscaled L1."""
    if not STAGED_ONLY:
        screenspace_points = _shared_zeros(tuple(pc._xyz.shape), pc._xyz.device).detach().requires_grad_(True)
        out = _one_call(cam, pc, deform, bg_color, scaling_modifier, stage, shs_grad_hook, screenspace_points, target,
                        step_scale=float(loss_scale))
        if out is not None:
            image, radii, depth, loss, visible = out         # visible = radii > 0, written by the forward itself
            return loss, image, screenspace_points, visible, radii, depth
    with torch.enable_grad():
        loss, image, screenspace_points, vis, radii, depth = render_with_l1(cam, pc, deform, bg_color, target,
                                                                            scaling_modifier, stage, shs_grad_hook)
        (loss * loss_scale if loss_scale != 1.0 else loss).backward()
    return loss.detach(), image.detach(), screenspace_points, vis, radii, depth


def render(cam: Camera, pc: GaussianParams, deform: Optional[nn.Module], bg_color: torch.Tensor,
           scaling_modifier: float = 1.0, stage: str = "fine", debug: bool = False, shs_grad_hook=None,
           regularise: bool = False, max_frame: int = MAX_FRAME, _image_only: bool = False):
    """-> (image, screenspace_points, visibility_filter, radii, depth), as the first five items of the
    reference's 7-tuple (renderer.py:221); with regularise=True the full 7-tuple (..., reg_dict, None): reg_dict holds
    the fine stage's momentum / scales / shs / knn_xyz / knn_long / knn_rot terms (:173-203; empty in the coarse
    stage, :152-153), computed from the deformation at t - dt and t + dt -- ONE batched query of 3 P points (with the centre query) here,
    two calls in the reference -- and the k = min(100, P) nearest neighbours of the deformed positions.  shs_grad_hook(grad), if given, is called during the backward pass as soon
    as dL/d(deformed SH coefficients) exists, i.e. before the deformation backward runs (data-parallel training
    starts the all-reduce of the SH gradients there, parallel.GradBucket.early_reduce)."""
    if torch.is_grad_enabled():
        # a fresh leaf (its .grad is the screen-space gradient the densifier reads) over a shared, read-only block of
        # zeros: the rasterizer never reads or writes the values, so no allocation + fill per view
        screenspace_points = _shared_zeros(tuple(pc._xyz.shape), pc._xyz.device).detach().requires_grad_(True)
    else:                                   # never read without a backward pass: no fill kernel for it
        screenspace_points = torch.empty_like(pc._xyz)
    if not (regularise or debug or STAGED_ONLY):
        # the common case in one library call per direction (render_call.py); same kernels, same results
        out = _one_call(cam, pc, deform, bg_color, scaling_modifier, stage, shs_grad_hook, screenspace_points, None)
        if out is not None:
            image, radii, depth, _ = out
            return image if _image_only else (image, screenspace_points, radii > 0, radii, depth)
    settings = GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=debug)
    rasterizer = GaussianRasterizer(raster_settings=settings)
    # GaussianParams keeps one (P,16,3) tensor; a GaussianModel concatenates f_dc and f_rest (get_features, :143-146)
    shs = pc._features if hasattr(pc, "_features") else pc.get_features
    means3D, opacity, scales, rotations = pc._xyz, pc._opacity, pc._scaling, pc._rotation
    if "coarse" not in stage and deform is not None:
        n = means3D.shape[0]
        if hasattr(cam, "time_force_columns"):
            time, force = cam.time_force_columns(n, means3D.device)
        else:
            time = torch.full((n, 1), cam.time, dtype=torch.float32, device=means3D.device)
            force = torch.full((n, 1), cam.force, dtype=torch.float32, device=means3D.device)
        extra = None
        if regularise:
            # the regularisers need the field at t - dt and t + dt too (reference: two more calls, :173-180): ONE query of
            # 3 P points -- every parameter gradient of the field is then produced once instead of being summed over
            # calls by autograd (one add kernel per parameter and call), and the launches are a third as many
            dt = 1.0 / (max_frame - 1)
            if n <= BATCHED_QUERY_MAX_POINTS:
                three = lambda t: torch.cat([t, t, t], dim=0)
                times = torch.cat([time, time - dt, time + dt], dim=0)
                out3 = deform.forward_dynamic(three(means3D), three(scales), three(rotations), three(opacity), three(shs),
                                              times, three(force), None)
                parts = [o.view(3, n, *o.shape[1:]).unbind(0) for o in out3]
            else:                           # the field's scratch grows with the rows of a call: three calls, as the reference
                calls = [deform.forward_dynamic(means3D, scales, rotations, opacity, shs, t_, force, None)
                         for t_ in (time, time - dt, time + dt)]
                parts = list(zip(*calls))
            means3D, scales, rotations, opacity, shs = (q[0] for q in parts)
            extra = parts
        else:
            means3D, scales, rotations, opacity, shs = deform.forward_dynamic(means3D, scales, rotations, opacity, shs,
                                                                              time, force, None)
        if shs_grad_hook is not None and shs.requires_grad:
            shs.register_hook(shs_grad_hook)
    raw_inputs = (pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features if hasattr(pc, "_features") else None)
    scales, rotations, opacity = activate(scales, rotations, opacity)    # exp / L2-normalise / sigmoid, one kernel
    reg_dict = {}
    if regularise and "coarse" not in stage and deform is not None:
        from .rigidity import temporal_regulation
        xyz0 = raw_inputs[0]
        m3, s3, r3, _, h3 = extra
        reg_dict = temporal_regulation(xyz0, means3D, m3[1], m3[2], scales, torch.exp(s3[1]), torch.exp(s3[2]), shs, h3[1], h3[2],
                                       rotations, r3[1], r3[2])            # torch.exp: pc.scaling_activation, renderer.py:181-182
    image, radii, depth = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=None,
                                     opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None)
    if _image_only:
        return image
    if regularise:
        return image, screenspace_points, radii > 0, radii, depth, reg_dict, None
    return image, screenspace_points, radii > 0, radii, depth


def render_batch(cams, pc: GaussianParams, deform: Optional[nn.Module], bg_color: torch.Tensor,
                 scaling_modifier: float = 1.0, stage: str = "fine", targets=None):
    """Several views of one step on one GPU -> [(image, screenspace_points, visibility_filter, radii, depth), ...], or,
    with `targets` (one ground-truth image per view), [(loss, image, screenspace_points, visibility_filter, radii,
    depth), ...] with every view's L1 loss computed inside its library call as render_with_l1 does.

    The reference renders the views of a batch one after the other (train.py:169-189), each with its own deformation
    query; BASELINE.json's multi-GPU configurations put 2-8 views of a step on every rank.  Here the deformation field
    is queried ONCE for all B views -- B * P rows with a per-row timestamp, which gs4d_deform_forward takes anyway --
    and so is its backward: the weight-gradient products, the plane-gradient pipeline and every launch of the field
    run once per step instead of once per view, and autograd adds up the views' parameter gradients in one reduction
    per tensor.  Each view's slice of the rows then goes through the composite call without a deformation field
    (activations + rasterizer [+ L1] in one library call per direction, binning issued before num_rendered has been
    read: render_call.py); the per-stage path is the fallback.  Results per view equal render()'s (identical images;
    gradients up to the order in which the views are summed)."""
    B = len(cams)
    if B == 0:
        return []
    if "coarse" in stage or deform is None or B == 1:
        if targets is None:
            return [render(c, pc, deform, bg_color, scaling_modifier, stage) for c in cams]
        return [render_with_l1(c, pc, deform, bg_color, t, scaling_modifier, stage) for c, t in zip(cams, targets)]
    shs = pc._features if hasattr(pc, "_features") else pc.get_features
    n, dev = pc._xyz.shape[0], pc._xyz.device
    cols = [c.time_force_columns(n, dev) if hasattr(c, "time_force_columns") else
            (torch.full((n, 1), c.time, dtype=torch.float32, device=dev), torch.full((n, 1), c.force, dtype=torch.float32, device=dev))
            for c in cams]
    times, forces = torch.cat([t for t, _ in cols], dim=0), torch.cat([f for _, f in cols], dim=0)
    tile = lambda t: t.expand(B, *t.shape).reshape(B * n, *t.shape[1:])        # B copies of the rows, one kernel
    m3, sc, rot, op, sh = deform.forward_dynamic(tile(pc._xyz), tile(pc._scaling), tile(pc._rotation), tile(pc._opacity),
                                                 tile(shs), times, forces, None)
    raw_view = [t.view(B, n, *t.shape[1:]).unbind(0) for t in (m3, sc, rot, op, sh)]
    grad = torch.is_grad_enabled()
    new_ssp = lambda: (_shared_zeros(tuple(pc._xyz.shape), dev).detach().requires_grad_(True) if grad else torch.empty_like(pc._xyz))
    out, staged = [], None
    for b, cam in enumerate(cams):
        screenspace_points = new_ssp()
        target = None if targets is None else targets[b]
        res = None
        if not STAGED_ONLY and isinstance(bg_color, torch.Tensor) and sh.dim() == 3 and sh.shape[1] == 16:
            call = RenderCall(cam, bg_color, scaling_modifier, pc.active_sh_degree, False, None, None, None, target, None)
            tensors = (raw_view[0][b], raw_view[1][b], raw_view[2][b], raw_view[3][b], raw_view[4][b])
            if call.applicable(tensors):
                res = render_call(call, tensors[0], tensors[1], tensors[2], tensors[3], tensors[4], screenspace_points)
        if res is not None:
            image, radii, depth, loss = res
        else:
            if staged is None:                              # one batched activation for the views that take the per-stage path
                a_sc, a_rot, a_op = activate(sc, rot, op)
                staged = [t.view(B, n, *t.shape[1:]).unbind(0) for t in (a_sc, a_rot, a_op)]
            settings = GaussianRasterizationSettings(
                image_height=int(cam.image_height), image_width=int(cam.image_width),
                tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
                scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                sh_degree=pc.active_sh_degree, campos=cam.camera_center, prefiltered=False, debug=False)
            image, radii, depth = GaussianRasterizer(raster_settings=settings)(
                means3D=raw_view[0][b], means2D=screenspace_points, shs=raw_view[4][b], colors_precomp=None,
                opacities=staged[2][b], scales=staged[0][b], rotations=staged[1][b], cov3D_precomp=None)
            loss = None
            if target is not None:
                from .fused import l1_loss
                loss = l1_loss(image, target)
        view = (image, screenspace_points, radii > 0, radii, depth)
        out.append(view if targets is None else (loss,) + view)
    return out


@torch.no_grad()
def render_for_image(cam: Camera, pc: GaussianParams, deform, bg_color, scaling_modifier=1.0, stage="fine"):
    return render(cam, pc, deform, bg_color, scaling_modifier, stage, _image_only=True)   # no visibility mask kernel


def render_set_fps(views, pc, deform, bg_color, stage="fine"):
    """The FPS loop of render.py:58-77: frames / (t_end - t_after_first_frame), each frame converted to 8 bit and
    copied to the host as the reference's to8b does (:64).  -> (fps, list of uint8 HxWx3 numpy frames)."""
    import time
    frames, t1 = [], None
    for idx, view in enumerate(views):
        if idx == 0:
            t1 = time.time()          # the reference starts its clock before the first frame's render call...
        img = render_for_image(view, pc, deform, bg_color, stage=stage)
        frames.append((255 * torch.clamp(img, 0, 1)).to(torch.uint8).permute(1, 2, 0).cpu().numpy())
    t2 = time.time()
    return (len(views) - 1) / max(t2 - t1, 1e-9), frames
