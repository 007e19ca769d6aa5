"""The hot path of one view as ONE autograd node over gs4d_render_forward / gs4d_render_backward (include/gs4d.h).

What the reference composes in Python -- deformation query, the three activations, the rasterizer call
(gaussian_renderer/renderer.py:123-147,207-216) and, optionally, the L1 loss (train.py:203) -- is issued by one C call
per direction here; `renderer.render()` takes this path whenever it applies (no regularisers, float32 HIP tensors) and
keeps the staged path (three autograd functions over the per-stage entry points) for everything else.  The device work
is the same kernels in the same order, so both paths give bit-identical results (tests/test_gpu_render_call.py).

Host plumbing only; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _cabi
from .scene.deformation import DeformDesc, DeformGrads

FP = C.c_void_p
FORWARD_ONLY = 2                       # GS4D_FORWARD_ONLY
LOSS_AFTER_BACKWARD = 4                # GS4D_LOSS_AFTER_BACKWARD: the backward call finishes the loss (render_step)
STAGE_RASTER_DONE, STAGE_DATA_DONE = 1, 2


class RenderDesc(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("scale_modifier", C.c_float), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("prefiltered", C.c_int), ("flags", C.c_int),
                ("background", FP), ("viewmatrix", FP), ("projmatrix", FP), ("cam_pos", FP),
                ("xyz", FP), ("log_scales", FP), ("rotations", FP), ("opacity_logit", FP), ("shs", FP),
                ("deform", C.POINTER(DeformDesc)), ("time", FP), ("force", FP), ("target", FP), ("l1_grad_scale", FP), ("visible", FP)]


class RenderState(C.Structure):
    _fields_ = [("arena", FP), ("arena_bytes", C.c_size_t), ("binning", FP), ("binning_bytes", C.c_size_t),
                ("num_rendered", C.c_int), ("layout_rendered", C.c_int), ("deferred_loss", FP)]


class RenderGrads(C.Structure):
    _fields_ = [("dL_dmeans2D", FP), ("d_xyz", FP), ("d_log_scales", FP), ("d_rotations", FP), ("d_opacity_logit", FP),
                ("d_shs", FP), ("deform", C.POINTER(DeformGrads))]


STAGE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
_NO_STAGE_FN = C.cast(None, STAGE_FN)


def _usable(t) -> bool:
    return t.dtype == torch.float32 and t.device.type == "cuda" and t.is_contiguous() and t.data_ptr() % 16 == 0


class RenderCall:
    """Everything of one render() call that is not a differentiable tensor: the camera, the per-point time / force
    columns, the optional L1 target and the data-parallel hooks."""
    __slots__ = ("H", "W", "tanfovx", "tanfovy", "bg", "view", "proj", "campos", "scale_modifier", "sh_degree", "debug",
                 "time", "force", "target", "deform_net", "shs_grad_hook", "params", "cfg")

    def __init__(self, cam, bg, scale_modifier, sh_degree, debug, deform_net, time, force, target, shs_grad_hook):
        self.H, self.W = int(cam.image_height), int(cam.image_width)
        self.tanfovx, self.tanfovy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
        self.bg, self.view, self.proj, self.campos = bg, cam.world_view_transform, cam.full_proj_transform, cam.camera_center
        self.scale_modifier, self.sh_degree, self.debug = float(scale_modifier), int(sh_degree), bool(debug)
        self.deform_net, self.time, self.force, self.target, self.shs_grad_hook = deform_net, time, force, target, shs_grad_hook
        self.params, self.cfg = deform_net.cached_plan() if deform_net is not None else ([], None)

    def applicable(self, tensors) -> bool:
        if not all(_usable(t) for t in tensors):
            return False
        if not all(_usable(t) for t in (self.bg, self.view, self.proj, self.campos)):
            return False
        if self.target is not None and not _usable(self.target):
            return False
        if self.deform_net is not None:
            plan = self.cfg["plan"]
            if not plan.in_place or not (_usable(self.time) and _usable(self.force)):
                return False
        return True


def _fill_desc(d: RenderDesc, call: RenderCall, P, M, xyz, scaling, rotation, opacity, features, flags):
    d.P, d.D, d.M, d.width, d.height = P, call.sh_degree, M, call.W, call.H
    d.scale_modifier, d.tan_fovx, d.tan_fovy, d.prefiltered, d.flags = call.scale_modifier, call.tanfovx, call.tanfovy, 0, flags
    d.background, d.viewmatrix, d.projmatrix, d.cam_pos = (call.bg.data_ptr(), call.view.data_ptr(), call.proj.data_ptr(),
                                                          call.campos.data_ptr())
    d.xyz, d.log_scales, d.rotations, d.opacity_logit, d.shs = (xyz.data_ptr(), scaling.data_ptr(), rotation.data_ptr(),
                                                                opacity.data_ptr(), features.data_ptr())
    if call.deform_net is not None:
        d.deform = C.pointer(call.cfg["plan"].desc)
        d.time, d.force = call.time.data_ptr(), call.force.data_ptr()
    else:
        d.deform = None
        d.time = d.force = None
    d.target = call.target.data_ptr() if call.target is not None else None


def _issue_forward(call: RenderCall, xyz, scaling, rotation, opacity, features, needs_grad: bool, l1_grad_scale=None,
                   visible=None):
    """gs4d_render_forward -> (color, radii, depth, loss or None, descriptor, state, arena).  l1_grad_scale (a device
    scalar, render_step): the forward's loss launch also leaves d(scale * L1)/dcolor for the backward."""
    L = _cabi.lib()
    dev = xyz.device
    P, M = xyz.shape[0], features.shape[1]
    flags = (1 if call.debug else 0) | (0 if needs_grad else FORWARD_ONLY)
    if l1_grad_scale is not None and call.target is not None and needs_grad:
        flags |= LOSS_AFTER_BACKWARD   # render_step: the backward follows at once
    d = RenderDesc()
    if call.deform_net is not None:
        call.cfg["plan"].desc.no_grad = 0 if needs_grad else 1
    _fill_desc(d, call, P, M, xyz, scaling, rotation, opacity, features, flags)
    d.l1_grad_scale = l1_grad_scale.data_ptr() if (l1_grad_scale is not None and call.target is not None) else None
    d.visible = visible.data_ptr() if visible is not None else None      # bool [P]: radii > 0, written by the preprocess launch
    color = torch.empty((3, call.H, call.W), dtype=torch.float32, device=dev)
    depth = torch.empty((1, call.H, call.W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev) if call.target is not None else None
    state = RenderState()
    arena = _cabi.ArenaAllocator(dev)
    with torch.cuda.device(dev):
        rc = L.gs4d_render_forward(C.byref(d), arena.callback, arena.ctx, C.c_void_p(color.data_ptr()),
                                   C.c_void_p(depth.data_ptr()), C.c_void_p(radii.data_ptr()),
                                   C.c_void_p(loss.data_ptr()) if loss is not None else None, C.byref(state),
                                   _cabi.current_stream(dev))
    arena.close()
    _cabi.check(rc, "gs4d_render_forward")
    return color, radii, depth, loss, d, state, arena


class _BackwardBuffers:
    """Everything gs4d_render_backward writes, allocated BEFORE the forward call is issued when the caller knows that a
    backward will follow (render_step): the host then has nothing to do between the two library calls."""
    __slots__ = ("d_xyz", "d_sc", "d_rot", "d_op", "d_shs", "d_m2d", "g", "dg", "views", "order_key")

    def __init__(self, call: RenderCall, dev, P, M, opacity_shape):
        self.d_xyz, self.d_sc, self.d_rot, self.d_op, self.d_shs = _cabi.empty_group(dev, (P, 3), (P, 3), (P, 4), tuple(opacity_shape),
                                                                                    (P, M, 3))
        self.d_m2d = torch.empty((P, 3), dtype=torch.float32, device=dev)
        g = self.g = RenderGrads()
        g.dL_dmeans2D, g.d_xyz, g.d_log_scales, g.d_rotations = (self.d_m2d.data_ptr(), self.d_xyz.data_ptr(), self.d_sc.data_ptr(),
                                                                 self.d_rot.data_ptr())
        g.d_opacity_logit, g.d_shs = self.d_op.data_ptr(), self.d_shs.data_ptr()
        self.dg, self.views, self.order_key = None, (), None
        if call.deform_net is not None:
            plan, cfg = call.cfg["plan"], call.cfg
            self.dg, self.views = plan.new_grads(dev)
            oc = cfg.get("hex_order_cache")
            if oc is not None:
                self.order_key = oc.bind(self.dg, P, int(plan.desc.L), dev)
            else:
                self.dg.hex_order, self.dg.hex_order_valid = None, 0
            g.deform = C.pointer(self.dg)

    def param_grads(self, call: RenderCall):
        if call.deform_net is None:
            return ()
        n_pl = call.cfg["plan"].n_planes
        return tuple(self.views[:n_pl]) + (None,) + tuple(self.views[n_pl:])


def _issue_backward(call: RenderCall, d: RenderDesc, state: RenderState, color, radii, g_color, g_depth, g_loss, buf: _BackwardBuffers):
    """gs4d_render_backward into `buf` (the data-parallel hooks of the call fire from inside it)."""
    L = _cabi.lib()
    dev = color.device
    stage_fn = _NO_STAGE_FN
    data_hook = call.cfg.get("data_grad_hook") if call.deform_net is not None else None
    if call.shs_grad_hook is not None or data_hook is not None:
        failure = []

        def on_stage(_ctx, stage):
            try:                                   # never let an exception cross the C boundary
                if stage == STAGE_RASTER_DONE and call.shs_grad_hook is not None:
                    call.shs_grad_hook(buf.d_shs)
                elif stage == STAGE_DATA_DONE and data_hook is not None:
                    n_pl = call.cfg["plan"].n_planes
                    data_hook(buf.d_xyz, buf.d_sc, buf.d_rot, list(buf.views[:n_pl]), buf.views[n_pl], buf.views[n_pl + 1])
            except Exception as ex:
                failure.append(ex)

        stage_fn = STAGE_FN(on_stage)
    scratch = _cabi.ArenaAllocator(dev)
    with torch.cuda.device(dev):
        rc = L.gs4d_render_backward(C.byref(d), C.byref(state), C.c_void_p(color.data_ptr()), C.c_void_p(radii.data_ptr()),
                                    _cabi.ptr(g_color), _cabi.ptr(g_depth), _cabi.ptr(g_loss), scratch.callback, scratch.ctx,
                                    C.byref(buf.g), stage_fn, None, _cabi.current_stream(dev))
    scratch.close()
    _cabi.check(rc, "gs4d_render_backward")
    if buf.order_key is not None:
        call.cfg["hex_order_cache"].recorded(buf.order_key)
    if stage_fn is not _NO_STAGE_FN and failure:
        raise failure[0]


class _RenderFunction(torch.autograd.Function):
    """(call, xyz, log_scales, rotations, opacity_logit, shs, means2D, *deformation parameters) ->
    (color [3,H,W], radii [P], depth [1,H,W], loss or None)."""

    @staticmethod
    def forward(ctx, call, xyz, scaling, rotation, opacity, features, means2D, *params):
        needs_grad = any(ctx.needs_input_grad)
        color, radii, depth, loss, d, state, arena = _issue_forward(call, xyz, scaling, rotation, opacity, features, needs_grad)
        ctx.num_rendered = state.num_rendered
        if needs_grad:
            ctx.call, ctx.desc, ctx.state = call, d, state
            ctx.versions = [p._version for p in params]
            ctx.set_materialize_grads(False)
            # first block = activation arena, last block = the binning chunk in use
            ctx.save_for_backward(xyz, scaling, rotation, opacity, features, color, radii, arena.tensors[0], arena.tensors[-1])
            ctx.params = params
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, loss

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_loss):
        L = _cabi.lib()
        call, d, state = ctx.call, ctx.desc, ctx.state
        xyz, scaling, rotation, opacity, features, color, radii, arena_t, binning_t = ctx.saved_tensors
        params = ctx.params
        n_extra = len(params)
        if g_color is None and g_loss is None and g_depth is None:
            return (None,) * (7 + n_extra)
        dev = xyz.device
        P, M = xyz.shape[0], features.shape[1]
        if [p._version for p in params] != ctx.versions:
            raise RuntimeError("a deformation parameter was modified in place between forward and backward")
        # the descriptor may have been re-used by a later forward of the same plan: restore what this backward reads
        _fill_desc(d, call, P, M, xyz, scaling, rotation, opacity, features, d.flags)
        if call.deform_net is not None:
            call.cfg["plan"].desc.no_grad = 0
        if g_color is not None and g_loss is not None:
            # the image feeds the fused loss AND something else: add the loss's share to the explicit gradient
            extra = torch.empty_like(color)
            _cabi.check(L.gs4d_l1_loss_backward(C.c_size_t(color.numel()), C.c_void_p(color.data_ptr()),
                                                C.c_void_p(call.target.data_ptr()), C.c_void_p(g_loss.contiguous().data_ptr()),
                                                C.c_void_p(extra.data_ptr()), _cabi.current_stream(dev)), "gs4d_l1_loss_backward")
            g_color, g_loss = g_color + extra, None
        if g_color is None and g_loss is None:
            g_color = torch.zeros_like(color)
        keep = []
        def f32c(t):
            if t is None:
                return None
            t = t if (t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0) else t.to(torch.float32).contiguous().clone()
            keep.append(t)
            return t
        g_color, g_depth, g_loss = f32c(g_color), f32c(g_depth), f32c(g_loss)
        buf = _BackwardBuffers(call, dev, P, M, opacity.shape)
        _issue_backward(call, d, state, color, radii, g_color, g_depth, g_loss, buf)
        return (None, buf.d_xyz, buf.d_sc, buf.d_rot, buf.d_op, buf.d_shs, buf.d_m2d) + buf.param_grads(call)


_SCALARS = {}


def _device_scalar(dev, value: float):
    key = (dev.index, float(value))
    t = _SCALARS.get(key)
    if t is None:
        if len(_SCALARS) > 64:
            _SCALARS.clear()
        t = _SCALARS[key] = torch.full((), float(value), dtype=torch.float32, device=dev)
    return t


def render_step(call: RenderCall, xyz, scaling, rotation, opacity, features, means2D, loss_scale: float = 1.0):
    """Forward, fused L1 and backward of one view issued back to back, without an autograd graph in between:
    d(loss_scale * L1) is ACCUMULATED into .grad of every tensor that requires it (the five Gaussian parameter tensors,
    means2D, the deformation parameters), exactly what `(loss_scale * loss).backward()` leaves there after
    render_call() -- same two library calls, same kernels, bit-identical gradients (tests/test_gpu_render_call.py).

    Why it exists: the forward call returns right after the host has read num_rendered, with ~0.2 ms of device work
    (binning, blend, loss) queued; autograd needs 0.15-0.3 ms of host time between that return and the backward call's
    first launch, so the GPU idles for 20-90 us per step at C2 (rocprofv3 kernel trace).  Here the
    gradient buffers exist before the forward call and the backward call follows it directly.
    -> (color, radii, depth, loss, visible); all detached, loss is the plain (unscaled) L1, visible = radii > 0 (written
    by the forward itself)."""
    assert call.target is not None
    dev = xyz.device
    P, M = xyz.shape[0], features.shape[1]
    with torch.no_grad():
        buf = _BackwardBuffers(call, dev, P, M, opacity.shape)
        g_loss = _device_scalar(dev, loss_scale)
        visible = torch.empty((P,), dtype=torch.bool, device=dev)
        color, radii, depth, loss, d, state, arena = _issue_forward(call, xyz, scaling, rotation, opacity, features, True, g_loss,
                                                                    visible)
        _issue_backward(call, d, state, color, radii, None, None, g_loss, buf)
        grads = (buf.d_xyz, buf.d_sc, buf.d_rot, buf.d_op, buf.d_shs, buf.d_m2d) + buf.param_grads(call)
        for p, g in zip((xyz, scaling, rotation, opacity, features, means2D) + tuple(call.params), grads):
            if g is None or not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g.view(p.shape) if g.shape != p.shape else g
            else:
                p.grad.add_(g.view(p.shape))
    return color, radii, depth, loss, visible


def render_call(call: RenderCall, xyz, scaling, rotation, opacity, features, means2D):
    """-> (color, radii, depth, loss or None).  The caller has checked call.applicable(...)."""
    return _RenderFunction.apply(call, xyz, scaling, rotation, opacity, features, means2D, *call.params)
