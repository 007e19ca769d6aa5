"""Fused hot-path elementwise operators over libgs4d.so (csrc/elementwise.hip):

  activate(log_scales, rotations, opacity_logit) -> (scales, rotations_n, opacity)
      the three activations of scene/gaussian_model.py:38-43 as applied in
      gaussian_renderer/renderer.py:170-172, one kernel per direction;
  l1_loss(network_output, gt)
      utils/loss_utils.py:20-21, a deterministic two-kernel reduction forward and one kernel backward.

Host plumbing only; there is no CPU path (a missing library raises).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected a float32 tensor, got {t.dtype}")
    if t.device.type != "cuda":
        raise RuntimeError("fused operators need tensors on a HIP device; there is no CPU path")
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_scales, rotations, opacity_logit):
        L = _cabi.lib()
        s_, r_, o_ = _f32c(log_scales.detach()), _f32c(rotations.detach()), _f32c(opacity_logit.detach())
        P = s_.shape[0]
        if s_.shape != (P, 3) or r_.shape != (P, 4) or o_.numel() != P:
            raise RuntimeError("activate expects log_scales [P,3], rotations [P,4], opacity [P,1]")
        dev = s_.device
        # outputs escape to the caller: separate allocations (views of one block would make autograd refuse in-place ops)
        scales, rot_n, opacity = (torch.empty(shp, dtype=torch.float32, device=dev) for shp in ((P, 3), (P, 4), tuple(o_.shape)))
        with torch.cuda.device(dev):
            rc = L.gs4d_activate_forward(C.c_int(P), _cabi.ptr(s_), _cabi.ptr(r_), _cabi.ptr(o_), _cabi.ptr(scales),
                                         _cabi.ptr(rot_n), _cabi.ptr(opacity), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_activate_forward")
        ctx.save_for_backward(scales, r_, opacity)
        return scales, rot_n, opacity

    @staticmethod
    def backward(ctx, g_scales, g_rot, g_opacity):
        L = _cabi.lib()
        scales, r_, opacity = ctx.saved_tensors
        P, dev = scales.shape[0], scales.device
        gs, gr, go = (None if g is None else _f32c(g) for g in (g_scales, g_rot, g_opacity))
        d_s, d_r, d_o = _cabi.empty_group(dev, (P, 3), (P, 4), tuple(opacity.shape))
        with torch.cuda.device(dev):
            rc = L.gs4d_activate_backward(C.c_int(P), _cabi.ptr(scales), _cabi.ptr(r_), _cabi.ptr(opacity), _cabi.ptr(gs),
                                          _cabi.ptr(gr), _cabi.ptr(go), _cabi.ptr(d_s), _cabi.ptr(d_r), _cabi.ptr(d_o),
                                          _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_activate_backward")
        return d_s, d_r, d_o


def activate(log_scales, rotations, opacity_logit):
    """-> (exp(log_scales), F.normalize(rotations), sigmoid(opacity_logit))"""
    return _Activate.apply(log_scales, rotations, opacity_logit)


class _L1Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        L = _cabi.lib()
        if a.shape != b.shape:
            raise RuntimeError(f"l1_loss: shapes differ, {tuple(a.shape)} vs {tuple(b.shape)}")
        a_, b_ = _f32c(a.detach()), _f32c(b.detach())
        dev = a_.device
        n = a_.numel()
        nscratch = int(L.gs4d_l1_loss_scratch_bytes()) // 4
        buf = torch.empty(nscratch, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)      # its own tensor: callers do `loss += reg`
        with torch.cuda.device(dev):
            rc = L.gs4d_l1_loss_forward(C.c_size_t(n), _cabi.ptr(a_), _cabi.ptr(b_), _cabi.ptr(buf),
                                        _cabi.ptr(loss), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_l1_loss_forward")
        ctx.save_for_backward(a_, b_)
        return loss

    @staticmethod
    def backward(ctx, g):
        L = _cabi.lib()
        a_, b_ = ctx.saved_tensors
        dev = a_.device
        g_ = _f32c(g.reshape(1))
        d_a = torch.empty_like(a_)
        with torch.cuda.device(dev):
            rc = L.gs4d_l1_loss_backward(C.c_size_t(a_.numel()), _cabi.ptr(a_), _cabi.ptr(b_), _cabi.ptr(g_),
                                         _cabi.ptr(d_a), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_l1_loss_backward")
        return d_a, None


def l1_loss(network_output, gt):
    """mean(|network_output - gt|); the gradient flows to network_output only (gt is data)."""
    return _L1Loss.apply(network_output, gt)
