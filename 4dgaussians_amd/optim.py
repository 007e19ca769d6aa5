"""Fused AdamW over libgs4d.so (csrc/train_step.hip): the optimizer the reference builds at
scene/gaussian_model.py:195-206 -- torch.optim.AdamW(l, lr=1e-4, eps=1e-8) over eight parameter groups with
their own learning rates -- as ONE kernel launch per step instead of torch's per-group foreach launches.

Drop-in for torch.optim.AdamW where the reference uses it: same constructor arguments, `param_groups`
(so GaussianModel.update_learning_rate, :228-242, keeps working), and the same state layout
(`step`, `exp_avg`, `exp_avg_sq`), so optimizer state_dicts interchange with torch's (checkpoints:
scene/gaussian_model.py:88-128).  amsgrad / maximize / capturable are not supported (the reference does not use them).
There is no CPU path: parameters must live on a HIP device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi


class AdamwTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_size_t), ("lr", C.c_double), ("weight_decay", C.c_double),
                ("bias_correction1", C.c_double), ("bias_correction2", C.c_double)]


def _dense_like(p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """`t` laid out in memory exactly like the dense parameter `p` (same strides), 16-byte aligned."""
    if t.stride() != p.stride() or t.data_ptr() % 16:
        out = torch.empty_like(p)          # preserve_format: p's strides
        out.copy_(t)
        return out
    return t


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("FusedAdamW: amsgrad is not supported")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _cabi.lib()
        keep = []
        by_hyper = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise RuntimeError("FusedAdamW needs float32 parameters on a HIP device; there is no CPU path")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdamW does not support sparse gradients")
                if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)) or p.data_ptr() % 16:
                    raise RuntimeError("FusedAdamW: parameters must be dense (contiguous or channels_last) and 16-byte aligned")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                step = float(st["step"])
                g = _dense_like(p, p.grad)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m.stride() != p.stride() or v.stride() != p.stride():      # e.g. a loaded state_dict
                    m, v = _dense_like(p, m), _dense_like(p, v)
                    st["exp_avg"], st["exp_avg_sq"] = m, v
                keep.append(g)
                t = AdamwTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(group["lr"]),
                                float(group["weight_decay"]), 1.0 - beta1 ** step, 1.0 - beta2 ** step)
                by_hyper.setdefault((p.device, float(beta1), float(beta2), float(group["eps"])), []).append(t)
        for (dev, beta1, beta2, eps), ts in by_hyper.items():
            arr = (AdamwTensor * len(ts))(*ts)
            with torch.cuda.device(dev):
                rc = L.gs4d_adamw_step(C.c_int(len(ts)), arr, C.c_double(beta1), C.c_double(beta2), C.c_double(eps),
                                       _cabi.current_stream(dev))
            _cabi.check(rc, "gs4d_adamw_step")
        return loss
