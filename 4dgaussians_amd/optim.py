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

import numpy as np
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


# AdamwTensor as a numpy record: the per-step table is filled column-wise instead of one ctypes struct per parameter
_REC = np.dtype([("param", "u8"), ("grad", "u8"), ("exp_avg", "u8"), ("exp_avg_sq", "u8"), ("numel", "u8"),
                 ("lr", "f8"), ("weight_decay", "f8"), ("bias_correction1", "f8"), ("bias_correction2", "f8")])
assert _REC.itemsize == C.sizeof(AdamwTensor)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("FusedAdamW: amsgrad is not supported")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._sig, self._plan = None, None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._sig, self._plan = None, None                 # new state tensors, new step counts

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._sig = None

    def _build_plan(self, active):
        """The step's launch table for the parameters that have a gradient: everything that does not change from step to
        step (validation, state creation, layout fixes, the batches of equal betas / eps) happens here, once per set of
        parameters -- it cost 0.8 ms of host time per step for the 60 tensors of a model, most of a C2 iteration."""
        batches = {}
        for gi, p in active:
            group = self.param_groups[gi]
            if p.device.type != "cuda" or p.dtype != torch.float32:
                raise RuntimeError("FusedAdamW needs float32 parameters on a HIP device; there is no CPU path")
            if not (p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)) or p.data_ptr() % 16:
                raise RuntimeError("FusedAdamW: parameters must be dense (contiguous or channels_last) and 16-byte aligned")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if m.stride() != p.stride() or v.stride() != p.stride() or m.data_ptr() % 16 or v.data_ptr() % 16:   # e.g. a loaded state_dict
                st["exp_avg"], st["exp_avg_sq"] = _dense_like(p, m), _dense_like(p, v)
            beta1, beta2 = group["betas"]
            batches.setdefault((p.device, float(beta1), float(beta2), float(group["eps"])), []).append((gi, p, st))
        plan = []
        for key, items in batches.items():
            rec = np.zeros(len(items), dtype=_REC)
            rec["numel"] = [p.numel() for _, p, _ in items]
            # the `step` entries of the torch state layout become zero-dim views of ONE tensor per batch: a single add
            # per step keeps all of them current (an add on each of 60 separate tensors was 70 us of host time)
            counts = [float(st["step"]) for _, _, st in items]
            base = torch.tensor(counts, dtype=torch.float32)
            for i, (_, _, st) in enumerate(items):
                st["step"] = base[i]
            plan.append(dict(key=key, items=items, rec=rec, gidx=np.array([gi for gi, _, _ in items]),
                             steps=np.array(counts, dtype=np.float64), base=base))
        return plan

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _cabi.lib()
        active = [(gi, p) for gi, group in enumerate(self.param_groups) for p in group["params"] if p.grad is not None]
        # the plan batches the parameters by (betas, eps) of their group: a later change to those is a new plan, as
        # torch.optim.AdamW honours it on the next step (lr and weight_decay are re-read every step below)
        hyper = [(float(g["betas"][0]), float(g["betas"][1]), float(g["eps"])) for g in self.param_groups]
        sig = [(id(p), id(self.state[p])) if p in self.state else (id(p), 0) for _, p in active] + hyper
        if sig != self._sig:
            self._plan = self._build_plan(active)
            self._sig = [(id(p), id(self.state[p])) for _, p in active] + hyper
        lrs = np.array([float(g["lr"]) for g in self.param_groups])
        wds = np.array([float(g["weight_decay"]) for g in self.param_groups])
        for b in self._plan:
            dev, beta1, beta2, eps = b["key"]
            rec = b["rec"]
            keep, pp, gp, mp, vp = [], [], [], [], []
            for _, p, st in b["items"]:
                g = p.grad
                if g.stride() != p.stride() or g.data_ptr() % 16 or g.is_sparse:
                    if g.is_sparse:
                        raise RuntimeError("FusedAdamW does not support sparse gradients")
                    g = _dense_like(p, g)
                    keep.append(g)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if m.stride() != p.stride() or v.stride() != p.stride():    # replaced behind the optimizer's back
                    m, v = _dense_like(p, m), _dense_like(p, v)
                    st["exp_avg"], st["exp_avg_sq"] = m, v
                pp.append(p.data_ptr()); gp.append(g.data_ptr()); mp.append(m.data_ptr()); vp.append(v.data_ptr())
            rec["param"], rec["grad"], rec["exp_avg"], rec["exp_avg_sq"] = pp, gp, mp, vp
            rec["lr"], rec["weight_decay"] = lrs[b["gidx"]], wds[b["gidx"]]
            steps = b["steps"] + 1.0            # committed (with the state's step tensors) only once the launch succeeded
            if steps[0] == steps[-1] and (steps == steps[0]).all():         # one value unless parameters joined later
                rec["bias_correction1"] = 1.0 - beta1 ** float(steps[0])
                rec["bias_correction2"] = 1.0 - beta2 ** float(steps[0])
            else:
                vals, inv = np.unique(steps, return_inverse=True)
                rec["bias_correction1"] = np.array([1.0 - beta1 ** float(x) for x in vals])[inv]
                rec["bias_correction2"] = np.array([1.0 - beta2 ** float(x) for x in vals])[inv]
            with torch.cuda.device(dev):
                rc = L.gs4d_adamw_step(C.c_int(len(rec)), C.c_void_p(rec.ctypes.data), C.c_double(beta1), C.c_double(beta2),
                                       C.c_double(eps), _cabi.current_stream(dev))
            _cabi.check(rc, "gs4d_adamw_step")
            b["steps"] = steps
            b["base"] += 1.0
            del keep
        return loss
