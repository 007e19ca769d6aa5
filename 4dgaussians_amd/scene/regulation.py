"""HexPlane grid regularisers over libgs4d.so (csrc/train_step.hip): the counterpart of
scene/regulation.py:22-29 (compute_plane_smoothness) and of GaussianModel._plane_regulation /
_time_regulation / _l1_regulation / _l2_regulation / compute_regulation (scene/gaussian_model.py:632-706).

The reference evaluates up to 4 x 6 separate torch expressions (slices, squares, means) per iteration; here the
whole weighted sum is one forward launch pair and one backward launch, reading the planes in place
(channels-last).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import torch

from .. import _cabi

SMOOTHNESS, L1, L2 = 0, 1, 2
MAX_TERMS = 32                                       # terms per gs4d_plane_reg_* call (REG_MAX_TERMS, csrc/train_step.hip)
PLANE_GRIDS, TIME_GRIDS = (0, 1, 3), (2, 4, 5)      # scene/gaussian_model.py: self.plane_grids / self.time_grids


class PlaneRegTerm(C.Structure):
    _fields_ = [("plane", C.c_void_p), ("grad", C.c_void_p), ("C", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("kind", C.c_int), ("weight", C.c_float)]


_TERM_CACHE = {}


class _PlaneReg(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kinds_weights, *planes):
        """kinds_weights: tuple of (plane index, kind, weight); planes: the [1,C,H,W] parameters."""
        L = _cabi.lib()
        dev = planes[0].device
        if dev.type != "cuda":
            raise RuntimeError("the plane regularisers run on a HIP device only; there is no CPU path")
        mem = [p.detach().contiguous(memory_format=torch.channels_last) for p in planes]
        # the term table only depends on where the planes live and on the weights: filled once, not on every step (24 terms
        # x 7 ctypes fields, twice per iteration, were a quarter of a millisecond of host time)
        key = (kinds_weights, tuple(p.data_ptr() for p in mem), tuple(p.shape for p in mem))
        terms = _TERM_CACHE.get(key)
        if terms is None:
            terms = (PlaneRegTerm * len(kinds_weights))()
            for t, (i, kind, w) in zip(terms, kinds_weights):
                p = mem[i]
                t.plane, t.grad, t.C, t.H, t.W, t.kind, t.weight = p.data_ptr(), None, p.shape[1], p.shape[2], p.shape[3], kind, w
            if len(_TERM_CACHE) >= 8:
                _TERM_CACHE.clear()
            _TERM_CACHE[key] = terms
        nscratch = int(L.gs4d_plane_reg_scratch_bytes()) // 4
        buf = torch.empty(nscratch, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)       # its own tensor: callers do `loss += reg`
        with torch.cuda.device(dev):
            rc = L.gs4d_plane_reg_forward(C.c_int(len(kinds_weights)), terms, _cabi.ptr(buf), _cabi.ptr(loss),
                                          _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_plane_reg_forward")
        ctx.kinds_weights, ctx.terms = kinds_weights, terms
        ctx.save_for_backward(*mem)
        return loss

    @staticmethod
    def backward(ctx, g):
        L = _cabi.lib()
        mem = ctx.saved_tensors
        dev = mem[0].device
        used = sorted({i for i, _, _ in ctx.kinds_weights})
        shapes = [(1, mem[i].shape[2], mem[i].shape[3], mem[i].shape[1]) for i in used]      # NHWC memory
        flat = _cabi.zeros_group(dev, *shapes)
        grads = {i: t for i, t in zip(used, flat)}
        terms = (PlaneRegTerm * len(ctx.kinds_weights))()
        C.memmove(terms, ctx.terms, C.sizeof(terms))                  # a private copy of the forward's table + the gradient pointers
        gptr = {i: t.data_ptr() for i, t in grads.items()}
        for t, (i, _, _) in zip(terms, ctx.kinds_weights):
            t.grad = gptr[i]
        g_ = g.reshape(1).to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            rc = L.gs4d_plane_reg_backward(C.c_int(len(ctx.kinds_weights)), terms, _cabi.ptr(g_), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_plane_reg_backward")
        out = [None] * len(mem)
        for i in used:
            out[i] = grads[i].permute(0, 3, 1, 2)
        return (None, *out)


def plane_regulation(planes, kinds_weights):
    """sum over (plane index, kind, weight) of weight * reg_kind(planes[index]) as one differentiable scalar."""
    kinds_weights = tuple((int(i), int(k), float(w)) for i, k, w in kinds_weights)
    if not kinds_weights:
        return torch.zeros((), device=planes[0].device)
    # one call takes at most MAX_TERMS terms: 3 multires levels with all four weights on are 36, 4 levels 48
    total = None
    for lo in range(0, len(kinds_weights), MAX_TERMS):
        part = _PlaneReg.apply(kinds_weights[lo:lo + MAX_TERMS], *planes)
        total = part if total is None else total + part
    return total


_EVERY_CACHE = {}


def _every_terms(index, args, time_smoothness_weight, l1_time_planes_weight, l2_time_planes_weight, plane_tv_weight):
    def terms(grid_ids, kind, weight):
        return [(index[(lv, gi)], int(kind), float(weight)) for (lv, gi) in index if gi in grid_ids]
    every = terms(PLANE_GRIDS, SMOOTHNESS, plane_tv_weight) if plane_tv_weight > 1e-7 else []
    if args.use_time:
        every += terms(TIME_GRIDS, SMOOTHNESS, time_smoothness_weight) if time_smoothness_weight > 1e-7 else []
        every += terms(TIME_GRIDS, L1, l1_time_planes_weight) if l1_time_planes_weight > 1e-7 else []
        every += terms(TIME_GRIDS, L2, l2_time_planes_weight) if l2_time_planes_weight > 1e-7 else []
    return tuple(every)


def _plane_regulation_checked(planes, kinds_weights):
    """plane_regulation for a term tuple that already holds (int, int, float) entries."""
    total = None
    for lo in range(0, len(kinds_weights), MAX_TERMS):
        part = _PlaneReg.apply(kinds_weights[lo:lo + MAX_TERMS], *planes)
        total = part if total is None else total + part
    return total


def compute_plane_smoothness(t):
    """scene/regulation.py:22-29."""
    return plane_regulation([t], [(0, SMOOTHNESS, 1.0)])


def compute_regulation(deformation, time_smoothness_weight, l1_time_planes_weight, l2_time_planes_weight,
                       plane_tv_weight, force_weight=0.0, detail=True):
    """GaussianModel.compute_regulation (scene/gaussian_model.py:679-706) for a deform_network: returns the
    reference's dict; `total_reg` is differentiable, the per-kind entries are detached values.  The reference
    gates every term on `weight > 1e-7` and the time terms on args.use_time; the force term only exists for
    blend_time_force = False, which this build (like the reference) does not run.
    detail=False skips the four per-kind logging values (each is one more forward launch pair)."""
    net = deformation.deformation_net
    args = net.args
    planes, index = [], {}
    for lv, grids in enumerate(net.grid.grids):
        if len(grids) == 3:
            continue
        for gi, p in enumerate(grids):
            index[(lv, gi)] = len(planes)
            planes.append(p)
    if not detail:                         # the training loop's call: the same term list on every iteration
        key = (tuple(index), bool(args.use_time), float(time_smoothness_weight), float(l1_time_planes_weight),
               float(l2_time_planes_weight), float(plane_tv_weight))
        every = _EVERY_CACHE.get(key)
        if every is None:
            every = _every_terms(index, args, time_smoothness_weight, l1_time_planes_weight, l2_time_planes_weight, plane_tv_weight)
            _EVERY_CACHE[key] = every
        out = defaultdict(lambda: torch.tensor(0))
        out["total_reg"] = _plane_regulation_checked(planes, every) if every else torch.tensor(0)
        return out

    def terms(grid_ids, kind, weight):
        return [(index[(lv, gi)], kind, weight) for (lv, gi) in index if gi in grid_ids]

    groups = {
        "plane_reg": terms(PLANE_GRIDS, SMOOTHNESS, plane_tv_weight) if plane_tv_weight > 1e-7 else [],
        "time_reg": terms(TIME_GRIDS, SMOOTHNESS, time_smoothness_weight) if args.use_time and time_smoothness_weight > 1e-7 else [],
        "l1_time_reg": terms(TIME_GRIDS, L1, l1_time_planes_weight) if args.use_time and l1_time_planes_weight > 1e-7 else [],
        "l2_time_reg": terms(TIME_GRIDS, L2, l2_time_planes_weight) if args.use_time and l2_time_planes_weight > 1e-7 else [],
    }
    out = defaultdict(lambda: torch.tensor(0))
    every = [t for g in groups.values() for t in g]
    out["total_reg"] = plane_regulation(planes, every) if every else torch.tensor(0)
    with torch.no_grad():
        for name, g in groups.items():
            out[name] = plane_regulation(planes, g).detach() if (g and detail) else torch.tensor(0)
    out["force_reg"] = torch.tensor(0)
    return out
