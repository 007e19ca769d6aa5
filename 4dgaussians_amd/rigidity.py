"""The k-nearest-neighbour rigidity / momentum regularisers of the fine stage:
gaussian_renderer/renderer.py:151-203 of the reference, computed from the three deformation queries of a view
(t - dt, t, t + dt) and the k = min(100, n) nearest neighbours of the deformed positions.

`neighbor_diff_norm(x, idx)` is the reference's `torch.norm(x[idx] - x[None, :, None], p=2, dim=-1)` as one kernel per
direction (csrc/rigidity.hip: no (n, k, dim) intermediates); `temporal_regulation` / `recurrent_regulation` assemble
the dictionaries of the two branches with the reference's expressions, term for term.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi
from .pointops import knn, knn_self_i32


def transpose_graph(idx: torch.Tensor, trusted: bool = False):
    """idx (..., n, k) -> (order int32 [n*k], row_start int32 [n+1]): the pairs i*k+s sorted by their target idx[i][s]
    and where each target's pairs begin.  Built once per neighbour graph (gs4d_transpose_graph: a radix sort over the
    bits the targets have; shared by every term of a view); it turns the backward's scatter into a gather.  trusted=True skips the range check of the indices (two
    reductions and two host synchronisations per view): for lists that come straight out of this package's knn()."""
    n, k = idx.shape[-2], idx.shape[-1]
    if n * k >= 2 ** 31:
        raise RuntimeError("transpose_graph: n * k must stay below 2^31")
    flat = idx.reshape(-1).to(torch.int32)
    if not trusted and flat.numel() and (int(flat.min()) < 0 or int(flat.max()) >= n):
        raise RuntimeError("transpose_graph: neighbour index out of range")
    flat = flat.contiguous()
    if flat.device.type != "cuda":
        raise RuntimeError("transpose_graph needs the neighbour lists on a HIP device; there is no CPU path")
    L = _cabi.lib()
    R = flat.numel()
    order = torch.empty(R, dtype=torch.int32, device=flat.device)
    row_start = torch.empty(n + 1, dtype=torch.int32, device=flat.device)
    if R == 0:
        return order, row_start.zero_()
    L.gs4d_transpose_graph_scratch_bytes.restype = C.c_size_t
    nbytes = int(L.gs4d_transpose_graph_scratch_bytes(C.c_int(n), C.c_longlong(R)))
    scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device=flat.device)
    sp = (scratch.data_ptr() + 255) & ~255
    with torch.cuda.device(flat.device):
        rc = L.gs4d_transpose_graph(C.c_int(n), C.c_longlong(R), _cabi.ptr(flat), _cabi.ptr(order), _cabi.ptr(row_start),
                                    C.c_void_p(sp), C.c_size_t(nbytes), _cabi.current_stream(flat.device))
    _cabi.check(rc, "transpose_graph")
    return order, row_start


class _NeighborDiffNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, graph_t):
        if x.device.type != "cuda":
            raise RuntimeError("neighbor_diff_norm runs on a HIP device only; there is no CPU path")
        n, dim = x.shape
        k = idx.shape[-1]
        x_ = x.detach().to(torch.float32).contiguous()
        idx_ = idx.reshape(n, k).to(torch.int64).contiguous()
        d = torch.empty((n, k), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = _cabi.lib().gs4d_neighbor_diff_norm_forward(C.c_int(n), C.c_int(k), C.c_int(dim), _cabi.ptr(x_), _cabi.ptr(idx_),
                                                             _cabi.ptr(d), _cabi.current_stream(x.device))
        _cabi.check(rc, "gs4d_neighbor_diff_norm_forward")
        ctx.save_for_backward(x_, idx_, d)
        ctx.graph_t = graph_t
        return d

    @staticmethod
    def backward(ctx, g):
        x_, idx_, d = ctx.saved_tensors
        n, dim = x_.shape
        k = idx_.shape[1]
        g_ = g.to(torch.float32).contiguous()
        dx = torch.zeros_like(x_)
        L = _cabi.lib()
        with torch.cuda.device(x_.device):
            if ctx.graph_t is None:                             # float atomics (no transposed graph at hand)
                rc = L.gs4d_neighbor_diff_norm_backward(C.c_int(n), C.c_int(k), C.c_int(dim), _cabi.ptr(x_), _cabi.ptr(idx_),
                                                        _cabi.ptr(d), _cabi.ptr(g_), _cabi.ptr(dx),
                                                        _cabi.current_stream(x_.device))
            else:
                order, row_start = ctx.graph_t
                coef = torch.empty_like(d)
                rc = L.gs4d_neighbor_diff_norm_backward_gather(
                    C.c_int(n), C.c_int(k), C.c_int(dim), _cabi.ptr(x_), _cabi.ptr(idx_), _cabi.ptr(d), _cabi.ptr(g_),
                    _cabi.ptr(order), _cabi.ptr(row_start), _cabi.ptr(coef), _cabi.ptr(dx), _cabi.current_stream(x_.device))
        _cabi.check(rc, "gs4d_neighbor_diff_norm_backward")
        return dx, None, None


def neighbor_diff_norm(x: torch.Tensor, idx: torch.Tensor, graph_t=None) -> torch.Tensor:
    """x (n, 3 | 4), idx (..., n, k) integer rows of neighbour indices into x -> distances shaped like idx:
    out[..., i, s] = || x[idx[..., i, s]] - x[i] ||_2, differentiable in x (zero distance -> zero gradient).
    graph_t = transpose_graph(idx), when given, makes the backward a deterministic gather instead of float atomics
    (worth building as soon as two terms share the neighbour lists)."""
    if x.dim() != 2 or x.shape[1] not in (3, 4):
        raise RuntimeError("neighbor_diff_norm: x must be (n, 3) or (n, 4)")
    if idx.shape[-2] != x.shape[0] or idx.numel() != x.shape[0] * idx.shape[-1]:
        raise RuntimeError("neighbor_diff_norm: idx must hold one row of neighbours per row of x")
    return _NeighborDiffNorm.apply(x, idx, graph_t).view(idx.shape)


class _RigidityTerms(torch.autograd.Function):
    """(vel_xyz, vel_long, rot_v1, rot_v2, idx, dist, graph_t) -> tensor [3] = (knn_xyz, knn_long, knn_rot) of
    renderer.py:194-203, one kernel forward, one backward (csrc/rigidity.hip); dist carries no gradient."""

    @staticmethod
    def forward(ctx, vel_xyz, vel_long, rot_v1, rot_v2, idx, dist, graph_t):
        dev = vel_xyz.device
        n, k = vel_xyz.shape[0], idx.shape[-1]
        f = lambda t: t.detach().to(torch.float32).contiguous()
        vx, vl, r1, r2 = f(vel_xyz), f(vel_long), f(rot_v1), f(rot_v2)
        if r1.data_ptr() % 16:
            r1 = r1.clone()
        if r2.data_ptr() % 16:
            r2 = r2.clone()
        i32 = idx.dtype == torch.int32                       # the query's own type: no widening (gs4d_rigidity_terms_*_i32)
        idx_ = idx.reshape(n, k).contiguous() if i32 else idx.reshape(n, k).to(torch.int64).contiguous()
        dist_ = f(dist).reshape(n, k)
        L = _cabi.lib()
        L.gs4d_rigidity_terms_scratch_bytes.restype = C.c_size_t
        scratch = torch.empty(int(L.gs4d_rigidity_terms_scratch_bytes()) // 4, dtype=torch.float32, device=dev)
        packed = torch.empty((n, 16), dtype=torch.float32, device=dev)
        sums = torch.empty(4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            fn = L.gs4d_rigidity_terms_forward_i32 if i32 else L.gs4d_rigidity_terms_forward
            rc = fn(C.c_int(n), C.c_int(k), _cabi.ptr(vx), _cabi.ptr(vl), _cabi.ptr(r1), _cabi.ptr(r2), _cabi.ptr(idx_),
                    _cabi.ptr(dist_), _cabi.ptr(packed), _cabi.ptr(scratch), _cabi.ptr(sums), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_rigidity_terms_forward")
        ctx.save_for_backward(packed, idx_, dist_, sums)
        ctx.graph_t = graph_t
        kn = float(k) * float(n)
        # knn_xyz = S_xyz / kn, knn_long = S_long / kn, knn_rot = (S_rot / kn) * S_w / kn
        return torch.stack([sums[0] / kn, sums[1] / kn, sums[2] / kn * sums[3] / kn])

    @staticmethod
    def backward(ctx, g):
        packed, idx_, dist_, sums = ctx.saved_tensors
        order, row_start = ctx.graph_t
        n, k = idx_.shape
        dev = packed.device
        kn = float(k) * float(n)
        g = g.to(torch.float32)
        coef = torch.stack([g[0] / kn, g[1] / kn, g[2] * sums[3] / kn / kn]).contiguous()     # stays on the device
        g_vx, g_vl = torch.empty((n, 3), device=dev), torch.empty((n, 3), device=dev)
        g_r1, g_r2 = torch.empty((n, 4), device=dev), torch.empty((n, 4), device=dev)
        with torch.cuda.device(dev):
            L = _cabi.lib()
            fn = L.gs4d_rigidity_terms_backward_i32 if idx_.dtype == torch.int32 else L.gs4d_rigidity_terms_backward
            rc = fn(C.c_int(n), C.c_int(k), _cabi.ptr(packed), _cabi.ptr(idx_), _cabi.ptr(dist_), _cabi.ptr(order), _cabi.ptr(row_start),
                    _cabi.ptr(coef), _cabi.ptr(g_vx), _cabi.ptr(g_vl), _cabi.ptr(g_r1), _cabi.ptr(g_r2), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_rigidity_terms_backward")
        return g_vx, g_vl, g_r1, g_r2, None, None, None


class _TemporalRegulation(torch.autograd.Function):
    """The thirteen tensors of temporal_regulation -> tensor [6] = (momentum, scales, shs, knn_xyz, knn_long, knn_rot),
    two launches of csrc/rigidity.hip per direction (gs4d_temporal_regulation_*): the second-difference terms, the kNN
    terms' inputs and their gradients never exist as tensors of their own."""

    @staticmethod
    def forward(ctx, k, *t13):
        dev = t13[1].device
        n = t13[1].shape[0]
        f = lambda t: t.detach().to(torch.float32).contiguous()
        ins = [f(t) for t in t13]
        for i in range(7, 13):                               # shs and rotations are read as float4
            if ins[i].data_ptr() % 16:
                ins[i] = ins[i].clone()
        shf = ins[7].numel() // n
        idx, dist, row_order = knn_self_i32(ins[1], k, return_order=True)   # neighbours of means3D_final (renderer.py:192)
        order, row_start = transpose_graph(idx, trusted=True)
        L = _cabi.lib()
        L.gs4d_temporal_regulation_scratch_bytes.restype = C.c_size_t
        scratch = torch.empty(int(L.gs4d_temporal_regulation_scratch_bytes(C.c_int(n))) // 4 + 4, dtype=torch.float32, device=dev)
        packed = torch.empty((n, 16), dtype=torch.float32, device=dev)
        out = torch.empty(6, dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * 13)(*[t.data_ptr() for t in ins])
        with torch.cuda.device(dev):
            rc = L.gs4d_temporal_regulation_forward(C.c_int(n), C.c_int(k), C.c_int(shf), ptrs, _cabi.ptr(idx), _cabi.ptr(dist),
                                                    _cabi.ptr(packed), _cabi.ptr(scratch), _cabi.ptr(out), _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_temporal_regulation_forward")
        ctx.save_for_backward(*ins, idx, dist, order, row_start, row_order, packed, scratch)
        ctx.k, ctx.shf, ctx.shapes = k, shf, [t.shape for t in t13]
        # six scalars, not one tensor for the caller to index: every out[i] in a loss expression costs the backward a
        # zero-filled [6] tensor, an index copy and an add (17 tiny kernels per iteration for the six terms)
        ctx.set_materialize_grads(False)
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        *ins, idx, dist, order, row_start, row_order, packed, scratch = ctx.saved_tensors
        n, dev = ins[1].shape[0], ins[1].device
        if all(g is None for g in gs):
            return (None,) * 14
        zero = None
        if any(g is None for g in gs):
            zero = torch.zeros((), dtype=torch.float32, device=dev)
        g_ = torch.stack([zero if g is None else g.to(torch.float32).reshape(()) for g in gs]).contiguous()
        grads = [torch.empty_like(t) for t in ins]
        ip = (C.c_void_p * 13)(*[t.data_ptr() for t in ins])
        gp = (C.c_void_p * 13)(*[t.data_ptr() for t in grads])
        with torch.cuda.device(dev):
            rc = _cabi.lib().gs4d_temporal_regulation_backward(
                C.c_int(n), C.c_int(ctx.k), C.c_int(ctx.shf), ip, _cabi.ptr(idx), _cabi.ptr(dist), _cabi.ptr(order),
                _cabi.ptr(row_start), _cabi.ptr(row_order), _cabi.ptr(packed), _cabi.ptr(scratch), _cabi.ptr(g_), gp,
                _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_temporal_regulation_backward")
        return (None, *[gr.view(sh) for gr, sh in zip(grads, ctx.shapes)])


def _neighbours(means3D_final, k):
    n = means3D_final.shape[0]
    k = min(int(k), n)
    p = means3D_final[None].contiguous().detach()
    idx, dist = knn(p, p, k)                               # renderer.py:161 / :192
    return k, n, idx, torch.exp(-100 * dist), transpose_graph(idx, trusted=True)     # weight: renderer.py:162 / :193


def temporal_regulation(means3D, means3D_final, means3D_prev, means3D_nxt, scales_final, scales_prev, scales_nxt,
                        shs_final, shs_prev, shs_nxt, rotations_final, rotations_prev, rotations_next, k=100, fused=True):
    """The non-recurrent branch, renderer.py:173-203.  *_final / *_prev / *_nxt: the deformation at t, t - dt, t + dt
    (scales activated as at :145,:181-182; rotations_final activated as at :146, rotations_prev / _next as the
    deformation returns them, exactly as the reference mixes them); means3D: the undeformed positions.
    fused=True computes the three kNN terms with gs4d_rigidity_terms_* (one kernel per direction); fused=False composes
    them from neighbor_diff_norm and torch expressions, line by line as the reference."""
    reg = {}
    if fused and means3D_final.device.type == "cuda":
        n = means3D_final.shape[0]
        out = _TemporalRegulation.apply(min(int(k), n), means3D, means3D_final, means3D_prev, means3D_nxt, scales_final, scales_prev,
                                        scales_nxt, shs_final, shs_prev, shs_nxt, rotations_final, rotations_prev, rotations_next)
        for name, term in zip(("momentum", "scales", "shs", "knn_xyz", "knn_long", "knn_rot"), out):
            reg[name] = term
        return reg
    momentum_all = torch.abs(means3D_nxt + means3D_prev - 2 * means3D_final)
    reg["momentum"] = momentum_all.mean()
    reg["scales"] = torch.abs(scales_nxt + scales_prev - 2 * scales_final).sum(dim=1).mean()
    reg["shs"] = torch.abs(shs_nxt + shs_prev - 2 * shs_final).mean()
    vel_xyz = (means3D_nxt - means3D_final) / 2
    vel_long = means3D_nxt - means3D
    rot_v1 = rotations_final - rotations_prev
    rot_v2 = rotations_next - rotations_final
    if fused:
        # the three kNN terms in one kernel per direction (no (n, k) tensors at all); :203 as written: the MEAN of the
        # rotation term scales the sum of the weights
        n = means3D_final.shape[0]
        k = min(int(k), n)
        idx, dist = knn_self_i32(means3D_final.detach().contiguous(), k)        # int32 lists, as the query writes them
        terms = _RigidityTerms.apply(vel_xyz, vel_long, rot_v1, rot_v2, idx, dist, transpose_graph(idx, trusted=True))
        reg["knn_xyz"], reg["knn_long"], reg["knn_rot"] = terms[0], terms[1], terms[2]
        return reg
    k, n, idx, weight, gt = _neighbours(means3D_final, k)
    mom_xyz_dist = neighbor_diff_norm(vel_xyz, idx, gt)
    mom_long_dist = neighbor_diff_norm(vel_long, idx, gt)
    reg["knn_xyz"] = (weight * mom_xyz_dist).sum() / k / n
    reg["knn_long"] = (weight * torch.abs(mom_long_dist - mom_xyz_dist)).sum() / k / n
    mom_rot_dist1 = neighbor_diff_norm(rot_v1, idx, gt)
    mom_rot_dist2 = neighbor_diff_norm(rot_v2, idx, gt)
    # :203 as written: the MEAN of the rotation term scales the sum of the weights
    reg["knn_rot"] = (weight * (mom_rot_dist1 + mom_rot_dist2).mean()).sum() / k / n
    return reg


def recurrent_regulation(means3D_in, means3D_canonical, means3D_final, rotations_in, rotations_final, scales_in,
                         scales_final, shs_in, shs_final, k=100):
    """The recurrent branch, renderer.py:154-171: *_in is the previous frame's state fed to the deformation,
    means3D_canonical is pc.get_xyz."""
    reg = {}
    vel_xyz = means3D_final - means3D_in
    vel_long = means3D_final - means3D_canonical
    vel_rot = rotations_final - rotations_in
    vel_scale = scales_final - scales_in
    vel_shs = shs_final - shs_in
    k, n, idx, weight, gt = _neighbours(means3D_final, k)
    mom_xyz_dist = neighbor_diff_norm(vel_xyz, idx, gt)
    mom_long_dist = neighbor_diff_norm(vel_long, idx, gt)
    mom_rot_dist = neighbor_diff_norm(vel_rot, idx, gt)
    reg["knn_xyz"] = (weight * mom_xyz_dist).sum() / k / n
    reg["knn_long"] = (weight * torch.abs(mom_long_dist - mom_xyz_dist)).sum() / k / n
    reg["knn_rot"] = (weight * mom_rot_dist).sum() / k / n
    reg["momentum"] = vel_xyz.abs().sum(dim=1).mean()
    reg["scales"] = vel_scale.abs().sum(dim=1).mean()
    reg["shs"] = vel_shs.mean()
    return reg
