"""k-nearest-neighbour query over libgs4d.so (csrc/knn.hip): the counterpart of the one pointops2 operator the
fork uses -- `knnquery` (pointops2/functions/pointops.py:34-49 over pointops2/src/knnquery/*) -- and of its
caller `knn` (utils/general_utils.py:144-158), which feeds the k = 100 rigidity regulariser
(gaussian_renderer/renderer.py:160-168,191-203).  Same names, arguments and return values.
There is no CPU path (a missing library or CPU tensors raise).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi


def knnquery(nsample, xyz, new_xyz, offset, new_offset, _point_order=None):
    """input: xyz (n, 3), new_xyz (m, 3) or None, offset (b) / new_offset (b): cumulative batch ends.
    output: idx int32 (m, nsample) -- indices into xyz --, dist float32 (m, nsample) = sqrt(dist2), ascending.
    _point_order (not in the reference; single batch only): an int32 tensor (n,) that receives the points' indices in
    Morton order (gs4d_knn_query_order)."""
    if new_xyz is None:
        new_xyz = xyz
    if xyz.device.type != "cuda" or new_xyz.device.type != "cuda":
        raise RuntimeError("knnquery needs tensors on a HIP device; there is no CPU path")
    if xyz.dtype != torch.float32 or xyz.dim() != 2 or xyz.size(1) != 3 or new_xyz.dtype != torch.float32 or \
            new_xyz.dim() != 2 or new_xyz.size(1) != 3:
        raise RuntimeError("knnquery: xyz and new_xyz must be float32 tensors of shape (n, 3)")
    assert xyz.is_contiguous() and new_xyz.is_contiguous()          # as the reference asserts
    if not 1 <= int(nsample) <= 128:
        raise RuntimeError("knnquery: nsample must be in 1..128")
    L = _cabi.lib()
    dev = xyz.device
    m = new_xyz.shape[0]
    ends = [int(v) for v in offset.tolist()]
    qends = [int(v) for v in new_offset.tolist()]
    # the query writes every (row, slot) when a batch holds at least nsample points; only short batches leave slots
    # untouched (zero there, as the reference's zero-initialised outputs)
    full = all(e - s_ >= int(nsample) for s_, e in zip([0] + ends[:-1], ends) if e > s_)
    make = torch.empty if full else torch.zeros
    idx = make((m, nsample), dtype=torch.int32, device=dev)
    dist2 = make((m, nsample), dtype=torch.float32, device=dev)
    same = new_xyz is xyz or (new_xyz.data_ptr() == xyz.data_ptr() and new_xyz.shape == xyz.shape)
    if _point_order is not None and (len(ends) != 1 or _point_order.shape != (ends[0],) or _point_order.dtype != torch.int32
                                     or not _point_order.is_contiguous() or _point_order.device != dev):
        raise RuntimeError("knnquery: _point_order needs one batch and a contiguous int32 tensor of n entries on the points' device")
    start = qstart = 0
    for end, qend in zip(ends, qends):
        n_b, m_b = end - start, qend - qstart
        if m_b > 0:
            if n_b <= 0:
                raise RuntimeError("knnquery: a batch with queries has no points")
            pts = xyz[start:end]
            qs = new_xyz[qstart:qend]
            scratch = _cabi.TensorAllocator(dev)
            with torch.cuda.device(dev):
                rc = L.gs4d_knn_query_order(scratch.callback, scratch.ctx, C.c_int(n_b), _cabi.ptr(pts), C.c_int(m_b),
                                            C.c_void_p(0) if (same and start == qstart) else _cabi.ptr(qs), C.c_int(int(nsample)),
                                            C.c_void_p(idx.data_ptr() + 4 * qstart * nsample),
                                            C.c_void_p(dist2.data_ptr() + 4 * qstart * nsample),
                                            _cabi.ptr(_point_order) if _point_order is not None else C.c_void_p(0),
                                            _cabi.current_stream(dev))
            scratch.close()
            _cabi.check(rc, "gs4d_knn_query")
            if start:
                idx[qstart:qend] += start                     # the reference's indices are global (into xyz)
        start, qstart = end, qend
    return idx, torch.sqrt(dist2)


def knn_self_i32(points, k, return_order=False):
    """The render path's query: the k nearest neighbours of every row of points (n, 3) among the same rows ->
    idx int32 (n, k) exactly as gs4d_knn_query writes it, dist float32 (n, k).  What knn() does for one batch, without
    the offset arithmetic and the widening to int64 (two passes over the 10 M-entry lists at C2).  return_order: also the
    points' indices in Morton order, int32 (n,) -- the row order the regulariser kernels visit the lists in."""
    n = points.shape[0]
    off = torch.tensor([n], dtype=torch.int32)
    if not return_order:
        return knnquery(int(k), points, points, off, off)
    order = torch.empty(n, dtype=torch.int32, device=points.device)
    return (*knnquery(int(k), points, points, off, off, _point_order=order), order)


def knn(x, src, k, transpose=False):
    """utils/general_utils.py:144-158: x (b, n, 3), src (b, m, 3) -> idx int64 (b, n, k) into src, dists (b, n, k)."""
    if transpose:
        x = x.transpose(1, 2).contiguous()
        src = src.transpose(1, 2).contiguous()
    b, n, _ = x.shape
    m = src.shape[1]
    same = x.data_ptr() == src.data_ptr() and x.shape == src.shape
    x = x.reshape(-1, 3)
    src = x if same else src.reshape(-1, 3)
    x_offset = torch.cumsum(torch.full((b,), n, dtype=torch.long), dim=0).int()
    src_offset = torch.cumsum(torch.full((b,), m, dtype=torch.long), dim=0).int()
    idx, dists = knnquery(k, src, x, src_offset, x_offset)
    idx = idx.view(b, n, k) - (src_offset - m).to(idx.device)[:, None, None]
    return idx.long(), dists.view(b, n, k)
