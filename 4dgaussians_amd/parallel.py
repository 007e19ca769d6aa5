"""View-parallel data parallelism: one process per GPU, replicated
parameters, each rank renders its own views, gradients are summed with ONE flat-bucket
all-reduce per parameter family over RCCL/xGMI (torch.distributed backend "nccl"), gloo on CPU
for tests.  The reference has no distributed code; its batch loop is sequential (train.py:169-189).
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading
import time
from typing import Iterable, List

import torch
import torch.distributed as dist

# GS4D_TRACE=1: the same one-line-per-event log the library writes (csrc/api.hip), around the collective waits
_TRACE = os.environ.get("GS4D_TRACE", "") not in ("", "0")


def _trace(msg: str) -> None:
    print(f"[gs4d-trace {os.getpid()} {threading.get_native_id()} {time.monotonic() * 1e3:.3f}] {msg}", file=sys.stderr, flush=True)


def views_for_rank(batch: int, rank: int, world: int) -> List[int]:
    """Round-robin partition of the view batch: rank r renders views {r, r+N, ...}."""
    return list(range(rank, batch, world))


def all_reduce_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, visibility: torch.Tensor):
    """The three small exchanges a densifying iteration adds (collectives 2-4), so that every rank
    accumulates the SAME statistics and densify / prune then change P identically everywhere:
      * SUM of the raw screen-space gradients (P,3): the reference adds them over the batch BEFORE taking the
        norm (train.py:229-233 -> scene/gaussian_model.py:613-615),
      * MAX of the int32 radii (P,) (train.py:199, max_radii2D update at train.py:302),
      * OR of the visibility filters (P,) (train.py:200).
    radii and visibility travel as one int32 [2,P] MAX all-reduce.  Returns (grad_sum, radii_max, visible_any);
    without a process group the inputs come back unchanged."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return viewspace_grad, radii, visibility
    g = viewspace_grad.detach().clone().contiguous()
    work = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
    iv = torch.stack([radii.to(torch.int32), visibility.to(torch.int32)]).contiguous()
    dist.all_reduce(iv, op=dist.ReduceOp.MAX)
    work.wait()
    return g, iv[0].to(radii.dtype), iv[1].to(torch.bool)


def seed_all_ranks(tag: int = 0, device=None) -> int:
    """Rank 0 draws a seed, every rank adopts it: densify_and_split samples new positions with torch.normal
    (scene/gaussian_model.py:446), and replicas must draw the same samples to stay identical.  Call it right
    before densify() on a densifying iteration.  Returns the seed."""
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64) + int(tag)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        t = seed.to(dev)
        dist.broadcast(t, 0)
        seed = t.cpu()
    torch.manual_seed(int(seed.item()))        # seeds the CPU and every GPU generator of this process
    return int(seed.item())


class _RcclUniqueId(ctypes.Structure):                     # rccl.h: ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128
    _fields_ = [("internal", ctypes.c_char * 128)]


class RcclComm:
    """The gradient exchange straight on RCCL: one communicator per process, created next to the torch.distributed
    process group (which carries the 128-byte unique id to the other ranks and keeps serving barriers and the odd
    small collective), and one side stream the all-reduces run on.

    Why not dist.all_reduce: a step issues three collectives after the forward's host synchronisation, where the host
    is on the critical path, and a torch.distributed collective costs more host time (work object, events, Python
    wrappers) than ncclAllReduce on a stream (measured in a world of one rank: 1.68 vs 1.75 ms per step).  The
    library is the librccl.so torch itself has loaded.
    Ordering: all_reduce_sum() makes the side stream wait for everything issued so far on the caller's current
    stream, wait() makes the current stream wait for the side stream; the buffers must stay alive (the bucket does).
    Construction is collective: every rank of the default process group must call it, and either all ranks get a
    communicator or all get a RuntimeError (the outcome is agreed through the process group first)."""

    FLOAT32, SUM = 7, 0                                      # ncclFloat32, ncclSum (rccl.h)

    def __init__(self, device):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclComm needs an initialised torch.distributed process group to exchange the unique id")
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        lib, problem = None, ""
        try:
            lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
            lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_RcclUniqueId)]
            lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _RcclUniqueId, ctypes.c_int]
            lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
            lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            lib.ncclGetErrorString.restype = ctypes.c_char_p
            lib.ncclGetErrorString.argtypes = [ctypes.c_int]
        except (OSError, AttributeError) as e:
            lib, problem = None, str(e)
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = _RcclUniqueId()
        ok = lib is not None
        if ok and rank == 0 and lib.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            ok, problem = False, "ncclGetUniqueId failed"
        # element 0: this rank is able to go on; elements 1..128: the bytes of rank 0's unique id
        msg = torch.zeros(129, dtype=torch.int32, device=self.device)
        msg[0] = 1 if ok else 0
        if rank == 0:
            msg[1:] = torch.tensor(list(bytes(uid)), dtype=torch.int32)
        flag = msg[:1].clone()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        dist.broadcast(msg, 0)
        if int(flag.item()) != 1:
            raise RuntimeError("RcclComm: librccl.so is not usable on every rank" + (f" ({problem})" if problem else ""))
        raw = bytes(int(v) & 0xFF for v in msg[1:].tolist())
        ctypes.memmove(ctypes.byref(uid), raw, 128)
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
            self.stream = torch.cuda.Stream(self.device)
        # agree on the outcome, so that no rank goes on alone with a communicator the others do not have
        flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            if rc == 0:
                lib.ncclCommDestroy(comm)
            raise RuntimeError("RcclComm: ncclCommInitRank failed" + (f": {lib.ncclGetErrorString(rc).decode()}" if rc else " on another rank"))
        self._lib, self._comm, self._busy = lib, comm, False
        self._stream_handle = ctypes.c_void_p(self.stream.cuda_stream)

    def all_reduce_sum(self, t: torch.Tensor, inline: bool = False):
        """In-place SUM over the ranks of a dense float32 device tensor: asynchronous on the side stream, or, with
        inline=True, on the caller's current stream (no cross-stream events: right for a small exchange that nothing
        can overlap with anyway; it must not be issued while side-stream work of this communicator is pending on
        another rank in a different order, so callers issue it after wait())."""
        if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
            raise ValueError("RcclComm.all_reduce_sum: dense float32 tensor on the communicator's device expected")
        cur = torch.cuda.current_stream(self.device)
        if inline:
            handle = ctypes.c_void_p(cur.cuda_stream)
        else:
            self.stream.wait_stream(cur)
            handle = self._stream_handle
        ptr = t.data_ptr()
        rc = self._lib.ncclAllReduce(ptr, ptr, t.numel(), self.FLOAT32, self.SUM, self._comm, handle)
        if rc != 0:
            raise RuntimeError("ncclAllReduce failed: " + self._lib.ncclGetErrorString(rc).decode())
        if not inline:
            self._busy = True

    def wait(self):
        """Everything issued through all_reduce_sum() is ordered before what the current stream does next."""
        if self._busy:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self._busy = False

    def close(self):
        if self._comm is not None and self._comm.value:
            self.stream.synchronize()
            self._lib.ncclCommDestroy(self._comm)
            self._comm = None


class GradBucket:
    """Flat fp32 bucket over the parameters that actually receive gradients (the reference's dead
    parameters -- prev_frame_encoder.*, feature_out1, force_embedder, grid.feature_aggr.*,
    opacity_deform.* -- have grad None and are skipped, never assumed dense).

    Packing is one multi-tensor copy (torch._foreach_copy_) instead of one small copy kernel per parameter
    (~45 tensors: the launches cost more than the all-reduce itself), and nothing is copied back: after
    finish() every p.grad IS a view of the reduced bucket (slots start on 64-float boundaries so the views
    keep the 16-byte alignment the fused optimizer needs; the padding travels as zeros).

    Early parameters: a gradient that is final before the backward pass ends can start its all-reduce at once
    and overlap with the rest of the backward.  The SH coefficients are the case that matters: the deformation
    adds a residual to them (scene/deformation.py:191-197), so dL/d_features == dL/d(deformed shs), which
    exists as soon as the rasterizer's backward has run -- 19.2 of the 37 MB of a C2 step, hidden behind the
    deformation backward.  `early=[param]` puts those slots first; `early_reduce(param, grad)` (called from a
    tensor hook) copies and launches; all_reduce() then only handles the remaining slots.  The layout is
    learnt on the first step (which parameters are live is only known after a backward), so the first step
    -- and any step after the set of live parameters changed -- reduces everything at the end."""

    ALIGN = 64

    def __init__(self, params: Iterable[torch.nn.Parameter], early: Iterable[torch.nn.Parameter] = (), comm: "RcclComm" = None):
        self.comm = comm               # None: torch.distributed collectives; an RcclComm: ncclAllReduce on its side stream
        early = list(early)
        early_ids = {id(p) for p in early}
        rest = [p for p in params if p.requires_grad and id(p) not in early_ids]
        self.params = [p for p in early if p.requires_grad] + rest      # early slots first
        self.early_ids = early_ids
        self.flat = None
        self.live = None
        self.slots = None
        self.split = 0                 # floats of the early region
        self._live_mask = None
        self._pending = []
        self._early_done = set()       # slot indices already on their way

    def _layout(self, live):
        offs, total, split = [], 0, 0
        for p in live:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if id(p) in self.early_ids:
                split = total
        dev = live[0].device if live else "cpu"
        self.live = live
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.slots = [self.flat[o:o + p.numel()] for o, p in zip(offs, live)]
        self.split = split
        # the per-step work is kept to list lookups (the host is on the critical path after the forward's
        # synchronisation): slot index of every live parameter, slot bounds, and the views of the bucket, shaped and
        # strided like the gradients autograd produced, that are both the copy targets and what finish() installs
        self._index = {id(p): i for i, p in enumerate(live)}
        self._bounds = [(o, o + p.numel()) for o, p in zip(offs, live)]
        self._grad_views = [_like_memory_order(slot, p.grad) for slot, p in zip(self.slots, live)]

    # exchange even in a world of one rank: lets a one-GPU box run the real collectives (RCCL, stream ordering,
    # hooks on the autograd thread) of the multi-rank path; see tests/test_gpu_dp_single_rank.py
    exchange_in_world_of_one = False

    def _active(self) -> bool:
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size() > 1 or self.exchange_in_world_of_one

    INLINE_BYTES = 2 << 20         # end-of-step regions up to this size run on the compute stream (RcclComm only)

    def _reduce(self, region: torch.Tensor, at_end: bool = False):
        if _TRACE:
            _trace(f"issue all_reduce of {region.numel() * 4} bytes ({'end of step' if at_end else 'early'})")
        if self.comm is not None:
            if at_end and region.numel() * 4 <= self.INLINE_BYTES:
                # the small tail of the step: nothing is left to overlap with, so skip the side stream and its two
                # cross-stream events; ordered after the early collectives by waiting for the side stream first
                self.comm.wait()
                self.comm.all_reduce_sum(region, inline=True)
            else:
                self.comm.all_reduce_sum(region)
            if not self._pending:
                self._pending.append(self.comm)
        else:
            self._pending.append(dist.all_reduce(region, op=dist.ReduceOp.SUM, async_op=True))

    def early_reduce(self, param, grad, force: bool = False) -> bool:
        """Start the all-reduce of `param`'s gradient now (grad: the final gradient, a dense tensor of the
        parameter's shape).  Returns False -- and does nothing -- while the layout is not known yet.
        force=True runs the bookkeeping and the copy without a process group (single-process tests)."""
        if self.flat is None or not (force or self._active()) or id(param) not in self.early_ids:
            return False
        i = self._index.get(id(param))
        if i is None:
            return False
        self._grad_views[i].copy_(grad)
        if self._active():
            self._reduce(self.slots[i])
        self._early_done.add(i)
        return True

    def early_reduce_many(self, params, grads, force: bool = False) -> bool:
        """early_reduce for several parameters at once: one multi-tensor copy and ONE all-reduce over the bucket
        range that spans their slots.  The parameters must occupy consecutive slots (list them consecutively in
        `early=`) so that the range holds nothing else; returns False -- and does nothing -- otherwise, or while
        the layout is not known yet, or if one of them is not live."""
        if self.flat is None or not (force or self._active()) or not params:
            return False
        ids = [self._index.get(id(p)) for p in params]
        if any(i is None or id(p) not in self.early_ids for i, p in zip(ids, params)):
            return False
        lo, hi = min(ids), max(ids)
        if hi - lo + 1 != len(ids) or len(set(ids)) != len(ids):
            return False
        torch._foreach_copy_([self._grad_views[i] for i in ids], list(grads))
        if self._active():
            self._reduce(self.flat[self._bounds[lo][0]:self._bounds[hi][1]])
        self._early_done.update(ids)
        return True

    def all_reduce(self, async_op: bool = False):
        """Sum gradients over all ranks in place.  Returns the work handles when async_op."""
        grads = [p.grad for p in self.params]                  # ONE .grad read per parameter and step
        mask = [g is not None for g in grads]
        if self.flat is None or mask != self._live_mask:
            if self._early_done:
                # the layout changed under an early reduce: wait for it, then redo everything at the end
                for w in self._pending:
                    w.wait()
                self._pending, self._early_done = [], set()
            self._live_mask = mask
            self._layout([p for p, m in zip(self.params, mask) if m])
        if not self.live or not self._active():
            self._early_done = set()
            return None
        live_grads = [g for g in grads if g is not None]
        done = self._early_done
        todo = [i for i in range(len(live_grads)) if i not in done] if done else range(len(live_grads))
        # the copy targets are views with the gradients' own shape and strides (channels-last plane gradients stay
        # in memory order, nothing is re-laid out)
        if len(todo):
            torch._foreach_copy_([self._grad_views[i] for i in todo], [live_grads[i] for i in todo])
        # one all-reduce per run of consecutive slots that were not reduced early: the whole bucket when nothing
        # was, flat[split:] when every early parameter was (the two usual cases), several ranges otherwise
        total, run_begin = self.flat.numel(), None
        for i in range(len(live_grads)):
            if i in done:
                if run_begin is not None:
                    self._reduce(self.flat[run_begin:self._bounds[i][0]], at_end=True)
                    run_begin = None
            elif run_begin is None:
                run_begin = self._bounds[i][0]
        if run_begin is not None:
            self._reduce(self.flat[run_begin:total], at_end=True)
        if not async_op:
            self.finish()
            return None
        return self._pending

    def finish(self):
        if not self._pending:
            self._early_done = set()
            return
        for k, w in enumerate(self._pending):
            if _TRACE:
                t0 = time.monotonic()
                _trace(f"enter GradBucket.finish wait {k + 1}/{len(self._pending)} ({type(w).__name__})")
            w.wait()
            if _TRACE:
                _trace(f"exit  GradBucket.finish wait {k + 1} after {(time.monotonic() - t0) * 1e3:.3f} ms")
        self._pending, self._early_done = [], set()
        for p, view in zip(self.live, self._grad_views):
            p.grad = view

    @property
    def nbytes(self) -> int:
        """Gradient payload of one exchange (the bucket itself is padded to ALIGN-float slots)."""
        return 0 if self.live is None else sum(p.numel() for p in self.live) * 4


def _like_memory_order(flat: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """View the 1-D `flat` with the shape and memory format of `like` (inverse of _memory_order_view)."""
    if like.is_contiguous():
        return flat.view(like.shape)
    n, c, h, w = like.shape
    return flat.view(n, h, w, c).permute(0, 3, 1, 2)


def _memory_order_view(t: torch.Tensor) -> torch.Tensor:
    """1-D view of a dense tensor in its own memory order (works for contiguous and channels-last)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)  # a view: permuted tensor is contiguous
    raise ValueError("gradient is neither contiguous nor channels-last")
