"""View-parallel data parallelism: one process per GPU, replicated
parameters, each rank renders its own views, gradients are summed with ONE flat-bucket
all-reduce per parameter family over RCCL/xGMI (torch.distributed backend "nccl"), gloo on CPU
for tests.  The reference has no distributed code; its batch loop is sequential (train.py:169-189).
"""
from __future__ import annotations

from typing import Iterable, List

import torch
import torch.distributed as dist


def views_for_rank(batch: int, rank: int, world: int) -> List[int]:
    """Round-robin partition of the view batch: rank r renders views {r, r+N, ...}."""
    return list(range(rank, batch, world))


class GradBucket:
    """Flat fp32 bucket over the parameters that actually receive gradients (the reference's dead
    parameters -- prev_frame_encoder.*, feature_out1, force_embedder, grid.feature_aggr.*,
    opacity_deform.* -- have grad None and are skipped, never assumed dense)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.flat = None
        self.live = None

    def _ensure(self):
        live = [p for p in self.params if p.grad is not None]
        n = sum(p.numel() for p in live)
        if self.flat is None or self.flat.numel() != n or [id(p) for p in live] != [id(p) for p in (self.live or [])]:
            self.live = live
            dev = live[0].device if live else "cpu"
            self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        return self.live

    def all_reduce(self, async_op: bool = False):
        """Sum gradients over all ranks in place.  Returns the work handle when async_op."""
        live = self._ensure()
        if not live or not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        off = 0
        for p in live:
            n = p.numel()
            # .reshape(-1) would copy channels-last plane grads into logical order; keep memory order instead
            self.flat[off:off + n].copy_(_memory_order_view(p.grad))
            off += n
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        self._pending = work
        if not async_op:
            self.finish()
            return None
        return work

    def finish(self):
        work = getattr(self, "_pending", None)
        if work is None:
            return
        work.wait()
        self._pending = None
        off = 0
        for p in self.live:
            n = p.numel()
            _memory_order_view(p.grad).copy_(self.flat[off:off + n])
            off += n

    @property
    def nbytes(self) -> int:
        return 0 if self.flat is None else self.flat.numel() * 4


def _memory_order_view(t: torch.Tensor) -> torch.Tensor:
    """1-D view of a dense tensor in its own memory order (works for contiguous and channels-last)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)  # a view: permuted tensor is contiguous
    raise ValueError("gradient is neither contiguous nor channels-last")
