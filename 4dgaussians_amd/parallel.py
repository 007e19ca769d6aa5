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
    opacity_deform.* -- have grad None and are skipped, never assumed dense).

    Packing is one multi-tensor copy (torch._foreach_copy_) instead of one small copy kernel per parameter
    (~45 tensors: the launches cost more than the all-reduce itself), and nothing is copied back: after
    finish() every p.grad IS a view of the reduced bucket (slots start on 64-float boundaries so the views
    keep the 16-byte alignment the fused optimizer needs; the padding travels as zeros)."""

    ALIGN = 64

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.flat = None
        self.live = None
        self.slots = None
        self._pending = None

    def _ensure(self):
        live = [p for p in self.params if p.grad is not None]
        if self.flat is None or [id(p) for p in live] != [id(p) for p in (self.live or [])]:
            self.live = live
            offs, total = [], 0
            for p in live:
                offs.append(total)
                total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            dev = live[0].device if live else "cpu"
            self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
            self.slots = [self.flat[o:o + p.numel()] for o, p in zip(offs, live)]
        return self.live

    def all_reduce(self, async_op: bool = False):
        """Sum gradients over all ranks in place.  Returns the work handle when async_op."""
        live = self._ensure()
        if not live or not dist.is_initialized() or dist.get_world_size() == 1:
            return None
        # .reshape(-1) would copy channels-last plane grads into logical order; keep memory order instead
        torch._foreach_copy_(self.slots, [_memory_order_view(p.grad) for p in live])
        self._pending = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        if not async_op:
            self.finish()
            return None
        return self._pending

    def finish(self):
        work = self._pending
        if work is None:
            return
        work.wait()
        self._pending = None
        for p, slot in zip(self.live, self.slots):
            p.grad = _like_memory_order(slot, p.grad)

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
