"""World-size-2 gloo test of the view-parallel gradient exchange on CPU."""
import importlib
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("4dgaussians_amd.parallel")
    torch.manual_seed(0)
    a = torch.nn.Parameter(torch.zeros(5, 3))
    plane = torch.nn.Parameter(torch.zeros(1, 4, 3, 2).contiguous(memory_format=torch.channels_last))
    dead = torch.nn.Parameter(torch.zeros(7))            # never gets a gradient: must be skipped, not assumed dense
    a.grad = torch.full((5, 3), float(rank + 1))
    plane.grad = (torch.arange(24, dtype=torch.float32).reshape(1, 4, 3, 2) * (rank + 1)).contiguous(memory_format=torch.channels_last)
    b = par.GradBucket([a, plane, dead])
    b.all_reduce(async_op=True)
    b.finish()
    assert dead.grad is None
    assert torch.equal(a.grad, torch.full((5, 3), 3.0))
    assert torch.equal(plane.grad, torch.arange(24, dtype=torch.float32).reshape(1, 4, 3, 2) * 3)
    assert plane.grad.is_contiguous(memory_format=torch.channels_last)
    assert b.nbytes == (15 + 24) * 4
    assert par.views_for_rank(8, rank, world) == list(range(rank, 8, 2))
    # early parameters: step 1 learns the layout (everything reduced at the end), step 2 starts `e` early
    e = torch.nn.Parameter(torch.zeros(9))
    late = torch.nn.Parameter(torch.zeros(70))
    be = par.GradBucket([late, e, dead], early=[e])
    for step in range(3):
        ge = torch.full((9,), float(10 * step + rank + 1))
        started = be.early_reduce(e, ge)                 # what a tensor hook would do mid-backward
        assert started == (step > 0)
        e.grad = ge.clone()                              # autograd delivers the same values at the end
        late.grad = torch.arange(70, dtype=torch.float32) * (rank + 1) + step
        be.all_reduce()
        assert torch.equal(e.grad, torch.full((9,), float(20 * step + 3)))
        assert torch.equal(late.grad, torch.arange(70, dtype=torch.float32) * 3 + 2 * step)
        assert e.grad.data_ptr() % 16 == 0 and late.grad.data_ptr() % 16 == 0
        e.grad = late.grad = None
    assert be.split == 64 and be.nbytes == (9 + 70) * 4
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_allreduce_world2(tmp_path):
    out = str(tmp_path / "done")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
