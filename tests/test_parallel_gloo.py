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
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_allreduce_world2(tmp_path):
    out = str(tmp_path / "done")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
