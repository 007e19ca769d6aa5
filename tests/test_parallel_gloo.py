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
    # a group of early parameters in consecutive slots: one all-reduce over their range (padding travels as zeros)
    e1, e2 = torch.nn.Parameter(torch.zeros(9)), torch.nn.Parameter(torch.zeros(1, 4, 3, 2).contiguous(memory_format=torch.channels_last))
    e3, late2 = torch.nn.Parameter(torch.zeros(100)), torch.nn.Parameter(torch.zeros(5))
    bg = par.GradBucket([late2, e1, e2, e3, dead], early=[e1, e2, e3])
    for step in range(3):
        g1 = torch.full((9,), float(rank + 1 + step))
        g2 = (torch.arange(24, dtype=torch.float32).reshape(1, 4, 3, 2) * (rank + 1)).contiguous(memory_format=torch.channels_last)
        g3 = torch.arange(100, dtype=torch.float32) + rank
        assert bg.early_reduce(e1, g1) == (step > 0)                      # e1 on its own, then e2 + e3 as a group ...
        if step < 2:
            assert bg.early_reduce_many([e2, e3], [g2, g3]) == (step > 0)
        else:                                                             # ... or only e3: e2 goes at the end, between two early slots
            assert bg.early_reduce(e3, g3)
        assert not bg.early_reduce_many([e1, e3], [g1, g3])               # not consecutive: refused, nothing done
        e1.grad, e2.grad, e3.grad = g1.clone(), g2.clone(), g3.clone()
        late2.grad = torch.full((5,), float(rank))
        bg.all_reduce()
        assert torch.equal(e1.grad, torch.full((9,), float(3 + 2 * step)))
        assert torch.equal(e2.grad, torch.arange(24, dtype=torch.float32).reshape(1, 4, 3, 2) * 3)
        assert e2.grad.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(e3.grad, 2 * torch.arange(100, dtype=torch.float32) + 1)
        assert torch.equal(late2.grad, torch.full((5,), 1.0))
        e1.grad = e2.grad = e3.grad = late2.grad = None
    if rank == 0:
        open(out, "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def _densify_worker(rank, world, port, out):
    """Two replicas see different views (different screen-space gradients, radii, visibility); after the three
    statistic exchanges and a shared seed, densify + prune leave both with the SAME Gaussians."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("4dgaussians_amd.parallel")
    from test_gaussian_model_host import model
    m = model(80, seed=3)                                   # identical replicas
    g = torch.Generator().manual_seed(100 + rank)           # rank-specific "views"
    vs_grad = torch.randn(80, 3, generator=g) * 1e-3
    radii = torch.randint(0, 60, (80,), generator=g, dtype=torch.int32)
    vis = radii > 20
    gs, rmax, vany = par.all_reduce_densification_stats(vs_grad, radii, vis)
    both_g = [torch.zeros_like(vs_grad) for _ in range(world)]
    both_r = [torch.zeros_like(radii) for _ in range(world)]
    dist.all_gather(both_g, vs_grad)
    dist.all_gather(both_r, radii)
    assert torch.equal(gs, both_g[0] + both_g[1])
    assert torch.equal(rmax, torch.maximum(both_r[0], both_r[1])) and rmax.dtype == torch.int32
    assert torch.equal(vany, (both_r[0] > 20) | (both_r[1] > 20)) and vany.dtype == torch.bool
    m.max_radii2D[vany] = torch.max(m.max_radii2D[vany], rmax[vany].float())        # train.py:302
    m.add_densification_stats(gs, vany)
    torch.manual_seed(rank)                                 # the replicas' generators have drifted apart ...
    seed = par.seed_all_ranks(tag=7)                        # ... and are brought together before the split samples
    m.percent_dense = 0.05
    m.densify(0.0012, 0.005, 1.0, 40)
    m.prune(0.0012, 0.005, 1.0, 40)
    n = torch.tensor([m.get_xyz.shape[0], seed])
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    assert torch.equal(ns[0], ns[1]) and int(ns[0][0]) != 80
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc"):
        mine = getattr(m, name).detach().contiguous()
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        assert torch.equal(parts[0], parts[1]), name
    if rank == 0:
        open(out, "w").write("ok %d" % int(ns[0][0]))
    dist.barrier()
    dist.destroy_process_group()


def test_densification_stays_rank_consistent_world2(tmp_path):
    out = str(tmp_path / "done")
    mp.spawn(_densify_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read().startswith("ok")


def test_densification_stats_without_process_group():
    par = importlib.import_module("4dgaussians_amd.parallel")
    g, r, v = torch.ones(4, 3), torch.arange(4, dtype=torch.int32), torch.tensor([True, False, True, False])
    g2, r2, v2 = par.all_reduce_densification_stats(g, r, v)
    assert g2 is g and r2 is r and v2 is v


def test_grad_bucket_allreduce_world2(tmp_path):
    out = str(tmp_path / "done")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "ok"
