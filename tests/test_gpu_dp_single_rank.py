"""The multi-rank training path on the REAL backend (RCCL, torch.distributed "nccl") in a process group of one rank:
the SH-gradient tensor hook and the deformation backward's data-phase hook start their all-reduces from the autograd
thread, the rest follows at the end, and every gradient comes back as a view of the bucket with the values of a
plain single-process step.  A one-GPU box cannot run two RCCL ranks; this covers the work handles, the stream ordering
and the hook plumbing that the gloo tests on the CPU cannot (tests/test_parallel_gloo.py covers the arithmetic)."""
import pytest
import torch
import torch.distributed as dist

from conftest import pkg
from test_gpu_deformation import make_args
from oracle import deformation_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("step_call", [False, True], ids=["autograd", "render_l1_step"])
@pytest.mark.parametrize("direct", [False, True], ids=["torch_collectives", "rccl_direct"])
def test_hooks_bucket_and_rccl_in_a_world_of_one(tmp_path, direct, step_call):
    RD, F, syn, par = pkg("renderer"), pkg("fused"), pkg("synthetic"), pkg("parallel")
    cfg = R.DeformConfig(plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2])
    torch.manual_seed(5)
    dev = torch.device("cuda")
    deform = pkg("scene.deformation").deform_network(make_args(cfg)).to(dev)
    _, cam_np, raw = syn.make_workload("C1", P=3000)
    pc = RD.GaussianParams(raw, dev)
    cam = RD.Camera(syn.look_at_camera(128, 96), dev, time=0.3)
    params = list(pc.parameters()) + list(deform.parameters())
    net = deform.deformation_net
    planes = [p for level in net.grid.grids for p in level]
    plane_ids = {id(p) for p in planes}
    data_params = [pc._xyz, pc._scaling, pc._rotation, *planes, net.force_time_embed.weight, net.force_time_embed.bias]
    target = torch.rand(3, 96, 128, device=dev)
    bg = torch.ones(3, device=dev)

    image, *_ = RD.render(cam, pc, deform, bg)
    F.l1_loss(image, target).backward()
    want = [None if p.grad is None else p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None

    dist.init_process_group("nccl", init_method="file://" + str(tmp_path / "rendezvous"), rank=0, world_size=1)
    par.GradBucket.exchange_in_world_of_one = True
    try:
        comm = par.RcclComm(dev) if direct else None          # ncclAllReduce on a side stream vs dist.all_reduce
        bucket = par.GradBucket(params, early=[pc._features, *data_params], comm=comm)
        started = []
        net.data_grad_hook = lambda d_xyz, d_sc, d_rot, gp, gw, gb: started.append(
            bucket.early_reduce_many(data_params, [d_xyz, d_sc, d_rot, *gp, gw, gb]))
        shs_hook = lambda g: started.append(bucket.early_reduce(pc._features, g))
        for step in range(3):
            del started[:]
            if step_call:                                          # the bench's form: both library calls back to back, no autograd graph
                RD.render_l1_step(cam, pc, deform, bg, target, shs_grad_hook=shs_hook)
            else:
                image, *_ = RD.render(cam, pc, deform, bg, shs_grad_hook=shs_hook)
                F.l1_loss(image, target).backward()
            assert started == [step > 0, step > 0]                 # SH hook first, then the data phase
            bucket.all_reduce()
            if step == 1 and direct:                               # the exchange really ran: poison check through a sum
                probe = torch.full((1000,), 2.0, device=dev)
                comm.all_reduce_sum(probe)
                comm.wait()
                assert float(probe.sum()) == 2000.0                # one rank: SUM leaves the values as they are
            torch.cuda.synchronize()
            lo, hi = bucket.flat.data_ptr(), bucket.flat.data_ptr() + 4 * bucket.flat.numel()
            for p, w in zip(params, want):
                assert (p.grad is None) == (w is None)
                if w is None:
                    continue
                assert lo <= p.grad.data_ptr() < hi and p.grad.data_ptr() % 16 == 0     # a view of the bucket
                assert p.grad.shape == w.shape
                if id(p) in plane_ids:                             # float atomics: summation order varies run to run
                    assert float((p.grad - w).abs().max()) <= 1e-5 * float(w.abs().max())
                    assert p.grad.is_contiguous(memory_format=torch.channels_last)
                else:
                    assert torch.equal(p.grad, w)
            for p in params:
                p.grad = None
    finally:
        net.data_grad_hook = None
        par.GradBucket.exchange_in_world_of_one = False
        if direct and "comm" in locals() and comm is not None:
            comm.close()
        dist.destroy_process_group()
