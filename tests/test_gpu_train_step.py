"""GPU parity of the per-iteration pieces after the render path:

* FusedAdamW against torch.optim.AdamW itself (the reference's optimizer, scene/gaussian_model.py:206) run on the
  CPU in float32 on the same parameters / gradients / learning-rate schedule: parameters to 1e-6 * max(1, |p|)
  after six steps (a few float32 ulps: torch's vectorised CPU kernels and the HIP kernel round fused / unfused
  multiply-adds differently), moments to 1e-5 relative;
* the plane regularisers against the reference's torch expressions (scene/regulation.py:22-29,
  scene/gaussian_model.py:632-677 restated below): values to 1e-5 relative, gradients to 1e-5 of their max.
"""
import types

import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1000, 3), (1, 32, 17, 9), (64, 134), (64,), (5, 1), (1, 8, 4, 6), (7,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ps[1] = ps[1].contiguous(memory_format=torch.channels_last)      # a HexPlane plane
    ps[5] = ps[5].contiguous(memory_format=torch.channels_last)
    return ps


def test_fused_adamw_matches_torch_adamw():
    opt_mod = pkg("optim")
    cpu = [torch.nn.Parameter(p.clone()) for p in _params(0)]
    gpu = [torch.nn.Parameter(p.clone().cuda()) for p in _params(0)]
    assert gpu[1].is_contiguous(memory_format=torch.channels_last)

    def groups(ps):
        return [{"params": ps[:2], "lr": 1.6e-4, "name": "xyz"}, {"params": ps[2:5], "lr": 1e-3, "name": "deformation"},
                {"params": ps[5:], "lr": 2.5e-3, "name": "f_dc"}]

    ref = torch.optim.AdamW(groups(cpu), lr=0.0001, eps=1e-8)
    got = opt_mod.FusedAdamW(groups(gpu), lr=0.0001, eps=1e-8)
    assert got.defaults["weight_decay"] == ref.defaults["weight_decay"] == 1e-2
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        for grp_r, grp_g in zip(ref.param_groups, got.param_groups):       # per-iteration learning-rate schedule
            grp_r["lr"] = grp_g["lr"] = grp_r["lr"] * 0.9
        for i, (pc, pg) in enumerate(zip(cpu, gpu)):
            if i == 6 or (i == 3 and it < 2):          # a parameter that never / only later receives a gradient
                pc.grad = pg.grad = None
                continue
            gr = torch.randn(pc.shape, generator=g) * (10.0 if i == 0 else 0.1)
            pc.grad = gr.clone().contiguous(memory_format=torch.channels_last) if pc.dim() == 4 else gr.clone()
            # plane gradients arrive as channels-last views, others contiguous; one deliberately mis-strided
            pg.grad = gr.cuda() if i != 1 else gr.cuda().contiguous()
        ref.step()
        got.step()
    for i, (pc, pg) in enumerate(zip(cpu, gpu)):
        err = float((pg.detach().cpu() - pc.detach()).abs().max())
        assert err <= 1e-6 * max(1.0, float(pc.detach().abs().max())), (i, err)
        if pc in ref.state and len(ref.state[pc]):
            for k in ("exp_avg", "exp_avg_sq"):
                a, b = got.state[pg][k].cpu(), ref.state[pc][k]
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30, (i, k)
            assert float(got.state[pg]["step"]) == float(ref.state[pc]["step"])
    assert len(got.state[gpu[6]]) == 0
    # optimizer state interchanges with torch's (checkpoints, scene/gaussian_model.py:88-128)
    sd = got.state_dict()
    other = torch.optim.AdamW(groups([torch.nn.Parameter(p.detach().clone()) for p in gpu]), lr=0.0001, eps=1e-8)
    other.load_state_dict(sd)
    got2 = opt_mod.FusedAdamW(groups(gpu), lr=0.0001, eps=1e-8)
    got2.load_state_dict(ref.state_dict())
    for p in gpu:
        p.grad = torch.ones_like(p)
    got2.step()                                       # CPU-loaded state is moved / re-laid out, not rejected


def test_fused_adamw_honours_betas_and_eps_changed_between_steps():
    """torch.optim.AdamW reads betas / eps from the group on every step; the fused optimizer's cached launch plan must
    notice such a change (it batches tensors by those values) instead of stepping on with the old ones."""
    opt_mod = pkg("optim")
    cpu = [torch.nn.Parameter(p.clone()) for p in _params(3)[:3]]
    gpu = [torch.nn.Parameter(p.clone().cuda()) for p in _params(3)[:3]]
    ref = torch.optim.AdamW([{"params": cpu[:2]}, {"params": cpu[2:]}], lr=1e-3, eps=1e-8)
    got = opt_mod.FusedAdamW([{"params": gpu[:2]}, {"params": gpu[2:]}], lr=1e-3, eps=1e-8)
    g = torch.Generator().manual_seed(5)
    for it in range(5):
        if it == 2:
            for o in (ref, got):
                o.param_groups[0]["betas"] = (0.5, 0.9)
                o.param_groups[1]["eps"] = 1e-3
        for pc, pg in zip(cpu, gpu):
            gr = torch.randn(pc.shape, generator=g)
            pc.grad = gr.clone().contiguous(memory_format=torch.channels_last) if pc.dim() == 4 else gr.clone()
            pg.grad = pc.grad.cuda()
        ref.step()
        got.step()
    for i, (pc, pg) in enumerate(zip(cpu, gpu)):
        err = float((pg.detach().cpu() - pc.detach()).abs().max())
        assert err <= 1e-6 * max(1.0, float(pc.detach().abs().max())), (i, err)
        assert float(got.state[pg]["step"]) == float(ref.state[pc]["step"]) == 5.0


def test_fused_adamw_refuses_cpu_parameters():
    opt_mod = pkg("optim")
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="HIP device"):
        opt_mod.FusedAdamW([p]).step()


# ---- reference expressions (scene/regulation.py:22-29, scene/gaussian_model.py:632-677) in torch
def ref_smoothness(t):
    h = t.shape[-2]
    first = t[..., 1:, :] - t[..., :h - 1, :]
    second = first[..., 1:, :] - first[..., :h - 2, :]
    return torch.square(second).mean()


def ref_regulation(grids, tsw, l1w, l2w, ptw):
    total = 0.0
    for level in grids:
        for gi in (0, 1, 3):
            total = total + ptw * ref_smoothness(level[gi])
        for gi in (2, 4, 5):
            total = total + tsw * ref_smoothness(level[gi]) + l1w * torch.abs(1 - level[gi]).mean() \
                + l2w * torch.square(1 - level[gi]).mean()
    return total


@pytest.mark.parametrize("multires", [[1, 2], [1, 2, 4]])       # 24 terms in one call; 36 terms = two chunks
def test_plane_regulation_matches_reference_expressions(multires):
    from test_gpu_deformation import make_args
    from oracle import deformation_ref as R
    reg = pkg("scene.regulation")
    cfg = R.DeformConfig(plane_channels=16, resolution=[12, 10, 14, 9], multires=multires)
    torch.manual_seed(5)
    net = pkg("scene.deformation").deform_network(make_args(cfg)).cuda()
    grids = net.deformation_net.grid.grids
    weights = dict(time_smoothness_weight=0.01, l1_time_planes_weight=1e-4, l2_time_planes_weight=2e-4, plane_tv_weight=2e-4)
    out = reg.compute_regulation(net, **weights)
    cpu = [[p.detach().cpu().contiguous().double().requires_grad_(True) for p in level] for level in grids]
    want = ref_regulation(cpu, 0.01, 1e-4, 2e-4, 2e-4)
    assert abs(float(out["total_reg"].detach()) - float(want.detach())) <= 1e-5 * abs(float(want.detach()))
    parts = float(out["time_reg"]) + float(out["plane_reg"]) + float(out["l1_time_reg"]) + float(out["l2_time_reg"])
    assert abs(parts - float(want.detach())) <= 1e-5 * abs(float(want.detach()))
    loss = out["total_reg"] * 3.0
    loss += 0.0                                      # callers accumulate in place (train.py `loss += tv_loss`): must not raise
    loss.backward()
    (want * 3.0).backward()
    for lg, lc in zip(grids, cpu):
        for pg, pc in zip(lg, lc):
            assert pg.grad is not None
            err = float((pg.grad.cpu().double() - pc.grad).abs().max())
            assert err <= 1e-5 * float(pc.grad.abs().max()), err
    # gating as in the reference: zero weights and use_time = False drop their terms
    z = reg.compute_regulation(net, 0.0, 0.0, 0.0, 0.0)
    assert float(z["total_reg"]) == 0.0
    net.deformation_net.args.use_time = False
    only_plane = reg.compute_regulation(net, **weights)
    want_plane = sum(2e-4 * ref_smoothness(level[gi].detach()) for level in cpu for gi in (0, 1, 3))
    assert abs(float(only_plane["total_reg"]) - float(want_plane)) <= 1e-5 * float(want_plane)
    # the single-plane helper of scene/regulation.py
    t = torch.rand(1, 8, 3, 5, device="cuda").contiguous(memory_format=torch.channels_last)
    assert abs(float(reg.compute_plane_smoothness(t)) - float(ref_smoothness(t.cpu().double()))) <= 1e-6


def test_adamw_step_counts_survive_state_dict_round_trip():
    """The step counts are kept outside the `step` tensors between steps: state_dict() shows the true count, and an
    optimizer restored from it continues exactly like the original."""
    om = pkg("optim")
    g = torch.Generator().manual_seed(3)
    mk = lambda: [torch.nn.Parameter(torch.randn(33, 8, generator=g).cuda()), torch.nn.Parameter(torch.randn(64, generator=g).cuda())]
    pa = mk()
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    grads = [[torch.randn(p.shape, generator=g).cuda() for p in pa] for _ in range(5)]
    oa = om.FusedAdamW([{"params": pa[:1], "lr": 1e-2}, {"params": pa[1:], "lr": 3e-3}])
    ob = om.FusedAdamW([{"params": pb[:1], "lr": 1e-2}, {"params": pb[1:], "lr": 3e-3}])
    for it in range(3):
        for opt, ps in ((oa, pa), (ob, pb)):
            for p, gr in zip(ps, grads[it]):
                p.grad = gr.clone()
            opt.step()
    sd = oa.state_dict()
    assert all(float(st["step"]) == 3.0 for st in sd["state"].values())
    assert all(float(oa.state[p]["step"]) == 3.0 for p in pa)
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oc = om.FusedAdamW([{"params": pc[:1], "lr": 1e-2}, {"params": pc[1:], "lr": 3e-3}])
    oc.load_state_dict(sd)
    for it in range(3, 5):
        for opt, ps in ((ob, pb), (oc, pc)):
            for p, gr in zip(ps, grads[it]):
                p.grad = gr.clone()
            opt.step()
    for b_, c_ in zip(pb, pc):
        assert torch.equal(b_, c_)
    assert all(float(st["step"]) == 5.0 for st in oc.state_dict()["state"].values())
