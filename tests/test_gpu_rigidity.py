"""k-NN rigidity / momentum regularisers: gs4d_neighbor_diff_norm_* and the two dictionaries of
gaussian_renderer/renderer.py:151-203 against the CPU restatement (oracle/rigidity_ref.py: the reference's torch
expressions with an exact brute-force neighbour search).  Tolerances: values 1e-5 relative, gradients 2e-5 of the
tensor's largest entry (float32 sums over n*k pairs in a different order)."""
import pytest
import torch

from conftest import pkg
from oracle import rigidity_ref as RR

pytestmark = pytest.mark.gpu

NAMES = ("means3D", "means3D_final", "means3D_prev", "means3D_nxt", "scales_final", "scales_prev", "scales_nxt",
         "shs_final", "shs_prev", "shs_nxt", "rotations_final", "rotations_prev", "rotations_next")


def _scene(n, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale
    base = r(n, 3, scale=0.05)                                 # dense cloud: exp(-100 d) weights are not all ~0
    t = {"means3D": base}
    for name in ("means3D_final", "means3D_prev", "means3D_nxt"):
        t[name] = base + r(n, 3, scale=0.01)
    for name in ("scales_final", "scales_prev", "scales_nxt"):
        t[name] = torch.exp(r(n, 3, scale=0.3) - 3)
    for name in ("shs_final", "shs_prev", "shs_nxt"):
        t[name] = r(n, 16, 3, scale=0.2)
    for name in ("rotations_final", "rotations_prev", "rotations_next"):
        t[name] = r(n, 4)
    return t


@pytest.mark.parametrize("gather", [False, True], ids=["atomics", "transposed_graph"])
@pytest.mark.parametrize("dim", [3, 4])
def test_neighbor_diff_norm_matches_torch_expression(dim, gather):
    rig = pkg("rigidity")
    n, k = 700, 100
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(n, dim, generator=g)
    idx = torch.randint(0, n, (1, n, k), generator=g)
    idx[0, :, 0] = torch.arange(n)                             # every point is its own first neighbour: distance 0
    up = torch.randn(1, n, k, generator=g)
    xc = x.clone().requires_grad_(True)
    want = RR.neighbor_diff_norm(xc, idx)
    (want * up).sum().backward()
    xg = x.cuda().requires_grad_(True)
    graph_t = rig.transpose_graph(idx.cuda()) if gather else None
    if gather:
        order, row_start = graph_t
        assert order.dtype == torch.int32 and row_start.dtype == torch.int32 and int(row_start[-1]) == n * k
        assert torch.equal(torch.sort(order.long()).values.cpu(), torch.arange(n * k))          # a permutation of the pairs
        tgt = idx.reshape(-1)[order.long().cpu()]
        assert bool((tgt[1:] >= tgt[:-1]).all())                                                 # grouped by target
    got = rig.neighbor_diff_norm(xg, idx.cuda(), graph_t)
    assert got.shape == (1, n, k)
    (got * up.cuda()).sum().backward()
    assert torch.allclose(got.cpu(), want.detach(), rtol=1e-6, atol=1e-7)
    assert float(got.detach()[0, :, 0].abs().max()) == 0.0
    assert torch.isfinite(xg.grad).all()
    assert float((xg.grad.cpu() - xc.grad).abs().max()) <= 2e-5 * float(xc.grad.abs().max())


def test_neighbor_diff_norm_edge_cases():
    rig = pkg("rigidity")
    x = torch.randn(5, 3, device="cuda", requires_grad=True)
    idx = torch.zeros(1, 5, 1, dtype=torch.int64, device="cuda")               # k = 1
    d = rig.neighbor_diff_norm(x, idx)
    assert torch.allclose(d[0, :, 0], torch.norm(x.detach()[0][None] - x.detach(), dim=-1), atol=1e-7)
    d.sum().backward()
    assert torch.isfinite(x.grad).all()
    empty = rig.neighbor_diff_norm(torch.zeros(0, 3, device="cuda"), torch.zeros(1, 0, 4, dtype=torch.int64, device="cuda"))
    assert empty.shape == (1, 0, 4)
    with pytest.raises(RuntimeError):
        rig.transpose_graph(torch.full((1, 5, 2), 5, dtype=torch.int64, device="cuda"))          # index out of range
    with pytest.raises(RuntimeError):
        rig.neighbor_diff_norm(torch.zeros(4, 5, device="cuda"), idx)
    with pytest.raises(RuntimeError):
        rig.neighbor_diff_norm(torch.zeros(4, 3), torch.zeros(1, 4, 2, dtype=torch.int64))     # CPU tensors: no CPU path


@pytest.mark.parametrize("n,k,skew", [(1, 1, False), (5, 2, False), (513, 7, True), (700, 100, False), (100_000, 30, False),
                                      (131_073, 9, True), (300_000, 8, True)],
                         ids=["one", "tiny", "two_narrow_passes", "one_wide_pass_pair", "c2_bits", "just_over_17_bits", "three_passes"])
def test_transpose_graph_is_the_stable_sort(n, k, skew):
    """gs4d_transpose_graph against torch.sort(stable=True) of the targets: order and row_start bit for bit (the stable
    permutation is unique).  skew: most pairs point at a few targets and many targets have no pair at all; the sizes
    cover one, two and three digit passes, a last tile that is not full and digits narrower than nine bits."""
    rig = pkg("rigidity")
    g = torch.Generator().manual_seed(n + k)
    if skew:
        hubs = torch.randint(0, n, (max(n // 50, 1),), generator=g)
        idx = hubs[torch.randint(0, hubs.numel(), (n, k), generator=g)]
        idx[::3] = torch.randint(0, n, idx[::3].shape, generator=g)
        idx[0, 0] = n - 1
    else:
        idx = torch.randint(0, n, (n, k), generator=g)
    idx = idx.to(torch.int32).cuda()
    order, row_start = rig.transpose_graph(idx[None], trusted=True)
    flat = idx.reshape(-1).long()
    want_t, want_o = torch.sort(flat, stable=True)
    assert order.dtype == torch.int32 and order.shape == (n * k,)
    assert torch.equal(order.long(), want_o)
    want_rs = torch.searchsorted(want_t, torch.arange(n + 1, device="cuda"))
    assert row_start.dtype == torch.int32 and torch.equal(row_start.long(), want_rs)
    # a second call on the same stream reuses the allocator's blocks: same answer
    order2, row_start2 = rig.transpose_graph(idx[None], trusted=True)
    assert torch.equal(order2, order) and torch.equal(row_start2, row_start)


def test_temporal_regulation_loss_of_two_terms_only():
    """A loss that names only some of the six terms (the fused Function then receives no gradient for the others): the
    gradients equal those of the composed, line-by-line path."""
    rig = pkg("rigidity")
    n, k = 400, 20
    base = _scene(n, 11)
    grads = {}
    for fused in (True, False):
        dev = {a: v.detach().clone().cuda().requires_grad_(True) for a, v in base.items()}
        got = rig.temporal_regulation(*[dev[a] for a in NAMES], k=k, fused=fused)
        (got["knn_xyz"] + 2.0 * got["shs"]).backward()
        grads[fused] = {a: (torch.zeros_like(dev[a]) if dev[a].grad is None else dev[a].grad.clone()) for a in NAMES}
    for a in NAMES:
        scale = float(grads[False][a].abs().max())
        assert float((grads[True][a] - grads[False][a]).abs().max()) <= 2e-5 * scale + 1e-12, a
    dev = {a: v.detach().clone().cuda().requires_grad_(True) for a, v in base.items()}
    got = rig.temporal_regulation(*[dev[a] for a in NAMES], k=k, fused=True)
    assert all(got[t].dim() == 0 for t in ("momentum", "scales", "shs", "knn_xyz", "knn_long", "knn_rot"))


@pytest.mark.parametrize("fused", [True, False], ids=["fused_terms", "composed"])
def test_temporal_regulation_matches_reference_expressions(fused):
    rig = pkg("rigidity")
    n, k = 600, 100
    cpu = {a: v.clone().requires_grad_(True) for a, v in _scene(n, 7).items()}
    idx, dist = RR.knn_bruteforce(cpu["means3D_final"].detach(), k)
    want = RR.temporal_regulation(*[cpu[a] for a in NAMES], idx, dist)
    dev = {a: v.detach().clone().cuda().requires_grad_(True) for a, v in cpu.items()}
    got = rig.temporal_regulation(*[dev[a] for a in NAMES], k=k, fused=fused)
    # distinct weights per term, so that a gradient routed to the wrong term cannot cancel out
    wts = {"momentum": 0.7, "scales": 1.3, "shs": 0.9, "knn_xyz": 1.1, "knn_long": 0.6, "knn_rot": 1.7}
    for q in cpu.values():
        q.grad = None
    sum(wts[key] * want[key] for key in want).backward()
    sum(wts[key] * got[key] for key in got).backward()
    assert set(got) == set(want) == {"momentum", "scales", "shs", "knn_xyz", "knn_long", "knn_rot"}
    for key in want:
        assert abs(float(got[key]) - float(want[key])) <= 1e-5 * abs(float(want[key])) + 1e-9, key
    for a in NAMES:
        if cpu[a].grad is None:
            assert dev[a].grad is None or float(dev[a].grad.abs().max()) == 0.0, a
            continue
        assert float((dev[a].grad.cpu() - cpu[a].grad).abs().max()) <= 2e-5 * float(cpu[a].grad.abs().max()) + 1e-12, a


def test_recurrent_regulation_and_small_clouds():
    rig = pkg("rigidity")
    n = 40                                                      # fewer points than k: k = min(100, n), renderer.py:160
    g = torch.Generator().manual_seed(3)
    t = [torch.randn(n, 3, generator=g) * 0.05 for _ in range(3)] + [torch.randn(n, 4, generator=g) for _ in range(2)] + \
        [torch.rand(n, 3, generator=g) for _ in range(2)] + [torch.randn(n, 16, 3, generator=g) for _ in range(2)]
    cpu = [v.clone().requires_grad_(True) for v in t]
    idx, dist = RR.knn_bruteforce(cpu[2].detach(), n)
    want = RR.recurrent_regulation(*cpu, idx, dist)
    sum(want.values()).backward()
    dev = [v.clone().cuda().requires_grad_(True) for v in t]
    got = rig.recurrent_regulation(*dev, k=100)
    sum(got.values()).backward()
    for key in want:
        assert abs(float(got[key]) - float(want[key])) <= 1e-5 * abs(float(want[key])) + 1e-9, key
    for a, b in zip(dev, cpu):
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-5 * float(b.grad.abs().max()) + 1e-12


def test_render_with_regularisers_matches_three_separate_queries():
    """render(..., regularise=True): the reference's fine-stage 7-tuple.  The batched 3P-point query must give the
    same dictionary as the reference's formulation (two more forward_dynamic calls at t -+ 1/(MAX_FRAME-1), scales
    through exp, rotations of the centre query normalised and the other two raw, renderer.py:145-146,175-182) on the
    oracle's expressions, and every term must reach the deformation parameters."""
    from test_gpu_deformation import make_args
    from oracle import deformation_ref as R
    RD, syn = pkg("renderer"), pkg("synthetic")
    torch.manual_seed(11)
    dev = torch.device("cuda")
    deform = pkg("scene.deformation").deform_network(make_args(R.DeformConfig(plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2]))).to(dev)
    _, cam_np, raw = syn.make_workload("C1", P=1500)
    pc = RD.GaussianParams(raw, dev)
    cam = RD.Camera(syn.look_at_camera(96, 64), dev, time=0.5)
    out = RD.render(cam, pc, deform, torch.ones(3, device=dev), regularise=True)
    assert len(out) == 7 and out[6] is None
    reg = out[5]
    n = 1500
    full = lambda v: torch.full((n, 1), v, device=dev)
    q = lambda t: deform.forward_dynamic(pc._xyz, pc._scaling, pc._rotation, pc._opacity, pc._features, full(t), full(0.0), None)
    dt = 1.0 / (RD.MAX_FRAME - 1)
    (m, s, r, _, h), (mp, sp, rp, _, hp), (mn, sn, rn, _, hn) = q(0.5), q(0.5 - dt), q(0.5 + dt)
    c = lambda v: v.detach().cpu()
    idx, dist = RR.knn_bruteforce(c(m), 100)
    want = RR.temporal_regulation(c(pc._xyz), c(m), c(mp), c(mn), c(torch.exp(s)), c(torch.exp(sp)), c(torch.exp(sn)), c(h), c(hp), c(hn),
                                  c(torch.nn.functional.normalize(r)), c(rp), c(rn), idx, dist)
    for key in want:
        assert abs(float(reg[key]) - float(want[key])) <= 2e-5 * abs(float(want[key])) + 1e-9, key
    (out[0].mean() + sum(reg.values())).backward()
    assert pc._xyz.grad is not None and torch.isfinite(pc._xyz.grad).all()
    planes = [p for level in deform.deformation_net.grid.grids for p in level]
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in planes)
    # very large scenes query the field three times instead of once (BATCHED_QUERY_MAX_POINTS): same image, same
    # dictionary, same gradients (plane gradients are float-atomic sums: 1e-5 of their magnitude)
    batched_grads = {"xyz": pc._xyz.grad.clone(), "planes": [p.grad.clone() for p in planes]}
    for p in list(pc.parameters()) + list(deform.parameters()):
        p.grad = None
    saved = RD.BATCHED_QUERY_MAX_POINTS
    RD.BATCHED_QUERY_MAX_POINTS = 0
    try:
        out2 = RD.render(cam, pc, deform, torch.ones(3, device=dev), regularise=True)
    finally:
        RD.BATCHED_QUERY_MAX_POINTS = saved
    assert torch.equal(out2[0], out[0])
    for key in reg:
        assert abs(float(out2[5][key]) - float(reg[key])) <= 1e-6 * abs(float(reg[key])) + 1e-12, key
    (out2[0].mean() + sum(out2[5].values())).backward()
    tol = lambda a, b: float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12
    assert tol(pc._xyz.grad, batched_grads["xyz"])
    assert all(tol(p.grad, g) for p, g in zip(planes, batched_grads["planes"]))
    # the coarse stage has no regulariser terms (renderer.py:152-153)
    coarse = RD.render(cam, pc, deform, torch.ones(3, device=dev), stage="coarse", regularise=True)
    assert coarse[5] == {}
