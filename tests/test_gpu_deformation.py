"""GPU parity of the HIP deformation field (through the reference-shaped deform_network module and the
C ABI) against (a) golden vectors produced by the reference's own Python and (b) the CPU oracle at a
larger size.  Tolerances: forward 1e-5 (abs, relative to max(1, |x|max)), gradients 1e-4 relative to the
tensor's max (plane gradients are summed with float atomics)."""
import types

import numpy as np
import pytest
import torch

from conftest import pkg
from golden_utils import CASES, load_case

pytestmark = pytest.mark.gpu
ORDER = ("means3D", "scales", "rotations", "opacity", "shs")


def make_args(cfg):
    a = types.SimpleNamespace()
    a.net_width, a.defor_depth, a.posebase_pe = cfg.net_width, cfg.defor_depth, cfg.posebase_pe
    a.scale_rotation_pe, a.opacity_pe, a.grid_pe = cfg.scale_rotation_pe, cfg.opacity_pe, 0
    a.force_pe, a.time_pe, a.bounds = cfg.force_pe, cfg.time_pe, cfg.bounds
    a.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": cfg.plane_channels,
                        "resolution": list(cfg.resolution)}
    a.multires = list(cfg.multires)
    a.no_dx, a.no_ds, a.no_dr, a.no_do, a.no_dshs = cfg.no_dx, cfg.no_ds, cfg.no_dr, cfg.no_do, cfg.no_dshs
    a.empty_voxel, a.static_mlp, a.apply_rotation = False, False, cfg.apply_rotation
    a.use_time, a.use_force, a.blend_time_force, a.prev_frames, a.no_grid = cfg.use_time, cfg.use_force, True, 0, False
    return a


def build_net(cfg, sd):
    D = pkg("scene.deformation")
    net = D.deform_network(make_args(cfg))
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected                      # every live reference key exists here
    assert all(any(s in m for s in ("prev_frame_encoder", "feature_out1", "force_embedder", "feature_aggr", "opacity_deform"))
               for m in missing), missing                  # only the dead parameters were left out of the fixture
    return net.cuda()


def close(name, got, want, tol):
    got, want = got.detach().cpu().double(), want.double()
    err = float((got - want).abs().max()) / max(1.0, float(want.abs().max())) if tol == "fwd" else \
        float((got - want).abs().max()) / max(float(want.abs().max()), 1e-20)
    lim = 1e-5 if tol == "fwd" else 1e-4
    assert err <= lim, (name, err)


@pytest.mark.parametrize("case", CASES)
def test_matches_reference_golden(case):
    cfg, sd, ins, outs, ws, gins, grads = load_case(case)
    net = build_net(cfg, sd)
    x = {k: ins[k].cuda().requires_grad_(True) for k in ("points", "scales", "rotations", "opacity", "shs")}
    got = net.forward_dynamic(x["points"], x["scales"], x["rotations"], x["opacity"], x["shs"], ins["time"].cuda(),
                              ins["force"].cuda(), None)
    for n, g in zip(ORDER, got):
        close(n, g, outs[n], "fwd")
    sum((g * ws[n].cuda()).sum() for n, g in zip(ORDER, got)).backward()
    for k, want in gins.items():
        if want.numel():
            close("d" + k, x[k].grad, want, "bwd")
    params = dict(net.named_parameters())
    for k, want in grads.items():
        g = params[k].grad
        assert g is not None, k
        close(k, g, want, "bwd")


def test_matches_oracle_at_scale_mutant_shape():
    """C2's network shape (arguments/dnerf/others/mutant.py + dnerf_default.py) on 20k points vs the CPU oracle."""
    from oracle import deformation_ref as R
    cfg = R.DeformConfig(defor_depth=0, plane_channels=32, resolution=[64, 64, 64, 75], multires=[1, 2])
    torch.manual_seed(3)
    net = pkg("scene.deformation").deform_network(make_args(cfg))
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "deform" in n and "bias" in n:
                p.add_(0.02 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    P = 20000
    g = torch.Generator().manual_seed(5)
    ins = dict(points=torch.rand(P, 3, generator=g) * 3.4 - 1.7, scales=torch.randn(P, 3, generator=g) * 0.5 - 3,
               rotations=torch.randn(P, 4, generator=g), opacity=torch.randn(P, 1, generator=g),
               shs=torch.randn(P, 16, 3, generator=g) * 0.3)
    time, force = torch.full((P, 1), 0.61), torch.zeros(P, 1)
    # A ReLU input within float32 rounding of zero is a discrete decision that two correct float32
    # implementations (different summation order) may take differently; one flipped unit of one point moves
    # a weight gradient by ~1e-3 of its max.  Points whose closest ReLU input is below 1e-5 (|accumulated
    # rounding| is ~1e-6) are dropped from the comparison set, as the rasterizer tests do with `margin`.
    cap = {}
    with torch.no_grad():
        R.forward_dynamic(sd, cfg, ins["points"], ins["scales"], ins["rotations"], ins["opacity"], ins["shs"], time, force, cap)
    keep = cap["relu_margin"] >= 1e-5
    assert keep.float().mean() > 0.9, float(keep.float().mean())
    ins = {k: v[keep].contiguous() for k, v in ins.items()}
    P = int(keep.sum())
    time, force = time[:P], force[:P]
    ws = [torch.randn(s, generator=g) for s in ((P, 3), (P, 3), (P, 4), (P, 1), (P, 16, 3))]
    # oracle
    sdo = {k: v.clone().requires_grad_(v.dtype == torch.float32 and "poc" not in k and "aabb" not in k) for k, v in sd.items()}
    xo = {k: v.clone().requires_grad_(True) for k, v in ins.items()}
    want = R.forward_dynamic(sdo, cfg, xo["points"], xo["scales"], xo["rotations"], xo["opacity"], xo["shs"], time, force)
    sum((o * w).sum() for o, w in zip(want, ws)).backward()
    # HIP
    net = net.cuda()
    xg = {k: v.cuda().requires_grad_(True) for k, v in ins.items()}
    got = net.forward_dynamic(xg["points"], xg["scales"], xg["rotations"], xg["opacity"], xg["shs"], time.cuda(), force.cuda(), None)
    for n, a, b in zip(ORDER, got, want):
        close(n, a, b.detach(), "fwd")
    sum((o * w.cuda()).sum() for o, w in zip(got, ws)).backward()
    for k in ("points", "scales", "rotations", "shs"):
        close("d" + k, xg[k].grad, xo[k].grad, "bwd")
    for k, p in net.named_parameters():
        if sdo[k].grad is not None:
            assert p.grad is not None, k
            close(k, p.grad, sdo[k].grad, "bwd")


def test_unsupported_configurations_fail_loudly():
    from oracle import deformation_ref as R
    cfg = R.DeformConfig(plane_channels=8, resolution=[4, 4, 4, 3], multires=[1])
    a = make_args(cfg)
    a.no_do = False
    net = pkg("scene.deformation").deform_network(a).cuda()
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(NotImplementedError):
        net.forward_dynamic(z(4, 3), z(4, 3), z(4, 4), z(4, 1), z(4, 16, 3), z(4, 1), z(4, 1), None)
    a.no_do = True
    with pytest.raises(RuntimeError, match="HIP device"):
        net.cpu().forward_dynamic(*(t.cpu() for t in (z(4, 3), z(4, 3), z(4, 4), z(4, 1), z(4, 16, 3), z(4, 1), z(4, 1))), None)


def test_set_aabb_and_extra_deform_switches_take_effect_after_a_forward():
    """the per-step plan cached by the first forward_dynamic must not outlive Deformation.set_aabb
    (a NEW aabb Parameter) or enable_/disable_extra_deform (descriptor flags): compare with the oracle after each."""
    from oracle import deformation_ref as R
    cfg = R.DeformConfig(defor_depth=1, plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2])
    torch.manual_seed(11)
    args = make_args(cfg)
    net = pkg("scene.deformation").deform_network(args)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "deform" in n and "bias" in n:
                p.add_(0.05 * torch.randn_like(p))
    net = net.cuda()
    P = 3000
    g = torch.Generator().manual_seed(2)
    ins = [torch.rand(P, 3, generator=g) * 2 - 1, torch.randn(P, 3, generator=g) * 0.5 - 3, torch.randn(P, 4, generator=g),
           torch.randn(P, 1, generator=g), torch.randn(P, 16, 3, generator=g) * 0.3]
    time, force = torch.full((P, 1), 0.37), torch.zeros(P, 1)

    def both(cfg_now):
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            want = R.forward_dynamic(sd, cfg_now, *ins, time, force)
            got = net.forward_dynamic(*(t.cuda() for t in ins), time.cuda(), force.cuda(), None)
        for n, a, b in zip(ORDER, got, want):
            close(n, a, b, "fwd")
        return [t.cpu() for t in got]

    first = both(cfg)
    net.deformation_net.set_aabb([2.5, 2.0, 1.5], [-1.5, -2.0, -2.5])          # after a forward has cached its plan
    second = both(cfg)
    assert float((first[0] - second[0]).abs().max()) > 1e-4                     # the new bounds really are in use
    net.deformation_net.disable_extra_deform()
    import dataclasses
    third = both(dataclasses.replace(cfg, no_ds=True, no_dr=True, no_dshs=True))
    assert torch.equal(third[1], ins[1]) and torch.equal(third[4], ins[4])      # scales / shs now pass through
    net.deformation_net.enable_extra_deform()
    fourth = both(cfg)
    for a, b in zip(second, fourth):
        assert torch.equal(a, b)


def test_cached_point_order_changes_no_gradient():
    """The plane-gradient pipeline may walk the points in the order an EARLIER backward recorded (HexOrderCache,
    gs4d_deform_grads::hex_order): the order only decides which adds are combined before they reach memory.  Same inputs:
    the gradients of the recording pass and of a reusing pass agree to the rounding of the float atomics; after every
    point has moved somewhere else (the cached order is then as bad as a random one) they still agree with a fresh sort."""
    from oracle import deformation_ref as R
    cfg = R.DeformConfig(defor_depth=1, plane_channels=16, resolution=[32, 32, 32, 8], multires=[1, 2])
    torch.manual_seed(5)
    net = pkg("scene.deformation").deform_network(make_args(cfg))
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "deform" in n and "bias" in n:
                p.add_(0.05 * torch.randn_like(p))
    net = net.cuda()
    dn = net.deformation_net
    P = 20000
    g = torch.Generator().manual_seed(3)
    base = [torch.rand(P, 3, generator=g) * 2 - 1, torch.randn(P, 3, generator=g) * 0.5 - 3, torch.randn(P, 4, generator=g),
            torch.randn(P, 1, generator=g), torch.randn(P, 16, 3, generator=g) * 0.3]
    time, force = torch.full((P, 1), 0.37).cuda(), torch.zeros(P, 1).cuda()
    w = [torch.randn_like(t).cuda() for t in base]
    planes = [p for level in dn.grid.grids for p in level]

    def grads(points):
        x = [points.cuda().requires_grad_(True)] + [t.cuda().requires_grad_(True) for t in base[1:]]
        for p in net.parameters():
            p.grad = None
        out = net.forward_dynamic(*x, time, force, None)
        sum((o * wi).sum() for o, wi in zip(out, w)).backward()
        return [p.grad.clone() for p in planes] + [x[0].grad.clone()]

    def same(a, b, tol):
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= tol * max(float(v.abs().max()), 1e-20)

    recorded = grads(base[0])                                  # sorts, leaves the order in the cache
    cache = dn.__dict__["_hex_order_cache"]
    entry = cache.entries[(P, 2, 0)]
    assert entry[1] == 1
    reused = grads(base[0])                                    # walks the cached order
    assert entry[1] == 2
    same(reused, recorded, 2e-6)
    moved = base[0][torch.randperm(P, generator=g)]            # every point somewhere else: the cached order is stale
    stale = grads(moved)
    assert entry[1] == 3
    cache.clear()
    fresh = grads(moved)
    assert cache.entries[(P, 2, 0)][1] == 1
    # a stale (here: random) order flushes almost every point on its own: ~6x more float atomics per texel than the sorted
    # walk, hence more rounding noise between the two sums (2.2e-6 of the maximum seen; the sorted pairs stay within 2e-6)
    same(stale, fresh, 5e-6)
    entry = cache.entries[(P, 2, 0)]
    entry[1] = cache.refresh                                   # due for a refresh: the next backward sorts again
    again = grads(moved)
    assert entry[1] == 1
    same(again, fresh, 2e-6)
