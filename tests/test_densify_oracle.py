"""Adaptive density control: the product's GaussianModel.densify / prune / reset_opacity /
add_densification_stats against oracle/densify_ref.py, an independent numpy restatement of
scene/gaussian_model.py:297-300,343-458,583-597,613-615 that knows nothing of torch or of optimizer objects.

PARITY UNPINNED against recorded reference outputs (the reference holds none and its module is not importable here, see
the oracle's header).  Every array that the surgery only gathers / concatenates -- parameters, both Adam moments, the
densification statistics, the deformation table -- must match BIT FOR BIT, row order included; the split's new
positions (rotation matrix x sample) and log-scales, and the reset opacities, to 1e-6 (float32 expression order of
numpy vs torch).  The Gaussian samples of the split are recorded from the product's torch.normal call and handed to the
oracle: the draw is the framework's generator, not the algorithm.

The CPU test injects torch.optim.AdamW (the reference's optimizer); the GPU test runs the same scenario with the fused
HIP AdamW on the device."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import densify_ref as D
from test_gaussian_model_host import opt_args
from test_gpu_deformation import make_args
from oracle import deformation_ref as R

GM = pkg("scene.gaussian_model")
NAMES = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
         "rotation": "_rotation"}
EXTENT, PERCENT_DENSE = 4.0, 0.01


def build(n, device, optimizer_cls, seed=0):
    cfg = R.DeformConfig(plane_channels=8, resolution=[4, 4, 4, 3], multires=[1])
    torch.manual_seed(seed)
    m = GM.GaussianModel(3, make_args(cfg), device=device, optimizer_cls=optimizer_cls)
    g = torch.Generator().manual_seed(seed)
    P = lambda t: torch.nn.Parameter(t.to(device))
    m._xyz, m._features_dc, m._features_rest = P(torch.randn(n, 3, generator=g)), P(torch.randn(n, 1, 3, generator=g)), P(torch.randn(n, 15, 3, generator=g))
    m._opacity = P(torch.randn(n, 1, generator=g) * 3)
    # log-scales around log(percent_dense * extent) = -3.2 so that both the clone and the split branch fire
    m._scaling, m._rotation = P(torch.randn(n, 3, generator=g) * 0.8 - 3.6), P(torch.randn(n, 4, generator=g))
    m._deformation_table = (torch.rand(n, generator=g) > 0.3).to(device)
    m.max_radii2D = (torch.rand(n, generator=g) * 60).to(device)
    m.spatial_lr_scale = 5.0
    m._deformation = m._deformation.to(device)
    m.training_setup(opt_args())
    m.percent_dense = PERCENT_DENSE
    return m, g


def optimizer_step(m, g):
    for grp in m.optimizer.param_groups:
        if len(grp["params"]) == 1:
            p = grp["params"][0]
            p.grad = torch.randn(p.shape, generator=g).to(p.device)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)


def snapshot(m):
    s = {}
    for k, attr in NAMES.items():
        p = getattr(m, attr)
        st = m.optimizer.state[p]
        s[k] = p.detach().cpu().numpy().copy()
        s["m_" + k], s["v_" + k] = st["exp_avg"].cpu().numpy().copy(), st["exp_avg_sq"].cpu().numpy().copy()
    s["xyz_gradient_accum"], s["denom"] = m.xyz_gradient_accum.cpu().numpy().copy(), m.denom.cpu().numpy().copy()
    s["max_radii2D"], s["deformation_accum"] = m.max_radii2D.cpu().numpy().copy(), m._deformation_accum.cpu().numpy().copy()
    s["deformation_table"] = m._deformation_table.cpu().numpy().copy()
    return s


def compare(m, s, approx=(), what=""):
    got = snapshot(m)
    assert got["xyz"].shape == s["xyz"].shape, (what, got["xyz"].shape, s["xyz"].shape)
    for k, want in s.items():
        assert got[k].shape == want.shape and got[k].dtype == want.dtype, (what, k, got[k].shape, want.shape, got[k].dtype, want.dtype)
        if k in approx:
            np.testing.assert_allclose(got[k], want, rtol=2e-6, atol=1e-6 * float(np.abs(want).max()), err_msg=f"{what}: {k}")
        else:
            np.testing.assert_array_equal(got[k], want, err_msg=f"{what}: {k}")
    # the parameter objects the model holds are the ones the optimizer steps
    for grp in m.optimizer.param_groups:
        if grp["name"] in NAMES:
            assert grp["params"][0] is getattr(m, NAMES[grp["name"]]), grp["name"]


def scenario(device, optimizer_cls):
    n = 600
    m, g = build(n, device, optimizer_cls)
    optimizer_step(m, g)
    optimizer_step(m, g)
    s = snapshot(m)
    compare(m, s, what="initial")
    report = {}
    for rnd in range(2):
        n = m.get_xyz.shape[0]
        # three views' worth of screen-space statistics; some Gaussians are never visible (0 / 0 -> NaN -> 0)
        for _ in range(3):
            vg = torch.randn(n, 3, generator=g) * 4e-4
            flt = torch.rand(n, generator=g) > 0.4
            m.add_densification_stats(vg.to(device), flt.to(device))
            D.add_densification_stats(s, vg.numpy(), flt.numpy())
        m._deformation_accum = torch.rand(n, 3, generator=g).to(device)
        s["deformation_accum"] = m._deformation_accum.cpu().numpy().copy()
        # torch.norm and numpy's sqrt(sum of squares) differ in the last bit: compared to 1e-6, then the product's sums
        # are carried forward so that the threshold decisions below are taken on identical numbers
        compare(m, s, approx=("xyz_gradient_accum",), what=f"round {rnd}: statistics")
        s["xyz_gradient_accum"] = m.xyz_gradient_accum.cpu().numpy().copy()

        drawn = []
        real_normal = torch.normal

        def recording_normal(*a, **k):
            out = real_normal(*a, **k)
            drawn.append((k["std"].detach().cpu().numpy().copy(), out.detach().cpu().numpy().copy()))
            return out

        torch.normal = recording_normal
        try:
            m.densify(0.0004, 0.005, EXTENT, 40, 5, 5)
        finally:
            torch.normal = real_normal

        def replay(stds):
            assert len(drawn) == 1, "the split draws once (gaussian_model.py:446)"
            np.testing.assert_allclose(drawn[0][0], stds, rtol=1e-6)
            return drawn[0][1]

        sel = D.densify(s, 0.0004, EXTENT, PERCENT_DENSE, replay)
        report[f"round {rnd}"] = dict(P_before=n, cloned=int(sel["cloned"].sum()), split=int(sel["split"].sum()), P_after=s["xyz"].shape[0])
        assert sel["cloned"].sum() > 10 and sel["split"].sum() > 10, report       # both branches exercised
        compare(m, s, approx=("xyz", "scaling"), what=f"round {rnd}: densify")
        # carry the product's float32 rounding of the new rows forward so that later stages compare bit for bit again
        s["xyz"], s["scaling"] = m._xyz.detach().cpu().numpy().copy(), m._scaling.detach().cpu().numpy().copy()

        m.max_radii2D = (torch.rand(m.get_xyz.shape[0], generator=g) * 60).to(device)
        s["max_radii2D"] = m.max_radii2D.cpu().numpy().copy()
        m.prune(0.0004, 0.005, EXTENT, 40 if rnd else None)
        pm = D.prune(s, 0.005, EXTENT, 40 if rnd else None)
        report[f"round {rnd}"]["pruned"] = int(pm.sum())
        assert pm.sum() > 5, report
        compare(m, s, what=f"round {rnd}: prune")
        optimizer_step(m, g)                                   # state and parameters still line up
        s = snapshot(m)
    m.reset_opacity()
    D.reset_opacity(s)
    compare(m, s, approx=("opacity",), what="reset_opacity")
    print("densify vs oracle:", report)


def test_densify_prune_reset_match_numpy_oracle_cpu():
    scenario("cpu", torch.optim.AdamW)


@pytest.mark.gpu
def test_densify_prune_reset_match_numpy_oracle_gpu_fused_adamw():
    scenario(torch.device("cuda", 0), None)
