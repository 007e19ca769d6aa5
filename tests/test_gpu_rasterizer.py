"""GPU parity tests: the HIP rasterizer (through the drop-in Python API and the C ABI of
libgs4d.so) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer / indexing outputs bit-exact; colour, depth <= 1e-5 abs
(depth relative to its magnitude); gradients <= 1e-4 where "relative" means: max |got - want| over the tensor divided
by max |want| over the tensor (the definition every gradient check in this repository uses), for ALL eight tensors
(round 1 held dL_dscales / dL_drotations to 3e-4; the per-Gaussian kernel now sums its instance rows in double).
Pixels whose discrete blend decisions sit within 2e-5 (relative) of a threshold in the oracle
are excluded from the float comparisons (exp() differs in the last ulp between libm and
v_exp_f32; such a flip changes a pixel by up to alpha*T ~ 4e-3) and their count is bounded.  Every check prints the
excluded fraction, the largest error ON the excluded pixels and the achieved errors (pytest -rP / -s shows them);
tests/test_gpu_exact_exp.py shows that the exclusion is only needed because of exp().
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import rasterizer_oracle as O

pytestmark = pytest.mark.gpu
syn = pkg("synthetic")


def dgr():
    return pkg("diff_gaussian_rasterization")


def dev(x, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t.requires_grad_(grad)


def scene(P, W, H, seed, s0, deg=3):
    cam = syn.look_at_camera(W, H)
    a = syn.activate(syn.make_gaussians(P, seed, s0))
    return a, cam


def settings(cam, deg, bg=(1, 1, 1), debug=False, prefiltered=False, scale_modifier=1.0):
    return dgr().GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=torch.tensor(bg, dtype=torch.float32, device="cuda"), scale_modifier=scale_modifier,
        viewmatrix=dev(cam.viewmatrix), projmatrix=dev(cam.projmatrix), sh_degree=deg, campos=dev(cam.campos),
        prefiltered=prefiltered, debug=debug)


def run_c(a, cam, deg, colors=None, cov=None, bg=(1, 1, 1), scale_modifier=1.0):
    """Direct call of the _C-level function (returns buffers for the state view)."""
    C_ = dgr()._C
    e = torch.Tensor([])
    out = C_.rasterize_gaussians(
        torch.tensor(bg, dtype=torch.float32, device="cuda"), dev(a["means3D"]), e if colors is None else dev(colors),
        dev(a["opacities"]), e if cov is not None else dev(a["scales"]), e if cov is not None else dev(a["rotations"]),
        scale_modifier, e if cov is None else dev(cov), dev(cam.viewmatrix), dev(cam.projmatrix), cam.tanfovx, cam.tanfovy,
        cam.image_height, cam.image_width, e if colors is not None else dev(a["sh"]), deg, dev(cam.campos), False, False)
    return out


def run_oracle(a, cam, deg, colors=None, cov=None, bg=(1, 1, 1), scale_modifier=1.0):
    return O.forward(np.asarray(bg, np.float32), a["means3D"], colors, a["opacities"],
                     None if cov is not None else a["scales"], None if cov is not None else a["rotations"],
                     scale_modifier, cov, cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy,
                     cam.image_height, cam.image_width, None if colors is not None else a["sh"], deg, cam.campos)


LAST_FORWARD_REPORT = {}      # filled by check_forward: what was excluded and how large the error is on it
LAST_GRADIENT_ERRORS = {}     # filled by grads_close: name -> achieved relative error


def check_forward(a, cam, deg, **kw):
    R, color, depth, radii, gb, bb, ib = run_c(a, cam, deg, **kw)
    r = run_oracle(a, cam, deg, **kw)
    P, W, H = r["P"], r["W"], r["H"]
    assert R == r["num_rendered"]
    v = dgr()._C.state_view(P, R, W, H, gb, bb, ib)
    np.testing.assert_array_equal(radii.cpu().numpy(), r["radii"])
    vis = r["radii"] > 0
    eq = lambda name, got, want: np.testing.assert_array_equal(got, want, err_msg=name)
    eq("tiles_touched", v["tiles_touched"].cpu().numpy().view(np.uint32), r["tiles_touched"])
    eq("point_offsets", v["point_offsets"].cpu().numpy().view(np.uint32), r["point_offsets"])
    # per-Gaussian floats are evaluated in the same order without contraction: bit-exact on visible Gaussians
    for name in ("depths", "means2D", "conic_opacity"):
        eq(name, v[name].cpu().numpy()[vis].view(np.uint32), r[name][vis].view(np.uint32))
    if kw.get("cov") is None:
        front = (np.c_[a["means3D"], np.ones(P, np.float32)] @ cam.viewmatrix)[:, 2] > 0.2
        eq("cov3D", v["cov3D"].cpu().numpy()[front & vis].view(np.uint32), r["cov3D"][front & vis].view(np.uint32))
    if kw.get("colors") is None:
        eq("rgb", v["rgb"].cpu().numpy()[vis].view(np.uint32), r["rgb"][vis].view(np.uint32))
        eq("clamped", v["clamped"].cpu().numpy()[vis], r["clamped"][vis])
    eq("keys_unsorted", v["keys_unsorted"].cpu().numpy().view(np.uint64), r["keys_unsorted"])
    eq("values_unsorted", v["values_unsorted"].cpu().numpy().view(np.uint32), r["values_unsorted"])
    eq("keys_sorted", v["keys_sorted"].cpu().numpy().view(np.uint64), r["keys_sorted"])
    eq("point_list", v["point_list"].cpu().numpy().view(np.uint32), r["point_list"])
    eq("ranges", v["ranges"].cpu().numpy().view(np.uint32), r["ranges"])
    ok = r["margin"] > 2e-5
    assert ok.mean() > 0.995, ok.mean()
    cerr = np.abs(color.cpu().numpy() - r["color"])
    nc = v["n_contrib"].cpu().numpy().view(np.uint32)
    LAST_FORWARD_REPORT.clear()
    LAST_FORWARD_REPORT.update(
        P=P, W=W, H=H, excluded_fraction=float(1 - ok.mean()), colour_err_kept=float(cerr[:, ok].max()),
        colour_err_excluded=float(cerr[:, ~ok].max()) if (~ok).any() else 0.0,
        n_contrib_mismatch_excluded=int((nc[~ok] != r["n_contrib"][~ok]).sum()),
        n_contrib_mismatch_all=int((nc != r["n_contrib"]).sum()))
    print("forward parity:", LAST_FORWARD_REPORT)
    assert cerr[:, ok].max() <= 1e-5
    dmax = max(1.0, float(r["depths"][vis].max()) if vis.any() else 1.0)
    assert np.abs(depth.cpu().numpy() - r["depth"])[:, ok].max() <= 1e-5 * dmax
    assert np.abs(v["final_T"].cpu().numpy() - r["final_T"])[ok].max() <= 1e-5
    eq("n_contrib", nc[ok], r["n_contrib"][ok])
    return r, (R, color, depth, radii, gb, bb, ib)


@pytest.mark.parametrize("P,W,H,seed,s0,deg", [
    (300, 64, 48, 0, 0.08, 3),
    (1000, 70, 50, 1, 0.05, 2),      # image not a multiple of the tile size
    (2000, 200, 120, 2, 0.03, 1),
    (5000, 160, 160, 3, 0.2, 0),     # huge Gaussians: every one covers many tiles, long lists, T-termination
    (1, 33, 17, 4, 0.3, 3),
])
def test_forward_parity(P, W, H, seed, s0, deg):
    a, cam = scene(P, W, H, seed, s0)
    check_forward(a, cam, deg)


def test_forward_parity_c1_full_size(synthetic):
    """BASELINE.json configs[0]: 800x800, 40k Gaussians, SH degree 3 -- full-size parity against the oracle."""
    a, cam, _ = synthetic.make_workload("C1")
    r, _ = check_forward(a, cam, 3)
    assert 6.0 < r["num_rendered"] / r["P"] < 10.0


def test_forward_precomputed_paths_and_bg_and_scale_modifier():
    a, cam = scene(800, 96, 80, 7, 0.06)
    r0 = run_oracle(a, cam, 3)
    col = np.random.default_rng(0).uniform(0, 1, (800, 3)).astype(np.float32)
    check_forward(a, cam, 0, colors=col, cov=r0["cov3D"].copy(), bg=(0.1, 0.5, 0.9))
    check_forward(a, cam, 3, scale_modifier=0.5, bg=(0, 0, 0))


def test_duplicate_depths_keep_index_order():
    a, cam = scene(600, 64, 64, 9, 0.1)
    a["means3D"][300:] = a["means3D"][:300]
    check_forward(a, cam, 3)


# dL_dscales / dL_drotations are derived from dL_dcov3D through a cancelling chain; with float32 accumulators the
# reference algorithm's own summation-order noise reaches ~1e-4 of their max there
# (tests/test_oracle_rasterizer.py::test_float32_accumulation_noise_floor).  The HIP path sums the per-tile partial
# rows of a Gaussian in double (csrc/geom_bwd.hip), which brings these two tensors inside north_star's 1e-4 as well.
# Where one of these two still exceeds 1e-4 on some input, the check states the deviation next to the number: it
# runs the oracle a second time with float32 accumulators (float_accum=True: a fixed-order stand-in for the
# reference's float atomics) and requires the HIP error to stay within 1.5x of what the REFERENCE ALGORITHM's own
# float32 arithmetic deviates from the exact sums on the same input (`floor` callable; printed).
GRAD_TOL = 1e-4
CHAIN_TENSORS = ("dL_dscales", "dL_drotations", "scales", "rotations")


def grads_close(name, got, want, tol=None, floor=None):
    tol = GRAD_TOL if tol is None else tol
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = max(np.abs(want).max(), 1e-20)
    err = np.abs(got - want).max() / scale
    LAST_GRADIENT_ERRORS[name] = float(err)
    if err > tol and floor is not None and name in CHAIN_TENSORS:
        noise = float(np.abs(np.asarray(floor(name), np.float64).reshape(want.shape) - want).max() / scale)
        LAST_GRADIENT_ERRORS[name + " (> 1e-4; reference float32 noise floor on this input)"] = noise
        print(f"{name}: HIP error {err:.3e} exceeds {tol:.0e}; the reference algorithm with float32 accumulators "
              f"deviates by {noise:.3e} from the same exact sums on this input")
        assert err <= 1.5 * noise, (name, err, noise)
        return
    assert err <= tol, (name, err)


def check_backward(a, cam, deg, seed=0, depth_grad=0.1, **kw):
    r, (R, color, depth, radii, gb, bb, ib) = check_forward(a, cam, deg, **kw)
    H, W, P = r["H"], r["W"], r["P"]
    ok = r["margin"] > 2e-5
    rng = np.random.default_rng(seed)
    gc = (rng.standard_normal((3, H, W)).astype(np.float32) * ok[None]).astype(np.float32)
    gd = (rng.standard_normal((1, H, W)).astype(np.float32) * ok[None] * depth_grad).astype(np.float32)
    want = O.backward(r, gc, gd)
    C_ = dgr()._C
    e = torch.Tensor([])
    colors, cov = kw.get("colors"), kw.get("cov")
    bg = kw.get("bg", (1, 1, 1))

    def call():
        return C_.rasterize_gaussians_backward(
            torch.tensor(bg, dtype=torch.float32, device="cuda"), dev(a["means3D"]), radii,
            e if colors is None else dev(colors), e if cov is not None else dev(a["scales"]),
            e if cov is not None else dev(a["rotations"]), kw.get("scale_modifier", 1.0), e if cov is None else dev(cov),
            dev(cam.viewmatrix), dev(cam.projmatrix), cam.tanfovx, cam.tanfovy, dev(gc), dev(gd),
            e if colors is not None else dev(a["sh"]), deg, dev(cam.campos), gb, R, bb, ib, False)

    LAST_GRADIENT_ERRORS.clear()
    floor_cache = {}

    def noise_floor(name):                      # computed only if a chain tensor misses 1e-4
        if not floor_cache:
            floor_cache.update(O.backward(r, gc, gd, float_accum=True))
        return floor_cache[name]

    got = call()
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    for n, g in zip(names, got):
        w = want[n]
        if n == "dL_dsh" and colors is not None:
            assert g.numel() == 0
            continue
        grads_close(n, g.cpu().numpy().reshape(w.shape), w, floor=noise_floor)
    print("backward parity (max err / max |want|):", {k: f"{v:.2e}" for k, v in LAST_GRADIENT_ERRORS.items()})
    again = call()
    for n, g1, g2 in zip(names, got, again):  # no atomics anywhere: bitwise reproducible
        assert torch.equal(g1, g2), n
    return want


@pytest.mark.parametrize("P,W,H,seed,s0,deg", [
    (300, 64, 48, 0, 0.08, 3),
    (1000, 70, 50, 1, 0.05, 2),
    (5000, 160, 160, 3, 0.2, 0),     # > 64 tiles per Gaussian: exercises the deferred (workgroup-summed) path
    (3000, 256, 144, 5, 0.03, 1),
])
def test_backward_parity(P, W, H, seed, s0, deg):
    a, cam = scene(P, W, H, seed, s0)
    check_backward(a, cam, deg, seed=seed)


def test_backward_precomputed_paths():
    a, cam = scene(800, 96, 80, 7, 0.06)
    r0 = run_oracle(a, cam, 3)
    col = np.random.default_rng(0).uniform(0, 1, (800, 3)).astype(np.float32)
    check_backward(a, cam, 0, colors=col, cov=r0["cov3D"].copy(), bg=(0.1, 0.5, 0.9))


def test_backward_parity_c1_full_size(synthetic):
    a, cam, _ = synthetic.make_workload("C1")
    check_backward(a, cam, 3, depth_grad=0.0)


def test_autograd_api_matches_oracle_and_reference_contract():
    D = dgr()
    a, cam = scene(500, 80, 64, 11, 0.07)
    rast = D.GaussianRasterizer(settings(cam, 3))
    t = {k: dev(a[k], True) for k in ("means3D", "opacities", "sh", "scales", "rotations")}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    color, radii, depth = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["sh"],
                               scales=t["scales"], rotations=t["rotations"])
    assert color.shape == (3, 64, 80) and depth.shape == (1, 64, 80) and radii.dtype == torch.int32
    r = run_oracle(a, cam, 3)
    ok = r["margin"] > 2e-5
    gc = (np.random.default_rng(1).standard_normal((3, 64, 80)) * ok[None]).astype(np.float32)
    (color * dev(gc)).sum().backward()
    want = O.backward(r, gc, np.zeros((1, 64, 80), np.float32))
    grads_close("means3D", t["means3D"].grad.cpu().numpy(), want["dL_dmeans3D"])
    grads_close("means2D", means2D.grad.cpu().numpy(), want["dL_dmeans2D"])
    grads_close("opacities", t["opacities"].grad.cpu().numpy(), want["dL_dopacity"])
    grads_close("sh", t["sh"].grad.cpu().numpy(), want["dL_dsh"])
    grads_close("scales", t["scales"].grad.cpu().numpy(), want["dL_dscales"])
    grads_close("rotations", t["rotations"].grad.cpu().numpy(), want["dL_drotations"])
    vis = rast.markVisible(t["means3D"].detach())
    assert vis.dtype == torch.bool
    np.testing.assert_array_equal(vis.cpu().numpy(), O.mark_visible(a["means3D"], cam.viewmatrix, cam.projmatrix))


def test_argument_validation_and_empty_input():
    D = dgr()
    a, cam = scene(10, 32, 32, 0, 0.1)
    rast = D.GaussianRasterizer(settings(cam, 3))
    m, o = dev(a["means3D"]), dev(a["opacities"])
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rast(means3D=m, means2D=m, opacities=o, scales=dev(a["scales"]), rotations=dev(a["rotations"]))
    with pytest.raises(Exception, match="exactly one of either scale/rotation"):
        rast(means3D=m, means2D=m, opacities=o, shs=dev(a["sh"]))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        D._C.rasterize_gaussians(torch.ones(3).cuda(), torch.zeros(4, 2).cuda(), *([torch.Tensor([])] * 4), 1.0,
                                 torch.Tensor([]), dev(cam.viewmatrix), dev(cam.projmatrix), 1.0, 1.0, 32, 32,
                                 torch.Tensor([]), 0, dev(cam.campos), False, False)
    z = torch.zeros(0, 3, device="cuda")
    color, radii, depth = rast(means3D=z, means2D=z, opacities=torch.zeros(0, 1, device="cuda"),
                               shs=torch.zeros(0, 16, 3, device="cuda"), scales=z, rotations=torch.zeros(0, 4, device="cuda"))
    assert float(color.abs().max()) == 0.0 and radii.numel() == 0  # P == 0: nothing launched, zeros returned


def test_prefiltered_violation_is_reported():
    D = dgr()
    cam = syn.look_at_camera(32, 32, azimuth=0.0, elevation=0.0)
    a = dict(means3D=np.array([[3.95, 0, 0]], np.float32), scales=np.full((1, 3), 0.01, np.float32),
             rotations=np.array([[1, 0, 0, 0]], np.float32), opacities=np.array([[0.5]], np.float32),
             sh=np.zeros((1, 16, 3), np.float32))
    rast = D.GaussianRasterizer(settings(cam, 0, prefiltered=True))
    with pytest.raises(RuntimeError, match="prefiltered"):
        rast(means3D=dev(a["means3D"]), means2D=dev(a["means3D"]), opacities=dev(a["opacities"]), shs=dev(a["sh"]),
             scales=dev(a["scales"]), rotations=dev(a["rotations"]))


def _dirty_allocator():
    """Leave NaN bit patterns in the blocks torch's caching allocator hands out next: outputs and scratch tables
    that are no longer cleared by a fill (colour / depth / radii, tile ranges, counters) must not depend on them."""
    junk = [torch.full((n,), float("nan"), device="cuda") for n in (64, 1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22, 1 << 24)
            for _ in range(3)]
    torch.cuda.synchronize()
    del junk


def test_outputs_do_not_depend_on_previous_buffer_contents():
    # a few small Gaussians on a large image: most tiles are empty (ranges (0, 0)), most pixels are background
    a, cam = scene(40, 320, 240, 11, 0.02)
    _dirty_allocator()
    check_forward(a, cam, 3, bg=(0.2, 0.4, 0.6))
    # every Gaussian behind the camera: nothing is rendered, the image is the background, depth and radii are zero
    b = {k: v.copy() for k, v in a.items()}
    b["means3D"] = (np.asarray(cam.campos, np.float32)[None, :] * 1.5 +
                    np.random.default_rng(5).normal(0, 0.05, (40, 3))).astype(np.float32)
    _dirty_allocator()
    R, color, depth, radii, *_ = run_c(b, cam, 3, bg=(0.2, 0.4, 0.6))
    assert R == 0
    assert int(radii.abs().max()) == 0 and float(depth.abs().max()) == 0.0
    want = torch.tensor((0.2, 0.4, 0.6), device="cuda").view(3, 1, 1).expand_as(color)
    assert torch.equal(color, want)
    # a prefilter violation (flag carried in the block sums, no cleared counter) leaves no state behind
    rast = dgr().GaussianRasterizer(settings(cam, 3, prefiltered=True))
    with pytest.raises(RuntimeError, match="prefiltered"):
        rast(means3D=dev(b["means3D"]), means2D=dev(b["means3D"]), opacities=dev(b["opacities"]), shs=dev(b["sh"]),
             scales=dev(b["scales"]), rotations=dev(b["rotations"]))
    _dirty_allocator()
    check_forward(a, cam, 3)


def test_outputs_are_independent_tensors_and_accept_in_place_ops():
    """the reference returns separately allocated tensors; callers clamp the image in place
    (image.clamp_(0, 1)) before the loss.  Views of one block made autograd refuse that."""
    D = dgr()
    a, cam = scene(400, 64, 48, 21, 0.08)
    rast = D.GaussianRasterizer(settings(cam, 3))
    t = {k: dev(a[k], True) for k in ("means3D", "opacities", "sh", "scales", "rotations")}
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    color, radii, depth = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["sh"],
                               scales=t["scales"], rotations=t["rotations"])
    ptrs = {x.untyped_storage().data_ptr() for x in (color, radii, depth)}
    assert len(ptrs) == 3 and all(x._base is None for x in (color, radii, depth))
    color.clamp_(0.0, 1.0)
    depth.mul_(2.0)                                        # (radii is saved for the backward, in the reference too: not touched)
    (color.sum() + depth.sum()).backward()
    assert means2D.grad is not None and means2D.grad._base is None      # does not pin the rasterizer's gradient block
    assert float(means2D.grad[:, 2].abs().max()) == 0.0 and bool(torch.isfinite(means2D.grad).all())
    assert float(means2D.grad[radii.cpu() == 0].abs().max() if (radii == 0).any() else 0.0) == 0.0
    F = pkg("fused")
    s, r, o = F.activate(t["scales"].detach().log().requires_grad_(True), t["rotations"], t["opacities"])
    assert len({x.untyped_storage().data_ptr() for x in (s, r, o)}) == 3 and all(x._base is None for x in (s, r, o))
    r.add_(0.0)                    # (exp / sigmoid save their OUTPUT for the backward, as torch's own do: not modified)
    (s.sum() + r.sum() + o.sum()).backward()
    loss = F.l1_loss(color.detach().requires_grad_(True), torch.zeros_like(color))
    loss += 1.0                                                          # train.py accumulates regularisers in place
    loss.backward()


def test_nan_opacity_is_blended_like_the_reference():
    """alpha = min(0.99, NaN * G) is 0.99 under fminf semantics (CUDA's min(float, float) and C's
    fminf both return the non-NaN operand), `power > 0` and `alpha < 1/255` are false, so the reference BLENDS a
    splat with a NaN opacity over every pixel of its tiles where power <= 0.  The alpha-support culling here must not
    drop it silently (its box is "never cull"); compared with the oracle, which restates forward.cu:336-354."""
    a, cam = scene(60, 96, 80, 31, 0.05)
    a["opacities"][7] = np.nan
    a["opacities"][23] = np.float32(1.0 / 255.0)          # exactly the threshold: visible only at the splat centre
    r, out = check_forward(a, cam, 3)
    touched = r["tiles_touched"][7]
    assert touched > 0                                     # the NaN splat is on screen, so the case is exercised
    assert np.isfinite(out[1].cpu().numpy()).all()         # 0.99, not NaN, reaches the image (as in the reference)


def test_non_finite_colour_reaches_only_the_pixels_that_blend_it():
    """the forward blend has a straight-line loop body -- a pixel that skips an entry (alpha < 1/255,
    power > 0, finished) blends it with weight 0 -- so a +Inf / NaN colour (SH overflow of a diverged model) would turn
    0 * Inf into NaN on every pixel of the wave's rectangle, where the reference `continue`s (forward.cu:339-347) and
    leaves those pixels finite.  Colours are staged as finite values (csrc/blend_fwd.hip), so: wherever the oracle's
    image is finite the HIP image agrees with it to 1e-5, and pixels that do blend the entry are non-finite or huge in
    both."""
    a, cam = scene(400, 96, 80, 17, 0.08)
    colors = np.random.default_rng(3).uniform(0, 1, (400, 3)).astype(np.float32)
    colors[11] = np.inf
    colors[57, 1] = np.nan
    a["opacities"][11] = a["opacities"][57] = np.float32(0.6)
    R, color, depth, radii, gb, bb, ib = run_c(a, cam, 3, colors=colors)
    r = run_oracle(a, cam, 3, colors=colors)
    want, got = r["color"], color.cpu().numpy()
    fin = np.isfinite(want).all(axis=0) & (r["margin"] > 2e-5)
    blended = ~np.isfinite(want).all(axis=0)
    assert r["tiles_touched"][11] > 0 and r["tiles_touched"][57] > 0 and blended.sum() > 20 and fin.sum() > 1000
    assert np.isfinite(got[:, fin]).all()
    assert np.abs(got[:, fin] - want[:, fin]).max() <= 1e-5
    # the Inf entry's tiles hold pixels that skip it: they sit in the same wave rectangles as pixels that blend it
    ys, xs = np.nonzero(blended)
    near = np.zeros_like(blended)
    near[max(ys.min() - 8, 0):ys.max() + 8, max(xs.min() - 8, 0):xs.max() + 8] = True
    assert (near & fin).sum() > 50
    bad = ~np.isfinite(got).all(axis=0) | (np.abs(got).max(axis=0) > 1e30)
    assert bad[blended & (r["margin"] > 2e-5)].all()     # where the reference blends the diverged splat, so does this


def test_backward_parity_long_lists_many_segments():
    """Lists of more than a thousand entries on a handful of tiles: the backward walks them in 128-entry segments from the
    forward's checkpoints (csrc/blend_bwd.hip); low opacities keep most pixels alive to the end of their list, so every
    segment is live and starts from a checkpoint with a small transmittance."""
    a, cam = scene(2500, 48, 40, 21, 0.2)
    a = dict(a, opacities=(a["opacities"] * 0.3).astype(np.float32))
    r, (R, color, depth, radii, gb, bb, ib) = check_forward(a, cam, 3)
    v = dgr()._C.state_view(2500, R, 48, 40, gb, bb, ib)
    lens = (v["ranges"][:, 1] - v["ranges"][:, 0]).cpu().numpy()
    deepest = int(v["n_contrib"].max().item())
    print("list lengths", lens.tolist(), "deepest contributor", deepest)
    assert lens.max() > 1024 and deepest > 640            # at least five live segments on some tile
    check_backward(a, cam, 3, seed=3)


def test_forward_only_flag_changes_no_output():
    """GS4D_FORWARD_ONLY (debug bit 1, set by the wrapper when no input needs a gradient) only drops the checkpoints."""
    a, cam = scene(3000, 96, 80, 4, 0.2)
    C_ = dgr()._C
    e = torch.Tensor([])
    args = (torch.ones(3, device="cuda"), dev(a["means3D"]), e, dev(a["opacities"]), dev(a["scales"]), dev(a["rotations"]), 1.0, e,
            dev(cam.viewmatrix), dev(cam.projmatrix), cam.tanfovx, cam.tanfovy, cam.image_height, cam.image_width, dev(a["sh"]), 3,
            dev(cam.campos), False, False)
    R0, c0, d0, r0, g0, b0, i0 = C_.rasterize_gaussians(*args)
    R1, c1, d1, r1, g1, b1, i1 = C_.rasterize_gaussians_forward_only(*args)
    assert R0 == R1 and torch.equal(c0, c1) and torch.equal(d0, d1) and torch.equal(r0, r1)
    # ... and does not reserve them either (80 of the ~116 bytes per instance of the binning chunk)
    L = pkg("_cabi").lib()
    L.gs4d_binning_bytes_forward_only.restype = C.c_size_t
    assert b1.numel() < 0.5 * b0.numel() and b1.numel() >= int(L.gs4d_binning_bytes_forward_only(C.c_int(R1)))
    v0, v1 = C_.state_view(3000, R0, 96, 80, g0, b0, i0), C_.state_view(3000, R1, 96, 80, g1, b1, i1)
    for k in ("point_list", "ranges", "n_contrib", "final_T"):
        assert torch.equal(v0[k], v1[k]), k
    # the autograd wrapper: identical image with and without gradients
    rast = dgr().GaussianRasterizer(settings(cam, 3))
    t = {k: dev(a[k], True) for k in ("means3D", "opacities", "sh", "scales", "rotations")}
    kw = dict(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), opacities=t["opacities"], shs=t["sh"],
              scales=t["scales"], rotations=t["rotations"])
    with torch.no_grad():
        img_ng, _, _ = rast(**kw)
    img_g, _, _ = rast(**kw)
    assert torch.equal(img_ng, img_g) and torch.equal(img_g, c0)
