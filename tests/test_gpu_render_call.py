"""The one-call path (gs4d_render_forward / gs4d_render_backward through 4dgaussians_amd/render_call.py) against the
staged path (three autograd nodes over the per-stage entry points): the same kernels in the same order, so every output
and every gradient must agree BIT FOR BIT -- except the HexPlane plane gradients, the library's only float-atomic sums,
which are held to 1e-6 of the tensor's maximum.  The staged path is the one the oracle comparisons observe stage by
stage (tests/test_gpu_c2.py); this file carries those results over to the path render() takes by default."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def small_scene(P=3000, W=160, H=112, seed=5):
    syn, R, Wl, Dm = pkg("synthetic"), pkg("renderer"), pkg("workload"), pkg("scene.deformation")
    dev = torch.device("cuda", 0)
    raw = syn.make_gaussians(P, seed, 0.06)
    raw = dict(xyz=raw["xyz"], sh=raw["sh"], opacity_logit=raw["opacity_logit"], log_scales=raw["log_scales"], rot=raw["rot"])
    torch.manual_seed(seed)
    pc = R.GaussianParams(raw, dev)
    deform = Dm.deform_network(Wl.deform_args("C2"))
    deform.deformation_net.set_aabb([1.3] * 3, [-1.3] * 3)
    deform = deform.to(dev)
    cam = R.Camera(syn.look_at_camera(W, H, azimuth=0.9), dev, time=0.41)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    return R, pc, deform, cam, torch.ones(3, device=dev), target


def run(R, staged, fn, params):
    before = R.STAGED_ONLY
    R.STAGED_ONLY = staged
    try:
        for p in params:
            p.grad = None
        out = fn()
        grads = [None if p.grad is None else p.grad.clone() for p in params]
    finally:
        R.STAGED_ONLY = before
    return out, grads


def assert_same(params, names, g_staged, g_one):
    for n, a, b in zip(names, g_staged, g_one):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        if "grids" in n:                    # float atomics: order-dependent rounding
            tol = 1e-6 * float(a.abs().max()) + 1e-12
            assert float((a - b).abs().max()) <= tol, (n, float((a - b).abs().max()), tol)
        else:
            assert torch.equal(a, b), (n, float((a - b).abs().max()))


def named(pc, deform):
    items = [("pc." + n, p) for n, p in pc.named_parameters()] + list(deform.named_parameters())
    return [n for n, _ in items], [p for _, p in items]


@pytest.mark.parametrize("stage", ["fine", "coarse"])
def test_one_call_equals_staged(stage):
    R, pc, deform, cam, bg, target = small_scene()
    F = pkg("fused")
    names, params = named(pc, deform)
    gd = torch.rand(1, cam.image_height, cam.image_width, device=bg.device)

    def step():
        image, ssp, vis, radii, depth = R.render(cam, pc, deform, bg, stage=stage)
        (F.l1_loss(image, target) + 0.1 * (depth * gd).sum()).backward()
        return image.detach().clone(), radii.clone(), depth.detach().clone(), vis.clone(), ssp.grad.clone()

    a, ga = run(R, True, step, params)
    b, gb = run(R, False, step, params)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert int((a[1] > 0).sum()) > 500
    assert_same(params, names, ga, gb)
    live = [n for n, g in zip(names, gb) if g is not None]
    assert len(live) >= (5 if stage == "coarse" else 5 + 12 + 2 + 2 + 16), live


def test_fused_l1_equals_render_then_l1():
    R, pc, deform, cam, bg, target = small_scene(seed=7)
    F = pkg("fused")
    names, params = named(pc, deform)

    def separate():
        image, ssp, *_ = R.render(cam, pc, deform, bg)
        loss = F.l1_loss(image, target)
        loss.backward()
        return loss.detach().clone(), image.detach().clone(), ssp.grad.clone()

    def fused():
        loss, image, ssp, vis, radii, depth = R.render_with_l1(cam, pc, deform, bg, target)
        loss.backward()
        return loss.detach().clone(), image.detach().clone(), ssp.grad.clone()

    a, ga = run(R, True, separate, params)
    b, gb = run(R, False, fused, params)
    c, gc = run(R, True, fused, params)         # the staged fallback of render_with_l1
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert_same(params, names, ga, gb)
    assert_same(params, names, ga, gc)


def test_image_used_by_the_fused_loss_and_elsewhere():
    R, pc, deform, cam, bg, target = small_scene(seed=9)
    F = pkg("fused")
    names, params = named(pc, deform)
    w = torch.rand(3, cam.image_height, cam.image_width, device=bg.device)

    def separate():
        image, *_ = R.render(cam, pc, deform, bg)
        (F.l1_loss(image, target) + (image * w).mean()).backward()

    def fused():
        loss, image, *_ = R.render_with_l1(cam, pc, deform, bg, target)
        (loss + (image * w).mean()).backward()

    _, ga = run(R, True, separate, params)
    _, gb = run(R, False, fused, params)
    for n, a, b in zip(names, ga, gb):          # the two image gradients are added in a different order: not bitwise
        if a is not None:
            assert float((a - b).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-12, n


def test_no_grad_frame_and_fallbacks():
    R, pc, deform, cam, bg, target = small_scene(seed=11)
    before = R.STAGED_ONLY
    try:
        R.STAGED_ONLY = True
        a = R.render_for_image(cam, pc, deform, bg)
        R.STAGED_ONLY = False
        b = R.render_for_image(cam, pc, deform, bg)
        # a float64 background is outside the one-call path: render() must quietly take the staged path (which raises
        # its own error for it), never crash inside the composite call
        with pytest.raises(RuntimeError, match="float32"):
            R.render(cam, pc, deform, bg.double())
    finally:
        R.STAGED_ONLY = before
    assert torch.equal(a, b)


def test_stage_hooks_fire_with_final_gradients():
    R, pc, deform, cam, bg, target = small_scene(seed=13)
    F = pkg("fused")
    seen = {}

    def shs_hook(g):
        seen["shs"] = g.clone()

    def data_hook(d_xyz, d_sc, d_rot, plane_grads, g_ftw, g_ftb):
        seen["xyz"], seen["planes"], seen["ftw"] = d_xyz.clone(), [p.clone() for p in plane_grads], g_ftw.clone()

    deform.deformation_net.data_grad_hook = data_hook
    try:
        image, *_ = R.render(cam, pc, deform, bg, shs_grad_hook=shs_hook)
        F.l1_loss(image, target).backward()
    finally:
        deform.deformation_net.data_grad_hook = None
    torch.cuda.synchronize()
    assert torch.equal(seen["shs"], pc._features.grad) and torch.equal(seen["xyz"], pc._xyz.grad)
    planes = [p for level in deform.deformation_net.grid.grids for p in level]
    for a, p in zip(seen["planes"], planes):
        assert torch.equal(a, p.grad)
    assert torch.equal(seen["ftw"], deform.deformation_net.force_time_embed.weight.grad)


def test_step_call_equals_autograd_backward():
    """renderer.render_l1_step (forward + L1 + backward issued back to back, gradients written to .grad without an
    autograd graph) against render_with_l1(...) followed by (scale * loss).backward(): same library calls, so loss,
    image and every gradient agree bit for bit (plane gradients: float atomics, 1e-6); a second call accumulates; the
    staged fallback of the step call gives the same again."""
    R, pc, deform, cam, bg, target = small_scene(seed=11)
    names, params = named(pc, deform)
    for scale in (1.0, 0.25):
        def autograd():
            loss, image, ssp, vis, radii, depth = R.render_with_l1(cam, pc, deform, bg, target)
            (loss * scale if scale != 1.0 else loss).backward()
            return loss.detach().clone(), image.detach().clone(), ssp.grad.clone(), radii.clone(), vis.clone()

        def step():
            loss, image, ssp, vis, radii, depth = R.render_l1_step(cam, pc, deform, bg, target, loss_scale=scale)
            assert not loss.requires_grad and not image.requires_grad
            return loss.clone(), image.clone(), ssp.grad.clone(), radii.clone(), vis.clone()

        a, ga = run(R, False, autograd, params)
        b, gb = run(R, False, step, params)
        c, gc = run(R, True, step, params)         # the step call's fallback: staged render + autograd
        for x, y, z in zip(a, b, c):
            assert torch.equal(x, y) and torch.equal(x, z)
        assert_same(params, names, ga, gb)
        assert_same(params, names, ga, gc)
        assert sum(g is not None for g in gb) >= 5 + 12 + 2 + 2 + 16

    def twice():
        R.render_l1_step(cam, pc, deform, bg, target)
        R.render_l1_step(cam, pc, deform, bg, target)

    one, g1 = run(R, False, lambda: R.render_l1_step(cam, pc, deform, bg, target) and None, params)
    two, g2 = run(R, False, twice, params)
    for n, u, v in zip(names, g1, g2):
        if u is not None:
            assert float((v - 2 * u).abs().max()) <= 2e-6 * float(u.abs().max()) + 1e-12, n


def test_speculative_binning_launch_changes_no_result():
    """gs4d_render_forward issues binning and blend BEFORE it has read num_rendered, sized for the largest of the thread's
    recent counts + 1/8 (csrc/api.hip); the kernels take the count from device memory and do nothing when it exceeds that
    capacity, and the host then issues the two stages again with the exact size.  Far view (few instances) nine times,
    then a near view whose count is several times larger: the near view's launch must overflow, be re-issued, and give
    the staged path's image and gradients bit for bit; the far views in between run from the speculative launch alone."""
    R, pc, deform, _, bg, target = small_scene(W=320, H=224, seed=13)
    syn, cabi = pkg("synthetic"), pkg("_cabi")
    cam_near = R.Camera(syn.look_at_camera(320, 224, azimuth=0.9, radius=2.6), bg.device, time=0.41)
    cam_far = R.Camera(syn.look_at_camera(320, 224, azimuth=0.9, radius=20.0), bg.device, time=0.41)
    names, params = named(pc, deform)

    def step(cam):
        def f():
            loss, image, ssp, vis, radii, depth = R.render_with_l1(cam, pc, deform, bg, target)
            loss.backward()
            return loss.detach().clone(), image.detach().clone(), radii.clone(), ssp.grad.clone()
        return f

    far_staged, g_far_staged = run(R, True, step(cam_far), params)
    near_staged, g_near_staged = run(R, True, step(cam_near), params)
    n_far, n_near = int((far_staged[2] > 0).sum()), int((near_staged[2] > 0).sum())
    i0, r0 = cabi.speculation_stats()
    for _ in range(9):                                           # the capacity follows the last EIGHT counts: forget the near one
        far, g_far = run(R, False, step(cam_far), params)
    i1, r1 = cabi.speculation_stats()
    assert i1 - i0 == 9 and r1 == r0, (i0, r0, i1, r1)          # (the staged calls above left a count to guess from)
    for x, y in zip(far_staged, far):
        assert torch.equal(x, y)
    assert_same(params, names, g_far_staged, g_far)
    near, g_near = run(R, False, step(cam_near), params)
    i2, r2 = cabi.speculation_stats()
    assert i2 - i1 == 1 and r2 - r1 == 1, ("the near view was expected to overflow the far views' capacity", n_far, n_near, i2, r2)
    for x, y in zip(near_staged, near):
        assert torch.equal(x, y)
    assert_same(params, names, g_near_staged, g_near)
    far, g_far = run(R, False, step(cam_far), params)           # capacity now follows the near view: no re-issue
    i3, r3 = cabi.speculation_stats()
    assert i3 - i2 == 1 and r3 == r2
    for x, y in zip(far_staged, far):
        assert torch.equal(x, y)
    assert_same(params, names, g_far_staged, g_far)


def test_c2_headline_scene_one_call_equals_staged():
    """BASELINE.json configs[1] as bench.py builds it: the default path of render_with_l1 against the staged path."""
    Wl, R, F = pkg("workload"), pkg("renderer"), pkg("fused")
    dev = torch.device("cuda", 0)
    scene = Wl.build_scene("C2", dev)
    pc, deform, cam, bg, target = (scene[k] for k in ("pc", "deform", "cam", "bg", "target"))
    names, params = named(pc, deform)

    def staged():
        image, ssp, *_ = R.render(cam, pc, deform, bg)
        loss = F.l1_loss(image, target)
        loss.backward()
        return loss.detach().clone(), image.detach().clone(), ssp.grad.clone()

    def fused():
        loss, image, ssp, *_ = R.render_with_l1(cam, pc, deform, bg, target)
        loss.backward()
        return loss.detach().clone(), image.detach().clone(), ssp.grad.clone()

    a, ga = run(R, True, staged, params)
    b, gb = run(R, False, fused, params)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert_same(params, names, ga, gb)


def test_render_batch_equals_per_view_renders():
    """render_batch: ONE deformation query (B * P rows, per-row timestamps) + B rasterizer passes, against B render()
    calls: identical images / radii / depth, gradients equal up to the order in which the views are summed."""
    R, pc, deform, cam, bg, target = small_scene(seed=21)
    syn, F = pkg("synthetic"), pkg("fused")
    dev = bg.device
    cams = [cam] + [R.Camera(syn.look_at_camera(160, 112, azimuth=0.9 + 0.5 * j, elevation=0.2 + 0.1 * j), dev, time=0.1 + 0.27 * j)
                    for j in range(1, 4)]
    targets = [torch.rand(3, 112, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(40 + j)) for j in range(4)]
    names, params = named(pc, deform)

    def one_by_one():
        outs = [R.render(c, pc, deform, bg) for c in cams]
        sum(F.l1_loss(o[0], t) for o, t in zip(outs, targets)).backward()
        return [(o[0].detach().clone(), o[3].clone(), o[4].detach().clone(), o[1].grad.clone()) for o in outs]

    def batched():
        outs = R.render_batch(cams, pc, deform, bg)
        sum(F.l1_loss(o[0], t) for o, t in zip(outs, targets)).backward()
        return [(o[0].detach().clone(), o[3].clone(), o[4].detach().clone(), o[1].grad.clone()) for o in outs]

    def batched_with_targets():                     # every view's L1 inside its library call
        outs = R.render_batch(cams, pc, deform, bg, targets=targets)
        sum(o[0] for o in outs).backward()
        for o, t in zip(outs, targets):
            assert torch.equal(o[0].detach(), F.l1_loss(o[1].detach(), t))
        return [(o[1].detach().clone(), o[4].clone(), o[5].detach().clone(), o[2].grad.clone()) for o in outs]

    a, ga = run(R, False, one_by_one, params)
    for fn, staged in ((batched, False), (batched_with_targets, False), (batched, True), (batched_with_targets, True)):
        b, gb = run(R, staged, fn, params)          # staged: the per-stage fallback of render_batch
        for va, vb in zip(a, b):
            for x, y in zip(va, vb):
                assert torch.equal(x, y)
        for n, x, y in zip(names, ga, gb):
            assert (x is None) == (y is None), n
            if x is not None:
                assert float((x - y).abs().max()) <= 2e-6 * float(x.abs().max()) + 1e-12, (n, float((x - y).abs().max()), float(x.abs().max()))
