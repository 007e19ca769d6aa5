"""End-to-end GPU check of the rows around the render path: a GaussianModel created
from a point cloud (distCUDA2), trained for a few iterations with the reference's loop structure (train.py:137-330:
render, L1 + grid regularisers, backward, densification statistics, densify / prune, fused AdamW step), then saved,
reloaded and rendered by the FPS harness."""
import types

import numpy as np
import pytest
import torch

from conftest import pkg
from test_gaussian_model_host import opt_args
from test_gpu_deformation import make_args
from oracle import deformation_ref as R

pytestmark = pytest.mark.gpu


def test_train_densify_save_load_render(tmp_path):
    GM, RD, F, syn = pkg("scene.gaussian_model"), pkg("renderer"), pkg("fused"), pkg("synthetic")
    IU = pkg("utils.image_utils")
    cfg = R.DeformConfig(plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2])
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    # scene/dataset_readers.py:450-455: a random cloud in [-1.3, 1.3]^3 with random colours
    pcd = GM.BasicPointCloud(points=rng.uniform(-1.3, 1.3, (2000, 3)), colors=rng.uniform(0, 1, (2000, 3)), normals=np.zeros((2000, 3)))
    g = GM.GaussianModel(3, make_args(cfg))
    g.create_from_pcd(pcd, spatial_lr_scale=5.0)
    g._deformation.deformation_net.set_aabb([1.3, 1.3, 1.3], [-1.3, -1.3, -1.3])
    g.training_setup(opt_args())
    assert g.get_xyz.shape == (2000, 3) and g._features_rest.shape == (2000, 15, 3)
    assert float(g.get_opacity.detach().mean()) == pytest.approx(0.1, abs=1e-6)
    # initial scales = sqrt(mean squared distance to the 3 nearest neighbours), gaussian_model.py:169-170
    d2 = pkg("simple_knn._C").distCUDA2(g.get_xyz.detach())
    assert torch.allclose(g.get_scaling.detach()[:, 0], torch.sqrt(torch.clamp_min(d2, 1e-7)), rtol=1e-5)

    dev = torch.device("cuda")
    cams = [RD.Camera(syn.look_at_camera(160, 120, azimuth=0.5 + 0.4 * i), dev, time=0.1 + 0.2 * i) for i in range(4)]
    bg = torch.ones(3, device=dev)
    targets = [torch.rand(3, 120, 160, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) * 0.2 + 0.4 for i in range(4)]
    losses, n_points = [], []
    for it in range(1, 41):
        g.update_learning_rate(it)
        cam, target = cams[it % 4], targets[it % 4]
        image, viewspace, visibility, radii, depth = RD.render(cam, g, g._deformation, bg)
        loss = F.l1_loss(image, target) + g.compute_regulation(0.01, 0.0001, 0.0, 0.0001)["total_reg"]
        loss.backward()
        losses.append(float(loss.detach()))
        with torch.no_grad():
            g.max_radii2D[visibility] = torch.max(g.max_radii2D[visibility], radii[visibility].float())
            g.add_densification_stats(viewspace.grad, visibility)
            if it % 10 == 0:
                g.densify(0.00002, 0.005, 5.0, None)
                g.prune(0.00002, 0.005, 5.0, None)
        g.optimizer.step()
        g.optimizer.zero_grad(set_to_none=True)
        n_points.append(g.get_xyz.shape[0])
    assert np.mean(losses[-8:]) < np.mean(losses[:8])                 # it learns
    assert n_points[-1] != 2000 and len(set(n_points)) > 1             # densify / prune changed the model
    assert g.optimizer.state[g._xyz]["exp_avg"].shape == g._xyz.shape  # optimizer state followed
    assert all(torch.isfinite(p).all() for p in (g._xyz, g._scaling, g._rotation, g._opacity, g._features_dc))

    # save, load into a fresh model, render with the FPS harness: identical frames
    g.save_ply(str(tmp_path / "point_cloud.ply"))
    g.save_deformation(str(tmp_path))
    h = GM.GaussianModel(3, make_args(cfg))
    h.load_ply(str(tmp_path / "point_cloud.ply"))
    h.load_model(str(tmp_path))
    assert h.active_sh_degree == 3            # load_ply switches every SH band on (gaussian_model.py:341) ...
    h.active_sh_degree = g.active_sh_degree   # ... while this fork trains at degree 0 (train.py:143-145): compare like with like
    fps, frames = RD.render_set_fps(cams, g, g._deformation, bg)
    _, frames2 = RD.render_set_fps(cams, h, h._deformation, bg)
    assert fps > 0 and len(frames) == 4 and frames[0].shape == (120, 160, 3) and frames[0].dtype == np.uint8
    for a, b in zip(frames, frames2):
        np.testing.assert_array_equal(a, b)
    img = RD.render_for_image(cams[0], h, h._deformation, bg)
    assert float(IU.psnr(img[None], targets[0][None])) > 5.0


def test_early_sh_gradient_equals_final_feature_gradient():
    """parallel.GradBucket.early_reduce relies on dL/d_features == dL/d(deformed shs) (the deformation adds a residual
    to the SH coefficients): the gradient seen by render()'s shs_grad_hook, mid-backward, must be bit-identical to
    the gradient autograd finally delivers to the parameter, and the bucket must hand it back unchanged."""
    RD, F, syn, par = pkg("renderer"), pkg("fused"), pkg("synthetic"), pkg("parallel")
    cfg = R.DeformConfig(plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2])
    torch.manual_seed(1)
    dev = torch.device("cuda")
    deform = pkg("scene.deformation").deform_network(make_args(cfg)).to(dev)
    _, cam_np, raw = syn.make_workload("C1", P=3000)
    pc = RD.GaussianParams(raw, dev)
    cam = RD.Camera(syn.look_at_camera(128, 96), dev, time=0.4)
    params = list(pc.parameters()) + list(deform.parameters())
    bucket = par.GradBucket(params, early=[pc._features])
    target = torch.rand(3, 96, 128, device=dev)
    seen = []
    for step in range(2):
        def hook(g):
            seen.append((bucket.early_reduce(pc._features, g, force=True), g.detach().clone()))
        image, *_ = RD.render(cam, pc, deform, torch.ones(3, device=dev), shs_grad_hook=hook)
        F.l1_loss(image, target).backward()
        assert len(seen) == step + 1
        started, g_mid = seen[-1]
        assert started == (step > 0)                       # the layout is learnt by the first all_reduce()
        assert torch.equal(g_mid, pc._features.grad)
        if started:
            slot = [s for p, s in zip(bucket.live, bucket.slots) if p is pc._features][0]
            assert torch.equal(slot.view_as(g_mid), g_mid) and bucket.split >= g_mid.numel()
        bucket.all_reduce()                                # single process: builds the layout, reduces nothing
        for p in params:
            p.grad = None


def test_data_phase_gradients_are_final_and_phased_backward_is_bit_identical():
    """The deformation backward in two phases (gs4d_deform_backward_phases): what the data-phase hook sees for positions,
    scales, rotations, planes and force_time_embed must be bit-identical to what autograd finally puts into .grad, the
    weight gradients computed afterwards must equal those of the one-call backward, and the bucket's grouped early
    reduce must hold exactly those values."""
    RD, F, syn, par = pkg("renderer"), pkg("fused"), pkg("synthetic"), pkg("parallel")
    cfg = R.DeformConfig(plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2])
    torch.manual_seed(2)
    dev = torch.device("cuda")
    deform = pkg("scene.deformation").deform_network(make_args(cfg)).to(dev)
    _, cam_np, raw = syn.make_workload("C1", P=3000)
    pc = RD.GaussianParams(raw, dev)
    cam = RD.Camera(syn.look_at_camera(128, 96), dev, time=0.7)
    params = list(pc.parameters()) + list(deform.parameters())
    net = deform.deformation_net
    planes = [p for level in net.grid.grids for p in level]
    data_params = [pc._xyz, pc._scaling, pc._rotation, *planes, net.force_time_embed.weight, net.force_time_embed.bias]
    target = torch.rand(3, 96, 128, device=dev)
    bg = torch.ones(3, device=dev)

    def grads_of_one_step():
        image, *_ = RD.render(cam, pc, deform, bg)
        F.l1_loss(image, target).backward()
        out = [None if p.grad is None else p.grad.detach().clone() for p in params]
        for p in params:
            p.grad = None
        return out

    reference = grads_of_one_step()                            # one-call backward
    bucket = par.GradBucket(params, early=[pc._features, *data_params])
    seen = []

    def hook(d_xyz, d_sc, d_rot, gp, g_ftw, g_ftb):
        gs = [d_xyz, d_sc, d_rot, *gp, g_ftw, g_ftb]
        seen.append((bucket.early_reduce_many(data_params, gs, force=True), [g.detach().clone() for g in gs]))

    net.data_grad_hook = hook
    for step in range(2):
        image, *_ = RD.render(cam, pc, deform, bg)
        F.l1_loss(image, target).backward()
        assert len(seen) == step + 1
        started, mid = seen[-1]
        assert started == (step > 0)
        for p, g in zip(data_params, mid):
            assert torch.equal(p.grad, g)
        plane_ids = {id(p) for p in planes}
        for p, want in zip(params, reference):                 # the phased backward changes no value at all
            assert (p.grad is None) == (want is None)
            if want is None:
                continue
            if id(p) in plane_ids:                             # plane cells are summed with float atomics: order varies run to run
                assert float((p.grad - want).abs().max()) <= 1e-5 * float(want.abs().max())
            else:
                assert torch.equal(p.grad, want)
        if started:
            for p, g in zip(data_params, mid):
                slot = [s for q, s in zip(bucket.live, bucket.slots) if q is p][0]
                assert torch.equal(slot, par._memory_order_view(g))
        bucket.all_reduce()
        for p in params:
            p.grad = None
    net.data_grad_hook = None
