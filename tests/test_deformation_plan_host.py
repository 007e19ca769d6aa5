"""Host logic of the deformation module's per-step plan (4dgaussians_amd/scene/deformation.py::_Plan), on the CPU: the
words patched into gs4d_deform_grads must be the addresses of the gradient views handed to autograd, in the order
the C ABI reads them (include/gs4d.h), the views must have the parameters' shapes (planes channels-last), and the plan
must notice when a parameter's storage is replaced."""
import copy

import torch

from conftest import pkg
from test_gpu_deformation import make_args
from oracle import deformation_ref as R

D = pkg("scene.deformation")


def _net(depth=0):
    cfg = R.DeformConfig(plane_channels=16, resolution=[16, 16, 16, 8], multires=[1, 2], defor_depth=depth)
    torch.manual_seed(0)
    return D.deform_network(make_args(cfg)).deformation_net


def test_plan_gradient_words_match_the_views():
    for depth in (0, 3):
        net = _net(depth)
        params, cfg = net._params(), net._config()
        plan = D._Plan(cfg, params)
        assert plan.in_place and plan.matches(params)
        g, views = plan.new_grads("cpu")
        n = plan.n_planes
        assert n == 12 and len(views) == len(params) - 1          # every parameter but the frozen aabb
        for i in range(n):
            assert g.planes[i] == views[i].data_ptr()
        assert g.planes[n] is None or g.planes[n] == 0            # unused levels stay null
        assert g.ft_w == views[n].data_ptr() and g.ft_b == views[n + 1].data_ptr()
        for i in range(plan.n_trunk // 2):
            assert g.trunk_w[i] == views[n + 2 + 2 * i].data_ptr() and g.trunk_b[i] == views[n + 3 + 2 * i].data_ptr()
        k = n + 2 + plan.n_trunk
        for h in range(4):
            assert g.head_w1[h] == views[k + 4 * h].data_ptr() and g.head_b1[h] == views[k + 4 * h + 1].data_ptr()
            assert g.head_w2[h] == views[k + 4 * h + 2].data_ptr() and g.head_b2[h] == views[k + 4 * h + 3].data_ptr()
        others = [p for p in params if p is not net.grid.aabb]
        for v, p in zip(views, others):
            assert v.shape == p.shape and v.data_ptr() % 16 == 0 and float(v.abs().max()) == 0.0
        assert all(v.is_contiguous(memory_format=torch.channels_last) for v in views[:n])
        # views do not overlap: writing one leaves the others zero
        views[3].fill_(1.0)
        assert all(float(v.abs().max()) == 0.0 for i, v in enumerate(views) if i != 3)
        # a second allocation patches the same structure with the new addresses
        g2, views2 = plan.new_grads("cpu")
        assert g2 is g and g.planes[0] == views2[0].data_ptr() and views2[0].data_ptr() != 0


def test_plan_is_invalidated_by_new_storage_and_not_copied():
    net = _net()
    params = net._params()
    plan = D._Plan(net._config(), params)
    with torch.no_grad():
        params[0].add_(1.0)                                       # in-place update (an optimizer step): still valid
    assert plan.matches(params)
    params[0].data = params[0].data.clone()                       # new storage (.to(), load with assign, ...)
    assert not plan.matches(params)
    net.__dict__["_step_cache"] = (params, {"plan": plan})
    clone = copy.deepcopy(net)                                    # ctypes structures are never deep-copied / pickled
    assert "_step_cache" not in clone.__dict__
    net.float()                                                   # _apply drops the cache
    assert net.__dict__.get("_step_cache") is None


def test_step_cache_is_dropped_by_set_aabb_and_extra_deform_switches():
    """set_aabb installs a NEW aabb Parameter and the switches change the descriptor's flags; a plan cached
    by an earlier forward_dynamic must not survive either (the GPU twin of this test checks the outputs change)."""
    net = _net()
    fake = lambda: (net._params(), {"plan": D._Plan(net._config(), net._params())}, net._flags(), net.grid.aabb)
    net.__dict__["_step_cache"] = fake()
    net.set_aabb([2.0, 2.0, 2.0], [-2.0, -2.0, -2.0])
    assert net.__dict__.get("_step_cache") is None
    net.__dict__["_step_cache"] = fake()
    net.disable_extra_deform()
    assert net.__dict__.get("_step_cache") is None and net._flags()[1:4] == (True, True, True)
    net.__dict__["_step_cache"] = fake()
    net.enable_extra_deform()
    assert net.__dict__.get("_step_cache") is None and net._flags()[1:4] == (False, False, False)
    # a direct assignment that bypasses the methods is caught by the per-step comparison of flags / aabb identity
    cache = net.__dict__["_step_cache"] = fake()
    net.args.no_ds = True
    assert cache[2] != net._flags()
    net.args.no_ds = False
    net.grid.set_aabb([1.0, 1.0, 1.0], [-1.0, -1.0, -1.0])
    assert net.grid.aabb is not cache[3]
