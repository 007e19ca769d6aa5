"""Host logic of the rasterizer wrapper: the shared block of zeros that stands in for absent upstream gradients and
backs the screen-space leaf of every view (4dgaussians_amd/diff_gaussian_rasterization/__init__.py, renderer.py)."""
import torch

from conftest import pkg


def test_shared_zeros_are_cached_per_shape_and_leaves_over_them_keep_their_own_gradients():
    D = pkg("diff_gaussian_rasterization")
    dev = torch.device("cpu")
    z1 = D._shared_zeros((5, 3), dev)
    z2 = D._shared_zeros((5, 3), dev)
    assert z1 is z2 and z1.dtype == torch.float32 and float(z1.abs().max()) == 0.0
    assert D._shared_zeros((4, 3), dev) is not z1
    # two views of one frame each: fresh leaves over the same storage, gradients accumulate separately
    a = z1.detach().requires_grad_(True)
    b = z1.detach().requires_grad_(True)
    assert a.is_leaf and b.is_leaf and a.data_ptr() == b.data_ptr()
    (a * 2.0).sum().backward()
    (b * 3.0).sum().backward()
    assert float(a.grad.min()) == 2.0 and float(b.grad.min()) == 3.0
    assert float(z1.abs().max()) == 0.0
    # the cache is bounded
    for n in range(1, 20):
        D._shared_zeros((n,), dev)
    assert len(D._zeros_cache) <= 8
