"""Fused activations and L1 loss (csrc/elementwise.hip) against the framework ops they replace
(scene/gaussian_model.py:38-43, utils/loss_utils.py:20-21 of the reference), on the GPU.
Tolerances: forward 1e-6 relative (same libm-class exp), backward 1e-5 relative."""
import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol=1e-7):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item()
    assert err <= atol + rtol * scale, (err, scale)


@pytest.mark.parametrize("P", [1, 63, 4097, 100000])
def test_activate_matches_torch(P):
    F = pkg("fused")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(P)
    s = (torch.randn(P, 3, generator=g) * 1.5 - 3).to(dev).requires_grad_()
    r = torch.randn(P, 4, generator=g).to(dev).requires_grad_()
    o = (torch.randn(P, 1, generator=g) * 3).to(dev).requires_grad_()
    ws, wr, wo = (torch.randn(P, k, generator=g).to(dev) for k in (3, 4, 1))

    a = F.activate(s, r, o)
    (a[0] * ws).sum().add((a[1] * wr).sum()).add((a[2] * wo).sum()).backward()
    got = [t.clone() for t in (*a, s.grad, r.grad, o.grad)]
    for t in (s, r, o):
        t.grad = None
    b = (torch.exp(s), torch.nn.functional.normalize(r), torch.sigmoid(o))
    (b[0] * ws).sum().add((b[1] * wr).sum()).add((b[2] * wo).sum()).backward()
    want = [*b, s.grad, r.grad, o.grad]
    for x, y, tol in zip(got, want, (1e-6, 1e-6, 1e-6, 1e-5, 1e-5, 1e-5)):
        _close(x, y, tol)


def test_activate_partial_upstream_gradients():
    """only the opacity output is used downstream: the other upstream gradients arrive as None"""
    F = pkg("fused")
    dev = torch.device("cuda", 0)
    s = torch.randn(100, 3, device=dev, requires_grad=True)
    r = torch.randn(100, 4, device=dev, requires_grad=True)
    o = torch.randn(100, 1, device=dev, requires_grad=True)
    F.activate(s, r, o)[2].sum().backward()
    assert torch.count_nonzero(s.grad) == 0 and torch.count_nonzero(r.grad) == 0
    _close(o.grad, torch.sigmoid(o) * (1 - torch.sigmoid(o)), 1e-5)


@pytest.mark.parametrize("shape", [(3, 17, 31), (3, 800, 800), (1,)])
def test_l1_loss_matches_torch(shape):
    F = pkg("fused")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(7)
    a = torch.rand(*shape, generator=g).to(dev).requires_grad_()
    b = torch.rand(*shape, generator=g).to(dev)
    b.view(-1)[0] = a.detach().view(-1)[0]          # an exact tie: sign(0) = 0
    loss = F.l1_loss(a, b)
    (loss * 0.37).backward()
    got = (loss.detach().clone(), a.grad.clone())
    a.grad = None
    ref = torch.abs(a - b).mean()
    (ref * 0.37).backward()
    # float32 mean of n terms: the reduction orders differ, bound 1e-6 relative
    _close(got[0], ref, 1e-6)
    _close(got[1], a.grad, 1e-6)
    # determinism: same bits twice
    assert F.l1_loss(a, b).item() == loss.item()


def test_fused_ops_refuse_cpu_tensors():
    F = pkg("fused")
    with pytest.raises(RuntimeError, match="no CPU path"):
        F.l1_loss(torch.rand(4), torch.rand(4))
