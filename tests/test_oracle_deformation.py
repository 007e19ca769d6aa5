"""Pins oracle/deformation_ref.py against golden vectors produced by the reference's own
scene/hexplane.py + scene/deformation.py (tests/golden/make_deformation_golden.py)."""
import pytest
import torch

from golden_utils import CASES, load_case
from oracle import deformation_ref as R

ORDER = ("means3D", "scales", "rotations", "opacity", "shs")


@pytest.mark.parametrize("case", CASES)
def test_forward_and_gradients_match_reference_golden(case):
    cfg, sd, ins, outs, ws, gins, grads = load_case(case)
    sd = {k: v.clone().requires_grad_(v.dtype == torch.float32 and k in grads) for k, v in sd.items()}
    x = {k: ins[k].clone().requires_grad_(True) for k in ("points", "scales", "rotations", "opacity", "shs")}
    got = R.forward_dynamic(sd, cfg, x["points"], x["scales"], x["rotations"], x["opacity"], x["shs"], ins["time"], ins["force"])
    for name, g in zip(ORDER, got):
        assert (g - outs[name]).abs().max() <= 1e-6 * max(1.0, float(outs[name].abs().max())), name
    sum((g * ws[n]).sum() for n, g in zip(ORDER, got)).backward()
    for k, want in gins.items():
        if want.numel():
            assert (x[k].grad - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max())), k
    for k, want in grads.items():
        g = sd[k].grad
        assert g is not None, k
        assert (g - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max())), k


def test_poc_fre_layout():
    x = torch.tensor([[1.0, 2.0]])
    e = R.poc_fre(x, 2)
    want = torch.tensor([[1.0, 2.0, 1.0, 2.0, 2.0, 4.0]])
    assert torch.allclose(e[:, :2], x)
    assert torch.allclose(e[:, 2:6], want[:, 2:].sin()) and torch.allclose(e[:, 6:], want[:, 2:].cos())


def test_non_blend_path_is_rejected():
    cfg, sd, ins, *_ = load_case("plain_small")
    cfg.blend_time_force = False
    with pytest.raises(NotImplementedError):
        R.forward_dynamic(sd, cfg, ins["points"], ins["scales"], ins["rotations"], ins["opacity"], ins["shs"], ins["time"], ins["force"])
