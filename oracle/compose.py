"""The composed fine-stage iteration on the CPU oracle -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

One view of train.py:168-229 restricted to the hot path: deformation query (oracle/deformation_ref.py, the PyTorch-CPU
restatement of scene/deformation.py + scene/hexplane.py) -> activations (scene/gaussian_model.py:38-43 as applied at
gaussian_renderer/renderer.py:145-147) -> rasterizer forward (oracle/rasterizer_ref.c) -> L1 loss against a target
image (utils/loss_utils.py:20-21) -> rasterizer backward -> autograd through the activations and the deformation
field.  bench.py times it (cpu_baseline); tests/test_gpu_c2.py compares the GPU step's gradients with it.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import numpy as np
import torch

from . import deformation_ref as DR
from . import rasterizer_oracle as RO


def config_from_args(a) -> DR.DeformConfig:
    """DeformConfig from the reference-style args object a deform_network was built with."""
    kc = a.kplanes_config
    return DR.DeformConfig(net_width=a.net_width, defor_depth=a.defor_depth, posebase_pe=a.posebase_pe,
                           scale_rotation_pe=a.scale_rotation_pe, force_pe=a.force_pe, time_pe=a.time_pe,
                           plane_channels=kc["output_coordinate_dim"], resolution=list(kc["resolution"]),
                           multires=list(a.multires), no_dx=a.no_dx, no_ds=a.no_ds, no_dr=a.no_dr, no_do=a.no_do,
                           no_dshs=a.no_dshs, apply_rotation=a.apply_rotation, use_time=a.use_time, use_force=a.use_force)


def forward(cfg: DR.DeformConfig, state_dict: Dict[str, torch.Tensor], raw: Dict[str, np.ndarray], cam, t_view: float,
            force: float = 0.0, train: bool = True, sh_degree: int = 3, bg=(1.0, 1.0, 1.0), capture: Optional[dict] = None,
            forced_raster_inputs: Optional[dict] = None) -> dict:
    """Deformation query + activations + rasterizer forward for one view.  raw: xyz, log_scales, rot, opacity_logit, sh
    (float32 numpy, pre-activation); cam: synthetic.CameraParams.  Returns the deformed / activated Gaussians, the
    rasterizer oracle's forward dict ("fwd") and, with train=True, the live autograd graph for backward().

    Stage-wise comparison ("teacher forcing", tests/test_gpu_c2.py): the composed function is chaotic in its
    intermediate values -- a 1e-6 difference in a deformed depth swaps two splats in a tile's list and moves a pixel by
    alpha*T -- so two correct float32 implementations can only be compared stage by stage on identical inputs.
    forced_raster_inputs {means3D, scales, rotations, opacity, shs} (numpy: the OTHER implementation's activated
    deformation outputs) replaces the rasterizer's inputs."""
    sd = {k: v.detach().cpu().contiguous().clone().requires_grad_(
        train and v.dtype == torch.float32 and "poc" not in k and "aabb" not in k) for k, v in state_dict.items()}
    P = raw["xyz"].shape[0]
    x = {k: torch.from_numpy(np.ascontiguousarray(raw[k])).clone().requires_grad_(train)
         for k in ("xyz", "log_scales", "rot", "opacity_logit", "sh")}
    H, W = cam.image_height, cam.image_width
    t0 = time.perf_counter()
    with torch.set_grad_enabled(train):
        m3, sc, rot, op, sh = DR.forward_dynamic(sd, cfg, x["xyz"], x["log_scales"], x["rot"], x["opacity_logit"], x["sh"],
                                                 torch.full((P, 1), float(t_view)), torch.full((P, 1), float(force)), capture)
        sc_a, rot_a, op_a = torch.exp(sc), torch.nn.functional.normalize(rot), torch.sigmoid(op)
    ri = forced_raster_inputs or dict(means3D=m3.detach().numpy(), scales=sc_a.detach().numpy(),
                                      rotations=rot_a.detach().numpy(), opacity=op_a.detach().numpy(), shs=sh.detach().numpy())
    c32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32))
    f = RO.forward(np.asarray(bg, np.float32), c32(ri["means3D"]), None, c32(ri["opacity"]), c32(ri["scales"]),
                   c32(ri["rotations"]), 1.0, None, cam.viewmatrix, cam.projmatrix, cam.tanfovx, cam.tanfovy, H, W,
                   c32(ri["shs"]), sh_degree, cam.campos)
    return dict(fwd=f, means3D=m3.detach(), scales=sc_a.detach(), rotations=rot_a.detach(), opacity=op_a.detach(),
                shs=sh.detach(), seconds=time.perf_counter() - t0, _graph=(m3, sc_a, rot_a, op_a, sh), _x=x, _sd=sd, H=H, W=W)


def backward(ctx: dict, target: np.ndarray, pixel_weight: Optional[np.ndarray] = None,
             forced_raster_grads: Optional[dict] = None) -> dict:
    """L1 loss against `target` [3,H,W] (loss = mean |w*render - w*target| with the optional per-pixel weight w [H,W],
    which is how the parity test removes pixels whose discrete decisions sit within rounding of a threshold),
    rasterizer backward, autograd through the activations and the deformation field.  forced_raster_grads
    {dL_dmeans3D, dL_dscales, dL_drotations, dL_dopacity, dL_dsh} (the OTHER implementation's gradients at the
    rasterizer inputs) replaces what is pushed into the deformation graph (teacher forcing, see forward()).
    Adds loss, raster_grads, grads (five Gaussian parameter tensors), param_grads (live deformation parameters)."""
    t0 = time.perf_counter()
    f, H, W = ctx["fwd"], ctx["H"], ctx["W"]
    w = np.ones((H, W), np.float32) if pixel_weight is None else pixel_weight.astype(np.float32)
    diff = f["color"] * w[None] - target * w[None]
    out = dict(ctx)
    out["loss"] = float(np.abs(diff.astype(np.float64)).mean())
    gc = (np.sign(diff) * w[None] / np.float32(diff.size)).astype(np.float32)
    g = RO.backward(f, gc, np.zeros((1, H, W), np.float32))
    gsrc = forced_raster_grads or g
    m3, sc_a, rot_a, op_a, sh = ctx["_graph"]
    tg = lambda k, shape: torch.from_numpy(np.ascontiguousarray(np.asarray(gsrc[k], np.float32)).reshape(shape))
    torch.autograd.backward([m3, sc_a, rot_a, op_a, sh],
                            [tg("dL_dmeans3D", m3.shape), tg("dL_dscales", sc_a.shape), tg("dL_drotations", rot_a.shape),
                             tg("dL_dopacity", op_a.shape), tg("dL_dsh", sh.shape)])
    out["raster_grads"] = g
    out["grads"] = {k: v.grad for k, v in ctx["_x"].items()}
    out["param_grads"] = {k: v.grad for k, v in ctx["_sd"].items() if v.requires_grad and v.grad is not None}
    out["seconds"] = ctx["seconds"] + time.perf_counter() - t0
    return out


def step(cfg: DR.DeformConfig, state_dict: Dict[str, torch.Tensor], raw: Dict[str, np.ndarray], cam, t_view: float,
         force: float = 0.0, target: Optional[np.ndarray] = None, train: bool = True, sh_degree: int = 3,
         bg=(1.0, 1.0, 1.0)) -> dict:
    """One whole iteration (forward; with train=True and a target also backward): what bench.py times."""
    ctx = forward(cfg, state_dict, raw, cam, t_view, force, train and target is not None, sh_degree, bg)
    return backward(ctx, target) if (train and target is not None) else ctx
