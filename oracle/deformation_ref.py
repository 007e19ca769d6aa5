"""oracle/deformation_ref.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (PyTorch-CPU float32, differentiable through autograd) of the reference's
HexPlane deformation field, written functionally over a flat state_dict that uses the
reference's own parameter names.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.

Reference restated (paths relative to /root/reference):
  scene/hexplane.py:18-19     normalize_aabb (aabb[0] = xyz_max, aabb[1] = xyz_min -> max maps to -1)
  scene/hexplane.py:21-47     grid_sample_wrapper: bilinear, align_corners=True, padding_mode='border'
  scene/hexplane.py:100-131   interpolate_ms_features: planes (xy,xz,xt,yz,yt,zt) multiplied, levels concatenated
  scene/hexplane.py:192-215   get_density: time is appended un-normalised
  scene/deformation.py:128-141 query_time_force: t' = force_time_embed([t, f]) when blend_time_force
  scene/deformation.py:147-200 forward_dynamic: trunk + heads (each head: ReLU, Linear, ReLU, Linear on the
                               concatenation [hidden, pe_tail, time_emb, force_emb])
  scene/deformation.py:265-274, 291-298 deform_network.forward_dynamic and poc_fre
  utils/graphics_utils.py:110-133 batch_quaternion_multiply (+ normalisation)

PARITY STATUS: pinned.  tests/golden/deformation_*.npz hold inputs, outputs and gradients
produced by importing the reference's own scene/hexplane.py + scene/deformation.py in the build
container (generator: tests/golden/make_deformation_golden.py); tests/test_oracle_deformation.py
checks this restatement against them.

The bilinear sampling is written out by hand (index arithmetic + gathers) instead of calling
F.grid_sample, so that it is an independent statement of the sampling rule the HIP kernel follows.
Only blend_time_force=True is supported: the reference's non-blend path crashes.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import Dict, List

import torch


@dataclass
class DeformConfig:
    """The fields of the reference's args object that the deformation field reads
    (scene/deformation.py:24-38,148-192,231-238; defaults arguments/__init__.py:76-107)."""
    net_width: int = 64
    defor_depth: int = 1
    posebase_pe: int = 10
    scale_rotation_pe: int = 2
    opacity_pe: int = 2
    force_pe: int = 2
    time_pe: int = 2
    bounds: float = 1.6
    plane_channels: int = 32                      # kplanes_config['output_coordinate_dim']
    resolution: List[int] = field(default_factory=lambda: [64, 64, 64, 75])
    multires: List[int] = field(default_factory=lambda: [1, 2])
    no_dx: bool = False
    no_ds: bool = False
    no_dr: bool = False
    no_do: bool = True
    no_dshs: bool = False
    apply_rotation: bool = True
    use_time: bool = True
    use_force: bool = False
    blend_time_force: bool = True

    @property
    def feat_dim(self) -> int:
        return self.plane_channels * len(self.multires)

    @property
    def deform_extra_dim(self) -> int:
        return 1 + 2 * self.force_pe + 1 + 2 * self.time_pe


PLANES = list(itertools.combinations(range(4), 2))  # (x,y) (x,z) (x,t) (y,z) (y,t) (z,t)


def poc_fre(x: torch.Tensor, n_freq: int) -> torch.Tensor:
    """[x, sin(x*2^k) (dim-major, freq-minor), cos(...)]"""
    freqs = torch.tensor([2.0 ** i for i in range(n_freq)], dtype=x.dtype)
    e = (x.unsqueeze(-1) * freqs).flatten(-2)
    return torch.cat([x, e.sin(), e.cos()], -1)


def bilinear_border(plane: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """plane [1,C,Hh,Ww]; u -> width axis, v -> height axis, both in [-1,1] normalised coords
    (align_corners=True, border padding).  Returns [n, C]."""
    _, C, Hh, Ww = plane.shape
    x = ((u + 1) * 0.5 * (Ww - 1)).clamp(0, Ww - 1)
    y = ((v + 1) * 0.5 * (Hh - 1)).clamp(0, Hh - 1)
    x0 = x.floor()
    y0 = y.floor()
    wx = x - x0
    wy = y - y0
    x0i = x0.long()
    y0i = y0.long()
    x1i = (x0i + 1).clamp(max=Ww - 1)
    y1i = (y0i + 1).clamp(max=Hh - 1)
    p = plane[0].permute(1, 2, 0)  # [Hh, Ww, C]
    f00, f01 = p[y0i, x0i], p[y0i, x1i]
    f10, f11 = p[y1i, x0i], p[y1i, x1i]
    wx = wx[:, None]
    wy = wy[:, None]
    return f00 * (1 - wx) * (1 - wy) + f01 * wx * (1 - wy) + f10 * (1 - wx) * wy + f11 * wx * wy


def hexplane_features(sd: Dict[str, torch.Tensor], cfg: DeformConfig, xyz: torch.Tensor, t: torch.Tensor,
                      prefix: str = "deformation_net.grid.") -> torch.Tensor:
    aabb = sd[prefix + "aabb"]
    pts = (xyz - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0
    pts = torch.cat([pts, t], -1)
    feats = []
    for lvl in range(len(cfg.multires)):
        acc = 1.0
        for ci, (a, b) in enumerate(PLANES):
            plane = sd[f"{prefix}grids.{lvl}.{ci}"]
            acc = acc * bilinear_border(plane, pts[:, a], pts[:, b])
        feats.append(acc)
    return torch.cat(feats, -1)


def _linear(sd, name, x):
    return x @ sd[name + ".weight"].T + sd[name + ".bias"]


def _note_relu(cap, x):
    """Track, per point, how close any ReLU input comes to zero: an input within float32 rounding of 0 is
    a discrete decision that two correct float32 implementations may take differently (test masks)."""
    if cap is not None:
        a = x.detach().abs()
        m = torch.where(a == 0, torch.full_like(a, float("inf")), a).min(dim=1).values   # exact zeros are exact everywhere
        cap["relu_margin"] = m if "relu_margin" not in cap else torch.minimum(cap["relu_margin"], m)


def _head(sd, name, x, cap=None):
    _note_relu(cap, x)
    h1 = _linear(sd, name + ".1", torch.relu(x))
    _note_relu(cap, h1)
    return _linear(sd, name + ".3", torch.relu(h1))


def quat_mul_normalized(q1, q2):
    w = q1[:, 0] * q2[:, 0] - q1[:, 1] * q2[:, 1] - q1[:, 2] * q2[:, 2] - q1[:, 3] * q2[:, 3]
    x = q1[:, 0] * q2[:, 1] + q1[:, 1] * q2[:, 0] + q1[:, 2] * q2[:, 3] - q1[:, 3] * q2[:, 2]
    y = q1[:, 0] * q2[:, 2] - q1[:, 1] * q2[:, 3] + q1[:, 2] * q2[:, 0] + q1[:, 3] * q2[:, 1]
    z = q1[:, 0] * q2[:, 3] + q1[:, 1] * q2[:, 2] - q1[:, 2] * q2[:, 1] + q1[:, 3] * q2[:, 0]
    q = torch.stack((w, x, y, z), 1)
    return q / torch.norm(q, dim=1, keepdim=True)


def forward_dynamic(sd: Dict[str, torch.Tensor], cfg: DeformConfig, points, scales, rotations, opacity, shs, time,
                    force, capture=None):
    """deform_network.forward_dynamic with prev_frames=None -> (means3D, scales, rotations, opacity, shs).
    capture: optional dict that receives "relu_margin" [P] (see _note_relu)."""
    if not cfg.blend_time_force:
        raise NotImplementedError("only blend_time_force=True runs in the reference (SURVEY.md fact 3)")
    D = "deformation_net."
    point_emb = poc_fre(points, cfg.posebase_pe)
    scales_emb = poc_fre(scales, cfg.scale_rotation_pe)
    rot_emb = poc_fre(rotations, cfg.scale_rotation_pe)
    force_emb = poc_fre(force, cfg.force_pe)
    time_emb = poc_fre(time, cfg.time_pe)
    if not cfg.use_time:
        time_emb = time_emb * 0.0
    if not cfg.use_force:
        force_emb = force_emb * 0.0
    tprime = _linear(sd, D + "force_time_embed", torch.cat((time_emb[:, :1], force_emb[:, :1]), 1))
    feat = hexplane_features(sd, cfg, point_emb[:, :3], tprime)
    h = _linear(sd, D + "feature_out2.0", feat)
    for i in range(max(cfg.defor_depth - 1, 0)):
        _note_relu(capture, h)
        h = _linear(sd, D + f"feature_out2.{2 * i + 2}", torch.relu(h))
    pts = points if cfg.no_dx else points + _head(sd, D + "pos_deform", torch.cat((h, point_emb[:, 3:], time_emb, force_emb), 1), capture)
    sc = scales if cfg.no_ds else scales + _head(sd, D + "scales_deform", torch.cat((h, scales_emb[:, 3:], time_emb, force_emb), 1), capture)
    if cfg.no_dr:
        rot = rotations
    else:
        dr = _head(sd, D + "rotations_deform", torch.cat((h, rot_emb[:, 4:], time_emb, force_emb), 1), capture)
        rot = quat_mul_normalized(rotations, dr) if cfg.apply_rotation else rotations + dr
    op = opacity if cfg.no_do else opacity + _head(sd, D + "opacity_deform", torch.cat((h, time_emb, force_emb), 1), capture)
    sh = shs if cfg.no_dshs else shs + _head(sd, D + "shs_deform", torch.cat((h, time_emb, force_emb), 1), capture).reshape(shs.shape[0], 16, 3)
    return pts, sc, rot, op, sh
