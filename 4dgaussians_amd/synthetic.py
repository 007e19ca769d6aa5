"""Deterministic synthetic workloads for tests and bench.py.

The reference ships no data (its loader needs a private dataset), so every workload
is a seeded, shape-matched stand-in:

* Gaussians: xyz ~ U(-1.3, 1.3)^3 (the box of scene/dataset_readers.py:450-455),
  log-scales ~ N(log s0, 0.6^2), rotations = normalised N(0,1)^4, opacity logits
  ~ N(0, 2^2), f_dc ~ N(0, 0.5^2), f_rest ~ N(0, 0.05^2) in the (P,16,3) layout of
  GaussianModel.get_features (scene/gaussian_model.py:143-146).
* Camera: on a sphere of radius 4 looking at the origin, FoVx = 0.6911112 rad, matrices
  assembled exactly like scene/cameras.py:103-106 with utils/graphics_utils.py:38-68
  (znear 0.01, zfar 100): viewmatrix = W2C transposed, projmatrix = viewmatrix @ P^T,
  campos = inverse(viewmatrix)[3, :3].

Host-side plumbing only (numpy); no GPU work happens here.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

FOVX_DNERF = 0.6911112

# name -> (W, H, P, batch, sh_degree).  Index in this dict == config_index of BASELINE.json.
CONFIGS = {
    "C1": dict(W=800, H=800, P=40_000, batch=1, sh_degree=3),
    "C2": dict(W=800, H=800, P=100_000, batch=1, sh_degree=3),
    "C3": dict(W=536, H=960, P=100_000, batch=4, sh_degree=3),
    "C4": dict(W=1352, H=1014, P=300_000, batch=8, sh_degree=3),
    "C5": dict(W=3840, H=2160, P=5_000_000, batch=64, sh_degree=3),
}
# s0 = TILE_SCALE_K / focal_x gives mean tiles-per-Gaussian ~= 8 (calibrated with the CPU oracle,
# see tests/test_synthetic.py); the achieved R/P is recorded in every bench line.
TILE_SCALE_K = 11.0


def seed_for(config_name: str) -> int:
    return 1234 + list(CONFIGS).index(config_name)


def fov2focal(fov: float, pixels: int) -> float:
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal: float, pixels: int) -> float:
    return 2 * math.atan(pixels / (2 * focal))


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> np.ndarray:
    """utils/graphics_utils.py:48-68 (float32 element assignment)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class CameraParams:
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    tanfovx: float
    tanfovy: float
    viewmatrix: np.ndarray  # (4,4) float32, = W2C^T  (scene/cameras.py:103)
    projmatrix: np.ndarray  # (4,4) float32, = viewmatrix @ P^T (scene/cameras.py:104-105)
    campos: np.ndarray      # (3,)  float32


def look_at_camera(W: int, H: int, azimuth: float = 0.7, elevation: float = 0.35, radius: float = 4.0,
                   fovx: float = FOVX_DNERF) -> CameraParams:
    c = radius * np.array([math.cos(elevation) * math.cos(azimuth),
                           math.cos(elevation) * math.sin(azimuth),
                           math.sin(elevation)], dtype=np.float64)
    fwd = -c / np.linalg.norm(c)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    w2c_R = np.stack([right, down, fwd], axis=0)          # rows: camera axes in world coords
    t = -w2c_R @ c
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = w2c_R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    # getWorld2View2 with zero translate / unit scale round-trips through two inversions in float64
    Rt = np.float32(np.linalg.inv(np.linalg.inv(Rt)))
    view = np.ascontiguousarray(Rt.T)
    fovy = focal2fov(fov2focal(fovx, W), H)
    proj = np.ascontiguousarray(projection_matrix(0.01, 100.0, fovx, fovy).T)
    full = (view @ proj).astype(np.float32)
    campos = np.linalg.inv(view.astype(np.float32))[3, :3].astype(np.float32)
    return CameraParams(W, H, fovx, fovy, math.tan(fovx * 0.5), math.tan(fovy * 0.5),
                        view, np.ascontiguousarray(full), np.ascontiguousarray(campos))


def make_gaussians(P: int, seed: int, s0: float, sh_coeffs: int = 16) -> dict:
    """Raw (pre-activation) Gaussian parameters as float32 numpy arrays."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1.3, 1.3, size=(P, 3)).astype(np.float32)
    log_scales = (math.log(s0) + 0.6 * rng.standard_normal((P, 3))).astype(np.float32)
    rot = rng.standard_normal((P, 4)).astype(np.float32)
    opacity_logit = (2.0 * rng.standard_normal((P, 1))).astype(np.float32)
    sh = np.empty((P, sh_coeffs, 3), dtype=np.float32)
    sh[:, :1] = 0.5 * rng.standard_normal((P, 1, 3))
    sh[:, 1:] = 0.05 * rng.standard_normal((P, sh_coeffs - 1, 3))
    return dict(xyz=xyz, log_scales=log_scales, rot=rot, opacity_logit=opacity_logit, sh=sh)


def activate(g: dict) -> dict:
    """scene/gaussian_model.py:38-43 activations in float32 numpy (exp, L2-normalise, sigmoid)."""
    scales = np.exp(g["log_scales"]).astype(np.float32)
    n = np.sqrt((g["rot"].astype(np.float32) ** 2).sum(-1, keepdims=True)).astype(np.float32)
    rot = (g["rot"] / np.maximum(n, np.float32(1e-12))).astype(np.float32)
    op = (1.0 / (1.0 + np.exp(-g["opacity_logit"].astype(np.float32)))).astype(np.float32)
    return dict(means3D=g["xyz"], scales=scales, rotations=rot, opacities=op, sh=g["sh"])


def make_workload(name: str, P: int | None = None, scale_k: float = TILE_SCALE_K):
    """(activated Gaussian dict, CameraParams) for a BASELINE.json config, optionally with P overridden."""
    cfg = CONFIGS[name]
    cam = look_at_camera(cfg["W"], cfg["H"])
    s0 = scale_k / fov2focal(cam.FoVx, cam.image_width)
    g = make_gaussians(P or cfg["P"], seed_for(name), s0)
    return activate(g), cam, g
