"""Densification control on the CPU, numpy only -- TEST INFRASTRUCTURE (only tests/ may import this module).

A restatement of the reference's adaptive density control on plain arrays, independent of torch and of any optimizer
object: scene/gaussian_model.py:593-597 (densify), :460-498 (densify_and_clone), :434-458 (densify_and_split),
:583-590 (prune), :376-389 + :356-374 (prune_points / optimizer state), :391-432 (cat_tensors_to_optimizer /
densification_postfix), :297-300 + :343-354 (reset_opacity), :613-615 (add_densification_stats),
utils/general_utils.py:86-107 (build_rotation).

PARITY UNPINNED against recorded reference outputs: the reference holds no fixtures for this path and its module cannot
be imported here (it needs the CUDA-only simple_knn / rasterizer extensions and open3d at import time).  The statements
below follow the reference's lines one by one; the product's torch implementation is compared with them on the same
inputs (tests/test_densify_oracle.py), with the Gaussian samples of the split (torch.normal in the reference, :446)
recorded from the product run and injected here, since the draw itself is the framework's generator, not the algorithm.

State: a dict of float32 arrays
    xyz [P,3], f_dc [P,1,3], f_rest [P,15,3], opacity [P,1], scaling [P,3] (log), rotation [P,4]   -- parameters
    m_<name>, v_<name>                                                                             -- Adam moments
    xyz_gradient_accum [P,1], denom [P,1], max_radii2D [P], deformation_accum [P,3], deformation_table [P] bool
"""
from __future__ import annotations

from typing import Callable, Dict

import numpy as np

PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
f32 = np.float32


def build_rotation(r: np.ndarray) -> np.ndarray:
    """utils/general_utils.py:86-107, float32 throughout, same expression order."""
    r = r.astype(f32)
    norm = np.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = np.zeros((q.shape[0], 3, 3), f32)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    two, one = f32(2), f32(1)
    R[:, 0, 0] = one - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - w * z)
    R[:, 0, 2] = two * (x * z + w * y)
    R[:, 1, 0] = two * (x * y + w * z)
    R[:, 1, 1] = one - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - w * x)
    R[:, 2, 0] = two * (x * z - w * y)
    R[:, 2, 1] = two * (y * z + w * x)
    R[:, 2, 2] = one - two * (x * x + y * y)
    return R


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(f32)


def prune_points(s: Dict[str, np.ndarray], mask: np.ndarray) -> None:
    """gaussian_model.py:376-389: rows with mask=True leave every per-Gaussian array, Adam moments included (:356-374)."""
    valid = ~mask
    for k in PARAMS:
        s[k] = s[k][valid]
        s["m_" + k], s["v_" + k] = s["m_" + k][valid], s["v_" + k][valid]
    for k in ("deformation_accum", "xyz_gradient_accum", "deformation_table", "denom", "max_radii2D"):
        s[k] = s[k][valid]


def densification_postfix(s: Dict[str, np.ndarray], new: Dict[str, np.ndarray], new_table: np.ndarray) -> None:
    """gaussian_model.py:391-432: new rows go to the END of every parameter, their moments start at zero, and the
    densification statistics of ALL rows restart from zero."""
    for k in PARAMS:
        ext = new[k].astype(f32)
        s[k] = np.concatenate((s[k], ext), axis=0)
        s["m_" + k] = np.concatenate((s["m_" + k], np.zeros_like(ext)), axis=0)
        s["v_" + k] = np.concatenate((s["v_" + k], np.zeros_like(ext)), axis=0)
    n = s["xyz"].shape[0]
    s["deformation_table"] = np.concatenate((s["deformation_table"], new_table), axis=-1)
    s["xyz_gradient_accum"] = np.zeros((n, 1), f32)
    s["deformation_accum"] = np.zeros((n, 3), f32)
    s["denom"] = np.zeros((n, 1), f32)
    s["max_radii2D"] = np.zeros((n,), f32)


def densify_and_clone(s, grads, grad_threshold, scene_extent, percent_dense) -> np.ndarray:
    """gaussian_model.py:460-498.  Returns the selection mask."""
    sel = np.sqrt((grads.astype(f32) ** 2).sum(-1)) >= f32(grad_threshold)
    sel &= np.exp(s["scaling"]).max(axis=1) <= f32(percent_dense * scene_extent)
    densification_postfix(s, {k: s[k][sel] for k in PARAMS}, s["deformation_table"][sel])
    return sel


def densify_and_split(s, grads, grad_threshold, scene_extent, percent_dense, normal: Callable[[np.ndarray], np.ndarray],
                      N: int = 2) -> np.ndarray:
    """gaussian_model.py:434-458.  normal(stds) -> zero-mean Gaussian samples with the given per-element standard
    deviations (the torch.normal call of :446).  Returns the selection mask over the rows present at entry."""
    n_init = s["xyz"].shape[0]
    padded = np.zeros((n_init,), f32)
    padded[:grads.shape[0]] = grads.reshape(-1)              # rows cloned a moment ago carry no statistics (:437-438)
    scale = np.exp(s["scaling"])
    sel = (padded >= f32(grad_threshold)) & (scale.max(axis=1) > f32(percent_dense * scene_extent))
    if not sel.any():
        return sel
    stds = np.tile(scale[sel], (N, 1))
    samples = np.asarray(normal(stds), f32)
    rots = np.tile(build_rotation(s["rotation"][sel]), (N, 1, 1))
    new_xyz = np.einsum("nij,nj->ni", rots, samples).astype(f32) + np.tile(s["xyz"][sel], (N, 1))
    new = dict(xyz=new_xyz, f_dc=np.tile(s["f_dc"][sel], (N, 1, 1)), f_rest=np.tile(s["f_rest"][sel], (N, 1, 1)),
               opacity=np.tile(s["opacity"][sel], (N, 1)), scaling=np.log(np.tile(scale[sel], (N, 1)) / f32(0.8 * N)),
               rotation=np.tile(s["rotation"][sel], (N, 1)))
    k = int(sel.sum())
    densification_postfix(s, new, np.tile(s["deformation_table"][sel], N))
    prune_points(s, np.concatenate((sel, np.zeros(N * k, bool))))
    return sel


def densify(s, max_grad, extent, percent_dense, normal) -> Dict[str, np.ndarray]:
    """gaussian_model.py:593-597: average screen-space gradient (0 / 0 -> 0), clone the small, split the large."""
    with np.errstate(invalid="ignore", divide="ignore"):
        grads = s["xyz_gradient_accum"] / s["denom"]
    grads[np.isnan(grads)] = 0.0
    c = densify_and_clone(s, grads, max_grad, extent, percent_dense)
    p = densify_and_split(s, grads, max_grad, extent, percent_dense, normal)
    return dict(cloned=c, split=p)


def prune(s, min_opacity, extent, max_screen_size) -> np.ndarray:
    """gaussian_model.py:583-590."""
    mask = (_sigmoid(s["opacity"]) < f32(min_opacity)).reshape(-1)
    if max_screen_size:
        mask = mask | (s["max_radii2D"] > max_screen_size) | (np.exp(s["scaling"]).max(axis=1) > f32(0.1 * extent))
    prune_points(s, mask)
    return mask


def reset_opacity(s) -> None:
    """gaussian_model.py:297-300, :343-354: opacities capped at 0.01, their moments zeroed."""
    o = np.minimum(_sigmoid(s["opacity"]), f32(0.01))
    s["opacity"] = np.log(o / (f32(1) - o)).astype(f32)
    s["m_opacity"], s["v_opacity"] = np.zeros_like(s["opacity"]), np.zeros_like(s["opacity"])


def add_densification_stats(s, viewspace_grad: np.ndarray, update_filter: np.ndarray) -> None:
    """gaussian_model.py:613-615."""
    g = viewspace_grad[update_filter, :2].astype(f32)
    s["xyz_gradient_accum"][update_filter] += np.sqrt((g * g).sum(-1, keepdims=True))
    s["denom"][update_filter] += 1
