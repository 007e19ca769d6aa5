"""Independent dense float64 formulation of the rasterizer's image function.

Purpose: cross-check oracle/rasterizer_ref.c (SURVEY.md section 8(c) pin (ii)).  It
shares no code with the oracle: matrices are ordinary row-major torch tensors,
the image is evaluated densely (every Gaussian against every pixel, no tiles), and
gradients come from torch.autograd instead of hand-derived formulas.

The image function follows RAST/cuda_rasterizer/forward.cu:74-152,155-256,261-379:
EWA projection with the +0.3 low-pass, un-normalised quaternion, SH colours with
a clamp at 0, front-to-back blending ordered by float32 view depth (stable), with
the discrete rules alpha = min(0.99, o*G), skip alpha < 1/255, skip power > 0 and
stop when T*(1-alpha) < 1e-4.  Discrete decisions are taken without gradient.

Known, documented differences from the reference's backward that this file
reproduces on purpose so that autograd matches it:
  * the frustum clamp of t.x / t.y (forward.cu:82-87) is treated as a constant
    when active (backward.cu:175-176,262-263 zero those partials);
The 1e-7 regulariser in backward.cu:203 is NOT reproduced (its effect is
<= 1e-7 / denom^2 relative, far below the test tolerance for the scenes used).
Tile culling: the formulation itself is dense; the integer tile rectangle of each
Gaussian (which pixels can see it at all) is passed in as a boolean mask `cover`
computed by the test from the oracle's radii / means2D, because a pixel just outside
a Gaussian's tile rectangle can still have alpha >= 1/255 (up to 3.33 sigma) and the
reference deliberately drops it.
"""
from __future__ import annotations

import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, dirs):
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = C0 * sh[:, 0]
    if deg > 0:
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] \
            + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8]
    if deg > 2:
        res = res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] \
            + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] \
            + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14] \
            + C3[6] * x * (xx - 3 * yy) * sh[:, 15]
    return torch.clamp_min(res + 0.5, 0.0)


def render64(means3D, scales, rotations, opacities, sh, deg, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
             H, W, bg, scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None, cover=None):
    """All tensor args float64 (requires_grad where wanted).  viewmatrix/projmatrix are the
    reference's transposed ("row-vector") matrices: p_view = [p,1] @ viewmatrix."""
    P = means3D.shape[0]
    dt = torch.float64
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    pv = ph @ viewmatrix                     # (P,4) view space
    hom = ph @ projmatrix
    pw = 1.0 / (hom[:, 3:4] + 1e-7)
    ndc = hom[:, :3] * pw
    visible = (pv[:, 2].detach().float() > 0.2)  # float32 compare like the kernel (inputs are float32-representable)

    if cov3D_precomp is None:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(P, 3, 3)  # row-major
        S = torch.diag_embed(scales * scale_modifier)
        L = Rm @ S
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], -1).reshape(P, 3, 3)

    fx = W / (2 * tanfovx)
    fy = H / (2 * tanfovy)
    tz = pv[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz = pv[:, 0] / tz
    tytz = pv[:, 1] / tz
    cx = (txtz.detach() < -limx) | (txtz.detach() > limx)
    cy = (tytz.detach() < -limy) | (tytz.detach() > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx)).detach() * tz, pv[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy)).detach() * tz, pv[:, 1])
    # when clamped the reference keeps d/dtz of the J entries with t.x held fixed: detach the product
    tx = torch.where(cx, tx.detach(), tx)
    ty = torch.where(cy, ty.detach(), ty)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz),
                     zero, fy / tz, -(fy * ty) / (tz * tz)], -1).reshape(P, 2, 3)
    Wr = viewmatrix[:3, :3].T                # world->view rotation (row-major, column-vector convention)
    A = J @ Wr                               # (P,2,3)
    cov2 = A @ Sigma @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    conA, conB, conC = c_ / det, -b / det, a / det

    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5

    if colors_precomp is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = sh_to_rgb(deg, sh, d)
    else:
        rgb = colors_precomp
    depth = pv[:, 2]

    order = torch.argsort(depth.detach().float(), stable=True)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    T = torch.ones(H, W, dtype=dt)
    Cimg = torch.zeros(3, H, W, dtype=dt)
    Dimg = torch.zeros(H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    n_contrib = torch.zeros(H, W, dtype=torch.int64)
    for g in order.tolist():
        if not bool(visible[g]):
            continue
        dx = px[g] - xs
        dy = py[g] - ys
        power = -0.5 * (conA[g] * dx * dx + conC[g] * dy * dy) - conB[g] * dx * dy
        G = torch.exp(power)
        alpha = torch.clamp_max(opacities[g] * G, 0.99)
        with torch.no_grad():
            use = (~done) & (power <= 0) & (alpha >= 1.0 / 255.0)
            if cover is not None:
                use = use & cover[g]
            test_T = T * (1 - alpha)
            stop = use & (test_T < 1e-4)
            done = done | stop
            use = use & ~stop
        w = torch.where(use, alpha * T, torch.zeros_like(T))
        Cimg = Cimg + rgb[g][:, None, None] * w[None]
        Dimg = Dimg + depth[g] * w
        T = torch.where(use, T * (1 - alpha), T)
    color = Cimg + T[None] * bg[:, None, None]
    return color, Dimg[None], T
