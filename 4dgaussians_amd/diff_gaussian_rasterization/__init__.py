"""Drop-in for the reference's `diff_gaussian_rasterization` Python package
(RAST/diff_gaussian_rasterization/__init__.py, RAST = submodules/depth-diff-gaussian-rasterization):
GaussianRasterizationSettings (:158-170), GaussianRasterizer (:172-221), rasterize_gaussians
(:21-42) and the autograd function (:44-156), with the same argument order, validation
messages, return tuple `(color, radii, depth)`, gradient order and debug-snapshot behaviour.
The native layer underneath is libgs4d.so (hand-written gfx950 HIP) instead of the CUDA `_C`.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


def _call_with_snapshot(fn, args, debug, dump_name, note):
    if not debug:
        return fn(*args)
    cpu_args = cpu_deep_copy_tuple(args)  # copy before they can be corrupted
    try:
        return fn(*args)
    except Exception:
        torch.save(cpu_args, dump_name)
        print(note)
        raise


_zeros_cache = {}


def _shared_zeros(shape, device):
    """A read-only zero tensor for an absent upstream gradient (the library only reads it), kept per shape and
    device instead of being allocated and filled in every backward."""
    key = (tuple(shape), device.type, device.index)
    z = _zeros_cache.get(key)
    if z is None:
        if len(_zeros_cache) >= 8:
            _zeros_cache.clear()
        z = _zeros_cache[key] = torch.zeros(shape, dtype=torch.float32, device=device)
    return z


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        s = raster_settings
        args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh, s.sh_degree,
                s.campos, s.prefiltered, s.debug)
        # no input asks for a gradient (render_for_image, evaluation): the blend need not leave the backward's checkpoints
        fn = _C.rasterize_gaussians if any(ctx.needs_input_grad) else _C.rasterize_gaussians_forward_only
        num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = _call_with_snapshot(
            fn, args, s.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered
        # autograd would otherwise launch two fills per backward: zeros for the (unused) depth gradient and for
        # the non-differentiable radii
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        s = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = _shared_zeros((3, s.image_height, s.image_width), means3D.device)
        if grad_depth is None:
            grad_depth = _shared_zeros((1, s.image_height, s.image_width), means3D.device)
        args = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, grad_out_color, grad_depth, sh, s.sh_degree,
                s.campos, geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, s.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_with_snapshot(
            _C.rasterize_gaussians_backward, args, s.debug, "snapshot_bw.dump",
            "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    @torch.no_grad()
    def markVisible(self, positions):
        s = self.raster_settings
        return _C.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = lambda: torch.Tensor([])
        shs = empty() if shs is None else shs
        colors_precomp = empty() if colors_precomp is None else colors_precomp
        scales = empty() if scales is None else scales
        rotations = empty() if rotations is None else rotations
        cov3D_precomp = empty() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
