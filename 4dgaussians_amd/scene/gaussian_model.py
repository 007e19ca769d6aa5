"""Gaussian parameter container for the rows after the render path:
densify / clone / split / prune with optimizer-state surgery, and the on-disk formats.

Mirrors the parts of scene/gaussian_model.py a training loop touches, with the same attribute and method names:
  parameters and activations      :30-157   (get_xyz / get_scaling / get_rotation / get_opacity / get_features)
  capture / restore               :88-128   (the 14-item checkpoint tuple of train.py:266-272)
  create_from_pcd                 :159-185  (distCUDA2 initial scales, identity rotations, opacity logit(0.1))
  training_setup, LR schedules    :188-242  (eight AdamW groups; here the fused AdamW of optim.py)
  PLY layout, save / load         :244-256, :281-295, :302-341   (x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,scale_*,rot_*; all f4)
  deformation.pth family          :262-279
  optimizer surgery, densify      :343-458, :583-597, :613-615
  compute_regulation              :679-706  (scene/regulation.py of this package)
Not carried over: grow / add_point_by_mask / downsample_point (open3d point up-sampling, off by default:
`add_point=False`, arguments/__init__.py) raise NotImplementedError.

Host-side logic over torch tensors; the device work it triggers (distCUDA2, the optimizer step, the
regularisers) runs in libgs4d.so.  `device` and `optimizer_cls` exist so the surgery and the file formats can
be exercised without a GPU (tests); the defaults are the HIP path.
"""
from __future__ import annotations

import os
from collections import namedtuple

import numpy as np
import torch
from torch import nn

from .deformation import deform_network
from .regulation import compute_regulation as _compute_regulation

C0 = 0.28209479177387814                     # utils/sh_utils.py:26
BasicPointCloud = namedtuple("BasicPointCloud", ["points", "colors", "normals"])      # utils/graphics_utils.py:17-20


def RGB2SH(rgb):                             # utils/sh_utils.py:114-115
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):                      # utils/general_utils.py:19-20
    return torch.log(x / (1 - x))


def build_rotation(r):                       # utils/general_utils.py:86-107
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    r_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r_ * z)
    R[:, 0, 2] = 2 * (x * z + r_ * y)
    R[:, 1, 0] = 2 * (x * y + r_ * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r_ * x)
    R[:, 2, 0] = 2 * (x * z - r_ * y)
    R[:, 2, 1] = 2 * (y * z + r_ * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """utils/general_utils.py:36-70: log-linear interpolation from lr_init to lr_final with an optional warm-up."""
    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        else:
            delay_rate = 1.0
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return helper


# ---------------------------------------------------------------- PLY (binary little endian / ascii, float properties)
def write_ply(path, names, table: np.ndarray):
    """One `vertex` element of float32 properties, as plyfile writes PlyData([PlyElement.describe(...)])
    (scene/gaussian_model.py:291-295)."""
    table = np.ascontiguousarray(table, dtype="<f4")
    assert table.ndim == 2 and table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % table.shape[0]
    header += "".join("property float %s\n" % n for n in names) + "end_header\n"
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4"}


def read_ply(path):
    """-> dict property name -> numpy array of the FIRST element (the vertex table), for scalar properties."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, seen_element = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if seen_element:                    # only the first element is read; skip the rest of the header
                    while f.readline().strip() != b"end_header":
                        pass
                    break
                seen_element, count = True, int(tok[2])
            elif tok[0] == "property":
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            return {n: rows[:, i].astype(t) for i, (n, t) in enumerate(props)}
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        dt = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        return {n: np.asarray(data[n]) for n, _ in props}


class GaussianModel:
    def __init__(self, sh_degree: int, args, device="cuda", optimizer_cls=None):
        self.device = torch.device(device)
        self._optimizer_cls = optimizer_cls
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._deformation = deform_network(args)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self._deformation_table = torch.empty(0)
        self._deformation_accum = torch.empty(0)
        # blend_time_force = True is the only configuration that runs: gaussian_model.py:67-84
        self.time_grids, self.force_grids, self.plane_grids = [2, 4, 5], [], [0, 1, 3]
        self.args = args
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ---- checkpoint tuple (train.py:266-272)
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._deformation.state_dict(), self._deformation_table,
                self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity, self.max_radii2D,
                self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, deform_state, self._deformation_table, self._features_dc, self._features_rest,
         self._scaling, self._rotation, self._opacity, self.max_radii2D, xyz_gradient_accum, denom, opt_dict,
         self.spatial_lr_scale) = model_args
        self._deformation.load_state_dict(deform_state, strict=False)
        self.training_setup(training_args)
        self.xyz_gradient_accum = xyz_gradient_accum
        self.denom = denom
        try:
            self.optimizer.load_state_dict(opt_dict)
        except Exception:          # the reference swallows optimizer-state mismatches (gaussian_model.py:125-128)
            pass

    # ---- views of the parameters
    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    @property
    def get_aabb(self):
        return self._deformation.get_aabb

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- initialisation (scene/__init__.py:60)
    def create_from_pcd(self, pcd, spatial_lr_scale: float, time_line: int = 0):
        from ..simple_knn._C import distCUDA2
        dev = self.device
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).to(dtype=torch.float32, device=dev)
        fused_color = RGB2SH(torch.tensor(np.asarray(pcd.colors)).to(dtype=torch.float32, device=dev))
        features = torch.zeros((fused_color.shape[0], 3, (self.max_sh_degree + 1) ** 2), dtype=torch.float32, device=dev)
        features[:, :3, 0] = fused_color
        dist2 = torch.clamp_min(distCUDA2(pts.contiguous()), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((pts.shape[0], 4), device=dev)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((pts.shape[0], 1), dtype=torch.float32, device=dev))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._deformation = self._deformation.to(dev)
        self._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self.max_radii2D = torch.zeros((pts.shape[0]), device=dev)
        self._deformation_table = torch.gt(torch.ones((pts.shape[0]), device=dev), 0)

    # ---- optimizer
    def training_setup(self, training_args):
        dev = self._xyz.device
        n = self.get_xyz.shape[0]
        self.percent_dense = training_args.percent_dense
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        s = self.spatial_lr_scale
        l = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * s, "name": "xyz"},
            {"params": list(self._deformation.get_mlp_parameters()), "lr": training_args.deformation_lr_init * s, "name": "deformation"},
            {"params": list(self._deformation.get_grid_parameters()), "lr": training_args.grid_lr_init * s, "name": "grid"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        cls = self._optimizer_cls
        if cls is None:
            from ..optim import FusedAdamW as cls
        self.optimizer = cls(l, lr=0.0001, eps=1e-8)
        self.xyz_scheduler_args = get_expon_lr_func(training_args.position_lr_init * s, training_args.position_lr_final * s,
                                                    lr_delay_mult=training_args.position_lr_delay_mult,
                                                    max_steps=training_args.position_lr_max_steps)
        self.deformation_scheduler_args = get_expon_lr_func(training_args.deformation_lr_init * s, training_args.deformation_lr_final * s,
                                                            lr_delay_mult=training_args.deformation_lr_delay_mult,
                                                            max_steps=training_args.position_lr_max_steps)
        self.grid_scheduler_args = get_expon_lr_func(training_args.grid_lr_init * s, training_args.grid_lr_final * s,
                                                     lr_delay_mult=training_args.deformation_lr_delay_mult,
                                                     max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group["lr"] = self.xyz_scheduler_args(iteration)
            if "grid" in group["name"]:
                group["lr"] = self.grid_scheduler_args(iteration)
            elif group["name"] == "deformation":
                group["lr"] = self.deformation_scheduler_args(iteration)

    # ---- on-disk formats
    def construct_list_of_attributes(self):
        l = ["x", "y", "z", "nx", "ny", "nz"]
        l += ["f_dc_%d" % i for i in range(self._features_dc.shape[1] * self._features_dc.shape[2])]
        l += ["f_rest_%d" % i for i in range(self._features_rest.shape[1] * self._features_rest.shape[2])]
        l.append("opacity")
        l += ["scale_%d" % i for i in range(self._scaling.shape[1])]
        l += ["rot_%d" % i for i in range(self._rotation.shape[1])]
        return l

    def save_ply(self, path):
        xyz = self._xyz.detach().cpu().numpy()
        f_dc = self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        f_rest = self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
        table = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, self._opacity.detach().cpu().numpy(),
                                self._scaling.detach().cpu().numpy(), self._rotation.detach().cpu().numpy()), axis=1)
        write_ply(path, self.construct_list_of_attributes(), table)

    def load_ply(self, path):
        v = read_ply(path)
        dev = self.device
        by_index = lambda prefix: sorted([n for n in v if n.startswith(prefix)], key=lambda x: int(x.split("_")[-1]))
        xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
        features_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[:, :, None]
        extra = by_index("f_rest_")
        assert len(extra) == 3 * (self.max_sh_degree + 1) ** 2 - 3
        features_extra = np.stack([v[n] for n in extra], axis=1).reshape((xyz.shape[0], 3, (self.max_sh_degree + 1) ** 2 - 1))
        scales = np.stack([v[n] for n in by_index("scale_")], axis=1)
        rots = np.stack([v[n] for n in by_index("rot")], axis=1)
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
        self._xyz = nn.Parameter(t(xyz).requires_grad_(True))
        self._features_dc = nn.Parameter(t(features_dc).transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(t(features_extra).transpose(1, 2).contiguous().requires_grad_(True))
        self._opacity = nn.Parameter(t(v["opacity"][..., None]).requires_grad_(True))
        self._scaling = nn.Parameter(t(scales).requires_grad_(True))
        self._rotation = nn.Parameter(t(rots).requires_grad_(True))
        self.active_sh_degree = self.max_sh_degree

    def save_deformation(self, path):
        os.makedirs(path, exist_ok=True)
        torch.save(self._deformation.state_dict(), os.path.join(path, "deformation.pth"))
        torch.save(self._deformation_table, os.path.join(path, "deformation_table.pth"))
        torch.save(self._deformation_accum, os.path.join(path, "deformation_accum.pth"))

    def load_model(self, path):
        dev = self.device
        n = self.get_xyz.shape[0]
        weight_dict = torch.load(os.path.join(path, "deformation.pth"), map_location=dev, weights_only=True)
        self._deformation.load_state_dict(weight_dict)
        self._deformation = self._deformation.to(dev)
        self._deformation_table = torch.gt(torch.ones((n), device=dev), 0)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        if os.path.exists(os.path.join(path, "deformation_table.pth")):
            self._deformation_table = torch.load(os.path.join(path, "deformation_table.pth"), map_location=dev, weights_only=True)
        if os.path.exists(os.path.join(path, "deformation_accum.pth")):
            self._deformation_accum = torch.load(os.path.join(path, "deformation_accum.pth"), map_location=dev, weights_only=True)
        self.max_radii2D = torch.zeros((n), device=dev)

    # ---- optimizer-state surgery
    def reset_opacity(self):
        opacities_new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        self._opacity = self.replace_tensor_to_optimizer(opacities_new, "opacity")["opacity"]

    def replace_tensor_to_optimizer(self, tensor, name):
        out = {}
        for group in self.optimizer.param_groups:
            if group["name"] == name:
                stored = self.optimizer.state.get(group["params"][0], None)
                new = nn.Parameter(tensor.detach().clone().requires_grad_(True))
                if stored is not None:
                    stored["exp_avg"] = torch.zeros_like(tensor)
                    stored["exp_avg_sq"] = torch.zeros_like(tensor)
                    del self.optimizer.state[group["params"][0]]
                    self.optimizer.state[new] = stored
                group["params"][0] = new
                out[name] = new
        return out

    def _prune_optimizer(self, mask):
        out = {}
        for group in self.optimizer.param_groups:
            if len(group["params"]) > 1:
                continue
            old = group["params"][0]
            stored = self.optimizer.state.get(old, None)
            new = nn.Parameter(old.detach()[mask].clone().requires_grad_(True))
            if stored is not None and len(stored):
                stored["exp_avg"] = stored["exp_avg"][mask]
                stored["exp_avg_sq"] = stored["exp_avg_sq"][mask]
                del self.optimizer.state[old]
                self.optimizer.state[new] = stored
            group["params"][0] = new
            out[group["name"]] = new
        return out

    def _adopt(self, t):
        self._xyz, self._features_dc, self._features_rest = t["xyz"], t["f_dc"], t["f_rest"]
        self._opacity, self._scaling, self._rotation = t["opacity"], t["scaling"], t["rotation"]

    def prune_points(self, mask):
        valid = ~mask
        self._adopt(self._prune_optimizer(valid))
        self._deformation_accum = self._deformation_accum[valid]
        self.xyz_gradient_accum = self.xyz_gradient_accum[valid]
        self._deformation_table = self._deformation_table[valid]
        self.denom = self.denom[valid]
        self.max_radii2D = self.max_radii2D[valid]

    def cat_tensors_to_optimizer(self, tensors_dict):
        out = {}
        for group in self.optimizer.param_groups:
            if len(group["params"]) > 1:
                continue
            ext = tensors_dict[group["name"]]
            old = group["params"][0]
            stored = self.optimizer.state.get(old, None)
            new = nn.Parameter(torch.cat((old.detach(), ext.detach()), dim=0).requires_grad_(True))
            if stored is not None and len(stored):
                stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(ext)), dim=0)
                stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.optimizer.state[old]
                self.optimizer.state[new] = stored
            group["params"][0] = new
            out[group["name"]] = new
        return out

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation,
                              new_deformation_table):
        d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
             "scaling": new_scaling, "rotation": new_rotation}
        self._adopt(self.cat_tensors_to_optimizer(d))
        dev, n = self._xyz.device, self.get_xyz.shape[0]
        self._deformation_table = torch.cat([self._deformation_table, new_deformation_table], -1)
        self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        self.denom = torch.zeros((n, 1), device=dev)
        self.max_radii2D = torch.zeros((n), device=dev)

    @torch.no_grad()
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        dev = self._xyz.device
        n_init = self.get_xyz.shape[0]
        padded_grad = torch.zeros((n_init), device=dev)
        padded_grad[:grads.shape[0]] = grads.squeeze()
        sel = torch.where(padded_grad >= grad_threshold, True, False)
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values > self.percent_dense * scene_extent)
        if not sel.any():
            return
        stds = self.get_scaling[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        rots = build_rotation(self._rotation[sel]).repeat(N, 1, 1)
        new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[sel].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(self.get_scaling[sel].repeat(N, 1) / (0.8 * N))
        self.densification_postfix(new_xyz, self._features_dc[sel].repeat(N, 1, 1), self._features_rest[sel].repeat(N, 1, 1),
                                   self._opacity[sel].repeat(N, 1), new_scaling, self._rotation[sel].repeat(N, 1),
                                   self._deformation_table[sel].repeat(N))
        prune_filter = torch.cat((sel, torch.zeros(N * int(sel.sum()), device=dev, dtype=torch.bool)))
        self.prune_points(prune_filter)

    @torch.no_grad()
    def densify_and_clone(self, grads, grad_threshold, scene_extent, *unused, **unused_kw):
        sel = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
        sel = torch.logical_and(sel, torch.max(self.get_scaling, dim=1).values <= self.percent_dense * scene_extent)
        self.densification_postfix(self._xyz[sel], self._features_dc[sel], self._features_rest[sel], self._opacity[sel],
                                   self._scaling[sel], self._rotation[sel], self._deformation_table[sel])

    @torch.no_grad()
    def prune(self, max_grad, min_opacity, extent, max_screen_size):
        prune_mask = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            prune_mask = torch.logical_or(prune_mask, self.max_radii2D > max_screen_size)
            prune_mask = torch.logical_or(prune_mask, self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(prune_mask)

    @torch.no_grad()
    def densify(self, max_grad, min_opacity, extent, max_screen_size, density_threshold=5, displacement_scale=5,
                model_path=None, iteration=None, stage=None):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        self.densify_and_clone(grads, max_grad, extent)
        self.densify_and_split(grads, max_grad, extent)

    @torch.no_grad()
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    @torch.no_grad()
    def update_deformation_table(self, threshold):
        self._deformation_table = torch.gt(self._deformation_accum.max(dim=-1).values / 100, threshold)

    def grow(self, *a, **k):
        raise NotImplementedError("open3d point up-sampling (add_point) is outside this build's scope")

    add_point_by_mask = downsample_point = grow

    def compute_regulation(self, time_smoothness_weight, l1_time_planes_weight, l2_time_planes_weight, plane_tv_weight,
                           force_weight=0.0):
        return _compute_regulation(self._deformation, time_smoothness_weight, l1_time_planes_weight, l2_time_planes_weight,
                                   plane_tv_weight, force_weight)
