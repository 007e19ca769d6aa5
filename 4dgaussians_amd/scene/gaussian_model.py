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

import math
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


def build_rotation(r):
    """(w, x, y, z) quaternion rows, normalised here, -> rotation matrices [n, 3, 3] (what utils/general_utils.py:86-107
    returns), written as R = (1 - 2|v|^2) I + 2 v v^T + 2 w [v]x for the unit quaternion (w, v)."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, v = q[:, 0], q[:, 1:]
    cross = torch.zeros((q.shape[0], 3, 3), dtype=r.dtype, device=r.device)          # [v]x
    cross[:, 0, 1], cross[:, 0, 2], cross[:, 1, 2] = -v[:, 2], v[:, 1], -v[:, 0]
    cross = cross - cross.transpose(1, 2)
    eye = torch.eye(3, dtype=r.dtype, device=r.device)
    return (1.0 - 2.0 * (v * v).sum(dim=1))[:, None, None] * eye + 2.0 * v[:, :, None] * v[:, None, :] + 2.0 * w[:, None, None] * cross


class _LogLinearSchedule:
    """lr(step): log-linear interpolation from lr_init (step 0) to lr_final (step max_steps), scaled during the first
    lr_delay_steps by a sine ramp from lr_delay_mult to 1; 0 before step 0 and when both ends are 0 -- the schedule of
    utils/general_utils.py:36-70."""

    def __init__(self, lr_init, lr_final, lr_delay_steps, lr_delay_mult, max_steps):
        self.off = lr_init == 0.0 and lr_final == 0.0
        self.log_ends = (None, None) if self.off else (math.log(lr_init), math.log(lr_final))
        self.delay_steps, self.delay_mult, self.max_steps = lr_delay_steps, lr_delay_mult, max_steps

    def __call__(self, step):
        if self.off or step < 0:
            return 0.0
        t = min(max(step / self.max_steps, 0.0), 1.0)
        lr = math.exp(self.log_ends[0] * (1 - t) + self.log_ends[1] * t)
        if self.delay_steps > 0:
            ramp = math.sin(0.5 * math.pi * min(max(step / self.delay_steps, 0.0), 1.0))
            lr *= self.delay_mult + (1 - self.delay_mult) * ramp
        return lr


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    return _LogLinearSchedule(lr_init, lr_final, lr_delay_steps, lr_delay_mult, max_steps)


# optimizer group name -> attribute, for the six per-Gaussian parameter tensors (one tensor per group), and the
# per-Gaussian statistics that follow the rows of the model
_POINT_PARAMS = (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                 ("scaling", "_scaling"), ("rotation", "_rotation"))
_POINT_STATS = ("xyz_gradient_accum", "denom", "max_radii2D", "_deformation_accum")


def _gather_rows(sources, index, fresh, tails, zero_fresh):
    """One row gather over several per-Gaussian arrays (gs4d_gather_rows, include/gs4d.h): output row i of every array
    is input row index[i]; arrays flagged in zero_fresh get zeros in rows where fresh is set; an array with a tail takes
    its last len(tail) rows from it.  HIP tensors: ONE launch over a pointer table.  CPU tensors (the host-logic
    tests): the same definition with torch indexing."""
    n_out = int(index.numel())
    outs = [torch.empty((n_out,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in sources]
    if not sources or n_out == 0:
        return outs
    dev = sources[0].device
    n_tail = max([0] + [int(t.shape[0]) for t in tails if t is not None])
    if dev.type != "cuda":
        for o, t, tl, zf in zip(outs, sources, tails, zero_fresh):
            o.copy_(t.index_select(0, index))
            if tl is not None:
                o[n_out - tl.shape[0]:] = tl
            if zf and fresh is not None:
                o[fresh] = 0
        return outs
    import ctypes as C
    from .. import _cabi

    class GatherArray(C.Structure):
        _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("tail", C.c_void_p), ("row_words", C.c_int), ("zero_fresh", C.c_int)]

    keep, table = [], (GatherArray * len(sources))()
    for k, (o, t, tl, zf) in enumerate(zip(outs, sources, tails, zero_fresh)):
        if t.element_size() != 4:
            raise RuntimeError("gs4d_gather_rows moves 4-byte elements")
        t = t.contiguous()
        if tl is not None:
            if tl.shape[0] != n_tail:
                raise RuntimeError("every tail of one gather must have the same number of rows")
            tl = tl.to(dtype=t.dtype).contiguous()
        keep += [t, tl]
        words = 1
        for d in t.shape[1:]:
            words *= int(d)
        table[k].src, table[k].dst = t.data_ptr() if t.numel() else None, o.data_ptr()
        table[k].tail = tl.data_ptr() if tl is not None and tl.numel() else None
        table[k].row_words, table[k].zero_fresh = words, int(bool(zf))
    index = index.to(dtype=torch.int64).contiguous()
    fresh_u8 = None if fresh is None else fresh.to(torch.uint8).contiguous()
    with torch.cuda.device(dev):
        rc = _cabi.lib().gs4d_gather_rows(C.c_longlong(n_out), _cabi.ptr(index), _cabi.ptr(fresh_u8), C.c_int(n_tail),
                                          C.c_int(len(sources)), table, C.c_longlong(int(sources[0].shape[0])),
                                          _cabi.current_stream(dev))
    _cabi.check(rc, "gs4d_gather_rows")
    return outs


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
    def __init__(self, sh_degree: int, args, device="cuda", optimizer_cls=None, spatial_order: bool = False):
        self.device = torch.device(device)
        self._optimizer_cls = optimizer_cls
        # spatial_order (no reference counterpart; off: the reference's row order is kept): store the Gaussians in the
        # Morton order of their positions -- at create_from_pcd and after every densify -- see reorder_spatially()
        self.spatial_order = bool(spatial_order)
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
        """Point cloud -> initial Gaussians (scene/gaussian_model.py:159-185): positions = the points, DC colour = the
        point colours as SH band 0, every other SH coefficient 0, isotropic log-scale = log of the root mean squared
        distance to the 3 nearest neighbours (distCUDA2, clamped at 1e-7), identity rotation, opacity logit(0.1)."""
        from ..simple_knn._C import distCUDA2
        dev = self.device
        as_f32 = lambda a: torch.as_tensor(np.asarray(a)).to(dtype=torch.float32, device=dev)
        points = as_f32(pcd.points).contiguous()
        n, coeffs = points.shape[0], (self.max_sh_degree + 1) ** 2
        leaf = lambda t: nn.Parameter(t.contiguous().requires_grad_(True))
        self.spatial_lr_scale = spatial_lr_scale
        self._deformation = self._deformation.to(dev)
        self._xyz = leaf(points)
        self._features_dc = leaf(RGB2SH(as_f32(pcd.colors)).reshape(n, 1, 3))
        self._features_rest = leaf(torch.zeros((n, coeffs - 1, 3), dtype=torch.float32, device=dev))
        spacing = torch.sqrt(torch.clamp_min(distCUDA2(points), 1e-7))
        self._scaling = leaf(torch.log(spacing)[:, None].expand(n, 3))
        identity = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        identity[:, 0] = 1.0
        self._rotation = leaf(identity)
        self._opacity = leaf(inverse_sigmoid(torch.full((n, 1), 0.1, dtype=torch.float32, device=dev)))
        self.max_radii2D = torch.zeros(n, device=dev)
        self._deformation_table = torch.ones(n, dtype=torch.bool, device=dev)
        if self.spatial_order:
            self.reorder_spatially()

    # ---- optimizer
    def training_setup(self, training_args):
        """Eight AdamW groups with the reference's names, order and learning rates (scene/gaussian_model.py:188-225):
        the three spatial rates scale with spatial_lr_scale, f_rest trains at feature_lr / 20."""
        a, s = training_args, self.spatial_lr_scale
        dev, n = self._xyz.device, self._xyz.shape[0]
        self.percent_dense = a.percent_dense
        self.xyz_gradient_accum, self.denom = torch.zeros((n, 1), device=dev), torch.zeros((n, 1), device=dev)
        self._deformation_accum = torch.zeros((n, 3), device=dev)
        rates = {"xyz": a.position_lr_init * s, "deformation": a.deformation_lr_init * s, "grid": a.grid_lr_init * s,
                 "f_dc": a.feature_lr, "f_rest": a.feature_lr / 20.0, "opacity": a.opacity_lr, "scaling": a.scaling_lr,
                 "rotation": a.rotation_lr}
        members = {"deformation": list(self._deformation.get_mlp_parameters()), "grid": list(self._deformation.get_grid_parameters())}
        members.update({name: [getattr(self, attr)] for name, attr in _POINT_PARAMS})
        order = ("xyz", "deformation", "grid", "f_dc", "f_rest", "opacity", "scaling", "rotation")
        groups = [{"params": members[name], "lr": rates[name], "name": name} for name in order]
        cls = self._optimizer_cls
        if cls is None:
            from ..optim import FusedAdamW as cls
        self.optimizer = cls(groups, lr=0.0001, eps=1e-8)
        schedule = lambda lr0, lr1, mult: get_expon_lr_func(lr0 * s, lr1 * s, lr_delay_mult=mult, max_steps=a.position_lr_max_steps)
        self.xyz_scheduler_args = schedule(a.position_lr_init, a.position_lr_final, a.position_lr_delay_mult)
        self.deformation_scheduler_args = schedule(a.deformation_lr_init, a.deformation_lr_final, a.deformation_lr_delay_mult)
        self.grid_scheduler_args = schedule(a.grid_lr_init, a.grid_lr_final, a.deformation_lr_delay_mult)

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

    # Every edit of the set of Gaussians -- prune, clone, split, a spatial re-ordering -- is ONE index vector: row i of the
    # edited model is row index[i] of the current one.  _resample applies it to the six parameter tensors, their Adam
    # moments and the per-Gaussian statistics with a single gather (gs4d_gather_rows), where the reference copies masks
    # and concatenates tensor by tensor (scene/gaussian_model.py:356-432).  The results are the reference's: parameters
    # and moments of surviving rows are carried over bit for bit, new rows start with zero moments, densification
    # resets the statistics, pruning keeps them.
    def _resample(self, index, fresh=None, tails=None, reset_stats=False):
        tails = tails or {}
        n_src, n_out = self._xyz.shape[0], int(index.numel())
        groups = {} if self.optimizer is None else {g["name"]: g for g in self.optimizer.param_groups if len(g["params"]) == 1}
        sources, tail_list, zero_fresh, slots = [], [], [], []
        for name, attr in _POINT_PARAMS:
            old = getattr(self, attr)
            sources.append(old.detach()); tail_list.append(tails.get(name)); zero_fresh.append(False); slots.append(("param", name, attr))
            state = self.optimizer.state.get(old) if name in groups else None
            if state:
                for key in ("exp_avg", "exp_avg_sq"):
                    sources.append(state[key]); tail_list.append(None); zero_fresh.append(True); slots.append((key, name, attr))
        if not reset_stats:
            for attr in _POINT_STATS:
                t = getattr(self, attr)
                if t.dim() and t.shape[0] == n_src:
                    t2 = t if t.dim() > 1 else t[:, None]
                    sources.append(t2); tail_list.append(None); zero_fresh.append(True); slots.append(("stat", attr, t.dim()))
        outs = _gather_rows(sources, index, fresh, tail_list, zero_fresh)
        moments = {}
        for (kind, name, attr), out in zip(slots, outs):
            if kind == "stat":
                setattr(self, name, out if attr > 1 else out[:, 0])
            elif kind != "param":
                moments[(name, kind)] = out
        for (kind, name, attr), out in zip(slots, outs):
            if kind != "param":
                continue
            old, new = getattr(self, attr), nn.Parameter(out.requires_grad_(True))
            if name in groups:
                state = self.optimizer.state.pop(old, None)
                if state is not None:
                    if state:
                        state["exp_avg"], state["exp_avg_sq"] = moments[(name, "exp_avg")], moments[(name, "exp_avg_sq")]
                    self.optimizer.state[new] = state
                groups[name]["params"][0] = new
            setattr(self, attr, new)
        self._deformation_table = self._deformation_table[index] if self._deformation_table.shape[0] == n_src else self._deformation_table
        if reset_stats:
            dev = self._xyz.device
            self.xyz_gradient_accum, self.denom = torch.zeros((n_out, 1), device=dev), torch.zeros((n_out, 1), device=dev)
            self._deformation_accum, self.max_radii2D = torch.zeros((n_out, 3), device=dev), torch.zeros(n_out, device=dev)

    def _rows(self, n=None):
        return torch.arange(self._xyz.shape[0] if n is None else n, device=self._xyz.device)

    def prune_points(self, mask):
        """Drop the rows where mask is set (scene/gaussian_model.py:376-389); statistics and moments follow."""
        self._resample(torch.nonzero(~mask, as_tuple=False).squeeze(1))

    def densification_postfix(self, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation,
                              new_deformation_table):
        """Append explicitly given rows (scene/gaussian_model.py:412-432): zero moments, statistics reset."""
        n, k = self._xyz.shape[0], new_xyz.shape[0]
        dev = self._xyz.device
        index = torch.cat((self._rows(), torch.zeros(k, dtype=torch.long, device=dev)))
        fresh = torch.cat((torch.zeros(n, dtype=torch.bool, device=dev), torch.ones(k, dtype=torch.bool, device=dev)))
        table = torch.cat((self._deformation_table, new_deformation_table.to(self._deformation_table.dtype)), -1)
        self._resample(index, fresh, {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
                                      "scaling": new_scaling, "rotation": new_rotation}, reset_stats=True)
        self._deformation_table = table

    def _clone_selection(self, grads, grad_threshold, scene_extent):
        """Small Gaussians (largest scale <= percent_dense * extent) whose accumulated screen-space gradient norm reaches the
        threshold (scene/gaussian_model.py:460-470)."""
        return (torch.norm(grads, dim=-1) >= grad_threshold) & (self.get_scaling.max(dim=1).values <= self.percent_dense * scene_extent)

    def _split_and_apply(self, index, fresh, grads, grad_threshold, scene_extent, N):
        """index / fresh describe the model an earlier step leaves (identity, or originals + clones); large Gaussians of THAT
        model (largest scale > percent_dense * extent, gradient at the threshold; rows beyond the gradient vector count
        as 0) are replaced by N samples each, drawn from the Gaussian itself, with scales / (0.8 N)
        (scene/gaussian_model.py:434-458).  One gather applies everything."""
        dev = self._xyz.device
        scales = self.get_scaling[index]
        padded = torch.zeros(index.numel(), device=dev)
        padded[:grads.shape[0]] = grads.squeeze()
        sel = (padded >= grad_threshold) & (scales.max(dim=1).values > self.percent_dense * scene_extent)
        k = int(sel.sum())
        if k == 0:
            return index, fresh, None
        rows = index[sel]
        stds = scales[sel].repeat(N, 1)
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
        frames = build_rotation(self._rotation[rows]).repeat(N, 1, 1)
        new_xyz = torch.bmm(frames, samples.unsqueeze(-1)).squeeze(-1) + self.get_xyz[rows].repeat(N, 1)
        new_scaling = self.scaling_inverse_activation(stds / (0.8 * N))
        index = torch.cat((index[~sel], rows.repeat(N)))
        fresh = torch.cat((fresh[~sel], torch.ones(N * k, dtype=torch.bool, device=dev)))
        return index, fresh, {"xyz": new_xyz, "scaling": new_scaling}

    @torch.no_grad()
    def densify_and_split(self, grads, grad_threshold, scene_extent, N=2):
        n = self._xyz.shape[0]
        index, fresh, tails = self._split_and_apply(self._rows(), torch.zeros(n, dtype=torch.bool, device=self._xyz.device),
                                                    grads, grad_threshold, scene_extent, N)
        if tails is not None:
            self._resample(index, fresh, tails, reset_stats=True)

    @torch.no_grad()
    def densify_and_clone(self, grads, grad_threshold, scene_extent, *unused, **unused_kw):
        n, dev = self._xyz.shape[0], self._xyz.device
        picked = torch.nonzero(self._clone_selection(grads, grad_threshold, scene_extent), as_tuple=False).squeeze(1)
        fresh = torch.cat((torch.zeros(n, dtype=torch.bool, device=dev), torch.ones(picked.numel(), dtype=torch.bool, device=dev)))
        self._resample(torch.cat((self._rows(), picked)), fresh, reset_stats=True)

    @torch.no_grad()
    def prune(self, max_grad, min_opacity, extent, max_screen_size):
        """Drop transparent Gaussians and, once max_screen_size is given, those too large on screen or in the world
        (scene/gaussian_model.py:583-591)."""
        doomed = (self.get_opacity < min_opacity).squeeze()
        if max_screen_size:
            doomed = doomed | (self.max_radii2D > max_screen_size) | (self.get_scaling.max(dim=1).values > 0.1 * extent)
        self.prune_points(doomed)

    @torch.no_grad()
    def densify(self, max_grad, min_opacity, extent, max_screen_size, density_threshold=5, displacement_scale=5,
                model_path=None, iteration=None, stage=None):
        """Clone, then split (scene/gaussian_model.py:593-597), as ONE gather: the split is decided on the model the clone
        step leaves (originals followed by the clones, whose padded gradient is 0) and composed with it."""
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        n, dev = self._xyz.shape[0], self._xyz.device
        picked = torch.nonzero(self._clone_selection(grads, max_grad, extent), as_tuple=False).squeeze(1)
        index = torch.cat((self._rows(), picked))
        fresh = torch.cat((torch.zeros(n, dtype=torch.bool, device=dev), torch.ones(picked.numel(), dtype=torch.bool, device=dev)))
        index, fresh, tails = self._split_and_apply(index, fresh, grads, max_grad, extent, 2)
        if self.spatial_order:
            index, fresh, tails = self._spatially_sorted(index, fresh, tails)
        self._resample(index, fresh, tails, reset_stats=True)

    @staticmethod
    def _morton_order(xyz, bits: int = 10):
        lo, hi = xyz.min(dim=0).values, xyz.max(dim=0).values
        q = ((xyz - lo) / (hi - lo).clamp_min(1e-12) * ((1 << bits) - 1)).long().clamp_(0, (1 << bits) - 1)
        code = torch.zeros(xyz.shape[0], dtype=torch.long, device=xyz.device)
        for b in range(bits):
            for axis in range(3):
                code |= ((q[:, axis] >> b) & 1) << (3 * b + axis)
        return torch.argsort(code, stable=True)

    def _spatially_sorted(self, index, fresh, tails):
        """Compose a pending edit (index / fresh / tails of _resample) with the Morton order of the rows it produces: the
        gather then writes the edited model spatially sorted, at no extra pass.  The tail rows (new positions of a split)
        come after the permutation, so they are folded into ordinary rows first."""
        xyz = self._xyz.detach()[index]
        n_tail = 0 if not tails else int(next(iter(tails.values())).shape[0])
        if n_tail:
            xyz = xyz.clone()
            xyz[-n_tail:] = tails["xyz"]
        order = self._morton_order(xyz)
        if not n_tail:
            return index[order], fresh[order], tails
        # rows that take a tail value cannot move inside one gather: apply the edit first, sort in a second gather
        self._resample(index, fresh, tails, reset_stats=True)
        n = self._xyz.shape[0]
        return order, torch.zeros(n, dtype=torch.bool, device=order.device), None

    @torch.no_grad()
    def reorder_spatially(self, bits: int = 10):
        """Optional (no reference counterpart): store the Gaussians in the Morton order of their positions, so that
        neighbouring rows are neighbours in space -- the HexPlane gathers and the kNN regularisers then touch nearby plane
        texels / records together.  A pure permutation through the same gather as every other edit:
        parameters, moments and statistics move with their rows; renders are unchanged up to the order of tile-list ties."""
        order = self._morton_order(self._xyz.detach(), bits)
        self._resample(order)
        return order

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
