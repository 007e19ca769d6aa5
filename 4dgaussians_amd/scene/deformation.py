"""deform_network / Deformation, mirroring scene/deformation.py:18-298 of the reference.

Same class names, constructor (`deform_network(args)` reading the same args fields), parameter
names and shapes (a reference `deformation.pth` loads with load_state_dict), weight
initialisation (xavier on every Linear weight, biases left at nn.Linear's default except the
four "zero-init" heads whose biases stay 0, scene/deformation.py:92-112,253,285-289) and the
same `forward_dynamic(points, scales, rotations, opacity, shs, time, force, prev_frames)` ->
`(means3D, scales, rotations, opacity, shs)` contract.

The computation is ONE autograd node backed by gs4d_deform_forward / gs4d_deform_backward in
libgs4d.so (include/gs4d.h).  There is no PyTorch fallback: tensors must live on a HIP device.
Supported = what runs in the reference: blend_time_force=True, no_do=True, prev_frames=None,
net_width=64, grid_pe=0 (SURVEY.md "five facts" #3; scene/deformation.py:101 vs :190).
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.init as init

from .. import _cabi
from .hexplane import HexPlaneField

MAX_LEVELS, MAX_TRUNK = 4, 8
BWD_DATA, BWD_WEIGHTS = 1, 2          # GS4D_DEFORM_BWD_* of include/gs4d.h
FP = C.c_void_p


class DeformDesc(C.Structure):
    _fields_ = [("C", C.c_int), ("L", C.c_int), ("trunk_layers", C.c_int), ("pos_pe", C.c_int), ("sr_pe", C.c_int),
                ("time_pe", C.c_int), ("force_pe", C.c_int), ("no_dx", C.c_int), ("no_ds", C.c_int), ("no_dr", C.c_int),
                ("no_dshs", C.c_int), ("apply_rotation", C.c_int), ("use_time", C.c_int), ("use_force", C.c_int),
                ("planes", FP * (MAX_LEVELS * 6)), ("plane_w", C.c_int * (MAX_LEVELS * 6)),
                ("plane_h", C.c_int * (MAX_LEVELS * 6)), ("aabb", FP), ("ft_w", FP), ("ft_b", FP),
                ("trunk_w", FP * MAX_TRUNK), ("trunk_b", FP * MAX_TRUNK), ("head_w1", FP * 4), ("head_b1", FP * 4),
                ("head_w2", FP * 4), ("head_b2", FP * 4), ("no_grad", C.c_int)]


class DeformGrads(C.Structure):
    _fields_ = [("planes", FP * (MAX_LEVELS * 6)), ("ft_w", FP), ("ft_b", FP), ("trunk_w", FP * MAX_TRUNK),
                ("trunk_b", FP * MAX_TRUNK), ("head_w1", FP * 4), ("head_b1", FP * 4), ("head_w2", FP * 4),
                ("head_b2", FP * 4), ("hex_order", C.c_void_p), ("hex_order_valid", C.c_int), ("reserved_", C.c_int)]


class HexOrderCache:
    """The plane-gradient pipeline's point order, kept from one backward to the next (gs4d_deform_grads::hex_order,
    include/gs4d.h): the order only decides which gradient adds are combined before they reach memory, never the result,
    and the points move little per iteration, so the ranking atomics, the bin scan and the scatter run only every
    `refresh` backward passes (GS4D_HEX_ORDER_REFRESH, default 16; 0 switches the cache off) and whenever the number of
    points, the device or the level count changes.  One entry per point count (the regularised render queries 3 P points)."""

    def __init__(self):
        import os
        self.refresh = int(os.environ.get("GS4D_HEX_ORDER_REFRESH", "16"))
        self.entries = {}
        self.records = self.reuses = 0          # backward passes that ranked the points / walked a cached order

    def clear(self):
        self.entries.clear()

    def bind(self, g, P, L, dev):
        """Points `g` at the order buffer for (P, L, dev) and says whether it holds a usable order."""
        if self.refresh <= 0 or P == 0:
            g.hex_order, g.hex_order_valid = None, 0
            return
        key = (P, L, dev.index)
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= 4:
                self.entries.clear()
            # starts as the identity order of every (level, sort plane): should the recording backward fail half way
            # (an exception between its phases), what a later pass walks is still a permutation of the points -- the
            # order decides only how well adds are combined, a non-permutation would drop or double gradients
            e = self.entries[key] = [torch.arange(P, dtype=torch.int32, device=dev).repeat(L * 3), self.refresh]
        valid = e[1] < self.refresh
        if valid:
            e[1] += 1
            self.reuses += 1
        else:
            # stays stale until the recording pass has returned without an error (recorded()): a scene that runs in
            # several batches keeps batch-local indices here, which the identity above is not
            e[1] = self.refresh
            self.records += 1
        g.hex_order, g.hex_order_valid = e[0].data_ptr(), int(valid)
        return None if valid else key

    def recorded(self, key):
        """The backward that bind() asked to record the order (key = bind's return value) has been issued."""
        if key is not None and key in self.entries:
            self.entries[key][1] = 1


def _lib():
    L = _cabi.lib()
    L.gs4d_deform_workspace_bytes.restype = C.c_size_t
    L.gs4d_deform_backward_scratch_bytes.restype = C.c_size_t
    return L


def init_zero_(model):
    for m in model:
        if isinstance(m, nn.Linear):
            init.zeros_(m.weight)
            if m.bias is not None:
                init.zeros_(m.bias)


class _Plan:
    """What a parameter set fixes from step to step, built once: the filled gs4d_deform_desc, and the layout of the
    ONE allocation that receives every parameter gradient (offsets, the gs4d_deform_grads words to patch, shape /
    stride / offset of each gradient view).  A step then costs a pointer comparison instead of ~50 tensor
    conversions and ~100 ctypes field assignments in each direction -- host time that sits on the critical path
    after the rasterizer's host synchronisation.  Valid while every parameter keeps its storage (`matches`)."""

    def __init__(self, cfg, params):
        import numpy as np
        self.params = list(params)
        self.ptrs = [p.data_ptr() for p in self.params]
        n_planes = cfg["L"] * 6
        planes, aabb, ft_w, ft_b, trunk, heads = _DeformFunction._split(cfg, [p.detach() for p in self.params])
        # the fast path needs every parameter to be usable in place (float32, dense, planes channels-last)
        self.in_place = all(a.data_ptr() == b.data_ptr() and a.dtype == torch.float32
                            for a, b in zip(planes + [aabb, ft_w, ft_b] + trunk + heads, self.params))
        self.desc = _DeformFunction._desc(cfg, planes, aabb, ft_w, ft_b, trunk, heads)
        self.n_planes, self.n_trunk = n_planes, len(trunk)
        # gradient allocation: planes as [H][W][C] memory under the [1,C,H,W] shape, everything else row-major
        views, offs, total = [], [], 0
        for t in planes + [ft_w, ft_b] + trunk + heads:
            n = t.numel()
            if t.dim() == 4:
                _, c, h, w = t.shape
                views.append((tuple(t.shape), (h * w * c, 1, w * c, c), total))
            else:
                views.append((tuple(t.shape), tuple(t.contiguous().stride()), total))
            offs.append(total)
            total += (n + 63) // 64 * 64
        self.grad_views, self.grad_total = views, max(total, 1)
        # the words of gs4d_deform_grads (all 8-byte pointers) in the order of `views`
        self.grads = DeformGrads()
        words = np.frombuffer(memoryview(self.grads).cast("B"), dtype=np.uint64)
        field = {name: getattr(DeformGrads, name).offset // 8 for name, _ in DeformGrads._fields_}
        idx = [field["planes"] + i for i in range(n_planes)] + [field["ft_w"], field["ft_b"]]
        for i in range(len(trunk) // 2):
            idx += [field["trunk_w"] + i, field["trunk_b"] + i]
        for h in range(4):
            idx += [field["head_w1"] + h, field["head_b1"] + h, field["head_w2"] + h, field["head_b2"] + h]
        self._words, self._idx = words, np.asarray(idx, dtype=np.int64)
        self._byte_offs = np.asarray(offs, dtype=np.uint64) * np.uint64(4)

    def matches(self, params) -> bool:
        return len(params) == len(self.ptrs) and all(p.data_ptr() == q for p, q in zip(params, self.ptrs))

    def new_grads(self, device):
        """-> (gs4d_deform_grads pointing into a fresh zero-filled allocation, the gradient views in parameter order)."""
        import numpy as np
        flat = torch.zeros(self.grad_total, dtype=torch.float32, device=device)
        self._words[self._idx] = np.uint64(flat.data_ptr()) + self._byte_offs
        return self.grads, [torch.as_strided(flat, shp, st, off) for shp, st, off in self.grad_views]


class _DeformFunction(torch.autograd.Function):
    """points, scales, rotations, shs, time, force, cfg, n_planes, *params -> 4 outputs.
    params = planes..., aabb, ft_w, ft_b, trunk (w,b)..., heads (w1,b1,w2,b2) x 4."""

    @staticmethod
    def _desc(cfg, planes, aabb, ft_w, ft_b, trunk, heads):
        d = DeformDesc()
        d.C, d.L, d.trunk_layers = cfg["C"], cfg["L"], len(trunk) // 2
        d.pos_pe, d.sr_pe, d.time_pe, d.force_pe = cfg["pos_pe"], cfg["sr_pe"], cfg["time_pe"], cfg["force_pe"]
        d.no_dx, d.no_ds, d.no_dr, d.no_dshs = cfg["no_dx"], cfg["no_ds"], cfg["no_dr"], cfg["no_dshs"]
        d.apply_rotation, d.use_time, d.use_force = cfg["apply_rotation"], cfg["use_time"], cfg["use_force"]
        for i, p in enumerate(planes):
            d.planes[i] = p.data_ptr()
            d.plane_w[i], d.plane_h[i] = p.shape[3], p.shape[2]
        d.aabb, d.ft_w, d.ft_b = aabb.data_ptr(), ft_w.data_ptr(), ft_b.data_ptr()
        for i in range(len(trunk) // 2):
            d.trunk_w[i], d.trunk_b[i] = trunk[2 * i].data_ptr(), trunk[2 * i + 1].data_ptr()
        for h in range(4):
            d.head_w1[h], d.head_b1[h] = heads[4 * h].data_ptr(), heads[4 * h + 1].data_ptr()
            d.head_w2[h], d.head_b2[h] = heads[4 * h + 2].data_ptr(), heads[4 * h + 3].data_ptr()
        return d

    @staticmethod
    def _split(cfg, params):
        n_planes = cfg["L"] * 6
        planes = [p.contiguous(memory_format=torch.channels_last) for p in params[:n_planes]]
        aabb, ft_w, ft_b = (t.contiguous() for t in params[n_planes:n_planes + 3])
        n_trunk = cfg["trunk_layers"]
        trunk = [t.contiguous() for t in params[n_planes + 3:n_planes + 3 + 2 * n_trunk]]
        heads = [t.contiguous() for t in params[n_planes + 3 + 2 * n_trunk:]]
        assert len(heads) == 16
        return planes, aabb, ft_w, ft_b, trunk, heads

    @staticmethod
    def forward(ctx, points, scales, rotations, shs, time, force, cfg, *params):
        L = _lib()
        dev = points.device
        if dev.type != "cuda":
            raise RuntimeError("the deformation field runs on a HIP device only; there is no CPU path")
        P = points.shape[0]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        points_, scales_, rot_, shs_ = f32(points), f32(scales), f32(rotations), f32(shs)
        time_, force_ = f32(time).reshape(-1), f32(force).reshape(-1)
        if rot_.data_ptr() % 16:
            rot_ = rot_.clone()
        if shs_.data_ptr() % 16:
            shs_ = shs_.clone()
        plan = cfg.get("plan")
        if plan is not None and plan.in_place and len(params) == len(plan.ptrs):   # storage checked by forward_dynamic
            d = plan.desc
        else:
            plan = None
            planes, aabb, ft_w, ft_b, trunk, heads = _DeformFunction._split(cfg, [p.detach() for p in params])
            d = _DeformFunction._desc(cfg, planes, aabb, ft_w, ft_b, trunk, heads)
        d.no_grad = 0 if any(ctx.needs_input_grad) else 1      # under torch.no_grad() nothing will read the activations
        o_xyz, o_sc, o_rot, o_shs = (torch.empty_like(t) for t in (points_, scales_, rot_, shs_))
        ws = torch.empty(max(int(L.gs4d_deform_workspace_bytes(C.byref(d), C.c_int(P))), 256), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.gs4d_deform_forward(C.byref(d), C.c_int(P), _cabi.ptr(points_), _cabi.ptr(scales_), _cabi.ptr(rot_),
                                       _cabi.ptr(shs_), _cabi.ptr(time_), _cabi.ptr(force_), _cabi.ptr(o_xyz),
                                       _cabi.ptr(o_sc), _cabi.ptr(o_rot), _cabi.ptr(o_shs), _cabi.ptr(ws),
                                       _cabi.current_stream(dev))
        _cabi.check(rc, "gs4d_deform_forward")
        ctx.cfg = cfg
        ctx.plan = plan
        if plan is not None:
            # the parameters are not re-read through saved_tensors (50 unpacks per step); their versions are kept
            # instead, so that an in-place update between forward and backward is still refused
            ctx.versions = [p._version for p in params]
            ctx.save_for_backward(points_, scales_, rot_, time_, force_, ws)
        else:
            ctx.save_for_backward(points_, scales_, rot_, time_, force_, ws, *params)
        return o_xyz, o_sc, o_rot, o_shs

    @staticmethod
    def backward(ctx, g_xyz, g_sc, g_rot, g_shs):
        L = _lib()
        cfg = ctx.cfg
        points_, scales_, rot_, time_, force_, ws, *params = ctx.saved_tensors
        dev = points_.device
        P = points_.shape[0]
        plan = ctx.plan
        if plan is not None:
            if [p._version for p in plan.params] != ctx.versions:
                raise RuntimeError("a deformation parameter was modified in place between forward and backward")
            d = plan.desc
            g, views = plan.new_grads(dev)
            n_pl, n_tr = plan.n_planes, plan.n_trunk
            gp, g_ftw, g_ftb = views[:n_pl], views[n_pl], views[n_pl + 1]
            gt, gh = views[n_pl + 2:n_pl + 2 + n_tr], views[n_pl + 2 + n_tr:]
        else:
            planes, aabb, ft_w, ft_b, trunk, heads = _DeformFunction._split(cfg, [p.detach() for p in params])
            d = _DeformFunction._desc(cfg, planes, aabb, ft_w, ft_b, trunk, heads)
            # every parameter gradient lives in one allocation cleared by one fill; plane gradients are
            # channels-last like the planes themselves ([H][W][C] bytes under a [1,C,H,W] shape)
            nhwc = [(1, p.shape[2], p.shape[3], p.shape[1]) for p in planes]
            flat = _cabi.zeros_group(dev, *nhwc, ft_w.shape, ft_b.shape, *[t.shape for t in trunk], *[t.shape for t in heads])
            gp = [t.permute(0, 3, 1, 2) for t in flat[:len(planes)]]
            g_ftw, g_ftb = flat[len(planes)], flat[len(planes) + 1]
            gt = flat[len(planes) + 2:len(planes) + 2 + len(trunk)]
            gh = flat[len(planes) + 2 + len(trunk):]
            g = DeformGrads()
            for i, t in enumerate(gp):
                g.planes[i] = t.data_ptr()
            g.ft_w, g.ft_b = g_ftw.data_ptr(), g_ftb.data_ptr()
            for i in range(len(trunk) // 2):
                g.trunk_w[i], g.trunk_b[i] = gt[2 * i].data_ptr(), gt[2 * i + 1].data_ptr()
            for h in range(4):
                g.head_w1[h], g.head_b1[h] = gh[4 * h].data_ptr(), gh[4 * h + 1].data_ptr()
                g.head_w2[h], g.head_b2[h] = gh[4 * h + 2].data_ptr(), gh[4 * h + 3].data_ptr()
        gz = lambda t, like: (torch.zeros_like(like) if t is None else (t if t.dtype == torch.float32 and t.is_contiguous() else t.to(torch.float32).contiguous()))
        g_xyz, g_sc, g_rot = gz(g_xyz, points_), gz(g_sc, scales_), gz(g_rot, rot_)
        g_shs = gz(g_shs, torch.empty(P, 16, 3, device=dev))
        if g_rot.data_ptr() % 16:
            g_rot = g_rot.clone()
        if g_shs.data_ptr() % 16:
            g_shs = g_shs.clone()
        d_xyz, d_sc, d_rot = (torch.empty_like(t) for t in (points_, scales_, rot_))
        d_shs = g_shs            # shs' = shs + delta: the input gradient IS the upstream gradient (the C ABI allows the alias)
        scratch = torch.empty(max(int(L.gs4d_deform_backward_scratch_bytes(C.byref(d), C.c_int(P))), 256),
                              dtype=torch.uint8, device=dev)
        hook = cfg.get("data_grad_hook")
        order_cache = cfg.get("hex_order_cache")
        order_key = None
        if order_cache is not None:
            order_key = order_cache.bind(g, P, int(d.L), dev)
        else:
            g.hex_order, g.hex_order_valid = None, 0

        def run(phases):
            with torch.cuda.device(dev):
                rc = L.gs4d_deform_backward_phases(
                    C.byref(d), C.byref(g), C.c_int(P), _cabi.ptr(points_), _cabi.ptr(scales_), _cabi.ptr(rot_),
                    _cabi.ptr(time_), _cabi.ptr(force_), _cabi.ptr(g_xyz), _cabi.ptr(g_sc), _cabi.ptr(g_rot),
                    _cabi.ptr(g_shs), _cabi.ptr(d_xyz), _cabi.ptr(d_sc), _cabi.ptr(d_rot), _cabi.ptr(d_shs), _cabi.ptr(ws),
                    _cabi.ptr(scratch), _cabi.current_stream(dev), C.c_int(phases))
            _cabi.check(rc, "gs4d_deform_backward_phases")

        if hook is None:
            run(BWD_DATA | BWD_WEIGHTS)
        else:
            # data-parallel overlap: everything but the Linear weight gradients is final after the data phase and can
            # start its all-reduce while the weight-gradient products run (parallel.GradBucket.early_reduce_many)
            run(BWD_DATA)
            hook(d_xyz, d_sc, d_rot, gp, g_ftw, g_ftb)
            run(BWD_WEIGHTS)
        if order_cache is not None:
            order_cache.recorded(order_key)
        param_grads = list(gp) + [None, g_ftw, g_ftb] + list(gt) + list(gh)
        return (d_xyz, d_sc, d_rot, d_shs, None, None, None, *param_grads)


class Deformation(nn.Module):
    def __init__(self, D=8, W=256, grid_pe=0, args=None):
        super().__init__()
        self.D, self.W, self.args = D, W, args
        self.grid = HexPlaneField(args.bounds, args.kplanes_config, args.multires, W)
        self.blend_time_force = args.blend_time_force
        self.ratio = 0
        self.posbase_pe, self.scale_rotation_pe, self.opacity_pe = args.posebase_pe, args.scale_rotation_pe, args.opacity_pe
        self.grid_pe, self.force_pe, self.time_pe = args.grid_pe, args.force_pe, args.time_pe
        self.extra_point_dim = args.posebase_pe * 2 * 3
        self.extra_scale_dim = args.scale_rotation_pe * 2 * 3
        self.extra_rotation_dim = args.scale_rotation_pe * 2 * 4
        self.extra_opacity_dim = args.opacity_pe * 2 * 3
        self.prev_frames = args.prev_frames
        self.prev_frames_out_dim = 16
        # optional callable(d_points, d_scales, d_rotations, plane_grads, ft_w_grad, ft_b_grad), called from the
        # backward as soon as those gradients are final (before the weight gradients are computed); see parallel.py
        self.data_grad_hook = None
        self.create_net()

    @property
    def get_aabb(self):
        return self.grid.get_aabb

    def set_aabb(self, xyz_max, xyz_min):
        self.grid.set_aabb(xyz_max, xyz_min)      # replaces grid.aabb by a new Parameter: the cached plan is stale
        self._step_cache = None
        self.__dict__.pop("_hex_order_cache", None)   # other bounds, other cells: the cached point order is void

    def create_net(self):
        W = self.W
        self.prev_frame_encoder = nn.Sequential(  # unused unless prev_frames is given; kept for state_dict parity
            nn.Conv2d(3, 16, kernel_size=5, stride=3), nn.ReLU(), nn.Conv2d(16, 64, kernel_size=5, stride=3), nn.ReLU(),
            nn.Conv2d(64, 128, kernel_size=5, stride=4), nn.ReLU(), nn.Conv2d(128, 128, kernel_size=3, stride=2), nn.ReLU(),
            nn.Flatten(), nn.Linear(128 * 5 * 5, 128), nn.ReLU(), nn.Linear(128, self.prev_frames_out_dim))
        total_prev_frames_dim = self.prev_frames * self.prev_frames_out_dim
        grid_out_dim = self.grid.feat_dim + self.grid_pe * self.grid.feat_dim * 2
        extra = 1 + 2 * self.force_pe + 1 + 2 * self.time_pe
        self.feature_out1 = nn.Linear(grid_out_dim + total_prev_frames_dim, grid_out_dim)
        layers = [nn.Linear(grid_out_dim, W)]
        for _ in range(self.D - 1):
            layers += [nn.ReLU(), nn.Linear(W, W)]
        self.feature_out2 = nn.Sequential(*layers)
        head = lambda in_extra, out: nn.Sequential(nn.ReLU(), nn.Linear(W + in_extra + extra, W), nn.ReLU(), nn.Linear(W, out))
        self.pos_deform = head(self.extra_point_dim, 3)
        self.scales_deform = head(self.extra_scale_dim, 3)
        init_zero_(self.scales_deform)
        self.rotations_deform = head(self.extra_rotation_dim, 4)
        init_zero_(self.rotations_deform)
        self.opacity_deform = head(self.extra_opacity_dim, 1)
        init_zero_(self.opacity_deform)
        self.shs_deform = head(0, 16 * 3)
        init_zero_(self.shs_deform)
        self.force_embedder = nn.Linear(2, 1)
        if self.blend_time_force:
            self.force_time_embed = nn.Linear(2, 1)

    @property
    def get_empty_ratio(self):
        return self.ratio

    def _config(self):
        a = self.args
        if not self.blend_time_force:
            raise NotImplementedError("only blend_time_force=True is supported (the reference's other path raises "
                                      "IndexError at scene/hexplane.py:116)")
        if not a.no_do:
            raise NotImplementedError("no_do=False feeds opacity_deform the wrong input width in the reference "
                                      "(scene/deformation.py:101 vs :190) and cannot run there either")
        if self.W != 64 or self.grid_pe != 0:
            raise NotImplementedError("net_width must be 64 and grid_pe 0")
        kc = a.kplanes_config
        return dict(C=int(kc["output_coordinate_dim"]), L=len(a.multires), trunk_layers=max(self.D, 1),
                    pos_pe=int(self.posbase_pe), sr_pe=int(self.scale_rotation_pe), time_pe=int(self.time_pe),
                    force_pe=int(self.force_pe), no_dx=int(bool(a.no_dx)), no_ds=int(bool(a.no_ds)),
                    no_dr=int(bool(a.no_dr)), no_dshs=int(bool(a.no_dshs)), apply_rotation=int(bool(a.apply_rotation)),
                    use_time=int(bool(a.use_time)), use_force=int(bool(a.use_force)),
                    data_grad_hook=self.data_grad_hook)

    def _flags(self):
        a = self.args
        return (bool(a.no_dx), bool(a.no_ds), bool(a.no_dr), bool(a.no_dshs), bool(a.apply_rotation), bool(a.use_time),
                bool(a.use_force), bool(a.no_do), self.blend_time_force)

    def _params(self):
        planes = [p for level in self.grid.grids for p in level]
        trunk = []
        for m in self.feature_out2:
            if isinstance(m, nn.Linear):
                trunk += [m.weight, m.bias]
        heads = []
        for h in (self.pos_deform, self.scales_deform, self.rotations_deform, self.shs_deform):
            heads += [h[1].weight, h[1].bias, h[3].weight, h[3].bias]
        return planes + [self.grid.aabb, self.force_time_embed.weight, self.force_time_embed.bias] + trunk + heads

    def __getstate__(self):                        # the step cache holds ctypes structures: never pickled / deep-copied
        state = dict(super().__getstate__()) if hasattr(super(), "__getstate__") else dict(self.__dict__)
        state.pop("_step_cache", None)
        state.pop("_hex_order_cache", None)
        return state

    def _apply(self, fn, *a, **k):                 # .to() / .cuda() / .float(): parameters may get new storage
        self._step_cache = None
        self.__dict__.pop("_hex_order_cache", None)
        return super()._apply(fn, *a, **k)

    def cached_plan(self):
        """-> (params, cfg): the parameter list in the order the C ABI consumes it and the per-parameter-set plan
        (cfg["plan"]: filled gs4d_deform_desc, gradient layout), rebuilt only when a parameter changed storage."""
        cache = self.__dict__.get("_step_cache")
        # The cached plan is valid while (a) every cached parameter keeps its storage, (b) the aabb Parameter is still
        # the cached object (HexPlaneField.set_aabb installs a new one) and (c) the switches read from `args` are
        # unchanged (enable_extra_deform / disable_extra_deform or a direct assignment).
        if (cache is None or cache[2] != self._flags() or self.grid.aabb is not cache[3]
                or not cache[1]["plan"].matches(cache[0])):
            params = self._params()
            cfg = self._config()
            cfg["plan"] = _Plan(cfg, params)
            cache = self.__dict__["_step_cache"] = (params, cfg, self._flags(), self.grid.aabb)
        params, cfg = cache[0], cache[1]
        cfg["data_grad_hook"] = self.data_grad_hook
        oc = self.__dict__.get("_hex_order_cache")
        if oc is None:
            oc = self.__dict__["_hex_order_cache"] = HexOrderCache()
        cfg["hex_order_cache"] = oc
        return params, cfg

    def forward_dynamic(self, points, scales, rotations, opacity, shs, time, force, prev_frames=None):
        if prev_frames is not None:
            raise NotImplementedError("prev_frames conditioning is outside the hot path (SURVEY.md section 8)")
        params, cfg = self.cached_plan()
        out = _DeformFunction.apply(points, scales, rotations, shs, time, force, cfg, *params)
        # opacity passes through (no_do): the reference's opacity[:, :1] is the tensor itself when it is (P, 1)
        return out[0], out[1], out[2], opacity if opacity.shape[1] == 1 else opacity[:, :1], out[3]

    def get_mlp_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" not in n]

    def get_grid_parameters(self):
        return [p for n, p in self.named_parameters() if "grid" in n]

    def enable_extra_deform(self):
        self.args.no_ds = self.args.no_dr = self.args.no_dshs = False
        self._step_cache = None

    def disable_extra_deform(self):
        self.args.no_ds = self.args.no_dr = self.args.no_dshs = True
        self._step_cache = None


def initialize_weights(m):
    if isinstance(m, nn.Linear):
        init.xavier_uniform_(m.weight, gain=1)  # the reference re-inits the weight (twice) and never the bias


class deform_network(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.deformation_net = Deformation(W=args.net_width, D=args.defor_depth, grid_pe=args.grid_pe, args=args)
        for name, n in (("pos_poc", args.posebase_pe), ("rotation_scaling_poc", args.scale_rotation_pe),
                        ("opacity_poc", args.opacity_pe), ("force_poc", args.force_pe), ("time_poc", args.time_pe)):
            self.register_buffer(name, torch.FloatTensor([(2 ** i) for i in range(n)]))
        self.apply(initialize_weights)
        self.recur = False
        self.prev_frames = 0

    @property
    def get_aabb(self):
        return self.deformation_net.get_aabb

    @property
    def get_empty_ratio(self):
        return self.deformation_net.get_empty_ratio

    def forward_dynamic(self, points_in, scales_in, rotations_in, opacity_in, shs_in, times_sel_in, force_in, prev_frames=None):
        # the positional encodings (poc_fre, scene/deformation.py:291-298) are evaluated inside the HIP path
        return self.deformation_net.forward_dynamic(points_in, scales_in, rotations_in, opacity_in, shs_in,
                                                    times_sel_in, force_in, prev_frames)

    def forward(self, *args):
        return self.forward_dynamic(*args)

    def get_mlp_parameters(self):
        return self.deformation_net.get_mlp_parameters()

    def get_grid_parameters(self):
        return self.deformation_net.get_grid_parameters()
