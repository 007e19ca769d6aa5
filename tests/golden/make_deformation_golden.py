"""Generates tests/golden/deformation_*.npz by IMPORTING THE REFERENCE'S OWN PYTHON
(scene/hexplane.py + scene/deformation.py under /root/reference) on CPU.  Run in the build
container only (the reference does not exist on the GPU box); the .npz files it writes are
plain data: the seeded state_dict, the inputs, the five outputs of forward_dynamic and the
gradients of a fixed scalar loss w.r.t. inputs and parameters.

Recipe: `import scene...` would execute scene/__init__.py and pull in
open3d / plyfile / simple_knn, so stub packages `scene` and `utils` are registered in sys.modules
with __path__ pointing at the reference directories and the four needed files are loaded by path.

    python tests/golden/make_deformation_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    for pkg in ("scene", "utils"):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        sys.modules[pkg] = m
    mods = {}
    for name, rel in (("utils.graphics_utils", "utils/graphics_utils.py"), ("scene.grid", "scene/grid.py"),
                      ("scene.hexplane", "scene/hexplane.py"), ("scene.deformation", "scene/deformation.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["scene.deformation"]


class Args:
    pass


def make_args(cfg: dict):
    a = Args()
    a.net_width = cfg["net_width"]; a.defor_depth = cfg["defor_depth"]; a.posebase_pe = cfg["posebase_pe"]
    a.scale_rotation_pe = cfg["scale_rotation_pe"]; a.opacity_pe = cfg["opacity_pe"]; a.grid_pe = 0
    a.force_pe = cfg["force_pe"]; a.time_pe = cfg["time_pe"]; a.bounds = cfg["bounds"]
    a.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": cfg["plane_channels"],
                        "resolution": list(cfg["resolution"])}
    a.multires = list(cfg["multires"])
    a.no_dx = cfg["no_dx"]; a.no_ds = cfg["no_ds"]; a.no_dr = cfg["no_dr"]; a.no_do = cfg["no_do"]; a.no_dshs = cfg["no_dshs"]
    a.empty_voxel = False; a.static_mlp = False; a.apply_rotation = cfg["apply_rotation"]
    a.use_time = cfg["use_time"]; a.use_force = cfg["use_force"]; a.blend_time_force = True
    a.prev_frames = 0; a.no_grid = False
    return a


BASE = dict(net_width=64, defor_depth=1, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2, force_pe=2, time_pe=2,
            bounds=1.6, plane_channels=32, resolution=[64, 64, 64, 75], multires=[1, 2], no_dx=False, no_ds=False,
            no_dr=False, no_do=True, no_dshs=False, apply_rotation=True, use_time=True, use_force=False)

CASES = {
    # mutant-shaped network (arguments/dnerf/others/mutant.py + dnerf_default.py) on reduced planes so the fixture stays small
    "mutant_small": dict(BASE, defor_depth=0, plane_channels=32, resolution=[8, 6, 5, 7], multires=[1, 2]),
    # force_blend-shaped (arguments/dnerf/force_blend.py): deeper trunk, force used, 4 PE frequencies (no_do must stay True: the opacity head is fed the wrong width in the reference, scene/deformation.py:101 vs :190)
    "force_small": dict(BASE, defor_depth=5, plane_channels=16, resolution=[6, 5, 4, 9], multires=[1, 2], force_pe=4,
                        time_pe=4, use_force=True),
    # single level, non-normalised quaternion update path
    "plain_small": dict(BASE, defor_depth=1, plane_channels=8, resolution=[4, 4, 4, 3], multires=[1], apply_rotation=False),
}


def main():
    deformation = load_reference()
    for name, cfg in CASES.items():
        torch.manual_seed(1234)
        net = deformation.deform_network(make_args(cfg)).float()
        # nn.Linear biases of the "zero-init" heads stay 0 in the reference; give every parameter a seeded
        # non-trivial value so that each one influences the outputs and gets a gradient worth comparing
        g = torch.Generator().manual_seed(99)
        with torch.no_grad():
            for pname, p in net.named_parameters():
                if "grids" in pname or "aabb" in pname:
                    continue
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        net.deformation_net.set_aabb([1.3, 1.2, 1.4], [-1.3, -1.1, -1.2])
        P = 257
        pts = (torch.rand(P, 3, generator=g) * 3.0 - 1.5).requires_grad_(True)   # some points outside the aabb -> border clamp
        scales = (torch.randn(P, 3, generator=g) * 0.5 - 3.0).requires_grad_(True)
        rots = torch.randn(P, 4, generator=g).requires_grad_(True)
        opac = torch.randn(P, 1, generator=g).requires_grad_(True)
        shs = (torch.randn(P, 16, 3, generator=g) * 0.3).requires_grad_(True)
        time = torch.full((P, 1), 0.37)
        force = torch.full((P, 1), 0.8)
        out = net.forward_dynamic(pts, scales, rots, opac, shs, time, force, None)
        ws = [torch.randn(o.shape, generator=g) for o in out]
        loss = sum((o * w).sum() for o, w in zip(out, ws))
        loss.backward()
        blob = {"cfg_json": np.array(__import__("json").dumps(cfg))}
        # parameters that forward_dynamic(prev_frames=None) never touches (grad is None) are listed by name
        # only: prev_frame_encoder.*, feature_out1, force_embedder, grid.feature_aggr.*, opacity_deform.* (no_do)
        dead = [k for k, p in net.named_parameters() if p.grad is None and p.requires_grad]
        blob["dead_params"] = np.array(dead)
        for k, v in net.state_dict().items():
            if k not in dead:
                blob["sd/" + k] = v.detach().numpy()
        for k, p in net.named_parameters():
            if p.grad is not None:
                blob["grad/" + k] = p.grad.numpy()
        for k, v in dict(points=pts, scales=scales, rotations=rots, opacity=opac, shs=shs).items():
            blob["in/" + k] = v.detach().numpy()
            blob["gin/" + k] = v.grad.numpy() if v.grad is not None else np.zeros(0, np.float32)
        blob["in/time"] = time.numpy(); blob["in/force"] = force.numpy()
        for k, o, w in zip(("means3D", "scales", "rotations", "opacity", "shs"), out, ws):
            blob["out/" + k] = o.detach().numpy()
            blob["w/" + k] = w.numpy()
        path = os.path.join(HERE, f"deformation_{name}.npz")
        np.savez_compressed(path, **blob)
        print(name, "params", sum(p.numel() for p in net.parameters()), "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
