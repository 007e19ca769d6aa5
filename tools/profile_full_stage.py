"""Runs only the fork's complete fine-stage iteration (render with regularisers + grid regularisers + AdamW) a few
times at C2 -- a target for `rocprofv3 --kernel-trace --stats -- python3 tools/profile_full_stage.py`."""
import importlib, os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
syn = importlib.import_module("4dgaussians_amd.synthetic"); R = importlib.import_module("4dgaussians_amd.renderer")
Dm = importlib.import_module("4dgaussians_amd.scene.deformation"); F = importlib.import_module("4dgaussians_amd.fused")
reg = importlib.import_module("4dgaussians_amd.scene.regulation"); om = importlib.import_module("4dgaussians_amd.optim")
dev = torch.device("cuda")
_, cam0, raw = syn.make_workload("C2")
torch.manual_seed(syn.seed_for("C2"))
raw = dict(raw, log_scales=raw["log_scales"] + 0.4982)
if os.environ.get("FFS_MORTON"):      # experiment: Gaussians stored in Morton order of their positions
    import numpy as np
    x = np.asarray(raw["xyz"], dtype=np.float64)
    q = ((x - x.min(0)) / (x.max(0) - x.min(0) + 1e-12) * 1023).astype(np.uint64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; return (v | (v << 2)) & 0x09249249
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    o = np.argsort(code, kind="stable")
    raw = {k_: (v_[o] if hasattr(v_, "shape") and len(v_) == len(o) else v_) for k_, v_ in raw.items()}
pc = R.GaussianParams(raw, dev)
Wl = importlib.import_module("4dgaussians_amd.workload")
deform = Dm.deform_network(Wl.deform_args("C2")); deform.deformation_net.set_aabb([1.3] * 3, [-1.3] * 3); deform = deform.to(dev)
cam = R.Camera(syn.look_at_camera(800, 800, azimuth=0.7), dev, time=0.37)
bg = torch.ones(3, device=dev); target = torch.rand(3, 800, 800, device=dev)
params = list(pc.parameters()) + list(deform.parameters())
opt = om.FusedAdamW([{"params": [p for p in params if p.requires_grad], "lr": 1e-4}], lr=1e-4, eps=1e-8)
import time
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for it in range(N + 3):
    if it == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    image, _, _, _, _, rd, _ = R.render(cam, pc, deform, bg, regularise=True)
    loss = F.l1_loss(image, target) + reg.compute_regulation(deform, 0.01, 0.0001, 0.0, 0.0001, detail=False)["total_reg"] \
        + 0.000001 * rd["momentum"] + rd["knn_xyz"] + 0.1 * rd["knn_long"] + rd["knn_rot"] + 0.001 * rd["shs"] + 0.001 * rd["scales"]
    loss.backward()
    opt.step()
    for p in params:
        p.grad = None
torch.cuda.synchronize()
print(f"full fine-stage iteration: {(time.perf_counter() - t0) / N * 1e3:.3f} ms wall over {N} iterations (3 warm-up iterations before)")
