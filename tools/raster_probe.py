"""Times the rasterizer alone (forward, backward) on a BASELINE config; used for kernel tuning with rocprofv3."""
import argparse
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
syn = importlib.import_module("4dgaussians_amd.synthetic")
D = importlib.import_module("diff_gaussian_rasterization")

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--P", type=int, default=None)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--deg", type=int, default=3)
args = ap.parse_args()

a, cam, _ = syn.make_workload(args.config, P=args.P)
dev = lambda x, g=False: torch.from_numpy(np.ascontiguousarray(x)).cuda().requires_grad_(g)
s = D.GaussianRasterizationSettings(
    image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
    bg=torch.ones(3, device="cuda"), scale_modifier=1.0, viewmatrix=dev(cam.viewmatrix), projmatrix=dev(cam.projmatrix),
    sh_degree=args.deg, campos=dev(cam.campos), prefiltered=False, debug=False)
t = {k: dev(a[k], True) for k in ("means3D", "opacities", "sh", "scales", "rotations")}
m2d = torch.zeros_like(t["means3D"], requires_grad=True)
rast = D.GaussianRasterizer(s)
target = torch.rand(3, cam.image_height, cam.image_width, device="cuda")


def step(bwd=True):
    color, radii, depth = rast(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["sh"],
                               scales=t["scales"], rotations=t["rotations"])
    if bwd:
        loss = (color - target).abs().mean()
        loss.backward()
    return radii


for _ in range(3):
    step()
torch.cuda.synchronize()
for bwd in (False, True):
    t0 = time.perf_counter()
    for _ in range(args.iters):
        step(bwd)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print(f"{args.config} P={t['means3D'].shape[0]} {'fwd+bwd' if bwd else 'fwd'}: {dt*1e3:.3f} ms/iter ({1/dt:.1f} it/s)")
