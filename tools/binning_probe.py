"""usage (GPU box): python tools/binning_probe.py [config]: renders the bench scene a few times and prints how the forwards were
binned (gs4d_binning_stats) and the statistic the choice rests on: the tile rectangles around every 256 consecutive Gaussians,
summed, against num_rendered (csrc/binning.hip scan_block_sums_kernel)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wl = importlib.import_module("4dgaussians_amd.workload")
rd = importlib.import_module("4dgaussians_amd.renderer")
cabi = importlib.import_module("4dgaussians_amd._cabi")

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
dev = torch.device("cuda:0")
for order in (True, False):
    sc = wl.build_scene(name, dev, calibrate=False, spatial_order=order)
    for i in range(4):
        with torch.no_grad():
            out = rd.render(sc["cam"], sc["pc"], sc["deform"], sc["bg"])
        torch.cuda.synchronize()
        radii = out[3]
        print("spatial_order", order, "forward", i, "(direct, passes, adds, rendered)", cabi.binning_stats_full(), "visible", int((radii > 0).sum()))
