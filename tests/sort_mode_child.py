"""Helper of test_gpu_sort_modes.py (not collected by pytest): rasterizes two seeded scenes and prints one digest per
scene over the sorted keys, the point list and the tile ranges.  The sort strategy is a process-wide setting read from
the environment (GS4D_SORT, GS4D_SORT_SEGMENT_CAP), so the test runs this file in child processes."""
import hashlib
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digests():
    syn = importlib.import_module("4dgaussians_amd.synthetic")
    C = importlib.import_module("4dgaussians_amd.diff_gaussian_rasterization")._C
    out = []
    for P, W, H, seed, s0 in ((3000, 200, 136, 1, 0.08), (20000, 320, 240, 2, 0.03), (400, 64, 48, 3, 0.6)):
        cam = syn.look_at_camera(W, H)
        a = syn.activate(syn.make_gaussians(P, seed, s0))
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        e = torch.Tensor([])
        R, color, depth, radii, gb, bb, ib = C.rasterize_gaussians(
            torch.ones(3, device="cuda"), dev(a["means3D"]), e, dev(a["opacities"]), dev(a["scales"]), dev(a["rotations"]), 1.0, e,
            dev(cam.viewmatrix), dev(cam.projmatrix), cam.tanfovx, cam.tanfovy, H, W, dev(a["sh"]), 3, dev(cam.campos), False, False)
        v = C.state_view(P, R, W, H, gb, bb, ib)
        h = hashlib.sha256()
        for k in ("keys_sorted", "point_list", "ranges", "n_contrib"):
            h.update(v[k].cpu().numpy().tobytes())
        h.update(color.cpu().numpy().tobytes())
        lens = (v["ranges"][:, 1] - v["ranges"][:, 0])
        out.append(f"{R} {int(lens.max())} {h.hexdigest()}")
    return out


if __name__ == "__main__":
    print("\n".join(digests()))
