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


def digest_scene(a, cam, P, W, H):
    """(num_rendered, longest list, sha256 over sorted keys, point list, ranges, n_contrib and the image) of one forward."""
    C = importlib.import_module("4dgaussians_amd.diff_gaussian_rasterization")._C
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
    return f"{R} {int(lens.max())} {h.hexdigest()}"


def digests():
    syn = importlib.import_module("4dgaussians_amd.synthetic")
    out = []
    # the last three scenes: runs of EQUAL depths (the same position four times over; 700 Gaussians in one point, whose
    # tiles' lists are one single depth): the order inside such a run is the Gaussian index, whatever order the lists
    # were filled in (direct placement fills them with atomics).  The 1104-pixel-wide scene has 69 tile columns: rows in no
    # spatial order there exceed the 63-tile rectangle a workgroup counts in LDS (csrc/tile_place.h) and take the fallback.
    for P, W, H, seed, s0, copies in ((3000, 200, 136, 1, 0.08, 1), (20000, 320, 240, 2, 0.03, 1), (400, 64, 48, 3, 0.6, 1),
                                      (6000, 200, 136, 4, 0.05, 4), (700, 64, 48, 5, 0.3, 700), (5000, 1104, 160, 6, 0.04, 1)):
        cam = syn.look_at_camera(W, H)
        a = syn.activate(syn.make_gaussians(P, seed, s0))
        if copies > 1:
            a["means3D"] = np.ascontiguousarray(np.tile(a["means3D"][:P // copies], (copies, 1)))
        out.append(digest_scene(a, cam, P, W, H))
    return out


if __name__ == "__main__":
    print("\n".join(digests()))
