"""Helper of test_gpu_mailbox_paths.py (not collected by pytest): renders two seeded scenes through the C-level
rasterizer, twice each (the second forward of a thread sizes its binning chunk speculatively), and prints a digest of
every output plus the thread's num_rendered wait statistics.  GS4D_NO_MAILBOX / GS4D_MAILBOX_SPIN_US are read once per
process, so every mode needs a process of its own."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401  (puts the repository root on sys.path)


def run():
    import torch
    import test_gpu_rasterizer as T
    cabi = conftest.pkg("_cabi")
    out = {"scenes": []}
    for (P, W, H, seed, s0) in ((3000, 256, 144, 5, 0.03), (800, 96, 80, 9, 0.25), (3000, 256, 144, 5, 0.03)):
        a, cam = T.scene(P, W, H, seed, s0)
        for rep in range(2):
            R, color, depth, radii, gb, bb, ib = T.run_c(a, cam, 3)
            v = T.dgr()._C.state_view(P, R, W, H, gb, bb, ib)
            torch.cuda.synchronize()
            h = hashlib.sha256()
            for t in (color, depth, radii, v["point_list"], v["ranges"], v["n_contrib"], v["final_T"]):
                h.update(t.contiguous().cpu().numpy().tobytes())
            out["scenes"].append({"R": int(R), "digest": h.hexdigest()})
    # a prefilter violation must be reported on whichever path delivers the counters
    a, cam = T.scene(500, 64, 48, 2, 0.1)
    a["means3D"][3] = cam.campos + 0.05 * (a["means3D"][3] - cam.campos) / max(1e-6, float(abs(a["means3D"][3] - cam.campos).max()))
    e = torch.Tensor([])
    try:
        T.dgr()._C.rasterize_gaussians(torch.ones(3, device="cuda"), T.dev(a["means3D"]), e, T.dev(a["opacities"]), T.dev(a["scales"]),
                                       T.dev(a["rotations"]), 1.0, e, T.dev(cam.viewmatrix), T.dev(cam.projmatrix), cam.tanfovx,
                                       cam.tanfovy, cam.image_height, cam.image_width, T.dev(a["sh"]), 3, T.dev(cam.campos), True, False)
        out["prefilter"] = "no error"
    except RuntimeError as ex:
        out["prefilter"] = "filtered although prefiltered" if "prefiltered is set" in str(ex) else str(ex)
    out["fast"], out["slow"], out["regrow"] = cabi.forward_stats()
    return out


if __name__ == "__main__":
    print("RESULT " + json.dumps(run()))
