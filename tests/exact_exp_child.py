"""Helper of test_gpu_exact_exp.py (not collected by pytest): renders seeded scenes and counts, over ALL pixels (no
margin mask), those that disagree with the CPU oracle -- colour error > 1e-5 or a different n_contrib.  GS4D_EXACT_EXP
is read once per process by the blend kernels' launchers, so the exact-exp run needs a process of its own."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401  (puts the repository root on sys.path)


def count():
    import test_gpu_rasterizer as T
    out = []
    for name, (P, W, H, seed, s0) in {"small": (3000, 256, 144, 5, 0.03), "big": (5000, 160, 160, 3, 0.2)}.items():
        a, cam = T.scene(P, W, H, seed, s0)
        R, color, depth, radii, gb, bb, ib = T.run_c(a, cam, 3)
        r = T.run_oracle(a, cam, 3)
        v = T.dgr()._C.state_view(P, R, W, H, gb, bb, ib)
        cerr = np.abs(color.cpu().numpy() - r["color"]).max(axis=0)
        nc = v["n_contrib"].cpu().numpy().view(np.uint32)
        bad = (cerr > 1e-5) | (nc != r["n_contrib"])
        out.append(dict(scene=name, pixels=int(bad.size), disagreeing=int(bad.sum()),
                        low_margin=int((r["margin"] <= 2e-5).sum()),
                        disagreeing_outside_mask=int((bad & (r["margin"] > 2e-5)).sum()),
                        max_colour_err=float(cerr.max())))
    a, cam, _ = T.syn.make_workload("C1")
    R, color, depth, radii, gb, bb, ib = T.run_c(a, cam, 3)
    r = T.run_oracle(a, cam, 3)
    v = T.dgr()._C.state_view(r["P"], R, r["W"], r["H"], gb, bb, ib)
    cerr = np.abs(color.cpu().numpy() - r["color"]).max(axis=0)
    nc = v["n_contrib"].cpu().numpy().view(np.uint32)
    bad = (cerr > 1e-5) | (nc != r["n_contrib"])
    out.append(dict(scene="C1", pixels=int(bad.size), disagreeing=int(bad.sum()), low_margin=int((r["margin"] <= 2e-5).sum()),
                    disagreeing_outside_mask=int((bad & (r["margin"] > 2e-5)).sum()), max_colour_err=float(cerr.max())))
    return out


if __name__ == "__main__":
    print("RESULT " + json.dumps(count()))
