"""Helper of test_gpu_hexplane_batches.py (not collected by pytest).  GS4D_HEX_BATCH_BYTES is read once per process by
csrc/deform.hip, so a run with small plane-gradient batches needs a process of its own.  Runs the mutant-shaped
deformation field forward + backward on seeded inputs and saves every gradient to the .npz named on the command line."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest  # noqa: F401  (puts the repository root on sys.path)


def run(P=20000, second_pass=False):
    from conftest import pkg
    from oracle import deformation_ref as R
    from test_gpu_deformation import make_args
    cfg = R.DeformConfig(defor_depth=0, plane_channels=32, resolution=[64, 64, 64, 75], multires=[1, 2])
    torch.manual_seed(3)
    net = pkg("scene.deformation").deform_network(make_args(cfg)).cuda()
    g = torch.Generator().manual_seed(5)
    x = [torch.rand(P, 3, generator=g) * 3.4 - 1.7, torch.randn(P, 3, generator=g) * 0.5 - 3, torch.randn(P, 4, generator=g),
         torch.randn(P, 1, generator=g), torch.randn(P, 16, 3, generator=g) * 0.3]
    x = [t.cuda().requires_grad_(True) for t in x]
    # per-point timestamps: the time planes' cells differ from point to point as well
    tm, fc = torch.rand(P, 1, generator=g).cuda(), torch.zeros(P, 1).cuda()
    ws = [torch.randn(s, generator=g).cuda() for s in ((P, 3), (P, 3), (P, 4), (P, 1), (P, 16, 3))]
    def grads():
        for t in list(net.parameters()) + x:
            t.grad = None
        out = net.forward_dynamic(*x, tm, fc, None)
        sum((o * w).sum() for o, w in zip(out, ws)).backward()
        res = {"in%d" % i: t.grad.cpu().numpy() for i, t in enumerate(x) if t.grad is not None}
        res.update({k: p.grad.cpu().numpy() for k, p in net.named_parameters() if p.grad is not None})
        return res

    res = grads()                                   # ranks the points and records the order (batch by batch)
    if second_pass:
        again = grads()                             # walks the recorded order
        cache = net.deformation_net.__dict__["_hex_order_cache"]
        res.update({"reuse/" + k: v for k, v in again.items()})
        res["cache_counts"] = np.array([cache.records, cache.reuses])
    return res


if __name__ == "__main__":
    np.savez(sys.argv[1], **run(second_pass=True))
