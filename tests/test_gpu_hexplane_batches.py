"""The plane-gradient pipeline of the HexPlane backward (csrc/deform.hip: per-point sample gradients -> counting sort
by cell -> reduce in cell order) runs over BATCHES of points so that its scratch stays bounded (2 GiB by default; a
5 M-point, 128-channel scene needed 31 GB before).  (a) A child process with GS4D_HEX_BATCH_BYTES = 3 MB (10 batches of
2 048 points) must produce the gradients of the single-batch run of this process: everything but the plane gradients
bit for bit (batching only regroups the float atomics of the planes), the planes to 1e-5 of their max -- on its first
pass, which ranks the points, and on a second one that walks the order the first recorded (one block per batch).  (b) The
force_blend-shaped field (C5's network: 128 channels, depth-5 trunk) on 1.2 M points with batches forced down to 256 MB
(5 batches): size-independent properties -- the scratch the library asks for is bounded, gradients are finite,
reproducible to float-atomic noise, and LINEAR in the upstream gradient."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_small_batches_reproduce_single_batch_gradients(tmp_path):
    import hex_batch_child
    want = hex_batch_child.run()
    out = str(tmp_path / "g.npz")
    env = dict(os.environ, GS4D_HEX_BATCH_BYTES=str(3 << 20))
    r = subprocess.run([sys.executable, os.path.join(HERE, "hex_batch_child.py"), out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    assert set(got.files) == set(want) | {"reuse/" + k for k in want} | {"cache_counts"}
    # the child's second pass walked the point order its first pass recorded, batch by batch
    assert got["cache_counts"].tolist() == [1, 1]
    worst = 0.0
    for prefix in ("", "reuse/"):
        for k, w in want.items():
            if "grids" in k:                               # plane gradients: float atomics, regrouped by the batches
                err = float(np.abs(got[prefix + k] - w).max() / max(np.abs(w).max(), 1e-30))
                worst = max(worst, err)
                assert err <= 1e-5, (prefix + k, err)
            else:
                np.testing.assert_array_equal(got[prefix + k], w, err_msg=prefix + k)
    assert any("grids" in k for k in want)
    print("plane gradients, 10 batches vs 1: max rel err", worst)


def test_shared_corner_sets_change_no_result(tmp_path):
    """Round 3: the gather and the per-point backward kernel compute the corner sets of a point once per (point, plane)
    and share them through LDS (csrc/hexplane.hip, C >= 16); GS4D_HEX_CORNERS=private keeps the per-lane form.  Same
    arithmetic by another lane: a child in the per-lane form must reproduce this process's gradients -- everything but
    the plane gradients bit for bit (the input gradients depend on every forward value), the planes to the rounding of
    their float atomics."""
    import hex_batch_child
    want = hex_batch_child.run()
    out = str(tmp_path / "g.npz")
    env = dict(os.environ, GS4D_HEX_CORNERS="private")
    r = subprocess.run([sys.executable, os.path.join(HERE, "hex_batch_child.py"), out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = np.load(out)
    for k, w in want.items():
        if "grids" in k:
            err = float(np.abs(got[k] - w).max() / max(np.abs(w).max(), 1e-30))
            assert err <= 2e-6, (k, err)
        else:
            np.testing.assert_array_equal(got[k], w, err_msg=k)


def test_force_blend_shape_1p2m_points_bounded_scratch_properties():
    # this process runs its other tests with the default batch size; the property run needs several batches at a size
    # the box handles quickly, so it goes to a child as well
    code = r'''
import os, sys, json, importlib
import numpy as np, torch
sys.path.insert(0, %r)
import conftest
from conftest import pkg
Wl, D, cabi = pkg("workload"), pkg("scene.deformation"), pkg("_cabi")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = D.deform_network(Wl.deform_args("C5"))
net.deformation_net.set_aabb([1.3] * 3, [-1.3] * 3)
net = net.to(dev)
P = 1_200_000
g = torch.Generator().manual_seed(1)
x = [torch.rand(P, 3, generator=g) * 2.6 - 1.3, torch.randn(P, 3, generator=g) - 3, torch.randn(P, 4, generator=g), torch.randn(P, 1, generator=g),
     torch.randn(P, 16, 3, generator=g) * 0.3]
x = [t.to(dev).requires_grad_(True) for t in x]
tm, fc = torch.rand(P, 1, generator=g).to(dev), torch.rand(P, 1, generator=g).to(dev)
gs = [torch.randn(P, 3, generator=g).to(dev), torch.randn(P, 3, generator=g).to(dev), torch.randn(P, 4, generator=g).to(dev), torch.randn(P, 16, 3, generator=g).to(dev)]
def grads(scale):
    for p in list(net.parameters()) + x:
        p.grad = None
    o = net.forward_dynamic(*x, tm, fc, None)
    torch.autograd.backward([o[0], o[1], o[2], o[4]], [scale * t for t in gs])
    r = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    r["xyz"] = x[0].grad.clone()
    return r
a, b, c = grads(1.0), grads(1.0), grads(2.0)
rep = {"finite": all(bool(torch.isfinite(v).all()) for v in a.values()), "n": len(a)}
rel = lambda u, v: float((u - v).abs().max() / v.abs().max().clamp_min(1e-30))
rep["repro"] = max(rel(a[k], b[k]) for k in a)
rep["linear"] = max(rel(c[k], 2 * a[k]) for k in a)
rep["nonzero_planes"] = min(float(v.abs().max()) for k, v in a.items() if "grids" in k) > 0
rep["peak_alloc"] = int(torch.cuda.max_memory_allocated())
print("REPORT " + json.dumps(rep))
''' % HERE
    import json

    def child(batch_bytes):
        env = dict(os.environ, GS4D_HEX_BATCH_BYTES=str(batch_bytes))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("REPORT ")][-1][7:])

    rep = child(256 << 20)
    whole = child(1 << 40)                                  # one batch: the sample gradients of all 1.2 M points at once
    print("force_blend shape, 1.2 M points, 256 MB batches:", rep)
    print("the same in one batch:", whole)
    assert rep["finite"] and rep["nonzero_planes"] and rep["n"] > 40
    assert rep["repro"] <= 1e-5 and rep["linear"] <= 1e-5, rep
    # unbatched, the sample gradients alone are 1.2 M x 2 levels x 6 planes x 128 channels x 4 B = 7.4 GB
    assert whole["peak_alloc"] - rep["peak_alloc"] > 6.5e9, (rep["peak_alloc"], whole["peak_alloc"])
