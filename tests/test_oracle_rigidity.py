"""CPU checks of oracle/rigidity_ref.py (the restatement of gaussian_renderer/renderer.py:151-203 the GPU tests compare
against): its brute-force neighbour search against scipy's cKDTree, its gathered-difference norm against explicit
loops, a hand-computed two-point case, and the oddity of :203 (the MEAN of the rotation term multiplies the sum of the
weights).  The reference's own tests hold no vectors for these lines (parity unpinned against recorded outputs)."""
import math
import subprocess
import sys

import numpy as np
import torch
from scipy.spatial import cKDTree

from conftest import ROOT
from oracle import rigidity_ref as RR


def test_bruteforce_knn_matches_kdtree():
    g = torch.Generator().manual_seed(0)
    p = torch.rand(300, 3, generator=g)
    idx, dist = RR.knn_bruteforce(p, 17)
    d_ref, i_ref = cKDTree(p.numpy().astype(np.float64)).query(p.numpy().astype(np.float64), k=17)
    assert idx.shape == (1, 300, 17) and idx.dtype == torch.int64
    assert np.array_equal(idx[0].numpy(), i_ref)                      # random floats: no ties
    assert np.allclose(dist[0].numpy(), d_ref, rtol=1e-5, atol=1e-7)
    assert bool((idx[0, :, 0] == torch.arange(300)).all()) and float(dist[0, :, 0].max()) == 0.0


def test_neighbor_diff_norm_is_the_gathered_difference_norm():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(40, 4, generator=g)
    idx = torch.randint(0, 40, (1, 40, 5), generator=g)
    d = RR.neighbor_diff_norm(x, idx)
    for i in range(40):
        for s in range(5):
            j = int(idx[0, i, s])
            assert abs(float(d[0, i, s]) - math.sqrt(sum(float(x[j, c] - x[i, c]) ** 2 for c in range(4)))) < 1e-6


def test_two_point_case_by_hand():
    # two points at distance 0.01; k = 2: each point sees itself (d = 0, w = 1) and the other (w = exp(-1))
    z3, z4 = torch.zeros(2, 3), torch.zeros(2, 4)
    final = torch.tensor([[0.0, 0.0, 0.0], [0.01, 0.0, 0.0]])
    nxt = final + torch.tensor([[0.0, 0.0, 0.0], [0.0, 0.2, 0.0]])    # vel_xyz = (nxt - final) / 2 = (0, 0.1, 0) for point 1
    idx, dist = RR.knn_bruteforce(final, 2)
    rot_next = torch.tensor([[0.0, 0, 0, 0], [0.3, 0, 0, 0.4]])       # rot_v2 differs by norm 0.5 between the points
    reg = RR.temporal_regulation(final, final, final, nxt, torch.ones(2, 3), torch.ones(2, 3), torch.ones(2, 3),
                                 torch.zeros(2, 16, 3), torch.zeros(2, 16, 3), torch.zeros(2, 16, 3), z4, z4, rot_next, idx, dist)
    w = math.exp(-1.0)
    # pairs (0,0) (0,1) (1,1) (1,0): velocity differences 0, 0.1, 0, 0.1
    assert abs(float(reg["knn_xyz"]) - (2 * w * 0.1) / 2 / 2) < 1e-7
    # vel_long = nxt - means3D(= final) = (0, 0.2, 0): |0.2 - 0.1| on the two cross pairs
    assert abs(float(reg["knn_long"]) - (2 * w * 0.1) / 2 / 2) < 1e-7
    # :203: mean over the 4 pairs of (d1 + d2) = (0 + 0.5 + 0 + 0.5) / 4 = 0.25, times sum of weights (2 + 2 w), / k / n
    assert abs(float(reg["knn_rot"]) - 0.25 * (2 + 2 * w) / 2 / 2) < 1e-7
    assert abs(float(reg["momentum"]) - 0.2 / 6) < 1e-7               # |nxt + prev - 2 final| has one entry of 0.2 among 6
    assert float(reg["scales"]) == 0.0 and float(reg["shs"]) == 0.0


def test_bench_refuses_to_run_without_a_gpu():
    """The product path has no CPU fallback: bench.py must stop with a clear message, not measure something else."""
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "0"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
    assert not r.stdout.strip().startswith("{")
