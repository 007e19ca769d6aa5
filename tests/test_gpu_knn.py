"""distCUDA2 on the GPU against the CPU oracle (bit-exact: both evaluate the exact 3-NN)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def oracle_knn(pts):
    from oracle import knn_oracle
    return knn_oracle.brute(pts)


@pytest.mark.parametrize("P,seed", [(2000, 0), (1, 1), (3, 2), (4, 3), (1025, 4), (30000, 5)])
def test_distcuda2_matches_oracle(P, seed):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    if P > 100:
        pts[10] = pts[11]  # duplicate point -> zero distance
    from simple_knn._C import distCUDA2
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    want = oracle_knn(pts)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_distcuda2_clustered_points_far_from_origin():
    rng = np.random.default_rng(7)
    pts = (rng.standard_normal((5000, 3)) * 0.01 + np.array([5.0, -3.0, 2.0])).astype(np.float32)
    from simple_knn._C import distCUDA2
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), oracle_knn(pts).view(np.uint32))
