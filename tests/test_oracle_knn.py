"""Pins oracle/knn_ref.c: brute force == Morton/box restatement (bitwise), neighbour sets agree
with scipy's cKDTree, and the reference's edge cases (P < 4, duplicates, origin in bbox)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from oracle import knn_oracle as K


@pytest.mark.parametrize("P,seed", [(2000, 0), (5000, 1), (1025, 2), (7, 3)])
def test_brute_equals_morton_restatement(P, seed):
    pts = np.random.default_rng(seed).uniform(-1.3, 1.3, (P, 3)).astype(np.float32)
    b = K.brute(pts)
    m, codes, order = K.morton(pts)
    np.testing.assert_array_equal(b.view(np.uint32), m.view(np.uint32))
    assert (np.diff(codes[order].astype(np.int64)) >= 0).all() and codes.max() < (1 << 30)


def test_matches_kdtree():
    pts = np.random.default_rng(5).uniform(-1.3, 1.3, (3000, 3)).astype(np.float32)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    want = (d[:, 1:] ** 2).mean(1)
    np.testing.assert_allclose(K.brute(pts), want, rtol=2e-5)


def test_edge_cases():
    # fewer than 4 points: FLT_MAX sentinels leak into the mean exactly as in simple_knn.cu:154,182
    one = K.brute(np.zeros((1, 3), np.float32))
    assert np.isinf(one[0])
    three = K.brute(np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32))
    assert (three > 1e37).all() and np.isfinite(three).all()
    dup = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    np.testing.assert_allclose(K.brute(dup)[0], (0 + 1 + 1) / 3.0)
    m, _, _ = K.morton(dup + 7.0)  # bbox still contains the origin (init {0,0,0}, simple_knn.cu:191)
    np.testing.assert_array_equal(m.view(np.uint32), K.brute(dup + 7.0).view(np.uint32))


def test_knn_query_oracle_matches_numpy_brute_force():
    """oracle/knn_ref.c::gs4d_ref_knn_query (restating knnquery_cuda_kernel.cu) against a float32 numpy brute force of
    the same expression: same distances, ascending, (0, 1e10) in unfilled slots.  Which of several EQUAL-distance
    points sits at the k-th place depends on the reference heap's history (a tied point is dropped when it happens to
    be the root), so indices are checked as: no repeats, every index has exactly the reported distance, and rows whose
    distances are strictly increasing match the (distance, index) order exactly."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
    pts[40:60] = pts[:20]                                   # exact duplicates: ties
    qs = rng.uniform(-1, 1, (50, 3)).astype(np.float32)
    for k in (1, 7, 100):
        idx, d2 = K.knn_query(pts, qs, k)
        dx = (qs[:, None, 0] - pts[None, :, 0]); dy = (qs[:, None, 1] - pts[None, :, 1]); dz = (qs[:, None, 2] - pts[None, :, 2])
        full = ((dx * dx + dy * dy).astype(np.float32) + dz * dz).astype(np.float32)
        order = np.lexsort((np.broadcast_to(np.arange(300), full.shape), full), axis=1)[:, :k]
        want = np.take_along_axis(full, order, 1)
        np.testing.assert_array_equal(d2.view(np.uint32), want.view(np.uint32))
        for q in range(50):
            assert len(set(idx[q])) == k
            np.testing.assert_array_equal(full[q, idx[q]].view(np.uint32), d2[q].view(np.uint32))
            if np.all(np.diff(want[q]) > 0):
                np.testing.assert_array_equal(idx[q], order[q])
    idx, d2 = K.knn_query(pts[:5], pts[:5], 8)              # fewer points than neighbours
    assert (d2[:, 5:] == np.float32(1e10)).all() and (idx[:, 5:] == 0).all() and (d2[:, 0] == 0).all()
