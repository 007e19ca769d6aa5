"""GPU parity of the k-nearest-neighbour query against the CPU oracle
(oracle/knn_ref.c::gs4d_ref_knn_query, a restatement of pointops2/src/knnquery/knnquery_cuda_kernel.cu).
Bar: squared distances bit-exact (integer-deciding float arithmetic, -ffp-contract=off on both sides); index rows
identical wherever a row's distances are strictly increasing; everywhere: no repeated index in a row and every
index has exactly the reported distance (which of several EQUAL-distance points the reference keeps at the k-th
place depends on its heap's history; this build keeps the lower indices)."""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import knn_oracle as K

pytestmark = pytest.mark.gpu


def check(pts, qs, k, same):
    P = pkg("pointops")
    x = torch.from_numpy(pts).cuda()
    y = x if same else torch.from_numpy(qs).cuda()
    off = torch.tensor([pts.shape[0]], dtype=torch.int32)
    qoff = torch.tensor([(pts if same else qs).shape[0]], dtype=torch.int32)
    idx, dist = P.knnquery(k, x, None if same else y, off, qoff)
    widx, wd2 = K.knn_query(pts, pts if same else qs, k)
    assert idx.dtype == torch.int32 and idx.shape == widx.shape
    got_d = dist.cpu().numpy()
    np.testing.assert_array_equal(got_d.view(np.uint32), np.sqrt(wd2).view(np.uint32))
    gi = idx.cpu().numpy()
    strict = np.all(np.diff(wd2, axis=1) > 0, axis=1)
    np.testing.assert_array_equal(gi[strict], widx[strict])
    q = pts if same else qs
    filled = wd2 < np.float32(1e10)
    assert (gi >= 0).all() and (gi < pts.shape[0]).all()
    d = q[:, None, :] - pts[gi]                                   # recompute every reported pair's distance
    re = ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(np.float32) + d[..., 2] * d[..., 2]).astype(np.float32)
    np.testing.assert_array_equal(re[filled].view(np.uint32), wd2[filled].view(np.uint32))
    srt = np.sort(np.where(filled, gi, -1 - np.arange(gi.shape[1])[None, :]), 1)
    assert (np.diff(srt, axis=1) != 0).all()                      # no index twice in a row
    return gi, got_d


@pytest.mark.parametrize("n,k,seed", [(2000, 100, 0), (5000, 100, 1), (129, 100, 2), (64, 3, 3), (70000, 100, 4), (1, 1, 5)])
def test_knn_query_self(n, k, seed):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.3, 1.3, (n, 3)).astype(np.float32)
    if n > 4000 and n < 10000:                                  # clustered + exact duplicates
        pts[: n // 2] = (pts[: n // 2] * 0.01 + 0.5).astype(np.float32)
        pts[100:300] = pts[400:600]
    if n > 50000:                                               # oracle is O(n * m): sample the queries
        qs = pts[rng.choice(n, 3000, replace=False)]
        check(pts, qs, k, same=False)
    else:
        gi, d = check(pts, None, k, same=True)
        assert (d[:, 0] == 0).all()


def test_knn_query_other_queries_fewer_points_and_batches():
    rng = np.random.default_rng(9)
    pts = rng.normal(0, 1, (700, 3)).astype(np.float32)
    qs = rng.normal(0, 2, (333, 3)).astype(np.float32)
    check(pts, qs, 100, same=False)
    check(pts[:40], qs[:9], 100, same=False)                    # n < k: (0, 1e10) padding
    # utils/general_utils.py::knn with two batches: indices are per-batch, distances sqrt'ed
    P = pkg("pointops")
    x = torch.from_numpy(pts[:600].reshape(2, 300, 3)).cuda()
    idx, dist = P.knn(x, x, 16)
    assert idx.dtype == torch.int64 and idx.shape == (2, 300, 16) and dist.shape == (2, 300, 16)
    for b in range(2):
        widx, wd2 = K.knn_query(pts[300 * b:300 * (b + 1)], pts[300 * b:300 * (b + 1)], 16)
        np.testing.assert_array_equal(dist[b].cpu().numpy().view(np.uint32), np.sqrt(wd2).view(np.uint32))
        np.testing.assert_array_equal(idx[b].cpu().numpy(), widx)      # no ties in this data


def test_knn_query_argument_errors():
    P = pkg("pointops")
    x = torch.zeros(10, 3, device="cuda")
    off = torch.tensor([10], dtype=torch.int32)
    with pytest.raises(RuntimeError, match="nsample"):
        P.knnquery(129, x, None, off, off)
    with pytest.raises(RuntimeError, match="HIP device"):
        P.knnquery(3, x.cpu(), None, off, off)


def test_point_order_is_a_permutation_and_changes_no_list():
    """gs4d_knn_query_order: the Morton order of the points comes back as a permutation; lists and distances are those of
    the plain query, bit for bit."""
    po = pkg("pointops")
    g = torch.Generator().manual_seed(5)
    pts = torch.randn(5000, 3, generator=g).cuda()
    idx0, d0 = po.knn_self_i32(pts, 20)
    idx1, d1, order = po.knn_self_i32(pts, 20, return_order=True)
    assert torch.equal(idx0, idx1) and torch.equal(d0, d1)
    assert order.dtype == torch.int32 and order.shape == (5000,)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(5000, device="cuda"))
    # spatially coherent: consecutive points of the order are far closer than consecutive points of the input
    step = lambda p: float((p[1:] - p[:-1]).norm(dim=1).mean())
    assert step(pts[order.long()]) < 0.5 * step(pts)
