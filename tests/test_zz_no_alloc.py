"""The allocation-free entry point gs4d_knn_mean_dist2_static (include/gs4d.h) against the existing
gs4d_knn_mean_dist2, through the C ABI, bit for bit (both evaluate the exact 3-NN; integer / index work and the
float sums are the same instructions), on two calls in a row -- the second with a smaller input, so that stale
contents of the shared static scratch would show -- and its size limit."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def _both(pts):
    cabi = pkg("_cabi")
    L = cabi.lib()
    L.gs4d_debug_alloc_calls.restype = C.c_ulonglong
    P = pts.shape[0]
    x = torch.from_numpy(pts).cuda()
    want = torch.full((P,), -1.0, device="cuda")
    alloc = cabi.TensorAllocator(x.device)
    calls0 = L.gs4d_debug_alloc_calls()
    rc = L.gs4d_knn_mean_dist2(alloc.callback, alloc.ctx, C.c_int(P), cabi.ptr(x), cabi.ptr(want), cabi.current_stream(x.device))
    alloc.close()
    cabi.check(rc, "gs4d_knn_mean_dist2")
    assert L.gs4d_debug_alloc_calls() == calls0 + 1             # the counter sees the allocating entry point's request
    got = torch.full((P,), -2.0, device="cuda")
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()["allocation.all.allocated"]
    calls1 = L.gs4d_debug_alloc_calls()
    free_before, _ = torch.cuda.mem_get_info()                  # hipMemGetInfo: a hipMalloc inside the call would show here
    rc = L.gs4d_knn_mean_dist2_static(C.c_int(P), cabi.ptr(x), cabi.ptr(got), cabi.current_stream(x.device))
    cabi.check(rc, "gs4d_knn_mean_dist2_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == before     # nothing went through torch's allocator
    assert L.gs4d_debug_alloc_calls() == calls1                  # the library asked nobody for memory (callbacks, runtime)
    assert torch.cuda.mem_get_info()[0] == free_before          # and the device's free memory did not move
    return got.cpu().numpy(), want.cpu().numpy()


def test_static_scratch_variant_matches_the_allocating_entry_point_twice():
    rng = np.random.default_rng(42)
    big = rng.uniform(-1.3, 1.3, (50_000, 3)).astype(np.float32)
    big[10] = big[11]                                   # duplicate point: distance 0
    got, want = _both(big)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    small = (rng.standard_normal((1_537, 3)) * 0.01 + np.array([5.0, -3.0, 2.0])).astype(np.float32)
    got, want = _both(small)                            # second call, smaller input, same static scratch
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    from oracle import knn_oracle
    np.testing.assert_array_equal(got.view(np.uint32), knn_oracle.brute(small).view(np.uint32))
    tiny = rng.uniform(-1, 1, (3, 3)).astype(np.float32)   # P < 4: FLT_MAX leaks into the mean, as in the reference
    got, want = _both(tiny)
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))


def test_static_scratch_variant_rejects_sizes_beyond_its_maximum():
    cabi = pkg("_cabi")
    L = cabi.lib()
    nmax = int(L.gs4d_knn_static_max_points())
    assert nmax == 262144
    x = torch.zeros(nmax + 1, 3, device="cuda")
    out = torch.zeros(nmax + 1, device="cuda")
    rc = L.gs4d_knn_mean_dist2_static(C.c_int(nmax + 1), cabi.ptr(x), cabi.ptr(out), cabi.current_stream(x.device))
    assert rc == -1 and b"maximum" in L.gs4d_last_error()
    pts = torch.rand(nmax, 3, device="cuda") * 2 - 1     # exactly the maximum runs
    out = torch.zeros(nmax, device="cuda")
    rc = L.gs4d_knn_mean_dist2_static(C.c_int(nmax), cabi.ptr(pts), cabi.ptr(out), cabi.current_stream(x.device))
    cabi.check(rc, "gs4d_knn_mean_dist2_static at the maximum")
    from simple_knn._C import distCUDA2
    assert torch.equal(out, distCUDA2(pts))
