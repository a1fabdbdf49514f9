"""vo_knn2_hamming_static (the entry point that allocates nothing) against vo_knn2_hamming, through the C ABI:
two calls in a row on the GPU's shared static scratch, the second with smaller inputs; then the limits."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sets(rng, nq, nt):
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    if nt >= 8 and nq >= 8:
        t[nt - 3] = q[5]          # an exact match at a high index
        t[2] = t[nt - 1] = q[7]   # a tie between a low and the highest index: the lower one comes first
    return q, t


def test_static_knn2_matches_context_entry_point():
    from visual_odometry_amd import capi
    rng = np.random.default_rng(77)
    with capi.Context(device=0, max_width=64, max_height=64, max_features=4096, max_batch=1) as ctx:
        # bit-exact (integer distances and indices); the second call is smaller than the first, so stale records of
        # the first call sit behind its results in the shared scratch
        for nq, nt in ((4096, 3977), (1000, 333), (7, 1), (300, 0)):
            q, t = _sets(rng, nq, nt)
            want = ctx.knn2(q, t)
            got = capi.knn2_static(q, t)
            assert got.dtype == want.dtype and np.array_equal(got, want), (nq, nt)
            if nt >= 8:
                assert got["idx0"][5] == nt - 3 and got["dist0"][5] == 0
                assert got["idx0"][7] == 2 and got["idx1"][7] == nt - 1


def test_static_knn2_rejects_oversize_and_bad_arguments():
    from visual_odometry_amd import capi
    lib = capi.load()
    n = 4097
    q = np.zeros((n, 32), np.uint8)
    out = np.zeros(n, capi.KNN_DTYPE)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.vo_knn2_hamming_static(0, None, p(q), n, p(q), 10, p(out), 1) == capi.VO_ERR_ARG
    assert b"static limit of 4096" in lib.vo_static_last_error()
    assert lib.vo_knn2_hamming_static(0, None, p(q), 10, p(q), n, p(out), 1) == capi.VO_ERR_ARG
    assert lib.vo_knn2_hamming_static(0, None, None, 10, p(q), 10, p(out), 1) == capi.VO_ERR_ARG
    assert lib.vo_knn2_hamming_static(0, None, p(q), 0, p(q), 10, p(out), 1) == capi.VO_OK
