"""vo_knn2_hamming_static (the entry point that allocates nothing) against the oracle's kNN-2 and against
vo_knn2_hamming, through the C ABI: calls in a row on the GPU's shared static scratch, each smaller than the one
before; then a user stream with device pointers and no wait, recorded into a HIP graph; then the limits."""
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
    from oracle import oracle
    from visual_odometry_amd import capi
    rng = np.random.default_rng(77)
    with capi.Context(device=0, max_width=64, max_height=64, max_features=4096, max_batch=1) as ctx:
        # bit-exact (integer distances and indices); the second call is smaller than the first, so stale records of
        # the first call sit behind its results in the shared scratch
        for nq, nt in ((4096, 3977), (1000, 333), (7, 1), (300, 0)):
            q, t = _sets(rng, nq, nt)
            want = oracle.knn2(q, t)            # the CPU restatement of BFMatcher::knnMatch(k = 2)
            got = capi.knn2_static(q, t)
            for f in ("idx0", "dist0", "idx1", "dist1"):
                assert np.array_equal(got[f], want[f]), (nq, nt, f)
            same = ctx.knn2(q, t)               # and the entry point whose kernels it shares
            assert got.dtype == same.dtype and np.array_equal(got, same), (nq, nt)
            if nt >= 8:
                assert got["idx0"][5] == nt - 3 and got["dist0"][5] == 0
                assert got["idx0"][7] == 2 and got["idx1"][7] == nt - 1


def test_static_knn2_user_stream_device_pointers_no_wait_and_graph_capture():
    """What a caller under HIP graph capture does: a warm-up call (n_query = 0) before the capture, device pointers,
    wait = 0, its own stream -- first enqueued plainly, then recorded into a graph and replayed on new data."""
    import torch
    from oracle import oracle
    from visual_odometry_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(78)
    nq, nt = 1500, 1203
    q, t = _sets(rng, nq, nt)
    assert lib.vo_knn2_hamming_static(0, None, None, 0, None, 0, None, 0) == capi.VO_OK     # warm-up: address lookup
    stream = torch.cuda.Stream()
    qd, td = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    od = torch.zeros(nq * capi.KNN_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    sp = C.c_void_p(stream.cuda_stream)
    args = (0, sp, C.c_void_p(qd.data_ptr()), nq, C.c_void_p(td.data_ptr()), nt, C.c_void_p(od.data_ptr()), 0)
    assert lib.vo_knn2_hamming_static(*args) == capi.VO_OK, lib.vo_static_last_error()
    stream.synchronize()
    want = oracle.knn2(q, t)
    got = od.cpu().numpy().view(capi.KNN_DTYPE)
    for f in ("idx0", "dist0", "idx1", "dist1"):
        assert np.array_equal(got[f], want[f]), f

    # the same call recorded into a graph; replayed after the inputs changed in place
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        rc = lib.vo_knn2_hamming_static(*args)
    assert rc == capi.VO_OK, lib.vo_static_last_error()
    q2, t2 = _sets(rng, nq, nt)
    qd.copy_(torch.from_numpy(q2)); td.copy_(torch.from_numpy(t2)); od.zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    want2 = oracle.knn2(q2, t2)
    got2 = od.cpu().numpy().view(capi.KNN_DTYPE)
    for f in ("idx0", "dist0", "idx1", "dist1"):
        assert np.array_equal(got2[f], want2[f]), f


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
