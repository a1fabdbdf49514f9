"""ctypes front end of the CPU oracle (oracle/libvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The shipped package never imports this module.
PARITY UNPINNED -- see vo_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvo_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
KNN_DTYPE = np.dtype([("idx0", "<i4"), ("dist0", "<i4"), ("idx1", "<i4"), ("dist1", "<i4")])
MATCH_DTYPE = np.dtype([("query", "<i4"), ("train", "<i4"), ("distance", "<f4"),
                        ("px", "<f4"), ("py", "<f4"), ("cx", "<f4"), ("cy", "<f4")])


_SORT_LIB_PATH = os.path.join(_HERE, "libvo_oracle_stdsort.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvo_oracle.so"],
                              stdout=subprocess.DEVNULL)
    cpp = os.path.join(_HERE, "vo_oracle_stdsort.cpp")
    if force or not os.path.exists(_SORT_LIB_PATH) or os.path.getmtime(cpp) > os.path.getmtime(_SORT_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libvo_oracle_stdsort.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ora_rng_next.restype = C.c_uint32
        _lib.ora_find_fundamental_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                                     C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]
        _lib.ora_backproject.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]
        _lib.ora_solve_pnp_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                              C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.ora_pnp_iterative.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        _lib.ora_match_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.ora_match_filter_x.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.ora_fast_atan2.restype = C.c_float
        _lib.ora_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.ora_harris_response.restype = C.c_float
        _lib.ora_harris_response.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.ora_ic_angle.restype = C.c_float
        _lib.ora_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib.ora_orb_level_sizes.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p]
        _lib.ora_orb_quotas.argtypes = [C.c_int, C.c_float, C.c_int, C.c_void_p]
        _lib.ora_orb_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ora_e_hypotheses.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.ora_find_essential_ransac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.ora_recover_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]
        _lib.ora_undistort.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        _lib.ora_stereo_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                          C.c_float, C.c_float, C.c_float, C.c_double, C.c_double, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
        _lib.ora_stereo_depth_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                C.c_int]
        _lib.ora_stereo_depth_plane.restype = None
        _lib.ora_ratio_test.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gray(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img)
    h, w, cn = img.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_gray(_p(img), w, h, w * cn, cn, _p(out), w)
    return out


def fast_score_map(g: np.ndarray, thr: int) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_fast_score_map(_p(g), w, h, w, thr, _p(out), w)
    return out


def fast(g: np.ndarray, thr: int) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    cap = (w * h) // 4 + 16
    out = np.zeros(cap, KP_DTYPE)
    n = lib().ora_fast(_p(g), w, h, w, thr, _p(out), cap)
    return out[:n].copy()


def select_top(kps: np.ndarray, width: int, max_features: int) -> np.ndarray:
    """Canonical top-N of raster-ordered FAST corners: (response desc, raster asc)."""
    k = np.ascontiguousarray(kps, KP_DTYPE).copy()
    n = lib().ora_select_top(_p(k), len(k), width, max_features)
    return k[:n].copy()


_sort_lib = None


def select_top_stdsort(kps: np.ndarray, max_features: int) -> np.ndarray:
    """The reference's literal std::sort + resize (src/feature_detector.cpp:62-68), libstdc++."""
    global _sort_lib
    if _sort_lib is None:
        if not os.path.exists(_SORT_LIB_PATH):
            build()
        _sort_lib = C.CDLL(_SORT_LIB_PATH)
    k = np.ascontiguousarray(kps, KP_DTYPE).copy()
    n = _sort_lib.ora_select_top_stdsort(_p(k), len(k), max_features)
    return k[:n].copy()


def blur7(g: np.ndarray) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_blur7(_p(g), w, h, w, _p(out), w)
    return out


def equalize_hist(g: np.ndarray) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_equalize_hist(_p(g), w, h, w, _p(out), w)
    return out


def gaussian_taps_q8(n: int, sigma: float) -> np.ndarray:
    taps = np.zeros(31, np.int32)
    if not lib().ora_gaussian_taps_q8(n, C.c_double(sigma), _p(taps)):
        raise ValueError("kernel size must be odd and in 1..31")
    return taps[:n].copy()


def gaussian_blur(g: np.ndarray, n: int = 3, sigma: float = 1.0) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    if not lib().ora_gaussian_blur(_p(g), w, h, w, n, C.c_double(sigma), _p(out), w):
        raise ValueError("kernel size must be odd and in 1..31")
    return out


def roi_mask(g: np.ndarray) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    out = np.empty((h, w), np.uint8)
    lib().ora_roi_mask(_p(g), w, h, w, _p(out), w)
    return out


def undistort(g, fx, fy, cx, cy, dist):
    """cv::undistort(gray, K, dist) with dist = k1 k2 p1 p2 [k3 [k4 k5 k6]]."""
    g = np.ascontiguousarray(g, np.uint8)
    h, w = g.shape
    d = np.ascontiguousarray(dist, np.float64).reshape(-1)
    out = np.zeros_like(g)
    if not lib().ora_undistort(_p(g), w, h, w, fx, fy, cx, cy, _p(d), len(d), _p(out), w):
        raise ValueError("4, 5 or 8 distortion coefficients expected")
    return out


def process_image(img: np.ndarray, enhance=False, blur_size=3, sigma=1.0, hist_eq=False) -> dict:
    """ImageProcessor::processImage (reference src/image_processor.cpp:6-28)."""
    out = {"gray": gray(img) if img.ndim == 3 and img.shape[2] == 3 else np.ascontiguousarray(img).copy()}
    if enhance:
        out["enhanced"] = equalize_hist(out["gray"]) if hist_eq else out["gray"].copy()
        out["denoised"] = gaussian_blur(out["enhanced"], blur_size, sigma) if blur_size > 1 \
            else out["enhanced"].copy()
        out["masked"] = roi_mask(out["denoised"])
    return out


def brief(blur: np.ndarray, kps: np.ndarray) -> np.ndarray:
    blur = np.ascontiguousarray(blur)
    kps = np.ascontiguousarray(kps)
    h, w = blur.shape
    out = np.zeros((len(kps), 32), np.uint8)
    lib().ora_brief(_p(blur), w, h, w, _p(kps), len(kps), _p(out))
    return out


def detect(g: np.ndarray, max_features: int = 1000, fast_threshold: int = 20):
    g = np.ascontiguousarray(g)
    h, w = g.shape
    kps = np.zeros(max(max_features, 1), KP_DTYPE)
    desc = np.zeros((max(max_features, 1), 32), np.uint8)
    n = lib().ora_detect(_p(g), w, h, w, max_features, fast_threshold, _p(kps), _p(desc))
    return kps[:n].copy(), desc[:n].copy()


def orb_levels(w, h, scale_factor=1.2, nlevels=8):
    lw, lh = np.zeros(nlevels, np.int32), np.zeros(nlevels, np.int32)
    ls = np.zeros(nlevels, np.float32)
    lib().ora_orb_level_sizes(w, h, scale_factor, nlevels, _p(lw), _p(lh), _p(ls))
    return lw, lh, ls


def orb_quotas(nfeatures, scale_factor=1.2, nlevels=8):
    q = np.zeros(nlevels, np.int32)
    lib().ora_orb_quotas(nfeatures, scale_factor, nlevels, _p(q))
    return q


def resize_linear_exact(g: np.ndarray, dw: int, dh: int) -> np.ndarray:
    g = np.ascontiguousarray(g)
    h, w = g.shape
    out = np.empty((dh, dw), np.uint8)
    lib().ora_resize_linear_exact(_p(g), w, h, w, _p(out), dw, dh, dw)
    return out


def fast_atan2(y, x):
    return float(lib().ora_fast_atan2(y, x))


def harris_response(g, x, y):
    g = np.ascontiguousarray(g)
    return float(lib().ora_harris_response(_p(g), g.shape[1], x, y))


def ic_angle(g, x, y):
    g = np.ascontiguousarray(g)
    return float(lib().ora_ic_angle(_p(g), g.shape[1], x, y))


def orb_umax():
    u = np.zeros(32, np.int32)
    n = lib().ora_orb_umax(_p(u))
    return u[:n].copy()


def orb_detect(g: np.ndarray, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20, pyramid=False):
    """cv::ORB::detectAndCompute; returns (kps, desc[, levels, blurred levels])."""
    g = np.ascontiguousarray(g)
    h, w = g.shape
    cap = 2 * nfeatures + 4096
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    lw, lh, _ = orb_levels(w, h, scale_factor, nlevels)
    tot = int((lw.astype(np.int64) * lh).sum())
    pyr = np.zeros(tot, np.uint8) if pyramid else None
    blr = np.zeros(tot, np.uint8) if pyramid else None
    n = lib().ora_orb_detect(_p(g), w, h, w, nfeatures, scale_factor, nlevels, fast_threshold, _p(kps),
                             _p(desc), cap, _p(pyr) if pyramid else None, _p(blr) if pyramid else None)
    assert 0 <= n <= cap
    if not pyramid:
        return kps[:n].copy(), desc[:n].copy()
    lv, bl, off = [], [], 0
    for a, b in zip(lw, lh):
        lv.append(pyr[off:off + a * b].reshape(b, a))
        bl.append(blr[off:off + a * b].reshape(b, a))
        off += a * b
    return kps[:n].copy(), desc[:n].copy(), lv, bl


def knn2(q: np.ndarray, t: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q)
    t = np.ascontiguousarray(t)
    out = np.zeros(len(q), KNN_DTYPE)
    lib().ora_knn2_hamming(_p(q), len(q), _p(t), len(t), _p(out))
    return out


def ratio_test(knn, nt, ratio, qk, tk):
    out = np.zeros(max(len(knn), 1), MATCH_DTYPE)
    n = lib().ora_ratio_test(_p(knn), len(knn), nt, ratio, _p(qk), _p(tk), _p(out))
    return out[:n].copy()


def rng_stream(n: int, seed: int = 0xFFFFFFFFFFFFFFFF) -> np.ndarray:
    class R(C.Structure):
        _fields_ = [("state", C.c_uint64)]
    r = R(seed)
    return np.array([lib().ora_rng_next(C.byref(r)) for _ in range(n)], np.uint32)


def f7(m1: np.ndarray, m2: np.ndarray) -> np.ndarray:
    m1 = np.ascontiguousarray(m1, np.float32)
    m2 = np.ascontiguousarray(m2, np.float32)
    out = np.zeros((3, 9), np.float64)
    n = lib().ora_f7(_p(m1), _p(m2), _p(out))
    return out[:max(n, 0)].copy()


def f_error(p1, p2, F):
    p1 = np.ascontiguousarray(p1, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    F = np.ascontiguousarray(F, np.float64)
    err = np.zeros(len(p1), np.float32)
    lib().ora_f_error(_p(p1), _p(p2), len(p1), _p(F), _p(err))
    return err


def find_fundamental_ransac(p1, p2, thr=3.0, conf=0.99, max_iters=1000):
    p1 = np.ascontiguousarray(p1, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    n = len(p1)
    mask = np.zeros(max(n, 1), np.uint8)
    F = np.zeros(9, np.float64)
    iters = C.c_int(0)
    ok = lib().ora_find_fundamental_ransac(_p(p1), _p(p2), n, thr, conf, max_iters, _p(mask),
                                           _p(F), C.addressof(iters))
    return bool(ok), F.reshape(3, 3), mask[:n].copy(), iters.value


def e5(q1, q2) -> np.ndarray:
    """Five-point solutions for 5 normalised correspondences: (k, 3, 3)."""
    q1 = np.ascontiguousarray(q1, np.float64).reshape(5, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(5, 2)
    out = np.zeros((10, 9), np.float64)
    n = lib().ora_e5(_p(q1), _p(q2), _p(out))
    return out[:n].reshape(-1, 3, 3).copy()


def e_error(q1, q2, E) -> np.ndarray:
    q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 2)
    E = np.ascontiguousarray(E, np.float64).reshape(9)
    err = np.zeros(len(q1), np.float32)
    lib().ora_e_error(_p(q1), _p(q2), len(q1), _p(E), _p(err))
    return err


def e_hypotheses(q1, q2, idx, thr):
    """Batched five-point hypotheses from an (H, 5) index table: E (H, 10, 9), n_models (H), counts (H, 10)."""
    q1 = np.ascontiguousarray(q1, np.float64).reshape(-1, 2)
    q2 = np.ascontiguousarray(q2, np.float64).reshape(-1, 2)
    idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 5)
    H = len(idx)
    E = np.zeros((H, 10, 9), np.float64)
    nm = np.zeros(H, np.int32)
    cnt = np.zeros((H, 10), np.int32)
    lib().ora_e_hypotheses(_p(q1), _p(q2), len(q1), _p(idx), H, thr, _p(E), _p(nm), _p(cnt))
    return E, nm, cnt


def find_essential_ransac(p1, p2, fx, fy, cx, cy, prob=0.999, threshold=1.0, max_iters=1000):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = len(p1)
    mask = np.zeros(max(n, 1), np.uint8)
    E = np.zeros(9, np.float64)
    iters = C.c_int(0)
    ok = lib().ora_find_essential_ransac(_p(p1), _p(p2), n, fx, fy, cx, cy, prob, threshold, max_iters, _p(mask),
                                         _p(E), C.addressof(iters))
    return bool(ok), E.reshape(3, 3), mask[:n].copy(), iters.value


def decompose_essential(E):
    E = np.ascontiguousarray(E, np.float64).reshape(9)
    R1, R2, t = np.zeros(9), np.zeros(9), np.zeros(3)
    lib().ora_decompose_essential(_p(E), _p(R1), _p(R2), _p(t))
    return R1.reshape(3, 3), R2.reshape(3, 3), t


def recover_pose(E, p1, p2, fx, fy, cx, cy, mask=None, distance_thresh=50.0):
    """cv::recoverPose: (votes, R, t, mask) with mask the winner's 0/1 votes."""
    E = np.ascontiguousarray(E, np.float64).reshape(9)
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    n = len(p1)
    m = np.zeros(max(n, 1), np.uint8)
    if mask is not None:
        m[:n] = np.asarray(mask, np.uint8).reshape(-1)[:n]
    R, t = np.zeros(9), np.zeros(3)
    good = lib().ora_recover_pose(_p(E), _p(p1), _p(p2), n, fx, fy, cx, cy, distance_thresh,
                                  int(mask is not None), _p(m), _p(R), _p(t))
    return good, R.reshape(3, 3), t, m[:n].copy()


def stereo_match(lk, ld, rk, rd, fx, baseline, band=2.0, min_disp=1.0, max_disp=256.0, ratio=0.8, size=None):
    """Band-limited left/right matching with cross-check: (right_idx, disparity, depth[, depth plane])."""
    lk, rk = np.ascontiguousarray(lk, KP_DTYPE), np.ascontiguousarray(rk, KP_DTYPE)
    ld, rd = np.ascontiguousarray(ld, np.uint8), np.ascontiguousarray(rd, np.uint8)
    nl = len(lk)
    ridx = np.zeros(max(nl, 1), np.int32)
    disp = np.zeros(max(nl, 1), np.float32)
    depth = np.zeros(max(nl, 1), np.float32)
    lib().ora_stereo_match(_p(lk), _p(ld), nl, _p(rk), _p(rd), len(rk), band, min_disp, max_disp, ratio, fx, baseline,
                           _p(ridx), _p(disp), _p(depth))
    out = (ridx[:nl].copy(), disp[:nl].copy(), depth[:nl].copy())
    if size is None:
        return out
    w, h = size
    plane = np.zeros((h, w), np.float32)
    lib().ora_stereo_depth_plane(_p(lk), _p(depth), nl, w, h, _p(plane), w)
    return out + (plane,)


def match_filter(knn, nt, qk, tk, ratio=0.7, knn_rev=None):
    nq = len(knn)
    out = np.zeros(max(nq, 1), MATCH_DTYPE)
    mask = np.zeros(max(nq, 1), np.uint8)
    nr = C.c_int(0)
    n = lib().ora_match_filter_x(_p(knn), _p(knn_rev) if knn_rev is not None else None, nq, nt, _p(qk),
                                 _p(tk), ratio, _p(out), _p(mask), C.addressof(nr))
    return out[:n].copy(), mask[:nr.value].copy(), nr.value


def backproject(matches, depth, fx, fy, cx, cy):
    m = np.ascontiguousarray(matches).copy()
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = depth.shape
    pts3 = np.zeros((max(len(m), 1), 3), np.float32)
    n = lib().ora_backproject(_p(m), len(m), _p(depth), w, h, w, fx, fy, cx, cy, _p(pts3))
    if n == 0:
        return m, pts3[:0].copy()
    return m[:n].copy(), pts3[:n].copy()


def track_filter(matches, prev_age, n_curr):
    m = np.ascontiguousarray(matches).copy()
    prev_age = np.ascontiguousarray(prev_age, np.int32)
    curr_age = np.zeros(max(n_curr, 1), np.int32)
    n = lib().ora_track_filter(_p(m), len(m), _p(prev_age), len(prev_age), _p(curr_age), n_curr)
    return m[:n].copy(), curr_age[:n_curr].copy()


def solve_pnp_ransac(obj, img, fx, fy, cx, cy, iters=100, thr=8.0, conf=0.99):
    obj = np.ascontiguousarray(obj, np.float32)
    img = np.ascontiguousarray(img, np.float32)
    n = len(obj)
    rvec = np.zeros(3)
    tvec = np.zeros(3)
    inl = np.zeros(max(n, 1), np.int32)
    ninl = C.c_int(0)
    ok = lib().ora_solve_pnp_ransac(_p(obj), _p(img), n, fx, fy, cx, cy, iters, thr, conf,
                                    _p(rvec), _p(tvec), _p(inl), C.addressof(ninl))
    return bool(ok), rvec, tvec, inl[:ninl.value].copy()


def rodrigues(v):
    v = np.ascontiguousarray(v, np.float64)
    if v.size == 3:
        R = np.zeros(9)
        lib().ora_rodrigues_vec2mat(_p(v), _p(R))
        return R.reshape(3, 3)
    r = np.zeros(3)
    lib().ora_rodrigues_mat2vec(_p(v.reshape(9)), _p(r))
    return r
