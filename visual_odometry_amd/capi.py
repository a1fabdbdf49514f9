"""ctypes binding of libvo_hip.so (the C ABI declared in include/vo_hip.h).

The library is the product; this module only marshals numpy arrays to plain
pointers.  There is no fallback: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VO_LIB_PATH: a diagnostic build of the library, see profiles/diag)
LIB_PATH = os.environ.get("VO_LIB_PATH") or os.path.join(_HERE, "libvo_hip.so")

VO_OK = 0
VO_ERR_ARG = -1
VO_ERR_HIP = -2
VO_ERR_STATE = -3
VO_ERR_UNSUPPORTED = -4

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
KNN_DTYPE = np.dtype([("idx0", "<i4"), ("dist0", "<i4"), ("idx1", "<i4"), ("dist1", "<i4")])
MATCH_DTYPE = np.dtype([("query", "<i4"), ("train", "<i4"), ("distance", "<f4"),
                        ("px", "<f4"), ("py", "<f4"), ("cx", "<f4"), ("cy", "<f4")])


class VoCfg(C.Structure):
    _fields_ = [("max_width", C.c_int), ("max_height", C.c_int), ("max_features", C.c_int),
                ("max_batch", C.c_int), ("gray_b", C.c_int), ("gray_g", C.c_int),
                ("gray_r", C.c_int), ("gray_shift", C.c_int)]


class VoBatchCfg(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("channels", C.c_int),
                ("n_streams", C.c_int), ("frames_per_stream", C.c_int),
                ("max_features", C.c_int), ("fast_threshold", C.c_int), ("ratio", C.c_float),
                ("f_threshold", C.c_double), ("f_confidence", C.c_double), ("f_max_iters", C.c_int),
                ("enable_pose", C.c_int), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("pnp_iters", C.c_int), ("pnp_threshold", C.c_double),
                ("pnp_confidence", C.c_double), ("cross_check", C.c_int),
                ("orb_levels", C.c_int), ("orb_scale_factor", C.c_float)]


class VoStereoCfg(C.Structure):
    _fields_ = [("baseline", C.c_double), ("band", C.c_float), ("min_disparity", C.c_float),
                ("max_disparity", C.c_float), ("ratio", C.c_float)]


RESULT_DTYPE = np.dtype([("n_keypoints", "<i4"), ("n_ratio", "<i4"), ("f_found", "<i4"),
                         ("f_iters", "<i4"), ("f_inliers", "<i4"), ("n_matches", "<i4"),
                         ("n_tracked", "<i4"), ("n_3d", "<i4"), ("pnp_success", "<i4"),
                         ("pnp_inliers", "<i4"), ("pnp_iters", "<i4"), ("pnp_points", "<i4"),
                         ("rvec", "<f8", (3,)), ("tvec", "<f8", (3,))])


class VoFrameView(C.Structure):
    _fields_ = [("result", C.c_uint8 * 96), ("detect_ms", C.c_float), ("match_ms", C.c_float),
                ("n_keypoints", C.c_int32), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("n_matches", C.c_int32), ("n_tracked", C.c_int32), ("n_valid", C.c_int32),
                ("matches", C.c_void_p), ("tracked", C.c_void_p), ("valid", C.c_void_p), ("pts3", C.c_void_p),
                ("n_inliers", C.c_int32), ("inliers", C.c_void_p)]


class VoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libvo_hip error {code}: {msg}")
        self.code = code


# every symbol include/vo_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
EXPORTS = {
    "vo_create": (_P, [C.c_int, C.POINTER(VoCfg)]),
    "vo_destroy": (None, [_P]),
    "vo_last_error": (C.c_char_p, [_P]),
    "vo_abi_version": (C.c_int, []),
    "vo_gray": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "vo_fast_score": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "vo_blur7": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int]),
    "vo_detect": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "vo_knn2_hamming": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P]),
    "vo_knn2_hamming_static": (C.c_int, [C.c_int, _P, _P, C.c_int, _P, C.c_int, _P, C.c_int]),
    "vo_static_last_error": (C.c_char_p, []),
    "vo_find_fundamental_ransac": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_int,
                                             _P, _P, _P, _P]),
    "vo_match_features": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_float, _P, _P]),
    "vo_detect_orb": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, _P, _P,
                                C.c_int, _P]),
    "vo_orb_get_level": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "vo_find_essential_ransac": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_int, _P, _P, _P, _P]),
    "vo_essential_hypotheses": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _P,
                                          C.c_int, C.c_double, _P, _P, _P]),
    "vo_undistort": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _P,
                               C.c_int, _P, C.c_int]),
    "vo_stereo_depth": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_double, C.c_double, C.c_float, C.c_float,
                                  C.c_float, C.c_float, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "vo_recover_pose": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                  C.c_double, C.c_int, _P, _P, _P, _P]),
    "vo_image_process": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_double, _P, _P, _P, _P, C.c_int]),
    "vo_match_features_ex": (C.c_int, [_P, _P, _P, C.c_int, _P, _P, C.c_int, C.c_float, C.c_int, _P, _P]),
    "vo_solve_pnp_ransac": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_int, C.c_double, C.c_double, _P, _P, _P, _P, _P]),
    "vo_batch_configure": (C.c_int, [_P, C.POINTER(VoBatchCfg)]),
    "vo_batch_reset": (C.c_int, [_P]),
    "vo_batch_set_camera": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double]),
    "vo_batch_set_previous": (C.c_int, [_P, C.c_int, _P, _P, C.c_int]),
    "vo_batch_submit": (C.c_int, [_P, _P, C.c_int, C.c_size_t, _P, C.c_int, C.c_size_t]),
    "vo_batch_submit_depth16": (C.c_int, [_P, _P, C.c_int, C.c_size_t, _P, C.c_int, C.c_size_t]),
    "vo_batch_submit_stereo": (C.c_int, [_P, _P, _P, C.c_int, C.c_size_t, C.POINTER(VoStereoCfg)]),
    "vo_batch_get_stereo": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "vo_batch_wait": (C.c_int, [_P, _P]),
    "vo_batch_get_features": (C.c_int, [_P, C.c_int, _P, _P, _P]),
    "vo_batch_get_matches": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P]),
    "vo_batch_get_inliers": (C.c_int, [_P, C.c_int, _P, _P]),
    "vo_batch_set_timing": (C.c_int, [_P, C.c_int]),
    "vo_batch_set_packed_output": (C.c_int, [_P, C.c_int]),
    "vo_batch_get_packed": (C.c_int, [_P, C.c_int, _P]),
    "vo_batch_set_graph": (C.c_int, [_P, C.c_int]),
    "vo_host_alloc": (_P, [C.c_size_t]),
    "vo_host_free": (None, [_P]),
    "vo_set_matcher": (C.c_int, [_P, C.c_int]),
    "vo_batch_stage_count": (C.c_int, []),
    "vo_batch_stage_name": (C.c_char_p, [C.c_int]),
    "vo_batch_get_timing": (C.c_int, [_P, _P]),
}

_lib = None


def load():
    """Load libvo_hip.so and bind every exported symbol. Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                              "(make -C visual_odometry_amd/csrc)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def knn2_static(query: np.ndarray, train: np.ndarray, device: int = 0) -> np.ndarray:
    """vo_knn2_hamming_static: the kNN-2 search without a context and without any allocation in the library."""
    lib = load()
    query = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
    train = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
    out = np.zeros(len(query), KNN_DTYPE)
    rc = lib.vo_knn2_hamming_static(device, None, _ptr(query), len(query), _ptr(train), len(train), _ptr(out), 1)
    if rc != VO_OK:
        raise VoError(rc, lib.vo_static_last_error().decode())
    return out


class Context:
    """One vo_ctx: a camera stream's device state on one GPU."""

    def __init__(self, device=0, max_width=1280, max_height=720, max_features=2000, max_batch=1):
        self._lib = load()
        cfg = VoCfg(max_width, max_height, max_features, max_batch, 0, 0, 0, 0)
        self._h = self._lib.vo_create(device, C.byref(cfg))
        if not self._h:
            raise VoError(VO_ERR_HIP, self._lib.vo_last_error(None).decode())
        self.max_features = max_features
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vo_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != VO_OK:
            raise VoError(rc, self._lib.vo_last_error(self._h).decode())

    # ---- single-frame operators -------------------------------------
    def gray(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, np.uint8)
        h, w, cn = img.shape
        out = np.empty((h, w), np.uint8)
        self._check(self._lib.vo_gray(self._h, _ptr(img), w, h, w * cn, cn, _ptr(out), w))
        return out

    def fast_score(self, gray: np.ndarray, threshold: int) -> np.ndarray:
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        out = np.empty((h, w), np.uint8)
        self._check(self._lib.vo_fast_score(self._h, _ptr(gray), w, h, w, threshold, _ptr(out), w))
        return out

    def blur7(self, gray: np.ndarray) -> np.ndarray:
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        out = np.empty((h, w), np.uint8)
        self._check(self._lib.vo_blur7(self._h, _ptr(gray), w, h, w, _ptr(out), w))
        return out

    def detect(self, gray: np.ndarray, max_features: int = 1000, fast_threshold: int = 20):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        kps = np.zeros(max_features, KP_DTYPE)
        desc = np.zeros((max_features, 32), np.uint8)
        n = C.c_int(0)
        self._check(self._lib.vo_detect(self._h, _ptr(gray), w, h, w, max_features, fast_threshold,
                                        _ptr(kps), _ptr(desc), C.addressof(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def knn2(self, query: np.ndarray, train: np.ndarray) -> np.ndarray:
        query = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        train = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        out = np.zeros(len(query), KNN_DTYPE)
        self._check(self._lib.vo_knn2_hamming(self._h, _ptr(query), len(query), _ptr(train),
                                              len(train), _ptr(out)))
        return out

    def find_fundamental_ransac(self, p1, p2, threshold=3.0, confidence=0.99, max_iters=1000):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = len(p1)
        mask = np.zeros(max(n, 1), np.uint8)
        F = np.zeros(9, np.float64)
        found = C.c_int(0)
        iters = C.c_int(0)
        self._check(self._lib.vo_find_fundamental_ransac(
            self._h, _ptr(p1), _ptr(p2), n, threshold, confidence, max_iters, _ptr(mask), _ptr(F),
            C.addressof(found), C.addressof(iters)))
        return bool(found.value), F.reshape(3, 3), mask[:n].copy(), iters.value

    def find_essential_ransac(self, p1, p2, fx, fy, cx, cy, prob=0.999, threshold=1.0, max_iters=1000):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = len(p1)
        mask = np.zeros(max(n, 1), np.uint8)
        E = np.zeros(9, np.float64)
        found, iters = C.c_int(0), C.c_int(0)
        self._check(self._lib.vo_find_essential_ransac(self._h, _ptr(p1), _ptr(p2), n, fx, fy, cx, cy, prob,
                                                       threshold, max_iters, _ptr(mask), _ptr(E),
                                                       C.addressof(found), C.addressof(iters)))
        return bool(found.value), E.reshape(3, 3), mask[:n].copy(), iters.value

    def undistort(self, gray, fx, fy, cx, cy, dist):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        d = np.ascontiguousarray(dist, np.float64).reshape(-1)
        out = np.zeros_like(gray)
        self._check(self._lib.vo_undistort(self._h, _ptr(gray), w, h, w, fx, fy, cx, cy, _ptr(d), len(d), _ptr(out),
                                           w))
        return out

    def stereo_depth(self, lk, ld, rk, rd, fx, baseline, band=2.0, min_disp=1.0, max_disp=256.0, ratio=0.8,
                     size=None):
        """Band-limited left/right matching with cross-check: (right_idx, disparity, depth[, depth plane])."""
        lk, rk = np.ascontiguousarray(lk, KP_DTYPE), np.ascontiguousarray(rk, KP_DTYPE)
        ld, rd = np.ascontiguousarray(ld, np.uint8), np.ascontiguousarray(rd, np.uint8)
        nl = len(lk)
        ridx = np.zeros(max(nl, 1), np.int32)
        disp = np.zeros(max(nl, 1), np.float32)
        depth = np.zeros(max(nl, 1), np.float32)
        w, h = size if size is not None else (0, 0)
        plane = np.zeros((h, w), np.float32) if size is not None else None
        n = C.c_int(0)
        self._check(self._lib.vo_stereo_depth(self._h, _ptr(lk), _ptr(ld), nl, _ptr(rk), _ptr(rd), len(rk), fx,
                                              baseline, band, min_disp, max_disp, ratio, _ptr(ridx), _ptr(disp),
                                              _ptr(depth), _ptr(plane) if plane is not None else None, w, h,
                                              C.addressof(n)))
        out = (ridx[:nl].copy(), disp[:nl].copy(), depth[:nl].copy())
        assert n.value == int((out[0] >= 0).sum())
        return out if plane is None else out + (plane,)

    def recover_pose(self, E, p1, p2, fx, fy, cx, cy, mask=None, distance_thresh=50.0):
        """cv::recoverPose: (votes, R, t, mask of the winning candidate's votes)."""
        E = np.ascontiguousarray(E, np.float64).reshape(9)
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        n = len(p1)
        m = np.zeros(max(n, 1), np.uint8)
        if mask is not None:
            m[:n] = np.asarray(mask, np.uint8).reshape(-1)[:n]
        R, t = np.zeros(9, np.float64), np.zeros(3, np.float64)
        good = C.c_int(0)
        self._check(self._lib.vo_recover_pose(self._h, _ptr(E), _ptr(p1), _ptr(p2), n, fx, fy, cx, cy,
                                              distance_thresh, int(mask is not None), _ptr(m), _ptr(R), _ptr(t),
                                              C.addressof(good)))
        return good.value, R.reshape(3, 3), t, m[:n].copy()

    def essential_hypotheses(self, p1, p2, fx, fy, cx, cy, samples, threshold=1.0):
        p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
        samples = np.ascontiguousarray(samples, np.int32).reshape(-1, 5)
        H = len(samples)
        E = np.zeros((H, 10, 9), np.float64)
        nm = np.zeros(H, np.int32)
        cnt = np.zeros((H, 10), np.int32)
        self._check(self._lib.vo_essential_hypotheses(self._h, _ptr(p1), _ptr(p2), len(p1), fx, fy, cx, cy,
                                                      _ptr(samples), H, threshold, _ptr(E), _ptr(nm), _ptr(cnt)))
        return E, nm, cnt

    def detect_orb(self, gray: np.ndarray, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20,
                   cap=None):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = cap or self.max_features
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(self._lib.vo_detect_orb(self._h, _ptr(gray), w, h, w, nfeatures, scale_factor, nlevels,
                                            fast_threshold, _ptr(kps), _ptr(desc), cap, C.addressof(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def orb_level_size(self, level: int):
        """(width, height) of pyramid level `level` of the last ORB detection / configuration."""
        w, h = C.c_int(0), C.c_int(0)
        self._check(self._lib.vo_orb_get_level(self._h, 0, level, 0, None, 0, C.addressof(w), C.addressof(h)))
        return w.value, h.value

    def orb_level(self, frame: int, level: int, blurred=False) -> np.ndarray:
        w, h = C.c_int(0), C.c_int(0)
        self._check(self._lib.vo_orb_get_level(self._h, frame, level, int(blurred), None, 0, C.addressof(w),
                                               C.addressof(h)))
        out = np.empty((h.value, w.value), np.uint8)
        self._check(self._lib.vo_orb_get_level(self._h, frame, level, int(blurred), _ptr(out), w.value, None, None))
        return out

    def process_image(self, img: np.ndarray, enhance=False, blur_size=3, sigma=1.0, hist_eq=False) -> dict:
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape[:2]
        cn = 1 if img.ndim == 2 else img.shape[2]
        planes = [np.empty((h, w), np.uint8) for _ in range(4 if enhance else 1)]
        ptrs = [_ptr(p) for p in planes] + [None] * (4 - len(planes))
        self._check(self._lib.vo_image_process(self._h, _ptr(img), w, h, w * cn, cn, int(enhance), int(hist_eq),
                                               blur_size, sigma, *ptrs, w))
        return dict(zip(("gray", "enhanced", "denoised", "masked"), planes))

    def match(self, prev_kps, prev_desc, curr_kps, curr_desc, ratio=0.7, cross_check=False) -> np.ndarray:
        prev_kps = np.ascontiguousarray(prev_kps, KP_DTYPE)
        curr_kps = np.ascontiguousarray(curr_kps, KP_DTYPE)
        prev_desc = np.ascontiguousarray(prev_desc, np.uint8)
        curr_desc = np.ascontiguousarray(curr_desc, np.uint8)
        out = np.zeros(max(len(prev_kps), 1), MATCH_DTYPE)
        n = C.c_int(0)
        self._check(self._lib.vo_match_features_ex(self._h, _ptr(prev_kps), _ptr(prev_desc), len(prev_kps),
                                                   _ptr(curr_kps), _ptr(curr_desc), len(curr_kps), ratio,
                                                   int(cross_check), _ptr(out), C.addressof(n)))
        return out[:n.value].copy()

    def solve_pnp_ransac(self, obj, img, fx, fy, cx, cy, iters=100, thr=8.0, conf=0.99):
        obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
        img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
        n = len(obj)
        rvec = np.zeros(3)
        tvec = np.zeros(3)
        inl = np.zeros(max(n, 1), np.int32)
        ninl = C.c_int(0)
        ok = C.c_int(0)
        self._check(self._lib.vo_solve_pnp_ransac(self._h, _ptr(obj), _ptr(img), n, fx, fy, cx, cy, iters,
                                                  thr, conf, _ptr(rvec), _ptr(tvec), _ptr(inl),
                                                  C.addressof(ninl), C.addressof(ok)))
        return bool(ok.value), rvec, tvec, inl[:ninl.value].copy()

    # ---- batched pipeline ---------------------------------------------
    def batch_configure(self, width, height, channels, n_streams=1, frames_per_stream=1,
                        max_features=1000, fast_threshold=20, ratio=0.7, f_threshold=3.0,
                        f_confidence=0.99, f_max_iters=1000, camera=None, pnp_iters=100,
                        pnp_threshold=8.0, pnp_confidence=0.99, cross_check=False, orb_levels=0,
                        orb_scale_factor=1.2):
        """camera: (fx, fy, cx, cy) enables depth back-projection + PnP."""
        fx, fy, cx, cy = camera if camera is not None else (0.0, 0.0, 0.0, 0.0)
        cfg = VoBatchCfg(width, height, channels, n_streams, frames_per_stream, max_features,
                         fast_threshold, ratio, f_threshold, f_confidence, f_max_iters,
                         1 if camera is not None else 0, fx, fy, cx, cy, pnp_iters, pnp_threshold,
                         pnp_confidence, int(cross_check), orb_levels, orb_scale_factor)
        self._check(self._lib.vo_batch_configure(self._h, C.byref(cfg)))
        self._bcfg = cfg
        self._batch = n_streams * frames_per_stream

    def batch_reset(self):
        self._check(self._lib.vo_batch_reset(self._h))

    def batch_submit_ptr(self, frames_ptr, row_stride, frame_stride, depth_ptr=None, depth_row_stride=0,
                         depth_frame_stride=0, depth16=False):
        """Raw pointers (ints), e.g. device memory owned by the caller; depth16: the depth planes are uint16 mm."""
        fn = self._lib.vo_batch_submit_depth16 if depth16 else self._lib.vo_batch_submit
        self._check(fn(self._h, frames_ptr, row_stride, frame_stride, depth_ptr, depth_row_stride, depth_frame_stride))

    def batch_submit(self, frames: np.ndarray, depth: np.ndarray | None = None):
        """frames: (batch, H, W[, C]) uint8 host array; depth: (batch, H, W) float32, or uint16 millimetres."""
        frames = np.ascontiguousarray(frames, np.uint8)
        cfg = self._bcfg
        assert frames.shape[0] == self._batch and frames.shape[1] == cfg.height and frames.shape[2] == cfg.width
        row = cfg.width * cfg.channels
        self._keep = [frames]
        dptr, drow, dfr = None, 0, 0
        fn = self._lib.vo_batch_submit
        if depth is not None:
            if depth.dtype == np.uint16:   # 16-bit millimetres: the extension entry point
                depth = np.ascontiguousarray(depth)
                fn = self._lib.vo_batch_submit_depth16
            else:
                depth = np.ascontiguousarray(depth, np.float32)
            self._keep.append(depth)
            dptr, drow, dfr = _ptr(depth), cfg.width, cfg.width * cfg.height
        self._check(fn(self._h, _ptr(frames), row, row * cfg.height, dptr, drow, dfr))

    def batch_submit_stereo(self, left, right, baseline, band=2.0, min_disp=1.0, max_disp=256.0, ratio=0.8):
        """left / right: (batch, H, W[, C]) uint8 host arrays, or (pointer, row_stride, frame_stride) tuples of
        device memory sharing one layout."""
        st = VoStereoCfg(baseline, band, min_disp, max_disp, ratio)
        cfg = self._bcfg
        if isinstance(left, tuple):
            (lp, row, fr), rp = left, right[0]
            self._check(self._lib.vo_batch_submit_stereo(self._h, lp, rp, row, fr, C.byref(st)))
            return
        left, right = np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(right, np.uint8)
        assert left.shape == right.shape and left.shape[0] == self._batch
        row = cfg.width * cfg.channels
        self._keep = [left, right]
        self._check(self._lib.vo_batch_submit_stereo(self._h, _ptr(left), _ptr(right), row, row * cfg.height,
                                                     C.byref(st)))

    def batch_stereo(self, frame: int):
        """(right_idx, disparity, depth) per left key point of a frame of the last stereo batch."""
        n = len(self.batch_features(frame)[0])
        ridx = np.zeros(max(n, 1), np.int32)
        disp = np.zeros(max(n, 1), np.float32)
        depth = np.zeros(max(n, 1), np.float32)
        nm = C.c_int(0)
        self._check(self._lib.vo_batch_get_stereo(self._h, frame, _ptr(ridx), _ptr(disp), _ptr(depth),
                                                  C.addressof(nm)))
        assert nm.value == int((ridx[:n] >= 0).sum())
        return ridx[:n].copy(), disp[:n].copy(), depth[:n].copy()

    def batch_wait(self) -> np.ndarray:
        res = np.zeros(self._batch, RESULT_DTYPE)
        self._check(self._lib.vo_batch_wait(self._h, _ptr(res)))
        self._keep = None
        return res

    def batch_features(self, frame: int):
        kps = np.zeros(self.max_features, KP_DTYPE)
        desc = np.zeros((self.max_features, 32), np.uint8)
        n = C.c_int(0)
        self._check(self._lib.vo_batch_get_features(self._h, frame, _ptr(kps), _ptr(desc), C.addressof(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def batch_matches(self, frame: int, which: int = 0):
        out = np.zeros(self.max_features, MATCH_DTYPE)
        pts3 = np.zeros((self.max_features, 3), np.float32)
        n = C.c_int(0)
        self._check(self._lib.vo_batch_get_matches(self._h, frame, which, _ptr(out), _ptr(pts3),
                                                   C.addressof(n)))
        return out[:n.value].copy(), pts3[:n.value].copy()

    def batch_inliers(self, frame: int):
        out = np.zeros(self.max_features, np.int32)
        n = C.c_int(0)
        self._check(self._lib.vo_batch_get_inliers(self._h, frame, _ptr(out), C.addressof(n)))
        return out[:n.value].copy()

    def batch_set_packed_output(self, enable: bool):
        self._check(self._lib.vo_batch_set_packed_output(self._h, int(enable)))

    def batch_set_graph(self, enable: bool):
        self._check(self._lib.vo_batch_set_graph(self._h, int(enable)))

    def batch_packed(self, frame: int) -> dict:
        """Everything of one frame of the last batch from the packed output block (copies out of pinned memory)."""
        v = VoFrameView()
        self._check(self._lib.vo_batch_get_packed(self._h, frame, C.byref(v)))

        def arr(ptr, n, dtype, shape=None):
            if n <= 0:
                return np.zeros((0,) + (shape or ()), dtype)
            nbytes = n * np.dtype(dtype).itemsize * (int(np.prod(shape)) if shape else 1)
            a = np.frombuffer(C.string_at(ptr, nbytes), dtype=dtype)
            return a.reshape((n,) + shape) if shape else a
        return {"result": np.frombuffer(bytes(v.result), RESULT_DTYPE)[0], "detect_ms": v.detect_ms,
                "match_ms": v.match_ms, "kps": arr(v.kps, v.n_keypoints, KP_DTYPE),
                "desc": arr(v.desc, v.n_keypoints, np.uint8, (32,)),
                "matches": arr(v.matches, v.n_matches, MATCH_DTYPE), "tracked": arr(v.tracked, v.n_tracked, MATCH_DTYPE),
                "valid": arr(v.valid, v.n_valid, MATCH_DTYPE), "pts3": arr(v.pts3, v.n_valid, np.float32, (3,)),
                "inliers": arr(v.inliers, v.n_inliers, np.int32)}

    def set_matcher(self, kind: str):
        """'mfma' (default): all-pairs Hamming distances on the matrix cores; 'popcount': XOR + popcount on the
        vector ALU.  Identical results."""
        self._check(self._lib.vo_set_matcher(self._h, {"mfma": 0, "popcount": 1}[kind]))

    def batch_set_timing(self, enable: bool):
        self._check(self._lib.vo_batch_set_timing(self._h, int(enable)))

    def batch_timing(self) -> dict:
        n = self._lib.vo_batch_stage_count()
        ms = np.zeros(n, np.float32)
        self._check(self._lib.vo_batch_get_timing(self._h, _ptr(ms)))
        return {self._lib.vo_batch_stage_name(i).decode(): float(ms[i]) for i in range(n)}
