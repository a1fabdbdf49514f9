"""FrameProcessor::processFrame restated on top of the CPU oracle
(reference src/frame_processor.cpp:18-202).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import numpy as np

from . import oracle as O


class OracleStream:
    """One camera stream: keeps the previous features and track ages."""

    def __init__(self, camera=None, max_features=1000, fast_threshold=20, ratio=0.7,
                 pnp_iters=100, pnp_threshold=8.0, pnp_confidence=0.99, cross_check=False, orb_levels=0,
                 orb_scale_factor=1.2, feature_cap=None):
        self.cross_check = cross_check
        self.orb = (orb_levels, orb_scale_factor)
        self.feature_cap = feature_cap
        self.camera = camera  # (fx, fy, cx, cy) or None
        self.max_features = max_features
        self.fast_threshold = fast_threshold
        self.ratio = ratio
        self.pnp = (pnp_iters, pnp_threshold, pnp_confidence)
        self.prev = None
        self.prev_ages = np.zeros(0, np.int32)

    @staticmethod
    def imu_filter(valid, pts3, camera, R, t, angular_rate):
        """IMU-guided feature filter (reference src/frame_processor.cpp:99-173)."""
        fx, fy, cx, cy = camera
        rv = O.rodrigues(np.asarray(R, np.float64))
        Rm = O.rodrigues(rv)
        P = pts3.astype(np.float64) @ Rm.T + np.asarray(t, np.float64).reshape(3)
        z = np.where(P[:, 2] != 0, 1.0 / P[:, 2], 1.0)
        u = (P[:, 0] * z * fx + cx).astype(np.float32)
        v = (P[:, 1] * z * fy + cy).astype(np.float32)
        dx = (u - valid["px"]).astype(np.float64)
        dy = (v - valid["py"]).astype(np.float64)
        err = np.sqrt(dx * dx + dy * dy)
        n = len(err)
        thr = 10.0 + angular_rate * 5.0
        min_retain = min(max(100, int(n * 0.6)), n)
        keep = np.flatnonzero(err <= thr)
        if len(keep) < min_retain:
            order = sorted(range(n), key=lambda i: (err[i], i))
            keep = np.array(sorted(order[:min_retain]), np.int64)
        return keep

    def process(self, frame: np.ndarray, depth: np.ndarray | None = None, imu=None) -> dict:
        gray = frame if frame.ndim == 2 else O.gray(frame)
        if self.orb[0] > 0:
            kps, desc = O.orb_detect(gray, self.max_features, self.orb[1], self.orb[0], self.fast_threshold)
            if self.feature_cap is not None:
                kps, desc = kps[:self.feature_cap], desc[:self.feature_cap]
        else:
            kps, desc = O.detect(gray, self.max_features, self.fast_threshold)
        out = {"kps": kps, "desc": desc, "matches": np.zeros(0, O.MATCH_DTYPE), "n_ratio": 0,
               "tracked": np.zeros(0, O.MATCH_DTYPE), "valid": np.zeros(0, O.MATCH_DTYPE),
               "pts3": np.zeros((0, 3), np.float32), "pnp_success": False, "rvec": np.zeros(3),
               "tvec": np.zeros(3), "inliers": np.zeros(0, np.int32)}
        if self.prev is not None:
            pk, pd = self.prev
            if len(pk) and len(kps):
                knn = O.knn2(pd, desc)
                rev = O.knn2(desc, pd) if self.cross_check else None
                m, _, nr = O.match_filter(knn, len(kps), pk, kps, self.ratio, knn_rev=rev)
                out["matches"], out["n_ratio"] = m, nr
            tracked, ages = O.track_filter(out["matches"], self.prev_ages, len(kps))
            out["tracked"] = tracked
            self.prev_ages = ages
            if self.camera is not None and depth is not None and len(tracked):
                fx, fy, cx, cy = self.camera
                valid, pts3 = O.backproject(tracked, depth, fx, fy, cx, cy)
                if len(pts3):
                    if imu is not None and len(pts3) >= 100:
                        keep = self.imu_filter(valid, pts3, self.camera, *imu)
                        if len(keep) < len(pts3):
                            valid, pts3 = valid[keep], pts3[keep]
                    out["valid"], out["pts3"] = valid, pts3
                    if len(pts3) >= 4:
                        img = np.stack([valid["px"], valid["py"]], 1)
                        ok, rvec, tvec, inl = O.solve_pnp_ransac(pts3, img, fx, fy, cx, cy, *self.pnp)
                        out.update(pnp_success=ok, rvec=rvec, tvec=tvec, inliers=inl)
        else:
            self.prev_ages = np.zeros(len(kps), np.int32)
        self.prev = (kps, desc)
        return out
