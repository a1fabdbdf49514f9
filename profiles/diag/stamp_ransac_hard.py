#!/usr/bin/env python3
"""RANSAC stages on the moving-objects scene: distribution of the iteration counts over the frames of one launch
and, from the diagnostic build (make -C profiles/diag), the cycle stamps of the F-RANSAC and PnP kernels.

    python3 profiles/diag/stamp_ransac_hard.py [movers] [streams] [flat]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch  # first: the library must bind to the HIP runtime torch loads

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_odometry_amd import capi, synth  # noqa: E402


def main():
    diag = os.path.join(ROOT, "profiles", "diag", "libvo_hip_diag.so")
    use_diag = os.path.exists(diag) and os.environ.get("VO_NO_DIAG", "0") != "1"
    if use_diag:
        capi.LIB_PATH = diag
    lib = capi.load()

    movers = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    F = 16
    cam = synth.CONFIG_CAMERAS[os.environ.get("VO_DIAG_CAM", "hd720")]   # (VO_DIAG_CAM=kitti: the KITTI-shaped frames)
    flat = len(sys.argv) > 3 and sys.argv[3] == "flat"   # the flat-wall scene of the benchmark instead
    if flat:
        f, d, _ = synth.make_pingpong_batch(cam, S, F, 72000, channels=4, relief=0.0, movers=0, workers=16,
                                            tilt=(0.08, -0.05))
    else:
        f, d, _ = synth.make_pingpong_batch(cam, S, F, 52000, channels=4, relief=2.0, movers=movers, workers=16)
    frames = torch.from_numpy(f).cuda()
    depth = torch.from_numpy(d).cuda()
    ctx = capi.Context(0, cam.width, cam.height, 2064, S * F)
    ctx.batch_configure(cam.width, cam.height, 4, S, F, max_features=2000, camera=(cam.fx, cam.fy, cam.cx, cam.cy),
                        orb_levels=8)
    ctx.batch_set_timing(True)
    for _ in range(3):
        ctx.batch_submit_ptr(frames.data_ptr(), cam.width * 4, cam.width * 4 * cam.height, depth.data_ptr(), cam.width,
                             cam.width * cam.height)
        res = ctx.batch_wait()
    tm = ctx.batch_timing()
    print("stage ms:", {k: round(v, 3) for k, v in tm.items()})
    m = res["n_ratio"] > 0


    def hist(name, v, bins):
        h, _ = np.histogram(v, bins=bins)
        print(f"{name}: mean {v.mean():.1f} max {v.max()}  hist " +
              " ".join(f"[{bins[i]},{bins[i + 1]}):{h[i]}" for i in range(len(h))))


    hist("ratio survivors", res["n_ratio"][m], [0, 200, 400, 600, 800, 1000, 1200, 1600, 2100])
    hist("F iterations", res["f_iters"][m], [0, 9, 17, 33, 73, 137, 201, 329, 1001])
    hist("PnP points", res["pnp_points"][m], [0, 100, 200, 400, 600, 800, 1000, 2100])
    hist("PnP iterations", res["pnp_iters"][m], [0, 9, 19, 29, 39, 49, 69, 89, 101])

    if use_diag:
        raw = C.CDLL(capi.LIB_PATH)
        pn = np.zeros((256, 16), np.uint64)
        raw.vo_debug_pnp_stamps.argtypes = [C.c_void_p]
        assert raw.vo_debug_pnp_stamps(pn.ctypes.data) == 0
        t = pn[:, :15].astype(np.int64)
        # stamps: 0 start, 1 points loaded, 2 subsets chained (last round), 3 hypotheses solved, 4 scored + accepted,
        # 5 inlier list, 6 planarity, 7 DLT sums, 8 null vector, 9 initial pose, 10 Levenberg-Marquardt done;
        # inside the solver (last round): 11 start, 12 control points, 13 singular vectors, 14 betas + pose
        names = ["load points", "rng window + chain (last round)", "solve hypotheses (last round)",
                 "score + accept (last round)", "inlier mask + list", "planarity", "DLT sums", "null vector",
                 "polar + Rodrigues", "Levenberg-Marquardt"]
        for i, nme in enumerate(names):
            dlt = t[:, i + 1] - t[:, i]
            print(f"PnP {nme:36s} median {np.median(dlt):9.0f} p90 {np.percentile(dlt, 90):9.0f}")
        for i, nme in enumerate(["solver: control points", "solver: singular vectors", "solver: betas + pose"]):
            dlt = t[:, 12 + i] - t[:, 11 + i]
            print(f"PnP   {nme:34s} median {np.median(dlt):9.0f} p90 {np.percentile(dlt, 90):9.0f}")
        for nme, dlt in (("solver: subset + load", t[:, 11] - t[:, 2]), ("solver: pick + Rodrigues + barrier", t[:, 3] - t[:, 14])):
            print(f"PnP   {nme:34s} median {np.median(dlt):9.0f} p90 {np.percentile(dlt, 90):9.0f}")
        tot = t[:, 10] - t[:, 0]
        print("PnP whole block: median %d p90 %d max %d" % (np.median(tot), np.percentile(tot, 90), tot.max()))
        if os.environ.get("VO_DIAG_DUMP"):   # the null-vector phase (identical work in every block) block by block
            nb = min(256, S * F)
            print("PnP null vector by block:", (t[:nb, 8] - t[:nb, 7]).tolist())
            print("PnP start by block:", (t[:nb, 0] - t[:nb, 0].min()).tolist())
        if hasattr(raw, "vo_debug_pnp_lm"):
            lm = np.zeros((256, 8), np.uint64)
            raw.vo_debug_pnp_lm.argtypes = [C.c_void_p]
            assert raw.vo_debug_pnp_lm(lm.ctypes.data) == 0
            lm = lm[:min(256, S * F)].astype(np.int64)
            print("PnP LM: iterations median %d max %d, lambda steps median %d; cycles (median) sums %d, reduction %d, "
                  "6x6 solves %d, error evaluations %d" % (np.median(lm[:, 0]), lm[:, 0].max(), np.median(lm[:, 5]),
                                                           np.median(lm[:, 1]), np.median(lm[:, 2]), np.median(lm[:, 3]),
                                                           np.median(lm[:, 4])))
        if hasattr(raw, "vo_debug_pnp_blocks"):
            pb = np.zeros((1024, 8), np.uint64)
            raw.vo_debug_pnp_blocks.argtypes = [C.c_void_p]
            assert raw.vo_debug_pnp_blocks(pb.ctypes.data) == 0
            pb = pb[:S * F].astype(np.int64)
            t0 = pb[:, 0].min()
            own, end = pb[:, 1] - t0, pb[:, 2] - t0
            print("PnP blocks: handed over %d, items solved %d, items skipped %d, frames resumed %d" %
                  (pb[:, 3].sum(), pb[:, 4].sum(), pb[:, 5].sum(), pb[:, 6].sum()))
            print("PnP blocks: own frame done (cycles from the first start) median %d p90 %d max %d; block done median %d "
                  "p90 %d max %d; latest start %d" % (np.median(own[own > 0]), np.percentile(own[own > 0], 90), own.max(),
                                                      np.median(end), np.percentile(end, 90), end.max(),
                                                      (pb[:, 0] - t0).max()))
            h = pb[:, 3] > 0
            if h.any():
                print("PnP handed-over frames' blocks: done at", sorted(end[h].tolist())[:12])
        fr = np.zeros((256, 16), np.uint64)
        raw.vo_debug_fr_stamps.argtypes = [C.c_void_p]
        assert raw.vo_debug_fr_stamps(fr.ctypes.data) == 0
        t = fr[:, :10].astype(np.int64)
        ok = t[:, 9] > 0
        t = t[ok]
        names = ["load points", "(earlier rounds)", "rng window", "subset attempts", "chain (one lane)", "7-point solve",
                 "score", "accept (one lane)", "final mask + compaction"]
        dd = np.diff(t, axis=1)
        print("F-RANSAC blocks with stamps:", int(ok.sum()))
        for i, n in enumerate(names):
            print(f"{n:36s} median {np.median(dd[:, i]):9.0f}  p90 {np.percentile(dd[:, i], 90):9.0f} max {dd[:, i].max():9.0f}")
        tot = t[:, 9] - t[:, 0]
        print(f"{'whole kernel':36s} median {np.median(tot):9.0f} p90 {np.percentile(tot, 90):9.0f} max {tot.max():9.0f}")
        print("F kernel first start -> last end (blocks 0..255):", int(t[:, 9].max() - t[:, 0].min()))


if __name__ == "__main__":
    main()
