#!/usr/bin/env python3
"""Throughput of the secondary benchmark configurations on one GPU (bench.py measures the headline one):
  stereo1080 -- HD1080 stereo rig, 4000 key points per image from the full detector, left-right matching on the
                device (vo_batch_submit_stereo), temporal matching with cross-check, F-RANSAC, PnP;
  kitti      -- 1241x376 BGR sequences, 8 streams per launch, 2000 key points, full detector;
  estress    -- 5000 correspondences at 40 % outliers, 4096 five-point hypotheses scored (host call included).
One JSON line per configuration.  Inputs are resident in HBM for the pipeline configurations."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pingpong(n):
    return list(range(n)) + list(range(n - 2, 0, -1))


def run_pipeline(name, cam_name, channels, streams, frames, features, stereo, cross_check, steps, warmup):
    import torch
    from visual_odometry_amd import capi, synth
    cam = synth.CONFIG_CAMERAS[cam_name]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    n_distinct = frames // 2 + 1
    scene = synth.Scene(77, relief=2.0)
    order = pingpong(n_distinct)
    left = synth.make_sequence(cam, n_distinct, 77, channels, scene=scene)
    lf = np.concatenate([np.stack([left["frames"][i] for i in order])] * streams)
    B = streams * frames
    ctx = capi.Context(0, cam.width, cam.height, features + 64, B * (2 if stereo else 1))
    ctx.batch_configure(cam.width, cam.height, channels, n_streams=streams, frames_per_stream=frames,
                        max_features=features, camera=K, orb_levels=8, cross_check=cross_check)
    lf_d = torch.from_numpy(lf).cuda()
    row, fr = cam.width * channels, cam.width * channels * cam.height
    if stereo:
        right = synth.make_sequence(cam, n_distinct, 78, channels, scene=scene, baseline_mm=120.0)
        rf_d = torch.from_numpy(np.concatenate([np.stack([right["frames"][i] for i in order])] * streams)).cuda()

        def step():
            ctx.batch_submit_stereo((lf_d.data_ptr(), row, fr), (rf_d.data_ptr(), row, fr), 120.0, band=2.0,
                                    min_disp=1.0, max_disp=400.0, ratio=0.8)
            return ctx.batch_wait()
    else:
        dp_d = torch.from_numpy(np.concatenate([np.stack([left["depth"][i] for i in order])] * streams)).cuda()

        def step():
            ctx.batch_submit_ptr(lf_d.data_ptr(), row, fr, dp_d.data_ptr(), cam.width, cam.width * cam.height)
            return ctx.batch_wait()
    for _ in range(warmup):
        res = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    truth = [synth.relative_pose(left["poses"][order[k - 1]], left["poses"][order[k]]) for k in range(frames)]
    t_err = [float(np.linalg.norm(res["tvec"][b] - truth[b % frames][1])) for b in range(B) if res["pnp_success"][b]]
    out = {"config": name, "value": B * steps / dt, "unit": "frames/s" if not stereo else "stereo frames/s",
           "ms_per_step": 1e3 * dt / steps, "frames_per_launch": B, "image": [cam.width, cam.height, channels],
           "max_features": features, "mean_keypoints": float(res["n_keypoints"].mean()),
           "mean_matches": float(res["n_matches"].mean()), "mean_3d_points": float(res["n_3d"].mean()),
           "pose_success_rate": float(res["pnp_success"].mean()),
           "median_translation_error_mm": float(np.median(t_err)) if t_err else None,
           "ground_truth_step_mm": float(np.linalg.norm(truth[1][1])),
           "pnp_iterations": {"mean": float(res["pnp_iters"].mean()), "max": int(res["pnp_iters"].max()),
                              "above_8": int((res["pnp_iters"] > 8).sum())},
           "f_ransac_iterations_mean": float(res["f_iters"].mean()) if "f_iters" in res.dtype.names else None}
    ctx.close()
    print(json.dumps(out), flush=True)


def run_estress(steps):
    import torch  # noqa: F401  (binds the HIP runtime first)
    from visual_odometry_amd import capi
    rng = np.random.default_rng(5)
    n, H = 5000, 4096
    a = rng.uniform(-0.17, 0.17, 3)
    ca, sa, cb, sb, cc, sc = np.cos(a[0]), np.sin(a[0]), np.cos(a[1]), np.sin(a[1]), np.cos(a[2]), np.sin(a[2])
    R = (np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]]) @ np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]]) @
         np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]]))
    t = rng.normal(size=3); t /= np.linalg.norm(t)
    X = np.c_[rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(4, 12, n)]
    X2 = X @ R.T + t
    p1 = (700 * X[:, :2] / X[:, 2:] + [640, 360] + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    p2 = (700 * X2[:, :2] / X2[:, 2:] + [640, 360] + rng.normal(0, 0.5, (n, 2))).astype(np.float32)
    out_idx = rng.permutation(n)[:2000]
    p2[out_idx] = rng.uniform(0, [1280, 720], (2000, 2)).astype(np.float32)
    samples = np.stack([rng.permutation(n)[:5] for _ in range(H)]).astype(np.int32)
    ctx = capi.Context(0, 1280, 720, 2000, 1)
    ctx.essential_hypotheses(p1, p2, 700.0, 700.0, 640.0, 360.0, samples, 1.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        E, nm, cnt = ctx.essential_hypotheses(p1, p2, 700.0, 700.0, 640.0, 360.0, samples, 1.0)
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": "estress", "value": H / dt, "unit": "five-point hypotheses solved and scored /s",
                      "ms_per_call": 1e3 * dt, "correspondences": n, "hypotheses": H, "models": int(nm.sum()),
                      "best_support": int(cnt.max()), "note": "host call: uploads, two kernels, downloads"}), flush=True)
    ctx.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="stereo1080,kitti,estress")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    for cfg in args.configs.split(","):
        if cfg == "stereo1080":
            run_pipeline(cfg, "hd1080", 4, 8, 8, 4000, True, True, args.steps, args.warmup)
        elif cfg == "kitti":
            run_pipeline(cfg, "kitti", 3, 8, 16, 2000, False, False, args.steps, args.warmup)
        elif cfg == "estress":
            run_estress(args.steps)
