#!/usr/bin/env python3
"""Calibration load for the FETCH_SIZE counter on 1-byte gray input: the level-0 tile kernel over the benchmark's
512 HD720 frames, once on the BGRA frames (--channels 4) and once on an 8-bit gray copy of them (--channels 1).
Both runs stage the same tiles and aprons, so the gray run fetches a quarter of the BGRA run's bytes and writes the
same outputs; the ratio of the two FETCH_SIZE readings says whether the counter's gfx950 correction for the BGRA
kernel (x 2) carries over to the gray-input instantiation.  Run each under `rocprofv3 --pmc FETCH_SIZE`
(and WRITE_SIZE) and read the tile kernel's line of profiles/pmcstats.py."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, choices=(1, 4), default=4)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from bench import build_inputs
    from visual_odometry_amd import capi

    S, F = 32, 16
    cam, frames_h, _, _ = build_inputs(S, F, seed=2000)
    H, W, C = cam.height, cam.width, args.channels
    if C == 1:
        frames_h = np.ascontiguousarray(frames_h[..., 1])  # any 8-bit plane of the same size will do
    frames_d = torch.from_numpy(frames_h).cuda()
    ctx = capi.Context(device=0, max_width=W, max_height=H, max_features=2000, max_batch=S * F)
    ctx.batch_configure(W, H, C, n_streams=S, frames_per_stream=F, max_features=2000, fast_threshold=20)
    for _ in range(args.steps):
        ctx.batch_submit_ptr(frames_d.data_ptr(), W * C, W * C * H)
        res = ctx.batch_wait()
    print("channels", C, "mean key points", float(res["n_keypoints"].mean()))


if __name__ == "__main__":
    main()
