#!/usr/bin/env python3
"""Distribution of the RANSAC iteration counts over the frames of one bench launch (headline and moving-objects
scene), beside the stage times of that launch: how much of the two RANSAC kernels is tail (the slowest frames)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from visual_odometry_amd import capi
    S, F, feats = 32, 16, 2000
    out = {}
    for name, movers, seed in (("headline", 0, 2000), ("hard", bench.HARD_SCENE_MOVERS, 52000)):
        cam, f, d, _ = bench.build_inputs(S, F, seed=seed, movers=movers, workers=16)
        W, H, C = cam.width, cam.height, 4
        fd, dd = torch.from_numpy(f).cuda(), torch.from_numpy(d).cuda()
        ctx = capi.Context(device=0, max_width=W, max_height=H, max_features=feats + 64, max_batch=S * F)
        ctx.batch_configure(W, H, C, n_streams=S, frames_per_stream=F, max_features=feats, fast_threshold=20,
                            camera=(cam.fx, cam.fy, cam.cx, cam.cy), orb_levels=8)
        for _ in range(3):
            ctx.batch_submit_ptr(fd.data_ptr(), W * C, W * C * H, dd.data_ptr(), W, W * H)
            res = ctx.batch_wait()
        ctx.batch_set_timing(True)
        acc = {}
        for _ in range(5):
            ctx.batch_submit_ptr(fd.data_ptr(), W * C, W * C * H, dd.data_ptr(), W, W * H)
            res = ctx.batch_wait()
            for k, v in ctx.batch_timing().items():
                acc[k] = acc.get(k, 0.0) + v / 5
        q = [0, 10, 25, 50, 75, 90, 95, 99, 100]
        out[name] = {
            "stage_ms": {k: round(v, 4) for k, v in acc.items()},
            "f_iters_percentiles": dict(zip(q, np.percentile(res["f_iters"], q).tolist())),
            "pnp_iters_percentiles": dict(zip(q, np.percentile(res["pnp_iters"], q).tolist())),
            "n_ratio_percentiles": dict(zip(q, np.percentile(res["n_ratio"], q).tolist())),
            "pnp_points_percentiles": dict(zip(q, np.percentile(res["pnp_points"], q).tolist())),
            "f_iters_hist": np.bincount(np.minimum(res["f_iters"], 1000) // 64, minlength=17).tolist(),
            "pnp_iters_hist": np.bincount(np.minimum(res["pnp_iters"], 100) // 10, minlength=11).tolist(),
        }
        ctx.close()
        del fd, dd
    print(json.dumps(out))


if __name__ == "__main__":
    main()
