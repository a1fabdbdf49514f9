"""Repeat one launch of the moving-objects scene (the frames that hand their PnP work over to other blocks) many
times on the GPU and compare every launch with the first repeatable one bit for bit: result records, inlier lists
and poses.  The hand-over is lock-free (atomics, no block waits for another); which block solves which item
differs from launch to launch, the outputs must not.

    python3 tools/soak_hard.py [launches] [streams] [frames_per_stream] [movers] [relief]      (GPU box)

movers = 0 gives the benchmark's ordinary scene, relief = 0 the flat wall (the planar branch of the refinement).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import bench
    from visual_odometry_amd import capi
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    movers = int(sys.argv[4]) if len(sys.argv) > 4 else bench.HARD_SCENE_MOVERS
    relief = float(sys.argv[5]) if len(sys.argv) > 5 else 2.0
    cam, f, d, _ = bench.build_inputs(S, F, seed=52000, movers=movers, workers=16, relief=relief,
                                      tilt=(0.0, 0.0) if relief else (0.08, -0.05))
    dev = torch.device("cuda:0")
    fd, dd = torch.from_numpy(f).to(dev), torch.from_numpy(d).to(dev)
    B = S * F
    with capi.Context(device=0, max_width=cam.width, max_height=cam.height, max_features=2000, max_batch=B) as c:
        c.batch_configure(cam.width, cam.height, 4, n_streams=S, frames_per_stream=F, max_features=2000,
                          fast_threshold=20, camera=(cam.fx, cam.fy, cam.cx, cam.cy), orb_levels=8)
        row = cam.width * 4

        def launch():
            c.batch_submit_ptr(fd.data_ptr(), row, row * cam.height, dd.data_ptr(), cam.width, cam.width * cam.height)
            res = c.batch_wait()
            inl = [c.batch_inliers(b).tobytes() if res["pnp_success"][b] else b"" for b in range(0, B, 7)]
            return res.tobytes(), inl, res

        launch()                      # the streams' first frames have no predecessor in this one
        ref = launch()
        iters = ref[2]["pnp_iterations"] if "pnp_iterations" in ref[2].dtype.names else None
        bad = 0
        for k in range(n):
            cur = launch()
            if cur[0] != ref[0] or cur[1] != ref[1]:
                bad += 1
                print("launch", k, "differs", flush=True)
        print(f"movers {movers} relief {relief}: {n} launches of {B} frames, {bad} differ from the reference launch; pose success "
              f"{float(ref[2]['pnp_success'].mean()):.3f}" + (f", PnP iterations max {int(iters.max())}" if iters is not None else ""))
        return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
