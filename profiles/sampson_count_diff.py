#!/usr/bin/env python3
"""How far the five-point scoring kernel's inlier counts are from the oracle's on the essential-matrix stress case
(5000 correspondences, 4096 hypotheses): both evaluate the Sampson error in fp64 from matrices that agree to
1e-9, so a count differs only where a point sits within rounding of the threshold.  Prints the worst difference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_essential as T
    from oracle import oracle as O
    from visual_odometry_amd import capi
    O.build()
    ctx = capi.Context(device=0, max_width=1920, max_height=1080, max_features=4000, max_batch=8)
    for seed, n, n_out, H in ((11, 300, 90, 256), (5, 5000, 2000, 4096)):
        p1, p2, Et, is_out = T._scene(seed, n, n_out)
        samples = T._sample_table(O, n, H)
        E, nm, cnt = ctx.essential_hypotheses(p1, p2, 700.0, 700.0, 640.0, 360.0, samples, 1.0)
        E_r, nm_r, cnt_r = O.e_hypotheses(T._norm(p1), T._norm(p2), samples, 1.0 / 700.0)
        d = np.abs(cnt - cnt_r)
        print(f"n={n} H={H}: candidate matrices {int(nm.sum())}, counts that differ {int((d > 0).sum())} "
              f"({100.0 * (d > 0).mean():.4f} % of the table), largest difference {int(d.max())}, "
              f"largest matrix difference {np.abs(E - E_r).max():.2e}")
    ctx.close()


if __name__ == "__main__":
    main()
