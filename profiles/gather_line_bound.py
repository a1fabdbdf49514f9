#!/usr/bin/env python3
"""Lower bound on the memory traffic of the three gather stages of the full detector (Harris ranking, intensity-
centroid orientation, steered BRIEF), from the CPU oracle's key points on frames of the benchmark's streams.

Memory moves in lines, not bytes: a stage that reads a w-byte window around a key point on each of r rows pulls
at least one line per row, two when the window straddles a line boundary.  For each stage this script builds
the set of distinct lines the stage's windows touch in one frame (the windows the kernels actually load: 37 rows
x 44 bytes from a 4-byte aligned start for BRIEF, 31 rows x 32 bytes for the orientation disc, 9 rows x 12 bytes
from an aligned start for the Harris footprint of every ranked candidate) and prints the union in bytes for
128-byte and 64-byte lines beside the per-byte figure (2N x 81 + N x 709 for ranking + orientation, N x 544 for
BRIEF).  A kernel that fetches every line exactly once moves the union; measured traffic is to be read
against that, not against the per-byte figure.  CPU only; needs the oracle library (built by
__graft_entry__.build()).

    python3 profiles/gather_line_bound.py [frames] > profiles/rNN_gather_line_bound.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
from visual_odometry_amd import synth  # noqa: E402


def lines_of_windows(cx, cy, x0, width, rows_up, rows_down, pitch, line):
    """Distinct (row, line) pairs of windows [x0, x0 + width) on rows cy - rows_up .. cy + rows_down."""
    out = set()
    for a, y, w0 in zip(cx, cy, x0):
        first, last = w0 // line, (w0 + width - 1) // line
        for r in range(y - rows_up, y + rows_down + 1):
            for ln in range(first, last + 1):
                out.add(r * (pitch // line + 1) + ln)
    return len(out)


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cam = synth.CONFIG_CAMERAS["hd720"]
    nfeat = 2000
    quotas = oracle.orb_quotas(nfeat)
    lw, lh, ls = oracle.orb_levels(cam.width, cam.height)
    tot = {k: np.zeros(3) for k in ("harris", "orientation", "brief")}
    n_kp = n_cand = 0
    for s in range(n_frames):
        seq = synth.make_sequence(cam, 2, 2000 + s, channels=1, scene=synth.Scene(2000 + s, relief=2.0))
        g = seq["frames"][1]
        kps, _, lv, _ = oracle.orb_detect(g, nfeat, 1.2, 8, 20, pyramid=True)
        n_kp += len(kps)
        for l in range(8):
            pitch = (int(lw[l]) + 127) // 128 * 128
            k = kps[kps["octave"] == l]
            cx = np.rint(k["x"] / ls[l]).astype(int)
            cy = np.rint(k["y"] / ls[l]).astype(int)
            # ranked candidates: FAST corners 31 px inside the level, everything at or above the score of the
            # (2 x quota)-th strongest
            c = oracle.fast(lv[l], 20)
            x, y = c["x"].astype(int), c["y"].astype(int)
            inside = (x >= 31) & (x < lw[l] - 31) & (y >= 31) & (y < lh[l] - 31)
            c, x, y = c[inside], x[inside], y[inside]
            want = 2 * int(quotas[l])
            if len(c) > want:
                cut = np.sort(c["response"])[::-1][want - 1]
                keep = c["response"] >= cut
                x, y = x[keep], y[keep]
            n_cand += len(x)
            for j, line in enumerate((128, 64)):
                tot["harris"][j] += line * lines_of_windows(x, y, (x - 4) & ~3, 12, 4, 4, pitch, line)
                tot["orientation"][j] += line * lines_of_windows(cx, cy, cx - 16, 32, 15, 15, pitch, line)
                tot["brief"][j] += line * lines_of_windows(cx, cy, (cx - 18) & ~3, 44, 18, 18, pitch, line)
            tot["harris"][2] += 81 * len(x)
            tot["orientation"][2] += 709 * len(k)
            tot["brief"][2] += 544 * len(k)
    print(f"{n_frames} HD720 frames of the benchmark's scenes (seeds 2000..), {n_kp / n_frames:.0f} key points and "
          f"{n_cand / n_frames:.0f} ranked candidates per frame")
    print(f"{'stage':14s} {'per-byte figure':>16s} {'distinct 128-B lines':>22s} {'distinct 64-B lines':>21s}   (KB per frame)")
    for k, v in tot.items():
        v = v / n_frames / 1e3
        print(f"{k:14s} {v[2]:16.0f} {v[0]:22.0f} {v[1]:21.0f}")


if __name__ == "__main__":
    main()
