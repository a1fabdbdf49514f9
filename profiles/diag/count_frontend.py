#!/usr/bin/env python3
"""Run a few batches of the one-level pipeline through the library given on the
command line (a cut variant from `make -C profiles/diag cuts`, or the product
library); meant to run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES`
so that the tile kernel's instruction counts can be split by phase."""
import os
import sys

import numpy as np
import torch  # first: the library must bind to the HIP runtime torch loads

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_odometry_amd import capi, synth  # noqa: E402

if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
capi.load()
cam = synth.CONFIG_CAMERAS["hd720"]
seq = synth.make_sequence(cam, 2, seed=2000, channels=4, scene=synth.Scene(2000, relief=2.0))
S, F = 16, 16
frames = torch.from_numpy(np.concatenate([seq["frames"]] * (S * F // 2))).cuda()
depth = torch.from_numpy(np.concatenate([seq["depth"]] * (S * F // 2))).cuda()
ctx = capi.Context(0, cam.width, cam.height, 2064, S * F)
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # 8: the full detector (level 0 + the upper-level launch)
ctx.batch_configure(cam.width, cam.height, 4, S, F, max_features=2000, camera=(cam.fx, cam.fy, cam.cx, cam.cy),
                    orb_levels=levels)
for _ in range(3):
    ctx.batch_submit_ptr(frames.data_ptr(), cam.width * 4, cam.width * 4 * cam.height, depth.data_ptr(), cam.width,
                         cam.width * cam.height)
    ctx.batch_wait()
print("done")
