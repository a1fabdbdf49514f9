#!/usr/bin/env python3
"""Phase-by-phase cycle breakdown of the per-level ranking kernel of the full detector from the diagnostic
build (make -C profiles/diag): thread 0 of the blocks of the first 32 frames, per pyramid level."""
import ctypes as C
import os
import sys

import numpy as np
import torch  # first: the library must bind to the HIP runtime torch loads

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_odometry_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "profiles", "diag", "libvo_hip_diag.so")
capi.load()
cam = synth.CONFIG_CAMERAS["hd720"]
seq = synth.make_sequence(cam, 2, seed=2000, channels=4, scene=synth.Scene(2000, relief=2.0))
S, F = 32, 16
frames = torch.from_numpy(np.concatenate([seq["frames"]] * (S * F // 2))).cuda()
depth = torch.from_numpy(np.concatenate([seq["depth"]] * (S * F // 2))).cuda()
ctx = capi.Context(0, cam.width, cam.height, 2064, S * F)
ctx.batch_configure(cam.width, cam.height, 4, S, F, max_features=2000, camera=(cam.fx, cam.fy, cam.cx, cam.cy),
                    orb_levels=8)
for _ in range(3):
    ctx.batch_submit_ptr(frames.data_ptr(), cam.width * 4, cam.width * 4 * cam.height, depth.data_ptr(), cam.width,
                         cam.width * cam.height)
    ctx.batch_wait()
out = np.zeros((8, 32, 16), np.uint64)
raw = C.CDLL(capi.LIB_PATH)
raw.vo_debug_os_stamps.argtypes = [C.c_void_p]
assert raw.vo_debug_os_stamps(out.ctypes.data) == 0
names = ["flatten tile lists", "border filter + score histogram", "cut score (scan)", "gather survivors",
         "Harris responses", "radix select (4 passes)", "second cut (compaction)", "pad the list",
         "rank by counting + write level list"]
t = out[:, :, :10].astype(np.int64)
d = np.diff(t, axis=2)
for lvl in (0, 3, 7):
    print(f"level {lvl}: whole block median {np.median(t[lvl, :, 9] - t[lvl, :, 0]):.0f} cycles")
    for i, n in enumerate(names):
        print(f"  {n:34s} median {np.median(d[lvl, :, i]):8.0f}")
