#!/usr/bin/env python3
"""Phase-by-phase cycle breakdown of the fused front-end kernel from the
diagnostic build (make -C profiles/diag): median cycles between consecutive
stamps of wave 0 over a window of 4096 blocks in the middle of a launch."""
import ctypes as C
import os
import sys

import numpy as np
import torch  # first: the library must bind to the HIP runtime torch loads

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_odometry_amd import capi, synth  # noqa: E402

capi.LIB_PATH = os.path.join(ROOT, "profiles", "diag", "libvo_hip_diag.so")
lib = capi.load()

cam = synth.CONFIG_CAMERAS["hd720"]
seq = synth.make_sequence(cam, 2, seed=1, channels=4)
S, F = 16, 16
frames = torch.from_numpy(np.concatenate([seq["frames"]] * (S * F // 2))).cuda()
depth = torch.from_numpy(np.concatenate([seq["depth"]] * (S * F // 2))).cuda()
ctx = capi.Context(0, cam.width, cam.height, 2000, S * F)
ctx.batch_configure(cam.width, cam.height, 4, S, F, max_features=2000,
                    camera=(cam.fx, cam.fy, cam.cx, cam.cy))
for _ in range(3):
    ctx.batch_submit_ptr(frames.data_ptr(), cam.width * 4, cam.width * 4 * cam.height, depth.data_ptr(), cam.width,
                         cam.width * cam.height)
    ctx.batch_wait()
out = np.zeros((4096, 16), np.uint64)
raw = C.CDLL(capi.LIB_PATH)
raw.vo_debug_fe_stamps.argtypes = [C.c_void_p]
assert raw.vo_debug_fe_stamps(out.ctypes.data) == 0
t = out[:, :10].astype(np.int64)
d = np.diff(t, axis=1)
names = ["issue loads", "wait loads + gray + LDS write", "barrier 1", "pre-test + queue", "barrier 2",
         "score / blur row pass", "barrier 3", "NMS + emit", "blur column pass + outputs"]
ok = (t[:, 9] > 0)
print("blocks with stamps:", int(ok.sum()))
for i, n in enumerate(names):
    col = d[ok, i]
    col = col[(col >= 0) & (col < 10 ** 7)]
    print(f"{n:32s} median {np.median(col):8.0f}  p90 {np.percentile(col, 90):8.0f} cycles")
tot = t[ok, 9] - t[ok, 0]
print(f"{'start -> end':32s} median {np.median(tot):8.0f}  p90 {np.percentile(tot, 90):8.0f} cycles")

# ---- PnP kernel (one block per frame): thread 0 of the first 256 blocks
pn = np.zeros((256, 16), np.uint64)
raw.vo_debug_pnp_stamps.argtypes = [C.c_void_p]
assert raw.vo_debug_pnp_stamps(pn.ctypes.data) == 0
t = pn[:, :11].astype(np.int64)
d = np.diff(t, axis=1)
names = ["load points", "rng window + subsets (last round)", "EPnP hypotheses (last round)",
         "score + accept (last round)", "inlier mask + list", "planarity", "DLT sums", "12x12 singular vectors",
         "polar + Rodrigues", "Levenberg-Marquardt"]
print("PnP, RANSAC loop total (stamp 1 -> 4): median", np.median(t[:, 4] - t[:, 1]))
for i, n in enumerate(names):
    print(f"{n:36s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} cycles")
print(f"{'whole kernel':36s} median {np.median(t[:, 10] - t[:, 0]):9.0f}")
q = pn[:, 11:15].astype(np.int64)
for i, n in enumerate(["  EPnP: control points + barycentric", "  EPnP: 10x12 singular vectors (16 lanes)",
                       "  EPnP: betas, Gauss-Newton, R|t (3 lanes)"]):
    print(f"{n:44s} median {np.median(q[:, i + 1] - q[:, i]):9.0f} cycles")

# ---- F-RANSAC kernel (one block per frame pair): thread 0 of the first 256 blocks, last round of the loop
fr = np.zeros((256, 16), np.uint64)
raw.vo_debug_fr_stamps.argtypes = [C.c_void_p]
assert raw.vo_debug_fr_stamps(fr.ctypes.data) == 0
t = fr[:, :10].astype(np.int64)
ok = t[:, 9] > 0
t = t[ok]
names = ["load points", "(earlier rounds)", "rng window", "subset attempts", "chain (one lane)", "7-point solve",
         "score", "accept (one lane)", "final mask + compaction"]
d = np.diff(t, axis=1)
print("F-RANSAC blocks with stamps:", int(ok.sum()), " iterations run:", res_iters if "res_iters" in dir() else "")
for i, n in enumerate(names):
    print(f"{n:36s} median {np.median(d[:, i]):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f} cycles")
print(f"{'whole kernel':36s} median {np.median(t[:, 9] - t[:, 0]):9.0f}")
