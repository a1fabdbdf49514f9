"""Latency of one HD720 BGRA + float32 depth frame from ordinary host memory through the ctypes binding
(submit + wait, median of 300), for comparing builds of the library: VO_LIB_PATH=<libvo_hip.so> python3 tools/lat_py.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_odometry_amd import capi, synth
cam = synth.CONFIG_CAMERAS["hd720"]
K = (cam.fx, cam.fy, cam.cx, cam.cy)
f, d, _ = synth.make_pingpong_batch(cam, 1, 8, 2000, channels=4, relief=2.0)
c = capi.Context(0, cam.width, cam.height, 1064, 1)
c.batch_configure(cam.width, cam.height, 4, 1, 1, max_features=1000, camera=K, orb_levels=0)
c.batch_set_packed_output(True)
ts = []
for it in range(340):
    k = it % 8
    t0 = time.perf_counter(); c.batch_submit(f[k:k + 1], d[k:k + 1]); c.batch_wait(); t1 = time.perf_counter()
    if it >= 40: ts.append(t1 - t0)
print(os.environ.get("VO_LIB_PATH", "default"), "median ms", round(float(np.median(ts)) * 1e3, 4), "p90", round(float(np.percentile(ts, 90)) * 1e3, 4))
