"""Times the essential-matrix stress configuration (5000 correspondences, 4096 five-point hypotheses)\non the GPU (operator call including the host copies) and on the CPU oracle."""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from visual_odometry_amd import capi
from oracle import oracle as O
import test_gpu_essential as T
ctx = capi.Context(0, 1280, 720, 2000, 1)
p1, p2, Et, io = T._scene(5, 5000, 2000)
s = T._sample_table(O, 5000, 4096)
for _ in range(3): ctx.essential_hypotheses(p1, p2, 700., 700., 640., 360., s, 1.0)
t0 = time.perf_counter()
for _ in range(10): ctx.essential_hypotheses(p1, p2, 700., 700., 640., 360., s, 1.0)
print("GPU call (incl. copies) ms", (time.perf_counter() - t0) * 100)
t0 = time.perf_counter(); O.e_hypotheses(T._norm(p1), T._norm(p2), s, 1 / 700.); print("oracle ms", (time.perf_counter() - t0) * 1e3)
