import numpy as np, sys, time
sys.path.insert(0, ".")
from visual_odometry_amd import capi, synth
cam = synth.CONFIG_CAMERAS["hd720"]
K = (cam.fx, cam.fy, cam.cx, cam.cy)
f, d, _ = synth.make_pingpong_batch(cam, 1, 8, 2000, channels=4, relief=2.0)
for levels, nf in ((0, 1000), (8, 2000)):
    c = capi.Context(0, cam.width, cam.height, nf + 64, 1)
    c.batch_configure(cam.width, cam.height, 4, 1, 1, max_features=nf, camera=K, orb_levels=levels)
    ts, tw = [], []
    for it in range(200):
        k = it % 8
        t0 = time.perf_counter(); c.batch_submit(f[k:k+1], d[k:k+1]); t1 = time.perf_counter(); c.batch_wait(); t2 = time.perf_counter()
        if it >= 20: ts.append(t1 - t0); tw.append(t2 - t0)
    print("levels", levels, "submit returns after (median us)", round(np.median(ts) * 1e6, 1), "wait returns after", round(np.median(tw) * 1e6, 1))
    c.close()
