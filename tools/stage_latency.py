import numpy as np, sys
sys.path.insert(0, "/root/repo")
from visual_odometry_amd import capi, synth
cam = synth.CONFIG_CAMERAS["hd720"]
K = (cam.fx, cam.fy, cam.cx, cam.cy)
f, d, _ = synth.make_pingpong_batch(cam, 1, 8, 2000, channels=4, relief=2.0)
for levels, nf in ((0, 1000), (8, 2000)):
    c = capi.Context(0, cam.width, cam.height, nf + 64, 1)
    c.batch_configure(cam.width, cam.height, 4, 1, 1, max_features=nf, camera=K, orb_levels=levels)
    c.batch_set_timing(True)
    acc = {}
    n = 0
    for it in range(40):
        k = it % 8
        c.batch_submit(f[k:k+1], d[k:k+1]); c.batch_wait()
        if it >= 8:
            n += 1
            for a, b in c.batch_timing().items(): acc[a] = acc.get(a, 0) + b
    print(levels, {a: round(b / n * 1e3, 1) for a, b in acc.items()}, "sum us", round(sum(acc.values()) / n * 1e3, 1))
    c.close()
