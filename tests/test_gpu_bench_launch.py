"""Parity at the launch the benchmark measures: HD720 BGRA + depth, full ORB detector with 8 pyramid levels,
2000 key points, 32 independent streams x 16 frames = 512 frames in ONE launch, two launches in a row (the
second continues every stream from the device-resident state of the first).  Every stream has its own scene.
Streams 0, 15 and 31 are replayed frame by frame on the CPU oracle and compared: features, descriptors and all
match lists bit for bit, poses to the tolerance below.  Batch entry 511 (stream 31, frame 15) is the last byte
range of every per-batch buffer, 12 % under 2^31 bytes of input."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

R_TOL = 1e-6  # rad
T_TOL = 1e-6  # relative to |t| (mm)


@pytest.mark.timeout(600)
def test_bench_launch_512_frames_distinct_streams(oracle):
    from oracle.pipeline import OracleStream
    from visual_odometry_amd import capi
    cam = synth.CONFIG_CAMERAS["hd720"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    S, F, N = 32, 16, 2000
    cap = N + 64
    frames, depth, _ = synth.make_pingpong_batch(cam, S, F, seed=2000, channels=4, relief=2.0, workers=16)
    assert frames.shape == (S * F, cam.height, cam.width, 4) and frames.nbytes > 1.8e9
    # the streams really differ
    assert not np.array_equal(frames[0], frames[15 * F]) and not np.array_equal(frames[0], frames[31 * F])
    check = (0, 15, 31)
    refs = {s: OracleStream(K, N, 20, orb_levels=8, feature_cap=cap) for s in check}
    c = capi.Context(device=0, max_width=cam.width, max_height=cam.height, max_features=cap, max_batch=S * F)
    try:
        c.batch_configure(cam.width, cam.height, 4, n_streams=S, frames_per_stream=F, max_features=N,
                          fast_threshold=20, camera=K, orb_levels=8)
        n_pose = 0
        for launch in range(2):
            c.batch_submit(frames, depth)
            res = c.batch_wait()
            assert (res["n_keypoints"] > 1500).all()
            # the first frame of a stream has nothing to match against -- in the first launch only
            first = res["n_matches"].reshape(S, F)[:, 0]
            assert (first == 0).all() if launch == 0 else (first > 100).all()
            for s in check:
                # the whole first launch is replayed (the oracle's state needs every frame); of the second launch
                # the carry-over frame (matched against frame 15 of launch 1 on the device) and its successor
                for f in range(F if launch == 0 else 2):
                    b = s * F + f
                    ref = refs[s].process(frames[b], depth[b])
                    k, d = c.batch_features(b)
                    assert np.array_equal(k, ref["kps"]) and np.array_equal(d, ref["desc"]), (launch, s, f)
                    assert res["n_ratio"][b] == ref["n_ratio"]
                    for which, name in ((0, "matches"), (1, "tracked"), (2, "valid")):
                        m, p3 = c.batch_matches(b, which)
                        assert np.array_equal(m, ref[name]), (launch, s, f, name)
                    assert np.array_equal(p3, ref["pts3"])
                    assert bool(res["pnp_success"][b]) == ref["pnp_success"]
                    if ref["pnp_success"]:
                        n_pose += 1
                        assert np.array_equal(c.batch_inliers(b), ref["inliers"])
                        assert np.abs(res["rvec"][b] - ref["rvec"]).max() < R_TOL
                        assert np.abs(res["tvec"][b] - ref["tvec"]).max() < T_TOL * max(1.0, np.linalg.norm(ref["tvec"]))
        assert n_pose == len(check) * (F - 1 + 2)
    finally:
        c.close()
