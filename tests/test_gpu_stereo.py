"""Stereo front end: band-limited left/right matching with cross-check, disparity -> depth and the sparse depth
plane, against the oracle; then the pipeline's pose from stereo depth alone against the rendered motion."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

BASELINE_MM = 120.0


def _nan_equal(a, b):
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def _detect_pair(ctx, oracle, cam, seed, n_kp, f=0):
    left, right = synth.make_stereo_sequence(cam, f + 1, seed=seed, channels=4, baseline_mm=BASELINE_MM)
    out = []
    for s in (left, right):
        g = oracle.gray(s["frames"][f])
        out.append(ctx.detect_orb(g, n_kp, 1.2, 8, 20))
    return left, right, out[0], out[1]


@pytest.mark.parametrize("cam_name,n_kp,band,ratio", [("hd720", 2000, 2.0, 0.8), ("hd1080", 4000, 1.0, 0.7),
                                                      ("kitti", 1500, 3.0, 1.0)])
def test_stereo_depth_matches_oracle(ctx, oracle, cam_name, n_kp, band, ratio):
    cam = synth.CONFIG_CAMERAS[cam_name]
    left, right, (kl, dl), (kr, dr) = _detect_pair(ctx, oracle, cam, 3100 + n_kp, n_kp)
    size = (cam.width, cam.height)
    ref = oracle.stereo_match(kl, dl, kr, dr, cam.fx, BASELINE_MM, band, 1.0, 256.0, ratio, size=size)
    got = ctx.stereo_depth(kl, dl, kr, dr, cam.fx, BASELINE_MM, band, 1.0, 256.0, ratio, size=size)
    assert np.array_equal(got[0], ref[0])          # right index per left key point
    assert np.array_equal(got[1], ref[1])          # disparity (float32, exact)
    assert _nan_equal(got[2], ref[2])              # depth (float32, exact)
    assert _nan_equal(got[3], ref[3])              # sparse depth plane
    m = got[0] >= 0
    assert m.sum() > 0.4 * len(kl)
    # one-to-one, inside the band, positive disparity
    assert len(np.unique(got[0][m])) == m.sum()
    assert np.abs(kl["y"][m] - kr["y"][got[0][m]]).max() <= band and got[1][m].min() >= 1.0
    # against the rendered depth: disparity is quantised to the key points' pixel grid
    true = left["depth"][0][np.rint(kl["y"][m]).astype(int), np.rint(kl["x"][m]).astype(int)]
    rel = np.abs(got[2][m] - true) / true
    assert np.median(rel) < 0.04 and (rel > 0.25).mean() < 0.02


def test_stereo_depth_edge_cases(ctx, oracle):
    cam = synth.CONFIG_CAMERAS["vga"]
    left, right, (kl, dl), (kr, dr) = _detect_pair(ctx, oracle, cam, 77, 500)
    # no right key points: nothing matches, the plane is all NaN
    ridx, disp, depth, plane = ctx.stereo_depth(kl, dl, kr[:0], dr[:0], cam.fx, BASELINE_MM, size=(cam.width, cam.height))
    assert (ridx == -1).all() and (disp == -1).all() and np.isnan(depth).all() and np.isnan(plane).all()
    # no left key points
    ridx, disp, depth = ctx.stereo_depth(kl[:0], dl[:0], kr, dr, cam.fx, BASELINE_MM)
    assert len(ridx) == 0
    # a disparity window that excludes everything
    ridx, _, _ = ctx.stereo_depth(kl, dl, kr, dr, cam.fx, BASELINE_MM, min_disp=500.0, max_disp=600.0)
    assert (ridx == -1).all()
    # swapping the images leaves no positive disparities
    ref = oracle.stereo_match(kr, dr, kl, dl, cam.fx, BASELINE_MM)
    got = ctx.stereo_depth(kr, dr, kl, dl, cam.fx, BASELINE_MM)
    assert np.array_equal(got[0], ref[0]) and (got[0] >= 0).sum() < 0.05 * len(kr)
    with pytest.raises(Exception):
        ctx.stereo_depth(kl, dl, kr, dr, cam.fx, -1.0)


def test_pose_from_stereo_depth(ctx, oracle):
    """Depth planes from the stereo matcher instead of the rendered depth: the pipeline's pose (full detector,
    ratio test, F-RANSAC, PnP) against the oracle on the same planes and against the rendered motion."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["hd720"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    n_frames = 3
    scene = synth.Scene(5150, relief=2.0)
    left = synth.make_sequence(cam, n_frames, 5150, 4, scene=scene)
    right = synth.make_sequence(cam, n_frames, 5151, 4, scene=scene, baseline_mm=BASELINE_MM)
    planes = []
    for f in range(n_frames):
        kl, dl = ctx.detect_orb(oracle.gray(left["frames"][f]), 2000, 1.2, 8, 20)
        kr, dr = ctx.detect_orb(oracle.gray(right["frames"][f]), 2000, 1.2, 8, 20)
        planes.append(ctx.stereo_depth(kl, dl, kr, dr, cam.fx, BASELINE_MM, size=(cam.width, cam.height))[3])
    planes = np.stack(planes)
    ctx.batch_configure(cam.width, cam.height, 4, n_streams=1, frames_per_stream=n_frames, max_features=2000,
                        camera=K, orb_levels=8)
    ctx.batch_submit(left["frames"], planes)
    res = ctx.batch_wait()
    ref = OracleStream(K, 2000, 20, orb_levels=8, feature_cap=ctx.max_features)
    for f in range(n_frames):
        r = ref.process(left["frames"][f], planes[f])
        assert bool(res["pnp_success"][f]) == r["pnp_success"]
        if f == 0:
            continue
        assert r["pnp_success"] and np.array_equal(ctx.batch_inliers(f), r["inliers"])
        assert np.abs(res["rvec"][f] - r["rvec"]).max() < 1e-6                                   # rad
        assert np.abs(res["tvec"][f] - r["tvec"]).max() < 1e-6 * max(1.0, np.linalg.norm(r["tvec"]))
        R_t, t_t = synth.relative_pose(left["poses"][f - 1], left["poses"][f])
        # 33 mm of motion seen through depths that carry the 1-px disparity quantisation (about 3 % at 5 m)
        assert np.linalg.norm(res["tvec"][f] - t_t) < 12.0
        assert np.abs(oracle.rodrigues(res["rvec"][f]) - R_t).max() < 3e-3


@pytest.mark.parametrize("detector_levels", [8, 0])
def test_stereo_batch_pipeline(oracle, detector_levels):
    """vo_batch_submit_stereo: left and right sequences of two rigs per batch, depth from the on-device stereo
    matcher; every frame against the oracle run on the oracle's own stereo depth planes, over two batches (so the
    planes must be back to all-NaN in between)."""
    from oracle.pipeline import OracleStream
    from visual_odometry_amd import capi
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    S, F, NB, N = 2, 2, 2, 800
    rigs = []
    for s in range(S):
        scene = synth.Scene(6100 + s, relief=2.0)
        rigs.append((synth.make_sequence(cam, F * NB, 6100 + s, 3, scene=scene),
                     synth.make_sequence(cam, F * NB, 6200 + s, 3, scene=scene, baseline_mm=BASELINE_MM)))
    ctx = capi.Context(0, cam.width, cam.height, N + 64, 2 * S * F)
    try:
        ctx.batch_configure(cam.width, cam.height, 3, n_streams=S, frames_per_stream=F, max_features=N, camera=K,
                            orb_levels=detector_levels)
        refs = [OracleStream(K, N, 20, orb_levels=detector_levels, feature_cap=ctx.max_features) for _ in range(S)]
        det = (lambda g: oracle.orb_detect(g, N, 1.2, 8, 20)) if detector_levels else (lambda g: oracle.detect(g, N, 20))
        for nb in range(NB):
            fr = [nb * F + f for f in range(F)]
            left = np.stack([rigs[s][0]["frames"][i] for s in range(S) for i in fr])
            right = np.stack([rigs[s][1]["frames"][i] for s in range(S) for i in fr])
            ctx.batch_submit_stereo(left, right, BASELINE_MM, band=2.0, min_disp=1.0, max_disp=200.0, ratio=0.8)
            res = ctx.batch_wait()
            for s in range(S):
                for f in range(F):
                    b = s * F + f
                    kl, dl = det(oracle.gray(left[b]))
                    kr, dr = det(oracle.gray(right[b]))
                    kl, dl, kr, dr = kl[:ctx.max_features], dl[:ctx.max_features], kr[:ctx.max_features], dr[:ctx.max_features]
                    ridx, disp, depth, plane = oracle.stereo_match(kl, dl, kr, dr, cam.fx, BASELINE_MM, 2.0, 1.0, 200.0,
                                                                   0.8, size=(cam.width, cam.height))
                    g_ridx, g_disp, g_depth = ctx.batch_stereo(b)
                    assert np.array_equal(g_ridx, ridx) and np.array_equal(g_disp, disp) and _nan_equal(g_depth, depth)
                    r = refs[s].process(left[b], plane)
                    k, d = ctx.batch_features(b)
                    assert np.array_equal(k.view(np.uint8), r["kps"].view(np.uint8)) and np.array_equal(d, r["desc"])
                    m, p3 = ctx.batch_matches(b, 2)
                    assert np.array_equal(m, r["valid"]) and np.array_equal(p3, r["pts3"])
                    assert bool(res["pnp_success"][b]) == r["pnp_success"]
                    if r["pnp_success"]:
                        assert np.array_equal(ctx.batch_inliers(b), r["inliers"])
                        assert np.abs(res["rvec"][b] - r["rvec"]).max() < 1e-6                     # rad
                        assert np.abs(res["tvec"][b] - r["tvec"]).max() < 1e-6 * max(1.0, np.linalg.norm(r["tvec"]))
            assert res["pnp_success"][1] and res["pnp_success"][F + 1]
        # too small a context is refused
        small = capi.Context(0, cam.width, cam.height, N + 64, S * F)
        try:
            small.batch_configure(cam.width, cam.height, 3, n_streams=S, frames_per_stream=F, max_features=N, camera=K)
            with pytest.raises(Exception):
                small.batch_submit_stereo(left, right, BASELINE_MM)
        finally:
            small.close()
    finally:
        ctx.close()


def test_stereo_batch_after_reconfigure_to_larger_batch(oracle):
    """A stereo batch of two frames, then the same context reconfigured to four frames of another size: the depth
    planes of the new layout must all start as NaN (no stale depth from the first configuration), so every frame
    still matches the oracle run on the oracle's own sparse planes."""
    from oracle.pipeline import OracleStream
    from visual_odometry_amd import capi
    big, small = synth.CONFIG_CAMERAS["vga"], synth.Camera(320, 240, 260.0, 260.0, 159.5, 119.5)
    N = 500
    ctx = capi.Context(0, big.width, big.height, N, 8)
    try:
        for cam, B, seed in ((small, 2, 6400), (big, 4, 6500)):
            K = (cam.fx, cam.fy, cam.cx, cam.cy)
            scene = synth.Scene(seed, relief=2.0)
            left = synth.make_sequence(cam, B, seed, 3, scene=scene)["frames"]
            right = synth.make_sequence(cam, B, seed + 1, 3, scene=scene, baseline_mm=BASELINE_MM)["frames"]
            ctx.batch_configure(cam.width, cam.height, 3, n_streams=1, frames_per_stream=B, max_features=N, camera=K)
            ctx.batch_submit_stereo(left, right, BASELINE_MM, band=2.0, min_disp=1.0, max_disp=200.0, ratio=0.8)
            res = ctx.batch_wait()
            ref = OracleStream(K, N, 20)
            for b in range(B):
                kl, dl = oracle.detect(oracle.gray(left[b]), N, 20)
                kr, dr = oracle.detect(oracle.gray(right[b]), N, 20)
                plane = oracle.stereo_match(kl, dl, kr, dr, cam.fx, BASELINE_MM, 2.0, 1.0, 200.0, 0.8,
                                            size=(cam.width, cam.height))[3]
                r = ref.process(left[b], plane)
                m, p3 = ctx.batch_matches(b, 2)
                assert np.array_equal(m, r["valid"]) and np.array_equal(p3, r["pts3"]), (cam.width, b)
                assert res["n_3d"][b] == len(r["pts3"])
    finally:
        ctx.close()


def test_stereo_depth_plane_rounds_like_the_backprojection(ctx, oracle):
    """Key points of upper pyramid levels have x.5 coordinates when the scale factor is 1.5 (odd level coordinates times 1.5): the sparse depth
    plane must be written at round-half-away (std::round), where backprojectAndFilter samples it, so every matched
    left key point finds its own depth again."""
    cam = synth.CONFIG_CAMERAS["vga"]
    scene = synth.Scene(6600, relief=2.0)
    left = synth.make_sequence(cam, 1, 6600, 1, scene=scene)["frames"][0]
    right = synth.make_sequence(cam, 1, 6601, 1, scene=scene, baseline_mm=BASELINE_MM)["frames"][0]
    kl, dl = ctx.detect_orb(left, 1000, 1.5, 3, 20)
    kr, dr = ctx.detect_orb(right, 1000, 1.5, 3, 20)
    assert (np.abs(kl["x"] - np.floor(kl["x"]) - 0.5) < 1e-6).sum() > 20     # half-pixel coordinates exist
    ref = oracle.stereo_match(kl, dl, kr, dr, cam.fx, BASELINE_MM, size=(cam.width, cam.height))
    got = ctx.stereo_depth(kl, dl, kr, dr, cam.fx, BASELINE_MM, size=(cam.width, cam.height))
    assert np.array_equal(got[0], ref[0]) and _nan_equal(got[3], ref[3])
    plane, matched = got[3], np.flatnonzero(got[0] >= 0)
    u = np.floor(np.abs(kl["x"][matched]) + 0.5).astype(int)                 # std::round for non-negative values
    v = np.floor(np.abs(kl["y"][matched]) + 0.5).astype(int)
    assert len(matched) > 100 and np.isfinite(plane[v, u]).all()
    # several key points may share a pixel: the nearest depth wins, so the sampled depth is never farther
    assert (plane[v, u] <= got[2][matched]).all()
