"""Parity of the pose operator and of the batched pipeline (many frames and
streams per call) against the CPU oracle chain."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu

R_TOL = 1e-6  # rad: tolerance asked of the recovered rotation
T_TOL = 1e-6  # relative to |t|


def _assert_pose_close(rvec, tvec, rvec_ref, tvec_ref):
    assert np.abs(rvec - rvec_ref).max() < R_TOL
    assert np.abs(tvec - tvec_ref).max() < T_TOL * max(1.0, np.linalg.norm(tvec_ref))


def _scene_points(rng, n, cam, outlier_frac, noise):
    X = np.stack([rng.uniform(-2000, 2000, n), rng.uniform(-1200, 1200, n), rng.uniform(2000, 6000, n)], 1)
    rv = np.array([0.012, -0.02, 0.015])
    t = np.array([30.0, -12.0, 55.0])
    th = np.linalg.norm(rv)
    k = rv / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    Y = X @ R.T + t
    uv = np.stack([Y[:, 0] / Y[:, 2] * cam.fx + cam.cx, Y[:, 1] / Y[:, 2] * cam.fy + cam.cy], 1)
    uv += rng.normal(0, noise, uv.shape)
    bad = rng.random(n) < outlier_frac
    uv[bad] = rng.uniform([0, 0], [cam.width, cam.height], (int(bad.sum()), 2))
    return X.astype(np.float32), uv.astype(np.float32), rv, t


@pytest.mark.parametrize("n,outliers,noise", [(400, 0.3, 0.5), (60, 0.0, 0.2), (1500, 0.5, 1.0), (6, 0.0, 0.0),
                                               (5, 0.0, 0.0), (4, 0.0, 0.0), (3, 0.0, 0.0)])
def test_solve_pnp_ransac(ctx, oracle, n, outliers, noise):
    cam = synth.CONFIG_CAMERAS["hd720"]
    rng = np.random.default_rng(n)
    X, uv, rv, t = _scene_points(rng, n, cam, outliers, noise)
    ok_r, rv_r, t_r, inl_r = oracle.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy)
    ok, rv_g, t_g, inl = ctx.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy)
    assert ok == ok_r
    assert np.array_equal(inl, inl_r)
    if ok:
        _assert_pose_close(rv_g, t_g, rv_r, t_r)
        if noise <= 1.0 and n >= 6:
            assert np.abs(rv_g - rv).max() < 5e-3 and np.abs(t_g - t).max() < 25.0  # near ground truth


@pytest.mark.parametrize("n,outliers,iters", [
    (300, 0.45, 100),   # bound stays near 100 after the first round: every remaining hypothesis goes to the item list
    (300, 0.6, 100),    # the bound never drops: all 100 iterations run
    (120, 0.35, 100),   # a dozen or two iterations: a short item list
    (300, 0.5, 1000),   # the bound starts above what the item list holds: rounds go on in the block until it fits
    (300, 0.2, 30),     # at most 22 iterations left after the first round
    (300, 0.3, 18), (300, 0.3, 19),   # the smallest budgets on either side of the hand-over rule
    (9, 0.3, 100), (7, 0.15, 100),    # tiny point sets: every draw repeats indices, RNG windows end early
])
def test_solve_pnp_ransac_many_hypotheses(ctx, oracle, n, outliers, iters):
    """Frames that need more than two rounds of hypotheses leave their block and are finished by the follow-up
    launches (work items over the whole chip, then a sequential replay): iteration count, inlier list and pose must
    be those of the sequential algorithm."""
    cam = synth.CONFIG_CAMERAS["hd720"]
    for seed in range(4):
        rng = np.random.default_rng(1000 * n + seed)
        X, uv, rv, t = _scene_points(rng, n, cam, outliers, 0.7)
        ok_r, rv_r, t_r, inl_r = oracle.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy, iters=iters)
        ok, rv_g, t_g, inl = ctx.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy, iters=iters)
        assert ok == ok_r
        assert np.array_equal(inl, inl_r)
        if ok:
            _assert_pose_close(rv_g, t_g, rv_r, t_r)


def _planar_points(rng, n, cam, plane, outlier_frac, noise, jitter=0.0):
    """Points on a plane z = plane(x, y) (+- jitter mm: measured depth is never exactly flat) seen from a known pose."""
    x, y = rng.uniform(-1500, 1500, n), rng.uniform(-900, 900, n)
    X = np.stack([x, y, plane(x, y) + rng.normal(0, 1.0, n) * jitter], 1)
    rv = np.array([0.012, -0.02, 0.015])
    t = np.array([30.0, -12.0, 55.0])
    th = np.linalg.norm(rv)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    Y = X @ R.T + t
    uv = np.stack([Y[:, 0] / Y[:, 2] * cam.fx + cam.cx, Y[:, 1] / Y[:, 2] * cam.fy + cam.cy], 1)
    uv += rng.normal(0, noise, uv.shape)
    bad = rng.random(n) < outlier_frac
    uv[bad] = rng.uniform([0, 0], [cam.width, cam.height], (int(bad.sum()), 2))
    return X.astype(np.float32), uv.astype(np.float32), rv, t


PLANES = {
    "wall": lambda x, y: 3000 + 0.5 * x - 0.2 * y,
    "floor": lambda x, y: 2500 + 1.5 * y,
    "facing": lambda x, y: 3500 + 0 * x,
}


@pytest.mark.parametrize("plane,n,outliers,noise,jitter", [
    ("wall", 400, 0.3, 0.5, 0.0), ("wall", 50, 0.0, 0.2, 0.0), ("wall", 7, 0.0, 0.1, 0.0),
    ("floor", 1500, 0.4, 1.0, 0.0), ("floor", 6, 0.0, 0.1, 0.0),
    ("facing", 300, 0.2, 0.5, 0.05), ("facing", 40, 0.0, 0.3, 0.3),
])
def test_solve_pnp_ransac_planar_points(ctx, oracle, plane, n, outliers, noise, jitter):
    """A planar inlier set (a wall, the floor): solvePnP(ITERATIVE) starts from a homography instead of the DLT
    (cvFindExtrinsicCameraParams2's planar branch, W[2]/W[1] < 1e-3) and succeeds."""
    cam = synth.CONFIG_CAMERAS["hd720"]
    rng = np.random.default_rng(n)
    X, uv, rv, t = _planar_points(rng, n, cam, PLANES[plane], outliers, noise, jitter)
    ok_r, rv_r, t_r, inl_r = oracle.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy)
    ok, rv_g, t_g, inl = ctx.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy)
    assert ok_r and ok
    assert np.array_equal(inl, inl_r) and len(inl) >= 0.5 * n
    _assert_pose_close(rv_g, t_g, rv_r, t_r)
    if n >= 40:
        assert np.abs(rv_g - rv).max() < 5e-3 and np.abs(t_g - t).max() < 25.0  # near ground truth


def test_solve_pnp_ransac_four_points_p3p(ctx, oracle):
    """Exactly four points: one P3P solve (solutions from the first three points, the fourth picks), no RANSAC."""
    cam = synth.CONFIG_CAMERAS["hd720"]
    rng = np.random.default_rng(4)
    n_ok = 0
    for trial in range(40):
        X, uv, rv, t = _scene_points(rng, 4, cam, 0.0, 0.0)
        ok_r, rv_r, t_r, inl_r = oracle.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy)
        ok, rv_g, t_g, inl = ctx.solve_pnp_ransac(X, uv, cam.fx, cam.fy, cam.cx, cam.cy)
        assert ok == ok_r and np.array_equal(inl, inl_r)
        if ok:
            n_ok += 1
            assert list(inl) == [0, 1, 2, 3]
            _assert_pose_close(rv_g, t_g, rv_r, t_r)
            # noise-free input: the true pose is among the solutions and the fourth point selects it
            assert np.abs(rv_g - rv).max() < 2e-3 and np.abs(t_g - t).max() < 10.0
    assert n_ok >= 38
    # three points or fewer: failure, as cv::solvePnPRansac reports
    ok, _, _, inl = ctx.solve_pnp_ransac(X[:3], uv[:3], cam.fx, cam.fy, cam.cx, cam.cy)
    assert not ok and len(inl) == 0


def test_solve_pnp_ransac_five_inliers_keep_minimal_sample_pose(ctx, oracle):
    """Five non-planar inliers cannot seed the DLT (six needed); OpenCV keeps the pose of the minimal sample and
    reports success."""
    cam = synth.CONFIG_CAMERAS["hd720"]
    rng = np.random.default_rng(99)
    X, uv, rv, t = _scene_points(rng, 5, cam, 0.0, 0.0)
    Xo = np.concatenate([X, rng.uniform([-2000, -1200, 2000], [2000, 1200, 6000], (6, 3)).astype(np.float32)])
    uvo = np.concatenate([uv, rng.uniform([0, 0], [cam.width, cam.height], (6, 2)).astype(np.float32)])
    ok_r, rv_r, t_r, inl_r = oracle.solve_pnp_ransac(Xo, uvo, cam.fx, cam.fy, cam.cx, cam.cy)
    ok, rv_g, t_g, inl = ctx.solve_pnp_ransac(Xo, uvo, cam.fx, cam.fy, cam.cx, cam.cy)
    assert ok == ok_r and np.array_equal(inl, inl_r)
    if ok_r and len(inl_r) == 5:
        _assert_pose_close(rv_g, t_g, rv_r, t_r)


def test_batch_pipeline_flat_wall(ctx, oracle):
    """A camera facing a flat wall (relief 0): every frame's PnP inlier set is planar, the pose comes out through
    the homography initialisation, bit-exact match lists and poses within tolerance of the oracle."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    F = 4
    for levels in (0, 8):
        seq = synth.make_sequence(cam, F, seed=31, channels=4, scene=synth.Scene(31, relief=0.0))
        ref = OracleStream(K, 1000, 20, orb_levels=levels, feature_cap=ctx.max_features)
        ctx.batch_configure(cam.width, cam.height, 4, n_streams=1, frames_per_stream=F, max_features=1000, camera=K,
                            orb_levels=levels)
        ctx.batch_submit(seq["frames"], seq["depth"])
        res = ctx.batch_wait()
        for f in range(F):
            r = ref.process(seq["frames"][f], seq["depth"][f])
            _compare_frame(ctx, res, f, r, pose=True)
            if f:
                assert r["pnp_success"] and res["pnp_success"][f]
                _, t_true = synth.relative_pose(seq["poses"][f - 1], seq["poses"][f])
                assert np.linalg.norm(res["tvec"][f] - t_true) < 15.0  # mm, on a 33 mm step (sideways motion and rotation trade off on a plane)


def _compare_frame(ctx, res, b, ref, pose):
    k, d = ctx.batch_features(b)
    assert np.array_equal(k, ref["kps"]) and np.array_equal(d, ref["desc"])
    assert res["n_keypoints"][b] == len(ref["kps"])
    assert res["n_ratio"][b] == ref["n_ratio"]
    m, _ = ctx.batch_matches(b, 0)
    assert np.array_equal(m, ref["matches"])
    tr, _ = ctx.batch_matches(b, 1)
    assert np.array_equal(tr, ref["tracked"])
    if pose:
        va, p3 = ctx.batch_matches(b, 2)
        assert np.array_equal(va, ref["valid"]) and np.array_equal(p3, ref["pts3"])
        assert bool(res["pnp_success"][b]) == ref["pnp_success"]
        if ref["pnp_success"]:
            assert np.array_equal(ctx.batch_inliers(b), ref["inliers"])
            _assert_pose_close(res["rvec"][b], res["tvec"][b], ref["rvec"], ref["tvec"])


def test_batch_pipeline_two_streams(ctx, oracle):
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    S, F, NB = 2, 3, 2
    seqs = [synth.make_sequence(cam, F * NB, seed=1001 + s, channels=4) for s in range(S)]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    refs = [OracleStream(K, 1000, 20) for _ in range(S)]
    ctx.batch_configure(cam.width, cam.height, 4, n_streams=S, frames_per_stream=F, max_features=1000,
                        fast_threshold=20, camera=K)
    n_pose = 0
    for nb in range(NB):
        frames = np.stack([seqs[s]["frames"][nb * F + f] for s in range(S) for f in range(F)])
        depth = np.stack([seqs[s]["depth"][nb * F + f] for s in range(S) for f in range(F)])
        ctx.batch_submit(frames, depth)
        res = ctx.batch_wait()
        for s in range(S):
            for f in range(F):
                ref = refs[s].process(seqs[s]["frames"][nb * F + f], seqs[s]["depth"][nb * F + f])
                _compare_frame(ctx, res, s * F + f, ref, pose=True)
                n_pose += int(ref["pnp_success"])
    assert n_pose == S * (F * NB - 1)


def test_batch_pipeline_multiview_filter_engages(ctx, oracle):
    """After three matched frames the track-age filter (age >= 3) starts to drop matches."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    seq = synth.make_sequence(cam, 6, seed=2024, channels=1)
    ref = OracleStream(None, 1000, 20)
    ctx.batch_configure(cam.width, cam.height, 1, n_streams=1, frames_per_stream=1, max_features=1000)
    dropped = 0
    for i in range(6):
        ctx.batch_submit(seq["frames"][i:i + 1])
        res = ctx.batch_wait()
        r = ref.process(seq["frames"][i])
        _compare_frame(ctx, res, 0, r, pose=False)
        dropped += len(r["matches"]) - len(r["tracked"])
    assert dropped > 0


def test_batch_pipeline_long_stream_launch(oracle):
    """66 frames of one stream in one launch (the track-age kernel stages the frame scalars in groups of 64),
    twice, so that ages cross both the group and the launch boundary; second stream offset by one frame."""
    from oracle.pipeline import OracleStream
    from visual_odometry_amd import capi
    cam = synth.CONFIG_CAMERAS["vga"]
    S, F, NB, D = 2, 66, 2, 5
    seq = synth.make_sequence(cam, D, seed=31, channels=1)
    walk = list(range(D)) + list(range(D - 2, 0, -1))  # 0 1 2 3 4 3 2 1 | 0 ...
    order = [[walk[(s + k) % len(walk)] for k in range(F * NB)] for s in range(S)]
    refs = [OracleStream(None, 500, 20) for _ in range(S)]
    c = capi.Context(device=0, max_width=cam.width, max_height=cam.height, max_features=500, max_batch=S * F)
    try:
        c.batch_configure(cam.width, cam.height, 1, n_streams=S, frames_per_stream=F, max_features=500)
        dropped = 0
        for nb in range(NB):
            frames = np.stack([seq["frames"][order[s][nb * F + f]] for s in range(S) for f in range(F)])
            c.batch_submit(frames)
            res = c.batch_wait()
            for s in range(S):
                for f in range(F):
                    r = refs[s].process(seq["frames"][order[s][nb * F + f]])
                    _compare_frame(c, res, s * F + f, r, pose=False)
                    dropped += len(r["matches"]) - len(r["tracked"])
        assert dropped > 0
    finally:
        c.close()


def test_batch_reset_makes_first_frame(ctx, oracle):
    cam = synth.CONFIG_CAMERAS["vga"]
    seq = synth.make_sequence(cam, 2, seed=77, channels=3)
    ctx.batch_configure(cam.width, cam.height, 3, 1, 1, max_features=500)
    ctx.batch_submit(seq["frames"][0:1]); r0 = ctx.batch_wait()
    ctx.batch_submit(seq["frames"][1:2]); r1 = ctx.batch_wait()
    assert r0["n_matches"][0] == 0 and r1["n_matches"][0] > 50
    ctx.batch_reset()
    ctx.batch_submit(seq["frames"][1:2]); r2 = ctx.batch_wait()
    assert r2["n_matches"][0] == 0 and r2["n_keypoints"][0] == r1["n_keypoints"][0]


def test_batch_pipeline_cross_check(ctx, oracle):
    """Mutual-nearest filtering between the ratio test and RANSAC, whole pipeline."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    seq = synth.make_sequence(cam, 4, seed=4242, channels=3)
    ref = OracleStream(K, 1000, 20, cross_check=True)
    ctx.batch_configure(cam.width, cam.height, 3, n_streams=1, frames_per_stream=4, max_features=1000,
                        camera=K, cross_check=True)
    ctx.batch_submit(seq["frames"], seq["depth"])
    res = ctx.batch_wait()
    for f in range(4):
        _compare_frame(ctx, res, f, ref.process(seq["frames"][f], seq["depth"][f]), pose=True)


def test_batch_pipeline_full_orb(ctx, oracle):
    """The batched pipeline with the full ORB detector (8-level pyramid) in front."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    S, F = 2, 2
    seqs = [synth.make_sequence(cam, 2 * F, seed=8800 + s, channels=4) for s in range(S)]
    refs = [OracleStream(K, 1500, 20, orb_levels=8, feature_cap=ctx.max_features) for _ in range(S)]
    ctx.batch_configure(cam.width, cam.height, 4, n_streams=S, frames_per_stream=F, max_features=1500,
                        camera=K, orb_levels=8)
    for nb in range(2):
        frames = np.stack([seqs[s]["frames"][nb * F + f] for s in range(S) for f in range(F)])
        depth = np.stack([seqs[s]["depth"][nb * F + f] for s in range(S) for f in range(F)])
        ctx.batch_submit(frames, depth)
        res = ctx.batch_wait()
        for s in range(S):
            for f in range(F):
                ref = refs[s].process(seqs[s]["frames"][nb * F + f], seqs[s]["depth"][nb * F + f])
                _compare_frame(ctx, res, s * F + f, ref, pose=True)
                assert len(np.unique(ref["kps"]["octave"])) == 8
                if nb or f:
                    assert ref["pnp_success"] and len(ref["matches"]) > 100


@pytest.mark.parametrize("shape,levels", [((200, 120), 0), ((30, 300), 0), ((260, 126), 3), ((330, 128), 4)])
def test_batch_of_frames_one_tile_wide_or_high(ctx, oracle, shape, levels):
    """Several frames per launch whose tile grid has a single column (width <= 128) or a single row (height <= 32):
    the block -> (frame, tile) maps divide by 1 there.  Features of every frame against the oracle."""
    h, w = shape
    big = synth.make_sequence(synth.CONFIG_CAMERAS["hd720"], 1, seed=h * 7 + w, channels=1)["frames"][0]
    S = 6
    frames = np.stack([np.ascontiguousarray(big[20 + 37 * s:20 + 37 * s + h, 60 + 101 * s:60 + 101 * s + w])
                       for s in range(S)])
    depth = np.full((S, h, w), 3000.0, np.float32)
    ctx.batch_configure(w, h, 1, n_streams=S, frames_per_stream=1, max_features=400, fast_threshold=20,
                        camera=(500.0, 500.0, w / 2, h / 2), orb_levels=levels)
    ctx.batch_submit(frames, depth)
    ctx.batch_wait()
    total = 0
    for s in range(S):
        if levels:
            k_ref, d_ref = oracle.orb_detect(frames[s], 400, 1.2, levels, 20)
            k_ref, d_ref = k_ref[:ctx.max_features], d_ref[:ctx.max_features]
        else:
            k_ref, d_ref = oracle.detect(frames[s], 400, 20)
        k, d = ctx.batch_features(s)
        assert np.array_equal(k, k_ref) and np.array_equal(d, d_ref), s
        total += len(k)
    if h > 70 and w > 70:
        assert total > 50  # the frames are large enough to hold key points inside the 31-px border


@pytest.mark.parametrize("levels", [0, 8])
def test_batch_is_reproducible_run_to_run(ctx, levels):
    """The same batch twice (streams reset in between): every output identical bit for bit, poses included --
    no result depends on the order in which blocks or atomics happen to run."""
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    S, F = 2, 3  # the shared context holds eight frames per batch
    seqs = [synth.make_sequence(cam, F, seed=7300 + s, channels=3, scene=synth.Scene(7300 + s, relief=2.0))
            for s in range(S)]
    frames = np.stack([seqs[s]["frames"][f] for s in range(S) for f in range(F)])
    depth = np.stack([seqs[s]["depth"][f] for s in range(S) for f in range(F)])
    ctx.batch_configure(cam.width, cam.height, 3, n_streams=S, frames_per_stream=F, max_features=1200, camera=K,
                        orb_levels=levels, cross_check=True)
    runs = []
    for _ in range(2):
        ctx.batch_reset()
        ctx.batch_submit(frames, depth)
        res = ctx.batch_wait()
        per_frame = []
        for b in range(S * F):
            k, d = ctx.batch_features(b)
            m, p3 = ctx.batch_matches(b, 2)
            per_frame.append((k.tobytes(), d.tobytes(), m.tobytes(), p3.tobytes(), ctx.batch_inliers(b).tobytes()))
        runs.append((res.tobytes(), per_frame))
    assert runs[0][0] == runs[1][0]
    assert runs[0][1] == runs[1][1]
    assert res["pnp_success"].reshape(S, F)[:, 1:].all()


@pytest.mark.parametrize("levels", [0, 8])
def test_batch_with_degenerate_streams(oracle, levels):
    """One launch whose streams run into every empty case of the chain side by side with an ordinary stream: blank
    frames (no key points at all), a blank frame between textured ones (nothing to match against, then nothing to
    match), depth that is zero / NaN / out of range everywhere (matches but no 3-D points, no pose), and a stream
    whose second frame shows something else entirely (a handful of matches at most)."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    F = 3
    a = synth.make_sequence(cam, F, seed=6100, channels=4)
    b = synth.make_sequence(cam, F, seed=6200, channels=4)
    blank = np.full_like(a["frames"][0], 90)
    bad_depth = a["depth"].copy()
    bad_depth[0][:] = 0.0
    bad_depth[1][:] = np.nan
    bad_depth[2][:] = 30000.0
    streams = [
        (a["frames"], a["depth"]),                                                  # ordinary
        (np.stack([blank] * F), a["depth"]),                                         # never a key point
        (np.stack([a["frames"][0], blank, a["frames"][2]]), a["depth"]),             # a blank frame in the middle
        (a["frames"], bad_depth),                                                    # no usable depth
        (np.stack([a["frames"][0], b["frames"][1], a["frames"][2]]), a["depth"]),    # unrelated middle frame
    ]
    S = len(streams)
    from visual_odometry_amd import capi
    ctx = capi.Context(device=0, max_width=cam.width, max_height=cam.height, max_features=864, max_batch=S * F)
    try:
        refs = [OracleStream(K, 800, 20, orb_levels=levels, feature_cap=ctx.max_features) for _ in range(S)]
        ctx.batch_configure(cam.width, cam.height, 4, n_streams=S, frames_per_stream=F, max_features=800,
                            fast_threshold=20, camera=K, orb_levels=levels)
        frames = np.stack([streams[s][0][f] for s in range(S) for f in range(F)])
        depth = np.stack([streams[s][1][f] for s in range(S) for f in range(F)])
        ctx.batch_submit(frames, depth)
        res = ctx.batch_wait()
        for s in range(S):
            for f in range(F):
                ref = refs[s].process(streams[s][0][f], streams[s][1][f])
                _compare_frame(ctx, res, s * F + f, ref, pose=True)
    finally:
        ctx.close()
    r = res.reshape(S, F)
    assert r["pnp_success"][0, 1:].all()
    assert (r["n_keypoints"][1] == 0).all() and (r["n_matches"][1] == 0).all() and not r["pnp_success"][1].any()
    assert r["n_keypoints"][2, 1] == 0 and r["n_matches"][2, 1] == 0 and r["n_matches"][2, 2] == 0
    assert (r["n_matches"][3, 1:] > 50).all() and not r["pnp_success"][3].any()
    assert r["n_matches"][4, 1] < 30
