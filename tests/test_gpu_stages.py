"""Parity of every single-frame operator of libvo_hip.so against the CPU oracle.

All stages here are integer work: the bar is bit-exact equality."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def test_gray_bgr_and_bgra(ctx, oracle, vga_seq):
    bgr = vga_seq["frames"][0]
    assert np.array_equal(ctx.gray(bgr), oracle.gray(bgr))
    bgra = np.concatenate([bgr, np.full(bgr.shape[:2] + (1,), 255, np.uint8)], axis=2)
    assert np.array_equal(ctx.gray(bgra), oracle.gray(bgra))


@pytest.mark.parametrize("shape", [(480, 640), (376, 1241), (97, 131), (16, 16)])
def test_gray_ragged_sizes(ctx, oracle, shape):
    rng = np.random.default_rng(5)
    for cn in (3, 4):
        img = rng.integers(0, 256, shape + (cn,), dtype=np.uint8)
        assert np.array_equal(ctx.gray(img), oracle.gray(img))


@pytest.mark.parametrize("thr", [0, 5, 20, 60, 255])
def test_fast_score_map(ctx, oracle, vga_seq, thr):
    g = oracle.gray(vga_seq["frames"][0])
    assert np.array_equal(ctx.fast_score(g, thr), oracle.fast_score_map(g, thr))


@pytest.mark.parametrize("shape", [(376, 1241), (97, 131), (16, 16), (7, 7), (33, 300)])
def test_fast_score_random_and_ragged(ctx, oracle, shape):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(ctx.fast_score(g, 10), oracle.fast_score_map(g, 10))
    # low-contrast blocks give plateaus and ties
    g2 = (np.kron(rng.integers(0, 4, (shape[0] // 4 + 1, shape[1] // 4 + 1)), np.ones((4, 4))) * 60).astype(np.uint8)
    g2 = np.ascontiguousarray(g2[:shape[0], :shape[1]])
    assert np.array_equal(ctx.fast_score(g2, 20), oracle.fast_score_map(g2, 20))


@pytest.mark.parametrize("shape", [(480, 640), (376, 1241), (97, 131), (16, 16), (7, 9)])
def test_blur7(ctx, oracle, shape):
    rng = np.random.default_rng(11)
    g = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(ctx.blur7(g), oracle.blur7(g))
    sat = np.full(shape, 255, np.uint8)
    assert np.array_equal(ctx.blur7(sat), oracle.blur7(sat))


@pytest.mark.parametrize("nfeat", [1000, 300, 4000])
def test_detect_matches_oracle(ctx, oracle, vga_seq, nfeat):
    g = oracle.gray(vga_seq["frames"][0])
    k_ref, d_ref = oracle.detect(g, nfeat, 20)
    k, d = ctx.detect(g, nfeat, 20)
    assert len(k) == len(k_ref) > 0
    assert np.array_equal(k, k_ref)
    assert np.array_equal(d, d_ref)


def test_detect_heavy_ties(ctx, oracle):
    # blocky image: thousands of corners share a handful of scores, so the
    # top-N cut falls inside a tie class and the raster tie-break decides
    rng = np.random.default_rng(3)
    g = (np.kron(rng.integers(0, 2, (60, 80)), np.ones((8, 8))) * 200 + 20).astype(np.uint8)
    k_ref, d_ref = oracle.detect(g, 500, 20)
    k, d = ctx.detect(g, 500, 20)
    assert np.array_equal(k, k_ref) and np.array_equal(d, d_ref)


def test_detect_empty_and_tiny(ctx, oracle):
    flat = np.full((120, 160), 77, np.uint8)
    k, d = ctx.detect(flat, 100, 20)
    assert len(k) == 0 and d.shape == (0, 32)
    rng = np.random.default_rng(1)
    small = rng.integers(0, 256, (40, 50), dtype=np.uint8)  # narrower than 2*31: border filter drops all
    k_ref, _ = oracle.detect(small, 100, 20)
    k, _ = ctx.detect(small, 100, 20)
    assert len(k) == len(k_ref) == 0


def test_knn2(ctx, oracle, vga_seq):
    g0, g1 = (oracle.gray(f) for f in vga_seq["frames"][:2])
    _, d0 = oracle.detect(g0, 1000, 20)
    _, d1 = oracle.detect(g1, 1000, 20)
    assert np.array_equal(ctx.knn2(d0, d1), oracle.knn2(d0, d1))


@pytest.mark.parametrize("matcher", ["mfma", "popcount"])
@pytest.mark.parametrize("nq,nt", [(1, 1), (5, 1), (3, 2), (33, 31), (129, 513), (700, 3999), (4000, 4000)])
def test_knn2_random_sizes_and_ties(ctx, oracle, nq, nt, matcher):
    """Both kernels (matrix cores / vector-ALU popcount) against the oracle, bit for bit."""
    ctx.set_matcher(matcher)
    try:
        rng = np.random.default_rng(nq * 31 + nt)
        # few distinct descriptors -> many exact distance ties; lowest index must win
        pool = rng.integers(0, 256, (17, 32), dtype=np.uint8)
        q = pool[rng.integers(0, 17, nq)]
        t = pool[rng.integers(0, 17, nt)]
        assert np.array_equal(ctx.knn2(q, t), oracle.knn2(q, t))
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
        assert np.array_equal(ctx.knn2(q, t), oracle.knn2(q, t))
        # extreme descriptors: all zeros / all ones (distances 0 and 256)
        q = np.zeros((nq, 32), np.uint8)
        q[::2] = 255
        t = np.zeros((nt, 32), np.uint8)
        t[1::3] = 255
        assert np.array_equal(ctx.knn2(q, t), oracle.knn2(q, t))
    finally:
        ctx.set_matcher("mfma")


@pytest.mark.parametrize("matcher", ["mfma", "popcount"])
def test_knn2_at_the_feature_limit(oracle, matcher):
    """6144 x 6144 descriptors, the largest set a context accepts (the matrix-core kernel's keys hold the train
    index below a scale of 6144), with clustered descriptors so that ties reach the highest indices."""
    from visual_odometry_amd import capi
    big = capi.Context(0, 640, 480, 6144, 2)
    try:
        big.set_matcher(matcher)
        rng = np.random.default_rng(6144)
        pool = rng.integers(0, 256, (40, 32), dtype=np.uint8)
        q = pool[rng.integers(0, 40, 6144)].copy()
        t = pool[rng.integers(0, 40, 6144)].copy()
        t[-1] = q[0]
        t[-2] = q[0]  # the nearest two of query 0 may be the last two trains
        flips = rng.integers(0, 256, (6144, 32), dtype=np.uint8) & rng.integers(0, 256, (6144, 32), dtype=np.uint8) & 3
        t[: 6000] ^= flips[: 6000]
        ref = oracle.knn2(q, t)
        got = big.knn2(q, t)
        assert np.array_equal(got, ref)
        assert ref["idx0"].max() >= 6000
    finally:
        big.close()


def _pair_points(oracle, seq, nfeat=1000):
    g0, g1 = (oracle.gray(f) for f in seq["frames"][:2])
    k0, d0 = oracle.detect(g0, nfeat, 20)
    k1, d1 = oracle.detect(g1, nfeat, 20)
    knn = oracle.knn2(d0, d1)
    good = oracle.ratio_test(knn, len(k1), 0.7, k0, k1)
    return k0, d0, k1, d1, good


def test_fundamental_ransac_mask(ctx, oracle, vga_seq):
    k0, d0, k1, d1, good = _pair_points(oracle, vga_seq)
    p1 = np.stack([good["px"], good["py"]], 1)
    p2 = np.stack([good["cx"], good["cy"]], 1)
    ok_r, F_r, mask_r, it_r = oracle.find_fundamental_ransac(p1, p2)
    ok, F, mask, it = ctx.find_fundamental_ransac(p1, p2)
    assert ok == ok_r and it == it_r
    assert np.array_equal(mask, mask_r)
    np.testing.assert_allclose(F, F_r, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("outlier_frac,n", [(0.4, 600), (0.6, 300), (0.0, 50), (0.8, 2000)])
def test_fundamental_ransac_with_outliers(ctx, oracle, vga_seq, outlier_frac, n):
    k0, d0, k1, d1, good = _pair_points(oracle, vga_seq)
    rng = np.random.default_rng(int(outlier_frac * 100) + n)
    sel = rng.integers(0, len(good), n)
    p1 = np.stack([good["px"], good["py"]], 1)[sel].astype(np.float32)
    p2 = np.stack([good["cx"], good["cy"]], 1)[sel].astype(np.float32)
    bad = rng.random(n) < outlier_frac
    p2[bad] = rng.uniform([0, 0], [640, 480], (int(bad.sum()), 2)).astype(np.float32)
    ok_r, F_r, mask_r, it_r = oracle.find_fundamental_ransac(p1, p2)
    ok, F, mask, it = ctx.find_fundamental_ransac(p1, p2)
    assert (ok, it) == (ok_r, it_r)
    assert np.array_equal(mask, mask_r)


def test_match_features_end_to_end(ctx, oracle, vga_seq):
    k0, d0, k1, d1, _ = _pair_points(oracle, vga_seq)
    knn = oracle.knn2(d0, d1)
    m_ref, _, _ = oracle.match_filter(knn, len(k1), k0, k1, 0.7)
    m = ctx.match(k0, d0, k1, d1, 0.7)
    assert len(m) == len(m_ref) > 50
    assert np.array_equal(m, m_ref)


def test_match_features_cross_check(ctx, oracle, vga_seq):
    k0, d0, k1, d1, _ = _pair_points(oracle, vga_seq)
    fwd, rev = oracle.knn2(d0, d1), oracle.knn2(d1, d0)
    m_ref, _, n_ratio = oracle.match_filter(fwd, len(k1), k0, k1, 0.7, knn_rev=rev)
    m_plain, _, n_plain = oracle.match_filter(fwd, len(k1), k0, k1, 0.7)
    m = ctx.match(k0, d0, k1, d1, 0.7, cross_check=True)
    assert np.array_equal(m, m_ref) and 50 < len(m)
    assert n_ratio <= n_plain
    # every kept match is a mutual nearest neighbour
    assert all(rev["idx0"][t] == q for q, t in zip(m["query"], m["train"]))


@pytest.mark.parametrize("shape,cn", [((480, 640), 3), ((375, 1241), 1), ((61, 67), 3), ((720, 1280), 3)])
@pytest.mark.parametrize("hist_eq,blur_size,sigma", [(False, 3, 1.0), (True, 3, 1.0), (True, 7, 0.0),
                                                      (False, 1, 1.0), (True, 31, 6.0), (False, 9, -1.0)])
def test_image_process(ctx, oracle, shape, cn, hist_eq, blur_size, sigma):
    """ImageProcessor::processImage, every plane bit-exact."""
    rng = np.random.default_rng(shape[0] * 7 + blur_size)
    h, w = shape
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img = (img.astype(np.int32) * 3 // 4 + 20).astype(np.uint8)  # leave empty bins at both ends
    if cn == 3:
        img = np.stack([img, np.roll(img, 1, 1), np.roll(img, 1, 0)], -1)
    ref = oracle.process_image(img, True, blur_size, sigma, hist_eq)
    got = ctx.process_image(img, True, blur_size, sigma, hist_eq)
    for k in ("gray", "enhanced", "denoised", "masked"):
        assert np.array_equal(got[k], ref[k]), k
    assert ref["masked"][: h // 4].max() == 0 and ref["masked"][h * 3 // 4 + 1:].max() == 0
    assert np.array_equal(ref["masked"][h // 4: h * 3 // 4 + 1], ref["denoised"][h // 4: h * 3 // 4 + 1])
    plain = ctx.process_image(img, False)
    assert list(plain) == ["gray"] and np.array_equal(plain["gray"], ref["gray"])


def test_image_process_rejects_even_kernel(ctx):
    img = np.zeros((64, 64), np.uint8)
    with pytest.raises(Exception):
        ctx.process_image(img, True, 4, 1.0)
    flat = ctx.process_image(np.full((64, 64), 37, np.uint8), True, 3, 1.0, True)
    assert (flat["enhanced"] == 37).all()  # cv::equalizeHist leaves a constant image alone


# ---- full ORB detector (pyramid, Harris ranking, orientation, steered BRIEF) ----

def _kp_bits(k):
    return np.ascontiguousarray(k).view(np.uint8)


def _same_blurred_level(got, ref, level):
    """Blurred pyramid levels agree wherever a descriptor can sample: key points lie 31 px inside a level and the
    rotated pattern reaches 18 px, so the library leaves the outermost 13 columns / rows at the right and bottom
    of the levels above level 0 undefined (it skips the tiles that start there)."""
    if level == 0:
        return np.array_equal(got, ref)
    h, w = ref.shape
    return got.shape == ref.shape and np.array_equal(got[:h - 13, :w - 13], ref[:h - 13, :w - 13])


@pytest.mark.parametrize("cam_name,nfeatures,levels,scale", [("vga", 1000, 8, 1.2), ("hd720", 2000, 8, 1.2),
                                                             ("kitti", 2000, 8, 1.2), ("vga", 500, 4, 1.5),
                                                             ("vga", 300, 3, 2.0), ("vga", 800, 1, 1.2)])
def test_detect_orb_bit_exact(ctx, oracle, cam_name, nfeatures, levels, scale):
    cam = synth.CONFIG_CAMERAS[cam_name]
    g = synth.make_sequence(cam, 1, seed=31 + nfeatures, channels=1)["frames"][0]
    k_ref, d_ref, lv, bl = oracle.orb_detect(g, nfeatures, scale, levels, 20, pyramid=True)
    k_ref, d_ref = k_ref[:ctx.max_features], d_ref[:ctx.max_features]
    k, d = ctx.detect_orb(g, nfeatures, scale, levels, 20)
    for l in range(levels):
        assert np.array_equal(ctx.orb_level(0, l), lv[l]), f"pyramid level {l}"
        assert _same_blurred_level(ctx.orb_level(0, l, blurred=True), bl[l], l), f"blurred level {l}"
    assert len(k) == len(k_ref) and len(k) >= min(nfeatures, 200)
    assert np.array_equal(k["octave"], k_ref["octave"])
    assert np.array_equal(k["x"], k_ref["x"]) and np.array_equal(k["y"], k_ref["y"])
    # Harris response: the tolerance asked for is 1e-4 relative; the float32 expression is in fact reproduced exactly
    assert np.allclose(k["response"], k_ref["response"], rtol=1e-4, atol=0)
    assert np.array_equal(_kp_bits(k), _kp_bits(k_ref))
    assert np.array_equal(d, d_ref)
    assert len(np.unique(k["octave"])) == levels


@pytest.mark.parametrize("shape,levels", [((289, 514), 3), ((95, 200), 4), ((97, 131), 2), ((600, 1067), 2),
                                          ((201, 357), 3), ((64, 190), 2), ((63, 300), 1), ((321, 159), 4)])
def test_detect_orb_odd_sizes(ctx, oracle, shape, levels):
    """Frames whose last tile column or row is almost empty, and levels with only a few rows (or none) inside the
    31-px border the detector drops: same key points and descriptors as the oracle."""
    h, w = shape
    big = synth.make_sequence(synth.CONFIG_CAMERAS["hd720"], 1, seed=h + w, channels=1)["frames"][0]
    g = np.ascontiguousarray(big[40:40 + h, 100:100 + w])
    k_ref, d_ref, lv, bl = oracle.orb_detect(g, 600, 1.2, levels, 20, pyramid=True)
    k, d = ctx.detect_orb(g, 600, 1.2, levels, 20)
    for l in range(levels):
        assert np.array_equal(ctx.orb_level(0, l), lv[l]), f"pyramid level {l}"
        assert _same_blurred_level(ctx.orb_level(0, l, blurred=True), bl[l], l), f"blurred level {l}"
    assert len(k) == len(k_ref)
    assert np.array_equal(_kp_bits(k), _kp_bits(k_ref))
    assert np.array_equal(d, d_ref)
    if min(h, w) > 150:
        assert len(k) > 100


def test_detect_orb_flat_and_tiny(ctx, oracle):
    flat = np.full((120, 160), 90, np.uint8)
    k, d = ctx.detect_orb(flat, 500, 1.2, 8, 20)
    assert len(k) == 0 and len(oracle.orb_detect(flat, 500)[0]) == 0
    with pytest.raises(Exception):
        ctx.detect_orb(np.zeros((20, 20), np.uint8), 500, 1.2, 8, 20)  # level 7 would be 6 px
    with pytest.raises(Exception):
        ctx.detect_orb(flat, 500, 2.5, 4, 20)


@pytest.mark.parametrize("n,n_out", [(8, 0), (9, 1), (12, 2), (14, 3), (14, 0)])
def test_find_fundamental_small_sets_take_lmeds(ctx, oracle, n, n_out):
    """8..14 correspondences: cv::findFundamentalMat(FM_RANSAC) switches to LMedS."""
    rng = np.random.default_rng(100 * n + n_out)
    X = np.c_[rng.uniform(-2000, 2000, n), rng.uniform(-1500, 1500, n), rng.uniform(3000, 9000, n)]
    ang = 0.03
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([120.0, -30.0, 40.0])
    def proj(P):
        return np.c_[700 * P[:, 0] / P[:, 2] + 640, 700 * P[:, 1] / P[:, 2] + 360]
    p1 = (proj(X) + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    p2 = (proj(X @ R.T + t) + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    p2[:n_out] += rng.uniform(40, 90, (n_out, 2)).astype(np.float32)
    ok_r, F_r, mask_r, it_r = oracle.find_fundamental_ransac(p1, p2)
    ok, F, mask, it = ctx.find_fundamental_ransac(p1, p2)
    assert ok == ok_r and it == it_r and np.array_equal(mask, mask_r)
    assert it_r >= 3 and mask_r.sum() >= 7
    assert np.abs(F - F_r).max() <= 1e-9 * np.abs(F_r).max()


def test_match_features_with_few_survivors(ctx, oracle, vga_seq):
    """A sparse pair (8..14 ratio survivors) still yields matches, through the LMedS branch."""
    k0, d0, k1, d1, _ = _pair_points(oracle, vga_seq)
    fwd = oracle.knn2(d0, d1)
    good = oracle.ratio_test(fwd, len(k1), 0.7, k0, k1)
    keep = good["query"][:11]
    ks, ds = k0[keep], d0[keep]
    knn = oracle.knn2(ds, d1)
    m_ref, _, n_ratio = oracle.match_filter(knn, len(k1), ks, k1, 0.7)
    assert 8 <= n_ratio <= 14 and len(m_ref) >= 7
    m = ctx.match(ks, ds, k1, d1, 0.7)
    assert np.array_equal(m, m_ref)


@pytest.mark.parametrize("cam_name,dist", [
    ("vga", [0.0, 0.0, 0.0, 0.0]),
    ("vga", [-0.28, 0.09, 0.0012, -0.0007, -0.015]),
    ("hd720", [0.12, -0.05, -0.002, 0.003, 0.01, 0.02, -0.01, 0.004]),
    ("kitti", [0.9, 0.4, 0.01, 0.01, 0.2]),  # strong pincushion: taps leave the image
])
def test_undistort_bit_exact(ctx, oracle, cam_name, dist):
    cam = synth.CONFIG_CAMERAS[cam_name]
    g = synth.make_sequence(cam, 1, seed=77, channels=1)["frames"][0]
    ref = oracle.undistort(g, cam.fx, cam.fy, cam.cx, cam.cy, dist)
    out = ctx.undistort(g, cam.fx, cam.fy, cam.cx, cam.cy, dist)
    assert np.array_equal(out, ref)
    if not any(dist):
        assert np.array_equal(out, g)
    with pytest.raises(Exception):
        ctx.undistort(g, cam.fx, cam.fy, cam.cx, cam.cy, [0.1, 0.2, 0.3])
