"""Worst-case corner density and the limits of the full detector: uniform noise at HD1080 with threshold 1 and
the largest key-point budget the library takes (hundreds of thousands of FAST corners per frame, every 128 x 32
tile list far above its usual fill, the cut score of the selection deep inside a huge class of equal scores),
through the single-frame operators and through one batched launch, against the oracle bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, N = 1920, 1080, 6144


@pytest.fixture(scope="module")
def noise():
    rng = np.random.default_rng(6144)
    return rng.integers(0, 256, (2, H, W), dtype=np.uint8)


@pytest.fixture(scope="module")
def big_ctx():
    from visual_odometry_amd import capi
    c = capi.Context(device=0, max_width=W, max_height=H, max_features=N, max_batch=2)
    yield c
    c.close()


def test_noise_frame_fast_detector_at_the_feature_limit(big_ctx, oracle, noise):
    k, d = big_ctx.detect(noise[0], N, 1)
    kr, dr = oracle.detect(noise[0], N, 1)
    # about one pixel in nine is a strict 3 x 3 maximum of the score map; the budget is a small part of that
    assert len(kr) > 5000
    assert np.array_equal(k, kr) and np.array_equal(d, dr)


def test_noise_frame_orb_detector_at_the_feature_limit(big_ctx, oracle, noise):
    k, d = big_ctx.detect_orb(noise[0], nfeatures=N - 64, scale_factor=1.2, nlevels=8, fast_threshold=1, cap=N)
    kr, dr = oracle.orb_detect(noise[0], nfeatures=N - 64, scale_factor=1.2, nlevels=8, fast_threshold=1)
    kr, dr = kr[:N], dr[:N]
    assert len(kr) >= N - 64
    assert np.array_equal(k, kr) and np.array_equal(d, dr)


@pytest.mark.parametrize("levels", [0, 8])
def test_noise_frames_one_batched_launch(big_ctx, oracle, noise, levels):
    n = N if levels == 0 else N - 64
    big_ctx.batch_configure(W, H, 1, n_streams=1, frames_per_stream=2, max_features=n, fast_threshold=1,
                            orb_levels=levels)
    big_ctx.batch_submit(noise, None)
    res = big_ctx.batch_wait()
    for f in range(2):
        k, d = big_ctx.batch_features(f)
        if levels == 0:
            kr, dr = oracle.detect(noise[f], n, 1)
        else:
            kr, dr = oracle.orb_detect(noise[f], nfeatures=n, scale_factor=1.2, nlevels=8, fast_threshold=1)
            kr, dr = kr[:N], dr[:N]
        assert res["n_keypoints"][f] == len(kr)
        assert np.array_equal(k, kr) and np.array_equal(d, dr), (levels, f)
    # unrelated noise frames: the matcher runs on the full 6144 x 6144 problem and finds next to nothing
    m, _ = big_ctx.batch_matches(1, 0)
    assert len(m) < 50


def test_orb_level_quota_limit(big_ctx, oracle, noise):
    """cv::ORB::create(nfeatures, scale, levels) takes any numbers; the full detector here keeps a level's
    candidates in LDS and takes at most 3840 key points from one level.  3000 / 2.0 / 3 asks level 0 for 1714 (the
    ordinary ranking variant), 4000 / 2.0 / 3 for 2286 and 6000 / 1.5 / 4 for 2492 (the largest variant); 6000 / 2.0 / 2
    asks for 4000 and is refused with a message that says so."""
    from visual_odometry_amd import capi
    img = noise[1][:720, :1280]
    for nf, sc, nl in ((3000, 2.0, 3), (4000, 2.0, 3), (6000, 1.5, 4)):
        k, d = big_ctx.detect_orb(img, nfeatures=nf, scale_factor=sc, nlevels=nl, fast_threshold=20, cap=N)
        kr, dr = oracle.orb_detect(img, nfeatures=nf, scale_factor=sc, nlevels=nl, fast_threshold=20)
        assert len(kr) >= min(nf, N) and np.array_equal(k, kr[:N]) and np.array_equal(d, dr[:N]), (nf, sc, nl)
    with pytest.raises(capi.VoError) as e:
        big_ctx.detect_orb(img, nfeatures=6000, scale_factor=2.0, nlevels=2, fast_threshold=20, cap=N)
    assert e.value.code == capi.VO_ERR_UNSUPPORTED and "3840" in str(e.value)


def test_orb_tie_class_beyond_the_small_ranking_lists(big_ctx, oracle):
    """A regular pattern of equal squares: at level 1 all 2808 corners inside the border share one FAST score, so the
    cut at 2 x quota = 724 keeps all of them -- more than the lists of the 256-thread ranking blocks hold (1536).  Those
    blocks must hand the level to the full-size variant; the result is the oracle's, also when only some frames of
    a batch are affected."""
    w, h = 1280, 720
    img = np.full((h, w), 60, np.uint8)
    for y in range(8, h - 9, 28):
        for x in range(8, w - 9, 28):
            img[y:y + 9, x:x + 9] = 200
    kr, dr, lv, _ = oracle.orb_detect(img, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20, pyramid=True)
    c = oracle.fast(lv[1], 20)
    inside = (c["x"] >= 31) & (c["x"] < lv[1].shape[1] - 31) & (c["y"] >= 31) & (c["y"] < lv[1].shape[0] - 31)
    assert 1536 < inside.sum() <= 4096 and len(np.unique(c["response"][inside])) == 1   # the premise
    k, d = big_ctx.detect_orb(img, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20, cap=N)
    assert len(kr) > 1000 and np.array_equal(k, kr) and np.array_equal(d, dr)
    # batches of 32 frames and more rank their levels in the small blocks: the pattern beside an ordinary frame
    from visual_odometry_amd import capi
    rng = np.random.default_rng(9)
    other = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ko, do = oracle.orb_detect(other, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20)
    with capi.Context(device=0, max_width=w, max_height=h, max_features=2000, max_batch=32) as c:
        c.batch_configure(w, h, 1, n_streams=16, frames_per_stream=2, max_features=2000, fast_threshold=20,
                          orb_levels=8)
        c.batch_submit(np.stack([other, img] * 16), None)
        c.batch_wait()
        for f in (0, 1, 30, 31):
            a, b = ((ko, do), (kr, dr))[f & 1]
            kk, dd = c.batch_features(f)
            assert np.array_equal(kk, a[:2000]) and np.array_equal(dd, b[:2000]), f


def test_largest_frame(oracle):
    """4096 x 4096 (2^24 pixels, the largest frame the library takes; corner keys hold y and x in 12 bits each):
    both detectors against the oracle, with key points in the far corner of the frame."""
    from visual_odometry_amd import capi
    w = h = 4096
    rng = np.random.default_rng(4096)
    # blocks of 4 x 4 equal pixels with some noise on top: corners at every scale of the pyramid
    img = np.kron(rng.integers(0, 256, (h // 4, w // 4), dtype=np.uint8), np.ones((4, 4), np.uint8))
    img = np.clip(img.astype(np.int16) + rng.integers(-6, 7, (h, w)), 0, 255).astype(np.uint8)
    with capi.Context(device=0, max_width=w, max_height=h, max_features=3000, max_batch=1) as c:
        k, d = c.detect(img, 2000, 20)
        kr, dr = oracle.detect(img, 2000, 20)
        assert len(kr) > 1800 and np.array_equal(k, kr) and np.array_equal(d, dr)
        k, d = c.detect_orb(img, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20, cap=3000)
        kr, dr = oracle.orb_detect(img, nfeatures=2000, scale_factor=1.2, nlevels=8, fast_threshold=20)
        assert len(kr) >= 2000 and np.array_equal(k, kr[:3000]) and np.array_equal(d, dr[:3000])
        assert ((kr["x"] > 3900) & (kr["y"] > 3900)).any()
