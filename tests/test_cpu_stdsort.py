"""Top-N truncation: the canonical order the GPU path implements against the reference's literal
std::sort + resize (reference src/feature_detector.cpp:62-68), run with the real libstdc++ in
oracle/libvo_oracle_stdsort.so.  std::sort is unstable, so corners of equal FAST score may come out in
any order and the corners that tie at the cut are an arbitrary subset of their class; everything else must
agree.  The test pins those structural facts and prints the measured divergence."""
import numpy as np
import pytest

from visual_odometry_amd import synth


def _divergence(O, gray, n_keep):
    k = O.fast(gray, 20)
    canon = O.select_top(k, gray.shape[1], n_keep)
    lit = O.select_top_stdsort(k, n_keep)
    pa = {(float(x), float(y)) for x, y in zip(canon["x"], canon["y"])}
    pb = {(float(x), float(y)) for x, y in zip(lit["x"], lit["y"])}
    cut = float(canon["response"][-1])
    only_lit = [i for i in range(len(lit)) if (float(lit["x"][i]), float(lit["y"][i])) not in pa]
    same_slot = int(((canon["x"] == lit["x"]) & (canon["y"] == lit["y"])).sum())
    return dict(candidates=len(k), cut=cut, tie_class=int((k["response"] == cut).sum()),
                tie_kept=int((canon["response"] == cut).sum()), set_diff=len(pb - pa), same_slot=same_slot,
                canon=canon, lit=lit, only_lit=only_lit)


@pytest.mark.parametrize("cam_name,seed,n_keep", [("vga", 1001, 1000), ("hd720", 2000, 1000), ("hd720", 2000, 2000)])
def test_stdsort_differs_from_canonical_only_in_ties(oracle, cam_name, seed, n_keep):
    cam = synth.CONFIG_CAMERAS[cam_name]
    relief = 2.0 if cam_name == "hd720" else 1.0   # hd720/2000: the benchmark's first stream
    seq = synth.make_sequence(cam, 1, seed=seed, channels=4, scene=synth.Scene(seed, relief=relief))
    d = _divergence(oracle, oracle.gray(seq["frames"][0]), n_keep)
    canon, lit = d["canon"], d["lit"]
    assert len(canon) == len(lit) == n_keep and d["candidates"] > n_keep
    # both are sorted by response, descending, and hold the same multiset of responses
    assert np.all(np.diff(lit["response"]) <= 0) and np.all(np.diff(canon["response"]) <= 0)
    assert np.array_equal(canon["response"], lit["response"])
    # every corner strictly above the cut score is in both; the sets differ only inside the tie class at the cut
    assert all(lit["response"][i] == d["cut"] for i in d["only_lit"])
    assert d["set_diff"] <= min(d["tie_kept"], d["tie_class"] - d["tie_kept"])
    # the canonical order is (response desc, raster asc); the literal one is not stable
    ras = canon["y"].astype(np.int64) * cam.width + canon["x"].astype(np.int64)
    same = canon["response"][1:] == canon["response"][:-1]
    assert np.all(ras[1:][same] > ras[:-1][same])
    print(f"\n[D2 divergence] {cam_name} seed {seed} keep {n_keep}: {d['candidates']} corners, cut score {d['cut']:.0f}, "
          f"{d['tie_kept']} of {d['tie_class']} tied corners kept; set difference {d['set_diff']} key points "
          f"({100.0 * d['set_diff'] / n_keep:.1f} %), {d['same_slot']} of {n_keep} in the same list position")


def test_stdsort_untouched_below_the_limit(oracle):
    cam = synth.CONFIG_CAMERAS["vga"]
    seq = synth.make_sequence(cam, 1, seed=5, channels=1)
    k = oracle.fast(seq["frames"][0], 20)
    assert np.array_equal(oracle.select_top_stdsort(k, len(k)), k)        # reference :62: no sort at all
    assert np.array_equal(oracle.select_top(k, cam.width, len(k)), k)
