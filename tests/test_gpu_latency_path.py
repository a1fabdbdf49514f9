"""The batch-of-1 latency path the FrameProcessor facade uses -- packed output (one D2H copy per frame), HIP graph
replay of the kernel chain, pinned host buffers -- must return exactly what the ordinary path returns: the same
streams are run with the options off and on and compared record by record, across replays, a re-configuration
(new geometry: the graph is captured again) and both detectors."""
import ctypes as C

import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def _run(ctx, cam, seqs, F, NB, levels, packed, graph, channels=4, pinned=None):
    """Every frame of NB launches of S streams x F frames; returns per (launch, batch entry) a dict of everything."""
    S = len(seqs)
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    ctx.batch_configure(cam.width, cam.height, channels, n_streams=S, frames_per_stream=F, max_features=600,
                        fast_threshold=20, camera=K, orb_levels=levels)
    ctx.batch_set_packed_output(packed)
    ctx.batch_set_graph(graph)
    out = []
    for nb in range(NB):
        frames = np.stack([seqs[s]["frames"][nb * F + f] for s in range(S) for f in range(F)])
        depth = np.stack([seqs[s]["depth"][nb * F + f] for s in range(S) for f in range(F)])
        if pinned is not None:
            fb, db = pinned
            np.copyto(fb, frames)
            np.copyto(db, depth)
            frames, depth = fb, db
        ctx.batch_submit(frames, depth)
        res = ctx.batch_wait()
        for b in range(S * F):
            if packed:
                p = ctx.batch_packed(b)
                assert p["result"].tobytes() == res[b].tobytes()
                rec = {k: p[k] for k in ("kps", "desc", "matches", "tracked", "valid", "pts3", "inliers")}
            else:
                k, d = ctx.batch_features(b)
                m, _ = ctx.batch_matches(b, 0)
                t, _ = ctx.batch_matches(b, 1)
                v, p3 = ctx.batch_matches(b, 2)
                rec = {"kps": k, "desc": d, "matches": m, "tracked": t, "valid": v, "pts3": p3,
                       "inliers": ctx.batch_inliers(b) if res["pnp_success"][b] else np.zeros(0, np.int32)}
            rec["result"] = res[b].copy()
            out.append(rec)
    return out


def _same(a, b):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x["result"].tobytes() == y["result"].tobytes(), i
        for k in ("kps", "desc", "matches", "tracked", "valid", "pts3"):
            assert np.array_equal(x[k], y[k]), (i, k)
        if x["result"]["pnp_success"]:
            assert np.array_equal(x["inliers"], y["inliers"]), i


@pytest.mark.parametrize("levels", [0, 4])
def test_packed_output_and_graph_replay_equal_the_plain_path(ctx, levels):
    cam = synth.CONFIG_CAMERAS["vga"]
    S, F, NB = 2, 2, 3
    seqs = [synth.make_sequence(cam, F * NB, seed=4100 + s, channels=4) for s in range(S)]
    plain = _run(ctx, cam, seqs, F, NB, levels, packed=False, graph=False)
    assert sum(int(r["result"]["pnp_success"]) for r in plain) == S * (F * NB - 1)
    ctx.batch_reset()
    _same(_run(ctx, cam, seqs, F, NB, levels, packed=True, graph=False), plain)
    ctx.batch_reset()
    _same(_run(ctx, cam, seqs, F, NB, levels, packed=False, graph=True), plain)
    ctx.batch_reset()
    _same(_run(ctx, cam, seqs, F, NB, levels, packed=True, graph=True), plain)


def test_graph_is_captured_again_after_a_new_geometry(ctx):
    """One context, graph on throughout: VGA, then a smaller gray geometry, then VGA again -- each segment equals the
    plain path on a fresh stream state."""
    cam = synth.CONFIG_CAMERAS["vga"]
    seqs = [synth.make_sequence(cam, 4, seed=4300, channels=4)]
    small = synth.Camera(width=320, height=200, fx=300.0, fy=300.0, cx=160.0, cy=100.0)
    gray = [{"frames": np.ascontiguousarray(seqs[0]["frames"][:, 100:300, 200:520, 1]),
             "depth": np.ascontiguousarray(seqs[0]["depth"][:, 100:300, 200:520])}]
    ref_a = _run(ctx, cam, seqs, 1, 4, 0, packed=False, graph=False)
    ref_b = _run(ctx, small, gray, 1, 4, 0, packed=False, graph=False, channels=1)
    _same(_run(ctx, cam, seqs, 1, 4, 0, packed=True, graph=True), ref_a)
    _same(_run(ctx, small, gray, 1, 4, 0, packed=True, graph=True, channels=1), ref_b)
    _same(_run(ctx, cam, seqs, 1, 4, 0, packed=True, graph=True), ref_a)


def test_pinned_host_buffers(ctx):
    """Frames and depth handed over in memory from vo_host_alloc: same results as from ordinary numpy arrays."""
    from visual_odometry_amd import capi
    lib = capi.load()
    cam = synth.CONFIG_CAMERAS["vga"]
    seqs = [synth.make_sequence(cam, 3, seed=4500, channels=4)]
    nf, nd = cam.height * cam.width * 4, cam.height * cam.width * 4
    pf, pd = lib.vo_host_alloc(nf), lib.vo_host_alloc(nd)
    assert pf and pd
    try:
        fb = np.ctypeslib.as_array((C.c_uint8 * nf).from_address(pf)).reshape(1, cam.height, cam.width, 4)
        db = np.ctypeslib.as_array((C.c_float * (nd // 4)).from_address(pd)).reshape(1, cam.height, cam.width)
        ref = _run(ctx, cam, seqs, 1, 3, 0, packed=False, graph=False)
        _same(_run(ctx, cam, seqs, 1, 3, 0, packed=True, graph=True, pinned=(fb, db)), ref)
    finally:
        lib.vo_host_free(pf)
        lib.vo_host_free(pd)


def test_packed_output_is_refused_for_large_batches():
    """The packed block is meant for the latency path: batch x (128 + 156 x feature capacity) bytes above 32 MiB is
    an argument error, and the context keeps working without it."""
    from visual_odometry_amd import capi
    cam = synth.CONFIG_CAMERAS["vga"]
    c = capi.Context(device=0, max_width=cam.width, max_height=cam.height, max_features=2064, max_batch=128)
    try:
        c.batch_configure(cam.width, cam.height, 1, n_streams=128, frames_per_stream=1, max_features=2000,
                          camera=(cam.fx, cam.fy, cam.cx, cam.cy))
        with pytest.raises(capi.VoError):
            c.batch_set_packed_output(True)
        g = synth.make_sequence(cam, 1, seed=9, channels=1)["frames"][0]
        c.batch_submit(np.stack([g] * 128), None)
        res = c.batch_wait()
        assert (res["n_keypoints"] == res["n_keypoints"][0]).all() and res["n_keypoints"][0] > 100
    finally:
        c.close()


def test_graph_is_dropped_when_a_single_frame_call_replans_the_pyramid(ctx):
    """A recorded batch chain holds the pyramid's buffer addresses and level tables.  A single-frame full-detector
    call at a larger geometry re-plans (and reallocates) them; the next graph batch must record the chain again
    instead of replaying the stale one.  Every segment equals the plain path."""
    cam = synth.CONFIG_CAMERAS["vga"]
    seqs = [synth.make_sequence(cam, 6, seed=4700, channels=4)]
    plain = _run(ctx, cam, seqs, 1, 6, 4, packed=False, graph=False)
    big = synth.make_sequence(synth.CONFIG_CAMERAS["hd1080"], 1, seed=4701, channels=1)["frames"][0]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    ctx.batch_configure(cam.width, cam.height, 4, n_streams=1, frames_per_stream=1, max_features=600,
                        fast_threshold=20, camera=K, orb_levels=4)
    ctx.batch_set_packed_output(True)
    ctx.batch_set_graph(True)
    got = []
    for i in range(6):
        if i in (2, 4):
            # between two graph batches: the context's largest frame through the single-frame operator
            kps, _ = ctx.detect_orb(big, nfeatures=3000, nlevels=8)
            assert len(kps) > 1000
        ctx.batch_submit(seqs[0]["frames"][i][None], seqs[0]["depth"][i][None])
        res = ctx.batch_wait()
        p = ctx.batch_packed(0)
        rec = {k: p[k] for k in ("kps", "desc", "matches", "tracked", "valid", "pts3", "inliers")}
        rec["result"] = res[0].copy()
        got.append(rec)
    ctx.batch_set_graph(False)
    ctx.batch_set_packed_output(False)
    _same(got, plain)


def test_stage_timing_is_refused_after_a_graph_replay(ctx):
    from visual_odometry_amd import capi
    cam = synth.CONFIG_CAMERAS["vga"]
    seq = synth.make_sequence(cam, 2, seed=4800, channels=4)
    ctx.batch_configure(cam.width, cam.height, 4, max_features=600, camera=(cam.fx, cam.fy, cam.cx, cam.cy))
    ctx.batch_set_timing(True)
    try:
        ctx.batch_set_graph(True)
        ctx.batch_submit(seq["frames"][:1], seq["depth"][:1])
        ctx.batch_wait()
        with pytest.raises(capi.VoError) as e:
            ctx.batch_timing()
        assert e.value.code == capi.VO_ERR_STATE
        ctx.batch_set_graph(False)
        ctx.batch_submit(seq["frames"][1:], seq["depth"][1:])
        ctx.batch_wait()
        t = ctx.batch_timing()
        assert len(t) == 12 and all(v >= 0 for v in t.values()) and sum(t.values()) > 0
    finally:
        ctx.batch_set_timing(False)
        ctx.batch_set_graph(False)


def test_failed_reconfigure_with_packed_output_leaves_the_context_as_it_was():
    """vo_batch_configure for a batch whose packed block would exceed 32 MiB fails before anything is committed: the
    old configuration keeps working."""
    from visual_odometry_amd import capi
    cam = synth.CONFIG_CAMERAS["vga"]
    c = capi.Context(device=0, max_width=cam.width, max_height=cam.height, max_features=2064, max_batch=128)
    try:
        K = (cam.fx, cam.fy, cam.cx, cam.cy)
        c.batch_configure(cam.width, cam.height, 1, n_streams=2, frames_per_stream=1, max_features=500, camera=K)
        c.batch_set_packed_output(True)
        seq = synth.make_sequence(cam, 2, seed=10, channels=1)
        c.batch_submit(np.stack([seq["frames"][0]] * 2), np.stack([seq["depth"][0]] * 2))
        first = c.batch_wait().copy()
        with pytest.raises(capi.VoError):
            c.batch_configure(cam.width, cam.height, 1, n_streams=128, frames_per_stream=1, max_features=500, camera=K)
        c.batch_submit(np.stack([seq["frames"][1]] * 2), np.stack([seq["depth"][1]] * 2))
        res = c.batch_wait()
        assert len(res) == 2 and (res["n_matches"] > 50).all() and (res["pnp_success"] == 1).all()
        assert (first["n_keypoints"] > 100).all()
    finally:
        c.close()


def test_depth16_gray_input_equals_float_depth(ctx, oracle):
    """vo_batch_submit_depth16: gray 8-bit frames with 16-bit millimetre depth (3 bytes per pixel over the link
    instead of 8) give what the float path gives on the same numbers -- and both what the oracle gives."""
    from oracle.pipeline import OracleStream
    cam = synth.CONFIG_CAMERAS["vga"]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    S, F = 2, 3
    seqs = [synth.make_sequence(cam, F, seed=4300 + s, channels=3) for s in range(S)]
    gray = np.concatenate([np.stack([oracle.gray(f) for f in q["frames"]]) for q in seqs])
    d16 = np.concatenate([np.clip(np.rint(q["depth"]), 0, 65535).astype(np.uint16) for q in seqs])
    d16[:, ::7, ::5] = 0                       # holes: no measurement
    d32 = d16.astype(np.float32)
    ctx.batch_set_packed_output(False)
    ctx.batch_set_graph(False)
    out = []
    for depth in (d32, d16):
        ctx.batch_configure(cam.width, cam.height, 1, n_streams=S, frames_per_stream=F, max_features=1000, camera=K)
        ctx.batch_submit(gray, depth)
        res = ctx.batch_wait()
        lists = [(ctx.batch_matches(b, 2), ctx.batch_inliers(b)) for b in range(S * F)]
        out.append((res, lists))
    (r32, l32), (r16, l16) = out
    assert r32.tobytes() == r16.tobytes()
    assert int(r32["pnp_success"].sum()) == S * (F - 1)
    for (m32, i32), (m16, i16) in zip(l32, l16):
        assert np.array_equal(m32[0], m16[0]) and np.array_equal(m32[1], m16[1]) and np.array_equal(i32, i16)
    # against the oracle, stream by stream
    for s in range(S):
        ref = OracleStream(K, 1000, 20)
        for f in range(F):
            b = s * F + f
            r = ref.process(gray[b], d32[b])
            m, p3 = l16[b][0]
            assert np.array_equal(m, r["valid"]) and np.array_equal(p3, r["pts3"])
            assert bool(r16["pnp_success"][b]) == r["pnp_success"]
            if r["pnp_success"]:
                assert np.array_equal(l16[b][1], r["inliers"])
                assert np.abs(r16["rvec"][b] - r["rvec"]).max() < 1e-6


@pytest.mark.parametrize("F", [2, 5])
def test_padded_host_planes_in_ordinary_memory(ctx, F):
    """Frames and depth planes handed over as padded rows and padded planes of ordinary host memory (the row-by-row
    packing of the staging pool: pieces and chunks end in the middle of rows and planes), several frames per launch:
    same results as from contiguous arrays."""
    cam = synth.CONFIG_CAMERAS["hd720"]
    seqs = [synth.make_sequence(cam, 2 * F, seed=4700, channels=4)]
    ref = _run(ctx, cam, seqs, F, 2, 0, packed=False, graph=False)
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    ctx.batch_configure(cam.width, cam.height, 4, n_streams=1, frames_per_stream=F, max_features=600,
                        fast_threshold=20, camera=K, orb_levels=0)
    ctx.batch_set_packed_output(False)
    ctx.batch_set_graph(False)
    row, drow = cam.width * 4 + 52, cam.width + 13            # bytes / elements
    plane, dplane = row * cam.height + 1000, drow * cam.height + 77
    rng = np.random.default_rng(3)
    out = []
    for nb in range(2):
        fb = rng.integers(0, 256, F * plane, dtype=np.uint8)   # garbage in the padding
        db = rng.random(F * dplane, dtype=np.float32) * 9000.0
        for f in range(F):
            v = fb[f * plane:f * plane + row * cam.height].reshape(cam.height, row)
            v[:, :cam.width * 4] = seqs[0]["frames"][nb * F + f].reshape(cam.height, cam.width * 4)
            d = db[f * dplane:f * dplane + drow * cam.height].reshape(cam.height, drow)
            d[:, :cam.width] = seqs[0]["depth"][nb * F + f]
        ctx.batch_submit_ptr(fb.ctypes.data, row, plane, db.ctypes.data, drow, dplane)
        res = ctx.batch_wait()
        for b in range(F):
            k, d = ctx.batch_features(b)
            m, _ = ctx.batch_matches(b, 0)
            t, _ = ctx.batch_matches(b, 1)
            v, p3 = ctx.batch_matches(b, 2)
            out.append({"kps": k, "desc": d, "matches": m, "tracked": t, "valid": v, "pts3": p3,
                        "inliers": ctx.batch_inliers(b) if res["pnp_success"][b] else np.zeros(0, np.int32),
                        "result": res[b].copy()})
    _same(out, ref)


@pytest.mark.parametrize("depth16", [False, True])
def test_depth_read_in_place_from_pinned_host_memory(ctx, depth16):
    """Frames already on the device, depth planes left in pinned host memory with padded rows and planes (what a
    host-fed pipeline does that uploads frames itself): the back-projection reads the depth values where they lie.
    Same results as with the depth planes on the device."""
    from visual_odometry_amd import capi
    lib = capi.load()
    hip = C.CDLL("libamdhip64.so.7")   # (the runtime the library itself is linked to)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def to_device(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0   # hipMemcpyHostToDevice
        return p

    cam = synth.CONFIG_CAMERAS["hd720"]
    F = 3
    seqs = [synth.make_sequence(cam, 2 * F, seed=4800, channels=4)]
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    esz = 2 if depth16 else 4
    drow = cam.width + 24
    dplane = drow * cam.height + 136
    base = lib.vo_host_alloc(F * dplane * esz + 4096)
    assert base
    pd = base + 1280   # (the planes start inside the allocation, not at its first byte)
    try:
        host = np.ctypeslib.as_array(((C.c_uint16 if depth16 else C.c_float) * (F * dplane)).from_address(pd))
        runs = []
        for in_place in (False, True):
            ctx.batch_configure(cam.width, cam.height, 4, n_streams=1, frames_per_stream=F, max_features=600,
                                fast_threshold=20, camera=K, orb_levels=0)
            ctx.batch_set_packed_output(False)
            ctx.batch_set_graph(False)
            out = []
            for nb in range(2):
                frames = np.stack([seqs[0]["frames"][nb * F + f] for f in range(F)])
                depth = np.stack([seqs[0]["depth"][nb * F + f] for f in range(F)])
                if depth16:
                    depth = np.clip(np.rint(depth), 0, 65535).astype(np.uint16)
                host[:] = 7 if depth16 else 123.0   # garbage in the padding (valid-looking depth values)
                for f in range(F):
                    host[f * dplane:f * dplane + drow * cam.height].reshape(cam.height, drow)[:, :cam.width] = depth[f]
                fd = to_device(np.ascontiguousarray(frames))
                dd = None if in_place else to_device(host)
                ctx.batch_submit_ptr(fd.value, cam.width * 4, cam.width * 4 * cam.height, pd if in_place else dd.value,
                                     drow, dplane, depth16=depth16)
                res = ctx.batch_wait()
                hip.hipFree(fd)
                if dd is not None:
                    hip.hipFree(dd)
                for b in range(F):
                    k, d = ctx.batch_features(b)
                    m, _ = ctx.batch_matches(b, 0)
                    t, _ = ctx.batch_matches(b, 1)
                    v, p3 = ctx.batch_matches(b, 2)
                    out.append({"kps": k, "desc": d, "matches": m, "tracked": t, "valid": v, "pts3": p3,
                                "inliers": ctx.batch_inliers(b) if res["pnp_success"][b] else np.zeros(0, np.int32),
                                "result": res[b].copy()})
            runs.append(out)
        assert sum(int(r["result"]["pnp_success"]) for r in runs[0]) >= 2 * F - 2   # (poses were computed)
        _same(runs[1], runs[0])
    finally:
        lib.vo_host_free(base)


def test_uploaded_depth_path_equals_depth_read_in_place():
    """VO_DEPTH_IN_PLACE=0 (the arrangement a plane too large for the staging area still gets: the depth plane
    uploaded on a second stream, the second part of the chain behind its event) returns what the default does.
    The switch is read once per process, so each variant runs in a process of its own and prints a digest."""
    import hashlib, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = (
        "import hashlib, numpy as np\n"
        "from visual_odometry_amd import capi, synth\n"
        "cam = synth.CONFIG_CAMERAS['vga']\n"
        "seq = synth.make_sequence(cam, 4, seed=4900, channels=4)\n"
        "ctx = capi.Context(0, cam.width, cam.height, 700, 2)\n"
        "ctx.batch_configure(cam.width, cam.height, 4, n_streams=1, frames_per_stream=2, max_features=600,\n"
        "                    fast_threshold=20, camera=(cam.fx, cam.fy, cam.cx, cam.cy), orb_levels=0)\n"
        "h = hashlib.sha256()\n"
        "for nb in range(2):\n"
        "    ctx.batch_submit(np.stack(seq['frames'][2 * nb:2 * nb + 2]), np.stack(seq['depth'][2 * nb:2 * nb + 2]))\n"
        "    res = ctx.batch_wait()\n"
        "    h.update(res.tobytes())\n"
        "    for b in range(2):\n"
        "        v, p3 = ctx.batch_matches(b, 2)\n"
        "        h.update(v.tobytes()); h.update(p3.tobytes())\n"
        "assert res['pnp_success'].all()\n"
        "print('digest', h.hexdigest())\n")
    digests = []
    for mode in ("1", "0"):
        env = dict(os.environ, VO_DEPTH_IN_PLACE=mode, PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", prog], env=env, cwd=root, capture_output=True, text=True, timeout=110)
        assert out.returncode == 0, out.stderr[-2000:]
        digests.append([l for l in out.stdout.splitlines() if l.startswith("digest")][-1])
    assert digests[0] == digests[1]
