#!/usr/bin/env python3
"""Throughput benchmark of the visual-odometry front end on MI355X.

A step = one pass of the hot path (BGRA -> FAST/ORB detect -> Hamming kNN +
ratio -> F-RANSAC -> track filter -> depth back-projection -> PnP RANSAC pose)
over one batch of synthetic HD720 frames that already sit in HBM.
One process per GPU; independent camera streams, no collective on the data
path ("replicas only": a single stream does not shard).

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec
VALU_PEAK_GLOPS = 39321.6    # 256 CU x 64 lanes x 2.4 GHz, 32-bit integer lane-ops/s (G)
MFMA_I8_PEAK_TOPS = 5033.2   # dense int8 MFMA: 1024 SIMDs x 1024 MAC/clk x 2 x 2.4 GHz (twice the bf16 rate)


HARD_SCENE_MOVERS = 30       # independently moving rectangles of the hard scene (>= 30 % of the ratio-test
                             # survivors then violate the camera's epipolar geometry)


def build_inputs(streams, frames, seed, movers=0, workers=1, relief=2.0, tilt=(0.0, 0.0), cam_name="hd720",
                 channels=4):
    """`streams` independent HD720 BGRA + depth streams (stream s is rendered from seed + s: its own scene,
    texture and noise), `frames` consecutive frames each, walking forth and back over frames/2 + 1 camera
    positions so that a launch can be repeated.  Returns the camera, the frames and depth planes in batch order
    and, per stream, the ground-truth T_prev_from_curr of each frame."""
    from visual_odometry_amd import synth
    cam = synth.CONFIG_CAMERAS[cam_name]
    f, d, truth = synth.make_pingpong_batch(cam, streams, frames, seed, channels=channels, relief=relief,
                                            movers=movers, workers=workers, tilt=tilt)
    return cam, f, d, truth


def kernel_source_hash():
    """SHA-256 over the kernel sources: profile-derived numbers are only quoted for the sources they were
    measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "visual_odometry_amd", "csrc", "*.hip")) +
                       glob.glob(os.path.join(ROOT, "visual_odometry_amd", "csrc", "*.h"))):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


def committed_traffic(frames_per_launch):
    """(per-stage counters, description of where they come from).  rocprofv3 cannot run inside this process, so
    HBM bytes per launch come from the newest PMC summary under profiles/ -- and only if it was taken at this
    launch size on exactly these kernel sources."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    if not cands:
        return {}, "none: no profiles/rNN_traffic.json"
    path = cands[-1]
    rel = os.path.relpath(path, ROOT)
    tj = json.load(open(path))
    if tj.get("frames_per_launch") != frames_per_launch:
        return {}, f"refused: {rel} was taken at {tj.get('frames_per_launch')} frames per launch"
    if tj.get("source_hash") != kernel_source_hash():
        return {}, f"refused: {rel} was taken on other kernel sources (hash mismatch)"
    return tj["stages"], f"{rel} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ passes, kernel sources {tj['source_hash'][:12]})"


def cpu_baseline(cam, frames, depth, n_frames, max_features, orb_levels):
    """The oracle pipeline on one host core, over a bounded sample of the same workload."""
    from oracle.pipeline import OracleStream
    ref = OracleStream((cam.fx, cam.fy, cam.cx, cam.cy), max_features, 20, orb_levels=orb_levels)
    period = len(frames)
    ref.process(frames[0], depth[0])  # first frame: detect only, not timed
    t0 = time.perf_counter()
    for i in range(1, n_frames + 1):
        ref.process(frames[i % period], depth[i % period])
    dt = time.perf_counter() - t0
    return n_frames / dt


def _cpu_worker(job):
    """One host core: its own stream (rendered here), the oracle pipeline over n_frames; returns (frames, seconds)."""
    seed, frames, n_frames, max_features, orb_levels = job
    cam, f, d, _ = build_inputs(1, frames, seed)  # its own scene
    from oracle.pipeline import OracleStream
    ref = OracleStream((cam.fx, cam.fy, cam.cx, cam.cy), max_features, 20, orb_levels=orb_levels)
    ref.process(f[0], d[0])
    t0 = time.perf_counter()
    for i in range(1, n_frames + 1):
        ref.process(f[i % frames], d[i % frames])
    return n_frames, time.perf_counter() - t0


def cpu_baseline_all_cores(workers, frames, n_frames, max_features, orb_levels):
    """One independent stream per host core, all at once (the aggregate a CPU deployment of the reference would
    reach): total frames over the slowest worker's time.  Runs before this process touches the GPU."""
    import concurrent.futures as cf
    import multiprocessing as mp
    jobs = [(9000 + w, frames, n_frames, max_features, orb_levels) for w in range(workers)]
    with cf.ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as ex:
        res = list(ex.map(_cpu_worker, jobs))
    return sum(r[0] for r in res) / max(r[1] for r in res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100,
                    help="timed steps (the default keeps the timed region above half a second)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=32, help="independent camera streams per GPU")
    ap.add_argument("--frames", type=int, default=16, help="consecutive frames per stream per step (even)")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--contexts", type=int, default=1,
                    help="vo_ctx objects per GPU (each owns streams/contexts camera streams and one HIP "
                         "stream); batches of different contexts overlap on the GPU")
    ap.add_argument("--detector", choices=("orb", "fast"), default="orb",
                    help="orb: full ORB detector, 8-level pyramid; fast: FAST corners + ORB::compute on one "
                         "level, the sequence the reference node executes")
    ap.add_argument("--matcher", choices=["mfma", "popcount"], default="mfma",
                    help="kernel for the all-pairs Hamming distances: int8 matrix cores (default) or XOR + popcount "
                         "on the vector ALU (identical results)")
    ap.add_argument("--cpu-frames", type=int, default=192, help="frames timed for the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=16,
                    help="host cores for the all-core CPU baseline (one stream each; 0 = skip; capped at the CPU count)")
    ap.add_argument("--render-workers", type=int, default=16, help="processes rendering the synthetic streams")
    ap.add_argument("--hard-scene", type=int, default=None,
                    help="also run the same launch on streams with independently moving objects (0 = skip; "
                         "default: on with one GPU, off with several)")
    ap.add_argument("--flat-scene", type=int, default=None,
                    help="also run the same launch on streams that look at a flat wall (0 = skip; default: on with "
                         "one GPU, off with several)")
    ap.add_argument("--config", choices=("hd720", "kitti8"), default="hd720",
                    help="hd720: the headline workload, the same number of streams on every GPU (weak scaling); "
                         "kitti8: eight KITTI-shaped 1241x376 BGR streams in total, dealt round-robin to the ranks "
                         "(strong scaling: 8, 4, 2, 1 streams per GPU at 1, 2, 4, 8 GPUs)")
    ap.add_argument("--host-fed", type=int, default=1, help="also measure the host-fed (PCIe-inclusive) rate (0 = skip)")
    ap.add_argument("--facade-latency", type=int, default=1,
                    help="also measure the single-stream latency through the C++ facade (0 = skip)")
    ap.add_argument("--graph", type=int, default=0,
                    help="replay each launch's kernel chain as a recorded HIP graph (vo_batch_set_graph) during the "
                         "timed steps")
    ap.add_argument("--cpu-fast", type=int, default=1,
                    help="also time the CPU baseline of the other detector (the one-level FAST sequence the "
                         "reference node runs) for the record (0 = skip)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process touches no GPU; it starts the N ranks as fresh child
        # processes of this same script, relays rank 0's one line and passes on a failing rank's status
        from visual_odometry_amd import shard
        sys.exit(shard.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if args.hard_scene is None:
        args.hard_scene = 0 if multi else 1
    if args.flat_scene is None:
        args.flat_scene = 0 if multi else 1

    # the all-core CPU baseline runs first, in child processes started before this process initialises the GPU
    cpu_all = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.cpu_frames > 0 and args.cpu_workers > 1:
        workers = max(1, min(args.cpu_workers, os.cpu_count() or 1))
        per_worker = max(16, args.cpu_frames // 2)
        lv = 8 if args.detector == "orb" else 0
        cpu_all = (cpu_baseline_all_cores(workers, args.frames, per_worker, args.features, lv), workers, per_worker)

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (which "
                         "starts the ranks itself) or under torch.distributed.run --nproc-per-node N")
    # VO_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo instead of RCCL (which refuses two ranks on one device) --
    # runs the N > 1 code path end to end on a one-GPU box; its throughput means nothing
    rehearsal = os.environ.get("VO_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    red_dev = "cpu" if rehearsal else "cuda"   # where the scalars of the cross-rank reductions live
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from visual_odometry_amd import capi

    from visual_odometry_amd import shard
    kitti = args.config == "kitti8"
    NC = max(1, args.contexts) if not kitti else 1
    assert args.streams % NC == 0, "--streams must be a multiple of --contexts"
    # the ranks of one node share its host cores: each renders with up to --render-workers processes out of its
    # share of them (a node with 16 cores per GPU keeps all 16 at any number of ranks)
    render_workers = max(1, min(args.render_workers, (os.cpu_count() or 1) // world))
    if kitti:
        # configs[3]: eight streams in all; rank r owns streams r, r + world, ... and renders exactly those (a stream's
        # seed does not depend on the number of ranks)
        if world > 8:
            raise SystemExit("--config kitti8 has eight streams: at most eight ranks")
        args.hard_scene = args.flat_scene = args.host_fed = args.facade_latency = 0
        args.cpu_frames = 0
        mine = shard.stream_seeds(8, world, rank, base_seed=4000)
        owned = shard.gather_stream_ids([s for s, _ in mine], dist)
        assert sorted(owned) == list(range(8)), "every stream is owned exactly once"
        S, F, C = len(mine), args.frames, 3
        parts = [build_inputs(1, F, seed=sd, cam_name="kitti", channels=3) for _, sd in mine]
        inputs = [(parts[0][0], np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]),
                   [p[3][0] for p in parts])]
    else:
        S, F, C = args.streams // NC, args.frames, 4   # streams per context
        # every stream of every context and rank has its own seed: 2000 + 1000 * rank + stream
        inputs = [build_inputs(S, F, seed=2000 + 1000 * rank + S * k, workers=render_workers) for k in range(NC)]
    Bc = S * F                               # frames per context per step
    B = Bc * NC                              # frames per step per GPU
    cam, frames_h, depth_h, truth = inputs[0]
    H, W = cam.height, cam.width
    K = (cam.fx, cam.fy, cam.cx, cam.cy)
    frames_d = [torch.from_numpy(i[1]).cuda() for i in inputs]
    depth_d = [torch.from_numpy(i[2]).cuda() for i in inputs]
    truths = [i[3] for i in inputs]
    ctxs = []
    levels = 8 if args.detector == "orb" else 0
    cap = args.features + (64 if levels else 0)  # room for boundary ties of the per-level ranking
    for _ in range(NC):
        c = capi.Context(device=local_rank, max_width=W, max_height=H, max_features=cap, max_batch=Bc)
        c.set_matcher(args.matcher)
        c.batch_configure(W, H, C, n_streams=S, frames_per_stream=F, max_features=args.features,
                          fast_threshold=20, camera=K, orb_levels=levels)
        ctxs.append(c)
    ctx = ctxs[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(fds, dds, steps, warmup, reps=5):
        """warmup + exactly `steps` timed steps over the given device frames (one tensor per context), then the
        per-kernel durations of one context's launch from HIP events on the library's own stream."""
        def step():
            # all contexts in flight at once: their kernels interleave on the GPU
            for c, fd, dd in zip(ctxs, fds, dds):
                c.batch_submit_ptr(fd.data_ptr(), W * C, W * C * H, dd.data_ptr(), W, W * H)
            return np.concatenate([c.batch_wait() for c in ctxs])

        r = None
        for c in ctxs:
            c.batch_set_graph(bool(args.graph))
        for _ in range(warmup):
            r = step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = step()
        barrier()
        dt = time.perf_counter() - t0
        fps_ = shard.aggregate_fps(B * steps, dt, dist, red_dev)
        dt = shard.max_over_ranks(dt, dist, red_dev)
        for c in ctxs:
            c.batch_set_graph(False)  # (per-stage events need plain launches)
        # (context 0 alone, so that one launch's duration is not stretched by another context's kernels)
        ctx.batch_set_timing(True)
        acc_ = {}
        for _ in range(reps):
            ctx.batch_submit_ptr(fds[0].data_ptr(), W * C, W * C * H, dds[0].data_ptr(), W, W * H)
            ctx.batch_wait()
            for k, v in ctx.batch_timing().items():
                acc_[k] = acc_.get(k, 0.0) + v / reps
        ctx.batch_set_timing(False)
        return fps_, dt, r, acc_

    fps, elapsed, res, acc = measure(frames_d, depth_d, args.steps, args.warmup)

    # the same launch on a hard scene: every stream has independently moving objects, so a third and more of the
    # ratio-test survivors are outliers for the F-RANSAC (hundreds of 7-point iterations instead of a handful)
    # and the PnP RANSAC sees what slips through.  Reported beside the headline, never as `value`.
    hard = None
    if args.hard_scene:
        hin = [build_inputs(S, F, seed=52000 + 1000 * rank + S * k, movers=HARD_SCENE_MOVERS, workers=render_workers)
               for k in range(NC)]
        hf = [torch.from_numpy(i[1]).cuda() for i in hin]
        hd = [torch.from_numpy(i[2]).cuda() for i in hin]
        for c in ctxs:
            c.batch_reset()
        h_fps, h_dt, h_res, h_acc = measure(hf, hd, max(3, args.steps // 2), max(2, args.warmup), reps=3)
        m = h_res["n_ratio"] > 0
        hard = {
            "value": h_fps, "unit": "frames/s", "steps": max(3, args.steps // 2),
            "ms_per_step": 1e3 * h_dt / max(3, args.steps // 2),
            "scene": f"{HARD_SCENE_MOVERS} independently moving textured rectangles per stream, own scene per stream",
            "stage_ms_per_launch": {k: round(v, 4) for k, v in h_acc.items()},
            "mean_keypoints": float(h_res["n_keypoints"].mean()),
            "mean_ratio_survivors": float(h_res["n_ratio"][m].mean()),
            "mean_matches": float(h_res["n_matches"][m].mean()),
            "outlier_fraction_of_ratio_survivors": float(1.0 - (h_res["f_inliers"][m] / h_res["n_ratio"][m]).mean()),
            "mean_f_ransac_iterations": float(h_res["f_iters"][m].mean()),
            "mean_pnp_iterations": float(h_res["pnp_iters"][m].mean()),
            "pose_success_rate": float(h_res["pnp_success"].mean()),
        }
        t_err = [float(np.linalg.norm(h_res["tvec"][b] - hin[b // Bc][3][(b % Bc) // F][b % F][1]))
                 for b in range(len(h_res)) if h_res["pnp_success"][b]]
        hard["median_translation_error_mm"] = float(np.median(t_err)) if t_err else None
        del hf, hd, hin

    # Host-fed throughput: the same launches with every frame and depth plane coming from (pinned) host memory,
    # the upload of launch k+1 running on a copy stream under the kernels of launch k.  PCIe-bound; never `value`.
    host_fed = None
    if args.host_fed and world == 1:
        hs = min(S, 8)                       # streams of the host-fed launches (keeps the pinned buffers below 1 GB)
        hb = hs * F
        n_h = max(4, args.steps // 2)
        copy_stream = torch.cuda.Stream()

        def host_fed_rate(fh, dh, channels, depth16, upload_depth):
            """frames/s of n_h launches of hb frames whose planes start in pinned host memory.  The frames of launch
            k+1 go up on a copy stream under the kernels of launch k.  The depth planes either go up with them
            (upload_depth) or stay where they are: the library reads the few hundred depth values a frame needs out
            of pinned host memory itself."""
            fp_ = torch.from_numpy(fh).pin_memory()
            dp_ = torch.from_numpy(dh).pin_memory()
            bufs = [(torch.empty_like(fp_, device="cuda"), torch.empty_like(dp_, device="cuda") if upload_depth else None)
                    for _ in range(2)]
            hctx = capi.Context(device=local_rank, max_width=W, max_height=H, max_features=cap, max_batch=hb)
            hctx.set_matcher(args.matcher)
            hctx.batch_configure(W, H, channels, n_streams=hs, frames_per_stream=F, max_features=args.features,
                                 fast_threshold=20, camera=K, orb_levels=levels)

            def upload(k):
                with torch.cuda.stream(copy_stream):
                    bufs[k % 2][0].copy_(fp_, non_blocking=True)
                    if upload_depth:
                        bufs[k % 2][1].copy_(dp_, non_blocking=True)

            def run_launch(k):
                fb, db = bufs[k % 2]
                hctx.batch_submit_ptr(fb.data_ptr(), W * channels, W * channels * H,
                                      db.data_ptr() if upload_depth else dp_.data_ptr(), W, W * H, depth16=depth16)

            upload(0)
            copy_stream.synchronize()
            run_launch(0); upload(1); hctx.batch_wait(); copy_stream.synchronize()   # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(1, n_h + 1):
                run_launch(k)          # kernels of launch k ...
                upload(k + 1)          # ... while launch k+1 comes up over PCIe
                hres = hctx.batch_wait()
                copy_stream.synchronize()
            torch.cuda.synchronize()
            dt_h = time.perf_counter() - t0
            moved = fp_.numel() * fp_.element_size() + (dp_.numel() * dp_.element_size() if upload_depth else 0)
            hctx.close()
            return {"value": hb * n_h / dt_h, "unit": "frames/s", "pcie_gb_per_s": moved * n_h / dt_h / 1e9,
                    "bytes_per_frame": moved // hb, "pose_success_rate": float(hres["pnp_success"].mean())}

        host_fed = host_fed_rate(frames_h[:hb], depth_h[:hb], C, False, False)
        host_fed.update({"frames_per_launch": hb, "launches": n_h,
                         "how": "BGRA frames + float32 depth in pinned host memory, double-buffered: the frames of the "
                                "next launch go up under the kernels of the current one; the depth planes are not "
                                "uploaded (the back-projection reads its values in place over the link)"})
        host_fed["depth_uploaded"] = host_fed_rate(frames_h[:hb], depth_h[:hb], C, False, True)
        # The same with the compact input format (an extension of this library, vo_batch_submit_depth16): gray 8-bit
        # frames and 16-bit millimetre depth, 3 bytes per pixel instead of 8
        gray_h = np.stack([ctx.gray(fr) for fr in frames_h[:hb]])
        d16_h = np.clip(np.rint(depth_h[:hb]), 0, 65535).astype(np.uint16).view(np.int16)
        host_fed["gray_depth16"] = host_fed_rate(gray_h, d16_h, 1, True, False)
        host_fed["gray_depth16"]["how"] = ("gray 8-bit frames + 16-bit millimetre depth (vo_batch_submit_depth16, an "
                                           "extension: the reference's camera interface delivers BGRA + float32), "
                                           "depth read in place")
        host_fed["gray_depth16"]["depth_uploaded"] = host_fed_rate(gray_h, d16_h, 1, True, True)

    # and on a flat wall (no depth relief at all): every PnP inlier set is planar, so the pose refinement starts from
    # a homography instead of the DLT
    flat = None
    if args.flat_scene:
        # the wall is seen at an angle: a camera that faces a wall squarely measures exactly equal depths, and an
        # exactly coplanar 5-point sample has no EPnP solution (in OpenCV as little as here; real depth is never
        # that flat)
        fin = [build_inputs(S, F, seed=72000 + 1000 * rank + S * k, workers=render_workers, relief=0.0,
                            tilt=(0.08, -0.05)) for k in range(NC)]
        ff = [torch.from_numpy(i[1]).cuda() for i in fin]
        fd_ = [torch.from_numpy(i[2]).cuda() for i in fin]
        for c in ctxs:
            c.batch_reset()
        steps_f = max(3, args.steps // 2)
        f_fps, f_dt, f_res, f_acc = measure(ff, fd_, steps_f, max(2, args.warmup), reps=3)
        flat = {"value": f_fps, "unit": "frames/s", "steps": steps_f, "ms_per_step": 1e3 * f_dt / steps_f,
                "scene": "flat wall seen at an angle (relief 0, slope 0.08 / -0.05), own texture per stream",
                "stage_ms_per_launch": {k: round(v, 4) for k, v in f_acc.items()},
                "mean_matches": float(f_res["n_matches"].mean()),
                "pose_success_rate": float(f_res["pnp_success"].mean())}
        t_err = [float(np.linalg.norm(f_res["tvec"][b] - fin[b // Bc][3][(b % Bc) // F][b % F][1]))
                 for b in range(len(f_res)) if f_res["pnp_success"][b]]
        flat["median_translation_error_mm"] = float(np.median(t_err)) if t_err else None
        del ff, fd_, fin

    if rank == 0:
        n_kp = float(res["n_keypoints"].mean())
        # Algorithmic work of one launch (= the whole batch of Bc frames of one context), per stage.
        # Image stages are priced in HBM bytes: every input byte read once, every output byte written once.
        px0 = W * H
        if levels:
            sizes = [ctx.orb_level_size(l) for l in range(levels)]
            px_up = float(sum(a * b for a, b in sizes[1:]))
            px_all = px0 + px_up
            px_last = sizes[-1][0] * sizes[-1][1]
        else:
            px_up, px_all = 0.0, float(px0)
        work = {
            # level 0: read C bytes/pixel, write the blurred plane (+ the gray plane that feeds the pyramid)
            "frontend_fast_nms_blur": ("hbm", Bc * px0 * (C + 1 + (1 if levels else 0)) / 1e9, HBM_PEAK_GBS, "GB/s"),
            # levels 1..L-1: read the previous level, write the level
            "pyramid_resize": ("hbm", Bc * (px_all - px_last + px_up) / 1e9 if levels else 0.0,
                               HBM_PEAK_GBS, "GB/s"),
            # levels 1..L-1: read the gray level, write its blurred version
            "frontend_upper_levels": ("hbm", Bc * 2 * px_up / 1e9, HBM_PEAK_GBS, "GB/s"),
            # per-level ranking + orientation (full detector): a 9x9 Harris footprint for each of the 2N best FAST
            # corners, the radius-15 disc (709 pixels) for each of the N key points kept
            "select_topn": ("hbm", Bc * (2 * n_kp * 81 + n_kp * 709) / 1e9 if levels else 0.0, HBM_PEAK_GBS, "GB/s"),
            "brief": ("hbm", Bc * n_kp * (512 + 32) / 1e9, HBM_PEAK_GBS, "GB/s"),
            # all-pairs Hamming distance as an int8 matrix product: 256 multiply-adds = 512 operations per
            # descriptor pair on the matrix cores (v_mfma_i32_32x32x32_i8)
            "knn2": ("mfma", Bc * n_kp * n_kp * 512 / 1e12, MFMA_I8_PEAK_TOPS, "TOP/s") if args.matcher == "mfma" else
                    # 8 XOR + 8 popcount-accumulate 32-bit lane operations per descriptor pair
                    ("valu", Bc * n_kp * n_kp * 16 / 1e9, VALU_PEAK_GLOPS, "Glane-op/s"),
        }
        dominant = max(acc, key=acc.get)
        roof = {"kernel": dominant, "ms_per_launch": acc[dominant], "bound": None, "achieved": None,
                "peak": None, "unit": None, "frac": None, "traffic": None}
        if dominant in work:
            bound, amount, peak, unit = work[dominant]
            ach = amount / (acc[dominant] * 1e-3)
            roof.update(bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak)
        # HBM bytes per launch from the PMC passes kept under profiles/ (rocprofv3 cannot run inside this
        # process): used only when they were taken at this launch size; null where the counter is uncalibrated
        stages_pmc, traffic_source = committed_traffic(Bc)
        measured = {k: v.get("bytes") for k, v in stages_pmc.items()}
        if measured.get("select_topn_rank") is not None and measured.get("select_topn_finish") is not None:
            # the stage is two kernels: per-level ranking (Harris footprints) and orientation (radius-15 discs)
            measured["select_topn"] = measured["select_topn_rank"] + measured["select_topn_finish"]
        busy = {k: v.get("valu_busy") for k, v in stages_pmc.items()}
        if busy.get(dominant) is not None:
            # measured vector-ALU occupancy of the dominant kernel (SQ_ACTIVE_INST_VALU pass kept under
            # profiles/): the image kernels are instruction-bound long before they are HBM-bound
            roof["valu_busy"] = busy[dominant]
        roof["traffic"] = measured.get(dominant)
        roof["traffic_source"] = traffic_source
        stage_roof = {}
        for k, (bound, amount, peak, unit) in work.items():
            if acc.get(k, 0.0) > 1e-4 and amount > 0:
                a_ = amount / (acc[k] * 1e-3)
                stage_roof[k] = {"bound": bound, "achieved": a_, "peak": peak, "unit": unit, "frac": a_ / peak}
                if bound == "hbm":
                    stage_roof[k]["algorithmic_bytes"] = amount * 1e9
                    stage_roof[k]["traffic"] = measured.get(k)
        out = {
            "metric": "frames/sec (detect+match+RANSAC pose) @1280x720, 2000 kp" if not kitti else
                      "frames/sec (detect+match+RANSAC pose) @1241x376 BGR, 2000 kp, 8 streams in total",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if kitti else "weak",
            "vs_baseline": None, "dtype": "u8/int32 (image), %s (match), f32 (Harris, angle), f64 (RANSAC)" %
                     ("int8 MFMA -> int32" if args.matcher == "mfma" else "u32 xor/popcount"),
            "data": "synthetic",
            "config": {"inputs": "hbm_resident",
                       "workload": ("kitti_1241x376_bgr_8_streams_total_2000kp_%s_ratio_Fransac_pnp" if kitti else
                                    "hd720_bgra_mono_stream_2000kp_%s_ratio_Fransac_pnp") %
                                   ("orb8level" if levels else "fast1level"),
                       "detector": args.detector, "matcher": args.matcher, "pyramid_levels": levels or 1,
                       "streams_per_gpu": S * NC, "streams_total": 8 if kitti else S * NC * world,
                       "distinct_streams": True, "contexts_per_gpu": NC, "frames_per_stream_per_step": F,
                       "frames_per_step": B, "frames_per_launch": Bc,
                       "max_features": args.features, "fast_threshold": 20,
                       "parallelism": "independent streams, one process per GPU, no collective"},
            "roofline": roof,
            "stage_roofline": stage_roof,
            "stage_ms_per_launch": {k: round(v, 4) for k, v in acc.items()},
            "mean_keypoints": n_kp, "mean_matches": float(res["n_matches"].mean()),
            "pose_success_rate": float(res["pnp_success"].mean()),
        }
        # accuracy of the last step's poses against the rendered camera motion (T_prev_from_curr, mm)
        t_err = [float(np.linalg.norm(res["tvec"][b] - truths[b // Bc][(b % Bc) // F][b % F][1]))
                 for b in range(len(res)) if res["pnp_success"][b]]
        out["median_translation_error_mm"] = float(np.median(t_err)) if t_err else None
        out["ground_truth_step_mm"] = float(np.linalg.norm(truth[0][1][1]))
        out["f_ransac_iterations_mean"] = float(res["f_iters"][res["n_ratio"] > 0].mean())
        if hard is not None:
            out["hard_scene"] = hard
        if flat is not None:
            out["flat_scene"] = flat
        if host_fed is not None:
            out["host_fed_fps"] = host_fed
        if args.facade_latency and world == 1:
            # the path the reference node takes: ONE stream, FrameProcessor::processFrame per frame through the C++
            # facade (tools/facade_bench, its own process), host frames in, every result out by value
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import facade_latency
                lat = {}
                for det, pinned in (("fast", False), ("fast", True), ("orb", False)):
                    r = facade_latency.run(iterations=200, detector=det, pinned=pinned)
                    lat[f"{det}_{'pinned' if pinned else 'pageable'}"] = {
                        "median_ms": r["median_ms"], "p90_ms": r["p90_ms"], "fps": r["fps"],
                        "mean_keypoints": r["mean_keypoints"], "pose_rate": r["pose_rate"]}
                lat["what"] = ("FrameProcessor::processFrame, one HD720 BGRA + float32 depth frame per call from host "
                               "memory, median of 200 frames; fast = the reference's effective detector (FAST, 1000 "
                               "key points), orb = DetectorMode::Orb with 2000 key points / 8 levels")
                out["facade_latency_ms"] = lat
            except Exception as e:  # the tool is optional for the throughput number
                out["facade_latency_ms"] = {"error": str(e)[:200]}
        if world == 1 and args.cpu_frames > 0:
            cpu_fps = cpu_baseline(cam, frames_h[:F], depth_h[:F], args.cpu_frames, args.features, levels)
            if cpu_all is not None:
                out["cpu_baseline_all_cores"] = {
                    "value": cpu_all[0], "unit": "frames/s", "cores": cpu_all[1], "kind": "port",
                    "sample": f"{cpu_all[1]} processes, one independent HD720 stream each, {cpu_all[2]} frames per "
                              "process, oracle C pipeline, total frames over the slowest process"}
            out["cpu_baseline"] = {"value": cpu_fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{args.cpu_frames} HD720 frames of the same stream and detector, "
                                             "oracle C pipeline, 1 thread"}
            if args.cpu_fast:
                # the other detector for the record: the bench's default is the 8-level detector, the reference node
                # runs the one-level FAST sequence (and vice versa with --detector fast)
                other = 0 if levels else 8
                n_other = max(32, args.cpu_frames // (1 if other == 0 else 2))
                out["cpu_baseline_other_detector"] = {
                    "value": cpu_baseline(cam, frames_h[:F], depth_h[:F], n_other, args.features, other),
                    "unit": "frames/s", "cores": 1, "kind": "port", "detector": "orb" if other else "fast",
                    "sample": f"{n_other} HD720 frames of the same stream, oracle C pipeline, 1 thread"}
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
