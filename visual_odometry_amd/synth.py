"""Seeded synthetic camera sequences for tests and benchmarks.

A textured height-field scene (value noise plus random rectangles and discs)
is viewed by a pinhole camera that moves a little between frames, so every
frame pair has a known relative pose, a consistent epipolar geometry and a
float32 depth map in millimetres -- the same kind of input the reference node
receives from the ZED camera (reference src/zed_interface.cpp:66-94: BGRA8
left image + float32 depth in mm).
"""
from __future__ import annotations

import dataclasses

import numpy as np

__all__ = ["Camera", "Scene", "make_sequence", "make_stereo_sequence", "make_streams", "pingpong",
           "make_pingpong_batch", "CONFIG_CAMERAS"]


@dataclasses.dataclass(frozen=True)
class Camera:
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float


# intrinsics used by the benchmark configurations
CONFIG_CAMERAS = {
    "vga": Camera(640, 480, 525.0, 525.0, 319.5, 239.5),
    "hd720": Camera(1280, 720, 700.0, 700.0, 639.5, 359.5),
    "hd1080": Camera(1920, 1080, 1050.0, 1050.0, 959.5, 539.5),
    "kitti": Camera(1241, 376, 718.856, 718.856, 607.19, 185.22),
}


def _value_noise(rng, shape, period):
    gh, gw = shape[0] // period + 3, shape[1] // period + 3
    grid = rng.random((gh, gw), dtype=np.float32)
    ys = np.arange(shape[0], dtype=np.float32) / period
    xs = np.arange(shape[1], dtype=np.float32) / period
    y0 = ys.astype(np.int32)
    x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    fy = fy * fy * (3 - 2 * fy)
    fx = fx * fx * (3 - 2 * fx)
    g00 = grid[y0][:, x0]
    g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]
    g11 = grid[y0 + 1][:, x0 + 1]
    return (g00 * (1 - fx) + g01 * fx) * (1 - fy) + (g10 * (1 - fx) + g11 * fx) * fy


class Scene:
    """Textured surface z = Z(u, v) (millimetres) with texture T(u, v)."""

    MM_PER_TEXEL = 4.0

    def __init__(self, seed: int, extent_mm=(16000.0, 10000.0), n_shapes: int = 900, relief: float = 1.0,
                 tilt=(0.0, 0.0)):
        self.relief = relief  # scales the height variation of the surface (1.0: +-1150 mm around 3.5 m)
        self.tilt = tilt      # slope of the mean plane in u and v (0, 0: the camera's first pose faces it squarely)
        rng = np.random.default_rng(seed)
        tw = int(extent_mm[0] / self.MM_PER_TEXEL)
        th = int(extent_mm[1] / self.MM_PER_TEXEL)
        tex = np.full((th, tw), 128.0, dtype=np.float32)
        for period, amp in ((64, 40.0), (16, 36.0), (5, 30.0)):
            tex += amp * (_value_noise(rng, (th, tw), period) - 0.5) * 2.0
        for _ in range(n_shapes):
            cx, cy = rng.integers(0, tw), rng.integers(0, th)
            w, h = rng.integers(6, 90), rng.integers(6, 90)
            val = float(rng.integers(10, 246))
            if rng.random() < 0.5:
                tex[max(0, cy - h):cy + h, max(0, cx - w):cx + w] = val
            else:
                r = int(min(w, h))
                yy, xx = np.ogrid[-r:r + 1, -r:r + 1]
                y0, x0 = max(0, cy - r), max(0, cx - r)
                sub = tex[y0:cy + r + 1, x0:cx + r + 1]
                m = (yy * yy + xx * xx <= r * r)[(y0 - (cy - r)):(y0 - (cy - r)) + sub.shape[0],
                                                 (x0 - (cx - r)):(x0 - (cx - r)) + sub.shape[1]]
                sub[m] = val
        self.tex = np.clip(tex, 0, 255)
        self.extent = extent_mm
        self.seed = seed

    def surface(self, u, v):
        z = 3500.0 + self.relief * (900.0 * np.sin(u / 1500.0) * np.cos(v / 1200.0)
                                    + 250.0 * np.sin(u / 500.0 + v / 700.0))
        if self.tilt[0] or self.tilt[1]:
            z = z + self.tilt[0] * u + self.tilt[1] * v
        return z

    def sample(self, u, v):
        tx = (u + self.extent[0] / 2) / self.MM_PER_TEXEL
        ty = (v + self.extent[1] / 2) / self.MM_PER_TEXEL
        th, tw = self.tex.shape
        tx = np.clip(tx, 0, tw - 1.001)
        ty = np.clip(ty, 0, th - 1.001)
        x0 = tx.astype(np.int32)
        y0 = ty.astype(np.int32)
        fx = (tx - x0).astype(np.float32)
        fy = (ty - y0).astype(np.float32)
        t = self.tex
        return ((t[y0, x0] * (1 - fx) + t[y0, x0 + 1] * fx) * (1 - fy)
                + (t[y0 + 1, x0] * (1 - fx) + t[y0 + 1, x0 + 1] * fx) * fy)

    def render(self, cam: Camera, R: np.ndarray, c: np.ndarray):
        """Render gray (float32) and depth (float32 mm) for world->camera Xc = R (Xw - c)."""
        xs = (np.arange(cam.width, dtype=np.float64) - cam.cx) / cam.fx
        ys = (np.arange(cam.height, dtype=np.float64) - cam.cy) / cam.fy
        dx, dy = np.meshgrid(xs, ys)
        Rt = R.T
        wx = Rt[0, 0] * dx + Rt[0, 1] * dy + Rt[0, 2]
        wy = Rt[1, 0] * dx + Rt[1, 1] * dy + Rt[1, 2]
        wz = Rt[2, 0] * dx + Rt[2, 1] * dy + Rt[2, 2]
        s = np.full_like(dx, 3500.0)
        for _ in range(12):
            u = c[0] + s * wx
            v = c[1] + s * wy
            z = c[2] + s * wz
            s = s + 0.8 * (self.surface(u, v) - z) / wz
        u = c[0] + s * wx
        v = c[1] + s * wy
        return self.sample(u, v).astype(np.float32), s.astype(np.float32)


def _rot(rx, ry, rz):
    cx_, sx = np.cos(rx), np.sin(rx)
    cy_, sy = np.cos(ry), np.sin(ry)
    cz_, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz_, -sz, 0], [sz, cz_, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _movers(scene: Scene, cam: Camera, n: int, seed: int):
    """n textured rectangles that slide over the image on their own (objects moving independently of the
    camera): per mover its texture patch, start position, velocity in px per frame and depth in mm."""
    rng = np.random.default_rng(seed + 104729)
    th, tw = scene.tex.shape
    out = []
    for _ in range(n):
        w = int(rng.integers(cam.width // 10, cam.width // 4))
        h = int(rng.integers(cam.height // 8, cam.height // 3))
        ty, tx = int(rng.integers(0, th - h)), int(rng.integers(0, tw - w))
        pos = np.array([rng.uniform(0, cam.width - w), rng.uniform(0, cam.height - h)])
        ang = rng.uniform(0, 2 * np.pi)
        vel = rng.uniform(4.0, 14.0) * np.array([np.cos(ang), np.sin(ang)])
        out.append((scene.tex[ty:ty + h, tx:tx + w], pos, vel, float(rng.uniform(1200.0, 2600.0))))
    return out


def make_sequence(cam: Camera, n_frames: int, seed: int, channels: int = 4,
                  noise_sigma: float = 2.0, scene: Scene | None = None, baseline_mm: float = 0.0,
                  movers: int = 0):
    """Return dict with 'frames' (n,H,W,C uint8 BGR[A] or n,H,W gray when channels==1),
    'depth' (n,H,W float32 mm), 'poses' (list of (R, c)) and 'camera'.  movers > 0 pastes that many
    independently moving textured rectangles over the rendered scene (their key points match from frame to
    frame but violate the camera's epipolar geometry: outliers for the RANSAC stages)."""
    scene = scene or Scene(seed)
    rng = np.random.default_rng(seed + 7919)
    sprites = _movers(scene, cam, movers, seed) if movers > 0 else []
    frames, depths, poses = [], [], []
    for k in range(n_frames):
        R = _rot(np.deg2rad(0.10 * k), np.deg2rad(0.25 * k), np.deg2rad(0.12 * k))
        # baseline_mm moves the camera centre along its own x axis (the right camera of a stereo rig)
        c = np.array([14.0 * k, 4.0 * k, 30.0 * k]) + R.T @ np.array([baseline_mm, 0.0, 0.0])
        gray, depth = scene.render(cam, R, c)
        for tex, pos, vel, z in sprites:
            x0, y0 = (int(round(v)) for v in pos + k * vel)
            xa, ya = max(x0, 0), max(y0, 0)
            xb, yb = min(x0 + tex.shape[1], cam.width), min(y0 + tex.shape[0], cam.height)
            if xa < xb and ya < yb:
                gray[ya:yb, xa:xb] = tex[ya - y0:yb - y0, xa - x0:xb - x0]
                depth[ya:yb, xa:xb] = z
        gray = gray + rng.normal(0.0, noise_sigma, gray.shape).astype(np.float32)
        g8 = np.clip(np.rint(gray), 0, 255).astype(np.uint8)
        if channels == 1:
            frames.append(g8)
        else:
            img = np.empty(g8.shape + (channels,), dtype=np.uint8)
            d = rng.integers(-3, 4, size=g8.shape + (2,), dtype=np.int16)
            img[..., 1] = g8
            img[..., 0] = np.clip(g8.astype(np.int16) + d[..., 0], 0, 255).astype(np.uint8)
            img[..., 2] = np.clip(g8.astype(np.int16) + d[..., 1], 0, 255).astype(np.uint8)
            if channels == 4:
                img[..., 3] = 255
            frames.append(img)
        depths.append(depth)
        poses.append((R, c))
    return {"frames": np.stack(frames), "depth": np.stack(depths), "poses": poses,
            "camera": cam}


def _stream_job(job):
    cam, n_frames, seed, channels, relief, movers, tilt = job
    seq = make_sequence(cam, n_frames, seed, channels, scene=Scene(seed, relief=relief, tilt=tilt), movers=movers)
    return seq["frames"], seq["depth"], seq["poses"]


def make_streams(cam: Camera, n_streams: int, n_frames: int, seed: int, channels: int = 4, relief: float = 1.0,
                 movers: int = 0, workers: int = 1, tilt=(0.0, 0.0)):
    """n_streams independent sequences (stream s: scene and noise of seed + s), rendered by `workers` processes.
    Returns a list of dicts like make_sequence's.  The workers are fresh interpreters (spawn), so this may be
    called from a process that already uses the GPU."""
    jobs = [(cam, n_frames, seed + s, channels, relief, movers, tilt) for s in range(n_streams)]
    if workers > 1 and n_streams > 1:
        import concurrent.futures as cf
        import multiprocessing as mp
        with cf.ProcessPoolExecutor(max_workers=min(workers, n_streams), mp_context=mp.get_context("spawn")) as ex:
            res = list(ex.map(_stream_job, jobs))
    else:
        res = [_stream_job(j) for j in jobs]
    return [{"frames": f, "depth": d, "poses": p, "camera": cam} for f, d, p in res]


def pingpong(n):
    """0, 1, ..., n-1, n-2, ..., 1: every consecutive pair (cyclically) is one camera step apart."""
    return list(range(n)) + list(range(n - 2, 0, -1))


def make_pingpong_batch(cam: Camera, n_streams: int, frames_per_stream: int, seed: int, channels: int = 4,
                        relief: float = 1.0, movers: int = 0, workers: int = 1, tilt=(0.0, 0.0)):
    """One launch of the batched pipeline: n_streams independent streams (seed + s), each walking forth and
    back over frames_per_stream / 2 + 1 rendered frames so that the launch can be repeated without a break in
    any stream.  Returns frames (S*F, H, W[, C]) and depth (S*F, H, W) in batch order (stream-major), and per
    stream the ground-truth T_prev_from_curr of each of its F frames."""
    assert frames_per_stream % 2 == 0
    n_distinct = frames_per_stream // 2 + 1
    order = pingpong(n_distinct)
    assert len(order) == frames_per_stream
    seqs = make_streams(cam, n_streams, n_distinct, seed, channels, relief, movers, workers, tilt)
    frames = np.concatenate([q["frames"][order] for q in seqs])
    depth = np.concatenate([q["depth"][order] for q in seqs])
    truth = [[relative_pose(q["poses"][order[k - 1]], q["poses"][order[k]]) for k in range(frames_per_stream)]
             for q in seqs]
    return frames, depth, truth


def make_stereo_sequence(cam: Camera, n_frames: int, seed: int, channels: int = 4, baseline_mm: float = 120.0):
    """Left and right sequences of a rectified stereo rig looking at the same scene."""
    scene = Scene(seed)
    left = make_sequence(cam, n_frames, seed, channels, scene=scene)
    right = make_sequence(cam, n_frames, seed + 1, channels, scene=scene, baseline_mm=baseline_mm)
    return left, right


def relative_pose(pose_prev, pose_curr):
    """T_prev_from_curr: X_prev = R X_curr + t (what the reference's PnP returns,
    reference src/vo_node.cpp:698-706)."""
    Rp, cp = pose_prev
    Rc, cc = pose_curr
    R = Rp @ Rc.T
    t = Rp @ (cc - cp)
    return R, t
