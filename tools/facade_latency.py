#!/usr/bin/env python3
"""Writes a short HD720 BGRA + depth sequence to a temporary dump and runs tools/facade_bench on it: the
single-stream, host-fed latency of FrameProcessor::processFrame.  Prints the tool's JSON line."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(iterations=200, detector="fast", frames=8, seed=2000, timeout=300, pinned=False):
    from visual_odometry_amd import synth
    cam = synth.CONFIG_CAMERAS["hd720"]
    f, d, _ = synth.make_pingpong_batch(cam, 1, frames, seed, channels=4, relief=2.0)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as fh:
        fh.write(np.ascontiguousarray(f).tobytes())
        fh.write(np.ascontiguousarray(d, np.float32).tobytes())
        path = fh.name
    try:
        out = subprocess.run([os.path.join(ROOT, "tools", "facade_bench"), path, str(cam.width), str(cam.height), "4",
                              str(frames), repr(cam.fx), repr(cam.fy), repr(cam.cx), repr(cam.cy), str(iterations),
                              detector] + (["pinned"] if pinned else []), capture_output=True, text=True,
                             timeout=timeout)
    finally:
        os.unlink(path)
    if out.returncode != 0:
        raise RuntimeError(f"facade_bench failed ({out.returncode}): {out.stderr[-500:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    print(json.dumps(run(detector=sys.argv[1] if len(sys.argv) > 1 else "fast", pinned="pinned" in sys.argv[2:])))
