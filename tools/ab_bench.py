"""Run bench.py against another build of the library: python tools/ab_bench.py <libvo_hip.so> [bench flags].

For A/B timing of two builds on the same GPU box in one session (clocks differ between boxes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import visual_odometry_amd.capi as capi  # noqa: E402

if __name__ == "__main__":
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
    sys.argv = [sys.argv[0]] + sys.argv[2:]
    import bench  # noqa: E402

    bench.main()
