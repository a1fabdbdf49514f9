"""Single-stream latency through the C++ facade with the depth plane read in place (VO_DEPTH_IN_PLACE unset) or
uploaded on a second stream (VO_DEPTH_IN_PLACE=0).  GPU box; every configuration runs in a child process of its
own (the switch is read once per process).  (The host-fed rates of both arrangements are on the bench line:
host_fed_fps and host_fed_fps.depth_uploaded.)"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.path.insert(0, HERE)
        import facade_latency
        for det, pinned in (("fast", False), ("fast", True), ("orb", False)):
            r = facade_latency.run(iterations=300, detector=det, pinned=pinned)
            print(sys.argv[1], det, "pinned" if pinned else "pageable", r["median_ms"], r["p90_ms"], r["pose_rate"], flush=True)
        sys.exit(0)
    for rep in range(2):
        for mode in ("1", "0"):
            env = dict(os.environ, VO_DEPTH_IN_PLACE=mode)
            subprocess.run([sys.executable, "-u", os.path.abspath(__file__), f"in_place {mode}"], env=env, check=True)
