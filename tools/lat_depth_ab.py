"""Single-stream latency through the C++ facade and the host-fed rate with the depth plane read in place
(VO_DEPTH_IN_PLACE unset) or uploaded (VO_DEPTH_IN_PLACE=0).  GPU box; every configuration runs in a child
process of its own (the switch is read once per process)."""
import json, os, subprocess, sys
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
    for mode in ("1", "0"):
        env = dict(os.environ, VO_DEPTH_IN_PLACE=mode)
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--cpu-frames", "0", "--cpu-workers", "0",
                              "--hard-scene", "0", "--flat-scene", "0", "--facade-latency", "0", "--steps", "20"],
                             env=env, check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        h = d["host_fed_fps"]
        print("in_place", mode, "host_fed", round(h["value"]), "pose", h["pose_success_rate"], "gray+depth16", round(h["gray_depth16"]["value"]),
              "pose", h["gray_depth16"]["pose_success_rate"], flush=True)
