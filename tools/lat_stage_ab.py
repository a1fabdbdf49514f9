"""Single-stream latency through the C++ facade: helper threads of the staging pool, graph replay
(GPU box).  Every configuration runs in a child process of its own (the switches are read once per process)."""
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
        for threads, graph in (("1", "0"), ("0", "0"), ("3", "0"), ("1", "1")):
            env = dict(os.environ, VO_STAGE_THREADS=threads, VO_GRAPH=graph)
            subprocess.run([sys.executable, "-u", os.path.abspath(__file__), f"helpers {threads} graph {graph}"], env=env, check=True)
