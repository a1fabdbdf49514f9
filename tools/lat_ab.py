"""Single-stream latency through the C++ facade: own staging + helper thread on/off, same box (GPU box)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import facade_latency
if __name__ == "__main__":
    for rep in range(2):
        for stage in ("0", "1"):
            os.environ["VO_STAGE"] = stage
            for det, pinned in (("fast", False), ("fast", True), ("orb", False)):
                r = facade_latency.run(iterations=300, detector=det, pinned=pinned)
                print("stage", stage, det, "pinned" if pinned else "pageable", r["median_ms"], r["p90_ms"], r["pose_rate"], flush=True)
