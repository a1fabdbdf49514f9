#!/usr/bin/env python3
"""HBM bytes per launch of the image stages from a FETCH_SIZE / WRITE_SIZE summary (pmcstats.py output of the two
--pmc passes of `python3 bench.py --cpu-frames 0 --steps 3 --warmup 1`): writes the JSON bench.py reads for
`roofline.traffic`.  FETCH_SIZE is doubled for the kernels whose loads are 16 bytes per lane (the level-0 tile
kernel on BGRA input, the resize kernel), as the counter tallies 128-byte requests at 64 bytes on gfx950.  The
gray-input instantiation of the tile kernel is taken as read: run on a gray copy of the BGRA frames (same tiles,
a quarter of the bytes) it reads 0.483 of the BGRA run's raw value, not 0.25, and 1.37 x the plane's bytes, which
is the apron re-read its 128x32 tiles imply (tools/calib_fetch.py, profiles/calib.sh).  Other kernels with narrow
loads are left uncalibrated (bytes = null)."""
import json
import sys


def main(path, frames_per_launch, out, sq_path=None, kstats_path=None):
    kb = {}
    for line in open(path):
        parts = line.split()
        if len(parts) < 3 or "=" not in parts[-1]:
            continue
        name, n = parts[0], int(parts[1])
        k, v = parts[-1].split("=")
        kb.setdefault(name, {})[k] = (float(v), n)

    def find(frag):
        for name, d in kb.items():
            if frag in name:
                return d
        return None

    stages = {}
    d = find("k_frontendILi4E")
    if d:
        stages["frontend_fast_nms_blur"] = {
            "bytes": (2 * d["FETCH_SIZE"][0] + d["WRITE_SIZE"][0]) * 1024,
            "note": "2 x FETCH_SIZE (16-byte loads) + WRITE_SIZE, one launch"}
    d = find("k_resize_exact")
    if d:
        n = d["FETCH_SIZE"][1] // max(1, find("k_frontendILi4E")["FETCH_SIZE"][1])  # launches per batch
        stages["pyramid_resize"] = {
            "bytes": (2 * d["FETCH_SIZE"][0] + d["WRITE_SIZE"][0]) * 1024 * n,
            "note": f"{n} launches; 2 x FETCH_SIZE (16-byte loads) + WRITE_SIZE"}
    d = find("k_frontendILi1E")
    if d:
        n = d["FETCH_SIZE"][1] // max(1, find("k_frontendILi4E")["FETCH_SIZE"][1])
        stages["frontend_upper_levels"] = {
            "bytes": (d["FETCH_SIZE"][0] + d["WRITE_SIZE"][0]) * 1024 * n,
            "fetch_bytes": d["FETCH_SIZE"][0] * 1024 * n, "write_bytes": d["WRITE_SIZE"][0] * 1024 * n,
            "note": f"{n} launches; FETCH_SIZE as read + WRITE_SIZE: on 1-byte gray input the counter reads the staged "
                    "bytes themselves (calibrated against the BGRA instantiation on the same tiles, "
                    "r01_fetch_calibration.txt)"}
    # vector-ALU occupancy of the same kernels: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs, so
    # busy = 4 * count / (1024 SIMDs * 2.4 GHz * average kernel duration from the kernel trace)
    if sq_path and kstats_path:
        sq = {}
        for line in open(sq_path):
            parts = line.split()
            if len(parts) > 2:
                sq[parts[0]] = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in parts[2:] if "=" in kv}
        dur = {}
        for line in open(kstats_path):
            if line.startswith('"'):
                name, rest = line[1:].split('",', 1)
                dur[name] = float(rest.split(",")[2]) * 1e-6  # avg_us
        for stage, frag in (("frontend_fast_nms_blur", "k_frontendILi4E"), ("frontend_upper_levels", "k_frontendILi1E"),
                            ("knn2", "k_knn2"), ("pyramid_resize", "k_resize_exact"), ("brief", "k_brief")):
            a = next((v for k, v in sq.items() if frag in k), None)
            d = next((v for k, v in dur.items() if frag in k), None)
            if a and d:
                stages.setdefault(stage, {"bytes": None})["valu_busy"] = round(
                    4 * a["SQ_ACTIVE_INST_VALU"] / (1024 * 2.4e9 * d), 3)
    json.dump({"frames_per_launch": frames_per_launch, "source": path, "stages": stages}, open(out, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], *(sys.argv[4:6] if len(sys.argv) >= 6 else (None, None)))
