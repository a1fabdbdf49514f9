#!/usr/bin/env python3
"""HBM bytes per launch of the image stages from a FETCH_SIZE / WRITE_SIZE summary (pmcstats.py output of the two
--pmc passes of `python3 bench.py --cpu-frames 0 --steps 3 --warmup 1`): writes the JSON bench.py reads for
`roofline.traffic`, stamped with the SHA-256 of the kernel sources the counters were taken on (bench.py refuses
the file when the tree's sources hash differently).

FETCH_SIZE is doubled for every kernel: a kernel that reads a 512 MiB buffer exactly once reports 262 200 KB with
aligned 16-byte loads, 262 200 KB with 4-byte loads and 270 300 KB with 16-byte loads at 4-byte alignment
(tools/fetch_probe.hip, profiles/r02_fetch_probe.txt) -- the counter tallies the 128-byte requests of gfx950 at
64 bytes whatever the load shape.  WRITE_SIZE reads true in the same probe.

  python3 profiles/traffic.py <pmc_hbm.txt> <frames per launch> <out.json> [<pmc_sq.txt> <kernel_stats.csv>]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(path, frames_per_launch, out, sq_path=None, kstats_path=None):
    from bench import kernel_source_hash
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
    lvl0 = find("k_frontendILi4E")
    steps = lvl0["FETCH_SIZE"][1] if lvl0 else 1          # one level-0 launch per batch
    for stage, frag in (("frontend_fast_nms_blur", "k_frontendILi4E"), ("pyramid_resize", "k_resize_exact"),
                        ("frontend_upper_levels", "k_frontend_levels"), ("brief", "k_brief_steered"),
                        ("knn2", "k_knn2"), ("select_topn_rank", "k_orb_select"), ("select_topn_finish", "k_orb_finish")):
        # (a stage may be several kernels: the ranking kernel runs as two instantiations, one per block size)
        ds = [d for name, d in kb.items() if frag in name and "FETCH_SIZE" in d and "WRITE_SIZE" in d]
        if not ds:
            continue
        fetch = write = 0.0
        launches = 0
        for d in ds:
            n = max(1, d["FETCH_SIZE"][1] // max(1, steps))   # launches per batch (the averages are per launch)
            fetch += 2 * d["FETCH_SIZE"][0] * 1024 * n
            write += d["WRITE_SIZE"][0] * 1024 * n
            launches += n
        stages[stage] = {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                         "note": f"{launches} launch(es) per batch; 2 x FETCH_SIZE + WRITE_SIZE"}
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
        for stage, frag in (("frontend_fast_nms_blur", "k_frontendILi4E"), ("frontend_upper_levels", "k_frontend_levels"),
                            ("knn2", "k_knn2"), ("pyramid_resize", "k_resize_exact"), ("brief", "k_brief_steered")):
            a = next((v for k, v in sq.items() if frag in k), None)
            d = next((v for k, v in dur.items() if frag in k), None)
            if a and d:
                st = stages.setdefault(stage, {"bytes": None})
                st["valu_busy"] = round(4 * a["SQ_ACTIVE_INST_VALU"] / (1024 * 2.4e9 * d), 3)
                if a.get("SQ_WAVES"):
                    st["valu_insts_per_wave"] = round(a["SQ_INSTS_VALU"] / a["SQ_WAVES"], 1)
    json.dump({"frames_per_launch": frames_per_launch, "source": os.path.basename(path),
               "source_hash": kernel_source_hash(), "stages": stages}, open(out, "w"), indent=1)
    print(json.dumps(stages, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3], *(sys.argv[4:6] if len(sys.argv) >= 6 else (None, None)))
