#!/usr/bin/env python3
"""Static instruction table of the tile kernels from the compiler's assembly: compiles k_image.hip for gfx950 to
assembly (device side only) and counts, per kernel, the instructions by unit (vector ALU, scalar ALU, LDS, vector
memory, waits / branches) in total and between consecutive s_barrier instructions in program order.  The kernels
hold three staging paths behind block-uniform branches, so the static numbers bound the per-path numbers from
above; the counters under profiles/ (rNN_tile_kernel_phase_counts.txt) give what a wave actually executes.

    python3 tools/asm_phase_table.py > profiles/rNN_tile_kernel_static_counts.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "visual_odometry_amd", "csrc", "k_image.hip")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mllvm",
         "-amdgpu-mfma-vgpr-form", "--cuda-device-only", "-S"]


def unit(m):
    if m.startswith("v_"):
        return "valu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_endpgm", "s_branch", "s_cbranch", "s_setpc", "s_sleep")):
        return "ctl"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k_image.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, SRC], stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    name, body, kernels = None, [], {}
    for line in text:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, body = m.group(1), []
            continue
        if line.startswith(".Lfunc_end") and name:
            kernels[name] = body
            name = None
            continue
        if name is not None:
            s = line.strip()
            if s and not s.startswith((".", ";")) and not s.endswith(":"):
                body.append(s.split()[0])
    cols = ["valu", "salu", "smem", "lds", "vmem", "ctl"]
    print("kernel / stretch between barriers (program order)".ljust(64) + "".join(c.rjust(7) for c in cols))
    for k, ins in kernels.items():
        if "k_frontend" not in k:
            continue
        parts, cur = [], []
        for m in ins:
            cur.append(m)
            if m == "s_barrier":
                parts.append(cur)
                cur = []
        parts.append(cur)

        def row(label, lst):
            c = {x: 0 for x in cols}
            for m in lst:
                u = unit(m)
                if u in c:
                    c[u] += 1
            print(label.ljust(64) + "".join(str(c[x]).rjust(7) for x in cols))
        row(k[:62], ins)
        for i, p in enumerate(parts):
            row(f"    stretch {i}", p)


if __name__ == "__main__":
    main()
