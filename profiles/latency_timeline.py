#!/usr/bin/env python3
"""Timeline of the last processFrame call in a rocprofv3 --kernel-trace --memory-copy-trace database of
tools/facade_bench: start and end of every copy and kernel in microseconds from the frame's first event."""
import sqlite3
import sys


def main(path, frames_back=2):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
    ev = []
    for name, s, e in db.execute(f"select s.kernel_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id"):
        ev.append((s, e, "K " + name.split("(")[0][:48]))
    if mc:
        cols = [r[1] for r in db.execute(f"pragma table_info({mc[0]})")]
        size_col = "size" if "size" in cols else None
        q = f"select start, end{', ' + size_col if size_col else ''} from {mc[0]}"
        for row in db.execute(q):
            ev.append((row[0], row[1], "C copy %s B" % (row[2] if size_col else "?")))
    ev.sort()
    # a frame ends with the copy behind its last kernel (k_finish_packed; k_carry without packed output): the events
    # between those kernels of two consecutive frames
    carries = [i for i, x in enumerate(ev) if "k_finish_packed" in x[2] or "k_carry" in x[2]]
    if len(carries) >= frames_back + 2:
        i0 = carries[-2 - frames_back] + 1
        i1 = carries[-1 - frames_back] + 1
        # the device-to-host copies of the earlier frame come right behind its carry kernel
        while i0 < i1 and ev[i0][0] - ev[carries[-2 - frames_back]][1] < 20000:
            i0 += 1
    else:
        i0, i1 = 0, len(ev)
    t0 = ev[i0][0]
    for s, e, n in ev[i0:i1]:
        print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f}  {n}")


if __name__ == "__main__":
    main(sys.argv[1])
