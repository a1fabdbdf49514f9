#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace results database
(the same numbers `--stats` prints): name, calls, total/avg/min/max in microseconds."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         f"max(d.end-d.start) from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name "
         f"order by 3 desc")
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows)
    print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
    for name, n, s, a, mn, mx in rows:
        name = name.split("(")[0][:70]
        print(f'"{name}",{n},{s/1e3:.1f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*s/tot:.2f}')


if __name__ == "__main__":
    main(sys.argv[1])
