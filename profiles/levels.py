#!/usr/bin/env python3
"""Per-launch-shape statistics of one kernel from a rocprofv3 results database: the tile kernel runs once per
pyramid level with a different grid, so grouping by grid separates the levels.
  levels.py <db> <kernel substring>        durations (kernel-trace db) and, when present, PMC counters per wave"""
import sqlite3
import sys
from collections import defaultdict


def main(path, pattern):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    def tab(prefix):
        t = [t for t in tabs if t.startswith(prefix)]
        return t[0] if t else None
    disp, sym = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    dcols = [r[1] for r in db.execute(f"pragma table_info({disp})")]
    gcols = [c for c in dcols if "grid" in c]
    names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {sym}")}
    rows = list(db.execute(f"select id, kernel_id, end - start, {', '.join(gcols)}"
                           f"{', event_id' if 'event_id' in dcols else ''} from {disp}"))
    shape = {}
    dur = defaultdict(list)
    ev2disp = {}
    for r in rows:
        if pattern not in names[r[1]]:
            continue
        g = tuple(r[3:3 + len(gcols)])
        shape[r[0]] = g
        dur[g].append(r[2])
        if "event_id" in dcols:
            ev2disp[r[-1]] = r[0]
    pmc_ev, pmc_info = tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    acc = defaultdict(lambda: defaultdict(float))
    if pmc_ev:
        cnames = {r[0]: r[1] for r in db.execute(f"select id, name from {pmc_info}")}
        cols = [r[1] for r in db.execute(f"pragma table_info({pmc_ev})")]
        key = "event_id" if "event_id" in cols else cols[1]
        for ev, pmc_id, value in db.execute(f"select {key}, pmc_id, value from {pmc_ev}"):
            d = ev2disp.get(ev, ev)
            if d in shape:
                acc[shape[d]][cnames[pmc_id]] += value
    print("grid", gcols)
    for g in sorted(dur, key=lambda g: -sum(dur[g])):
        n = len(dur[g])
        line = f"{g} launches={n} avg_us={sum(dur[g]) / n / 1e3:.1f}"
        if g in acc:
            a = acc[g]
            w = a.get("SQ_WAVES", 0) or 1
            line += " " + " ".join(f"{c}/wave={v / w:.1f}" for c, v in sorted(a.items()) if c != "SQ_WAVES")
            line += f" waves/launch={w / n:.0f}"
        print(line)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
