#!/usr/bin/env python3
"""Per-kernel sums of the PMC counters in a rocprofv3 --pmc results database:
kernel, dispatches, then counter=value-per-dispatch pairs."""
import sqlite3
import sys
from collections import defaultdict


def main(path, pattern=""):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    def tab(prefix):
        return [t for t in tabs if t.startswith(prefix)][0]
    disp, sym = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
    pmc_ev, pmc_info = tab("rocpd_pmc_event"), tab("rocpd_info_pmc")
    names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {sym}")}
    cnames = {r[0]: r[1] for r in db.execute(f"select id, name from {pmc_info}")}
    kid = {r[0]: r[1] for r in db.execute(f"select id, kernel_id from {disp}")}
    cols = [r[1] for r in db.execute(f"pragma table_info({pmc_ev})")]
    key = "event_id" if "event_id" in cols else cols[1]
    # pmc events reference dispatch events; map through the dispatch table's event_id when present
    dcols = [r[1] for r in db.execute(f"pragma table_info({disp})")]
    ev2disp = {r[1]: r[0] for r in db.execute(f"select id, event_id from {disp}")} if "event_id" in dcols else None
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for ev, pmc_id, value in db.execute(f"select {key}, pmc_id, value from {pmc_ev}"):
        d = ev2disp.get(ev) if ev2disp else ev
        if d is None or d not in kid:
            continue
        k = names[kid[d]].split("(")[0]
        acc[k][cnames[pmc_id]] += value
        ndisp[k].add(d)
    for k in sorted(acc, key=lambda k: -max(acc[k].values())):
        if pattern and pattern not in k:
            continue
        n = len(ndisp[k])
        print(k[:60], n, " ".join(f"{c}={v / n:.4g}" for c, v in sorted(acc[k].items())))


if __name__ == "__main__":
    main(*sys.argv[1:])
