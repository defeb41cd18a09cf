#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database: per-kernel calls / total / average duration (the
`--stats` table) and, when present, per-kernel sums of the PMC counters.

    python tools/rocpd_summary.py gpurun_out/prof_stats/bench_results.db [--csv out.csv]"""
import argparse
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "").replace("parrot::", "")
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv")
    ap.add_argument("--by-grid", action="store_true", help="one row per (kernel, grid): separates the layer shapes of one kernel")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    if a.by_grid:
        rows = cur.execute("select name, grid_x / workgroup_x, grid_y / workgroup_y, count(*), sum(duration), avg(duration) from kernels "
                           "group by 1, 2, 3 order by 5 desc").fetchall()
        total = sum(r[4] for r in rows) or 1
        pmc = {}
        try:
            for name, gx, gy, cname, val in cur.execute(
                    "select k.name, k.grid_x / k.workgroup_x, k.grid_y / k.workgroup_y, p.counter_name, sum(p.counter_value) from pmc_events p "
                    "join kernels k on p.dispatch_id = k.dispatch_id group by 1, 2, 3, 4"):
                pmc.setdefault((name, gx, gy), {})[cname] = val
        except sqlite3.Error:
            pass
        counters = sorted({c for v in pmc.values() for c in v})
        w = csv.writer(open(a.csv, "w", newline="") if a.csv else sys.stdout)
        w.writerow(["kernel", "wg_x", "wg_y", "calls", "total_ms", "avg_us", "pct"] + [c + "_per_call" for c in counters])
        for name, gx, gy, n, tot, avg in rows:
            w.writerow([short(name), gx, gy, n, round(tot / 1e6, 4), round(avg / 1e3, 2), round(100.0 * tot / total, 2)] +
                       [(pmc[(name, gx, gy)][c] / n if (name, gx, gy) in pmc and c in pmc[(name, gx, gy)] else "") for c in counters])
        return
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    pmc = {}
    try:
        for name, cname, val in cur.execute(
                "select k.name, p.counter_name, sum(p.counter_value) from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id group by 1, 2"):
            pmc.setdefault(name, {})[cname] = val
    except sqlite3.Error:
        pass
    counters = sorted({c for v in pmc.values() for c in v})
    hdr = ["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"] + [c + "_sum" for c in counters] + [c + "_per_call" for c in counters]
    out = []
    for name, n, tot, avg, mn, mx in rows:
        r = [short(name), n, round(tot / 1e6, 4), round(avg / 1e3, 2), round(mn / 1e3, 2), round(mx / 1e3, 2), round(100.0 * tot / total, 2)]
        r += [pmc.get(name, {}).get(c, "") for c in counters]
        r += [(pmc[name][c] / n if name in pmc and c in pmc[name] else "") for c in counters]
        out.append(r)
    w = csv.writer(open(a.csv, "w", newline="") if a.csv else sys.stdout)
    w.writerow(hdr)
    w.writerows(out)


if __name__ == "__main__":
    main()
