#!/usr/bin/env python3
"""Timeline digest of a rocprofv3 kernel trace (rocpd SQLite): per step of a steady loop, how much of the wall time had a kernel
running, how much of the kernel time overlapped (several streams), and the idle gaps between kernels.

    python tools/overlap_report.py trace.db [--skip 0.5]      (--skip: leading fraction of the kernels to drop = set-up and warm-up)"""
import argparse
import json
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--skip", type=float, default=0.5)
    ap.add_argument("--exclude", default="mfma_ceiling", help="substring of kernel names to leave out")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = cur.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
    rows = [r for r in rows if a.exclude not in r[0]]
    rows = rows[int(len(rows) * a.skip):]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    busy, cur_end, gaps = 0, rows[0][1], []
    for r in rows:
        s, e = r[1], r[2]
        if s > cur_end:
            gaps.append(s - cur_end)
            cur_end = s
        if e > cur_end:
            busy += e - cur_end
            cur_end = e
    ksum = sum(r[2] - r[1] for r in rows)
    gaps.sort()
    out = {"kernels": len(rows), "wall_ms": (t1 - t0) / 1e6, "busy_ms": busy / 1e6, "kernel_sum_ms": ksum / 1e6,
           "busy_frac": busy / (t1 - t0), "overlap_factor": ksum / busy, "gaps": len(gaps), "gap_total_ms": sum(gaps) / 1e6,
           "gap_median_us": gaps[len(gaps) // 2] / 1e3 if gaps else 0, "gap_p90_us": gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0,
           "gaps_over_20us": sum(1 for g in gaps if g > 20000), "gaps_over_20us_ms": sum(g for g in gaps if g > 20000) / 1e6}
    if qcol:
        out["queues"] = len({r[3] for r in rows})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
