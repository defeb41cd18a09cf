#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 kernel trace (rocpd SQLite) as a timeline: start offset, duration, queue, grid and
name of every kernel between two launches of the step's first kernel (default `tte_embed_kernel`; `voc_embed_kernel` for
the vocoder-only workload).  Used for the single-utterance (B = 1) critical path.

    python tools/b1_timeline.py trace.db [--first tte_embed_kernel] [--step -2]"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("parrot::", "").replace("Sch", "")[:58]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--first", default="tte_embed_kernel")
    ap.add_argument("--step", type=int, default=-2, help="which occurrence of the first kernel starts the step (negative: from the end)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    gcol = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0")
    wcol = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
    rows = cur.execute(f"select name, start, end, {qcol}, {gcol}, {wcol} from kernels order by start").fetchall()
    firsts = [i for i, r in enumerate(rows) if a.first in r[0]]
    i0 = firsts[a.step]
    i1 = firsts[a.step + 1] if a.step + 1 < 0 or a.step + 1 < len(firsts) and a.step >= 0 else len(rows)
    step = rows[i0:i1]
    t0 = step[0][1]
    queues = {q: n for n, q in enumerate(sorted({r[3] for r in step}))}
    print(f"# {len(step)} kernels, wall {(max(r[2] for r in step) - t0) / 1e3:.1f} us, kernel sum {sum(r[2] - r[1] for r in step) / 1e3:.1f} us")
    print("# start_us  dur_us  q  workgroups  kernel")
    prev_end = t0
    for name, s, e, q, g, w in step:
        wg = (g // w) if w else g
        gap = (s - prev_end) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {queues[q]}  {wg:6d}  {short(name)}" + (f"   [idle {gap:.1f}]" if gap > 3 else ""))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
