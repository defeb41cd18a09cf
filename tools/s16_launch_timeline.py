#!/usr/bin/env python3
"""Workgroup timeline of ONE conv_split16 launch (experiment build with -DS16_TRACE): when and where (XCD / SE / CU) every
workgroup ran -- rounds, tail, per-CU balance.
    tools/build_exp.sh s16trace -DS16_TRACE
    PARROT_HIP_LIB=build_exp/libparrot_s16trace.so python tools/s16_launch_timeline.py [--layer mrf0_k11d5] [--batch 64]"""
import argparse
import collections
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from microbench_conv import LAYERS  # noqa: E402
from parrot_tts_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="mrf0_k11d5")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--dump", default="", help="write the raw records (start, end in 10 ns ticks, HW_ID, XCC_ID per workgroup) to this .npy")
    a = ap.parse_args()
    name, cin, cout, k, dil, T = next(l for l in LAYERS if l[0] == a.layer)
    dev = "cuda:0"
    w = torch.randn(cout, cin, k) / (cin * k) ** 0.5
    b = torch.randn(cout) * 0.1
    x = torch.randn(a.batch, cin, T, device=dev)
    res = torch.randn(a.batch, cout, T, device=dev)
    out = torch.empty(a.batch, cout, T, device=dev)
    plan = ops.ConvPlan(w, b, dilation=dil, padding=dil * (k - 1) // 2, pre_act=1, pre_slope=0.1, tile_cfg=a.tile)
    lib = _lib.lib()
    for _ in range(20):  # (settle the clocks)
        plan(x, res, out=out)
    torch.cuda.synchronize()
    assert lib.parrot_debug_s16_wg_clear() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    plan(x, res, out=out)
    e1.record()
    torch.cuda.synchronize()
    n = 16384
    buf = (C.c_ulonglong * (4 * n))()
    lib.parrot_debug_s16_wg.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.parrot_debug_s16_wg(buf, n) == 0
    r = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4)
    if a.dump:
        np.save(a.dump, r[: int(np.nonzero(r[:, 0])[0].max()) + 1])
    r = r[r[:, 0] != 0]
    t0 = int(r[:, 0].min())
    st, en = (r[:, 0].astype(np.int64) - t0) / 100.0, (r[:, 1].astype(np.int64) - t0) / 100.0  # us
    hw, xcc = r[:, 2].astype(np.int64), r[:, 3].astype(np.int64) & 15
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    span = float(en.max())
    dur = en - st
    per_cu = collections.Counter(cu.tolist())
    order = np.argsort(st)
    q = len(order) // 4
    # concurrency profile: workgroups in flight at 20 points of the span
    pts = np.linspace(0, span, 21)[1:-1]
    conc = [int(((st <= p) & (en > p)).sum()) for p in pts]
    busy_cu = {}
    for c in per_cu:
        m = cu == c
        busy_cu[c] = float(dur[m].sum())
    print(json.dumps({
        "layer": name, "batch": a.batch, "event_us": round(e0.elapsed_time(e1) * 1e3, 1), "workgroups": int(len(r)), "span_us": round(span, 1),
        "cus_used": len(per_cu), "xcds_used": int(len(set(xcc.tolist()))), "wg_per_cu_min_max": [min(per_cu.values()), max(per_cu.values())],
        "wg_us_mean": round(float(dur.mean()), 1), "wg_us_first_quarter_by_start": round(float(dur[order[:q]].mean()), 1),
        "wg_us_last_quarter_by_start": round(float(dur[order[-q:]].mean()), 1),
        "mean_concurrency": round(float(dur.sum()) / span, 1), "concurrency_at_5pct_steps": conc,
        "last_start_us": round(float(st.max()), 1),
        "cu_busy_wg_us_min_mean_max": [round(min(busy_cu.values()), 1), round(sum(busy_cu.values()) / len(busy_cu), 1), round(max(busy_cu.values()), 1)],
    }))


if __name__ == "__main__":
    main()
