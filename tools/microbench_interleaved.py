#!/usr/bin/env python3
"""Like microbench_conv.py, but the layers are launched round-robin (as the pipeline does: every launch meets
cold weights / a different activation tensor) and timed per launch with HIP events.
    python tools/microbench_interleaved.py --layers mrf0_k3,mrf0_k7d3,mrf0_k11d5 --rounds 100"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import ops  # noqa: E402
from tools.microbench_conv import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=100)
    ap.add_argument("--layers", type=str, default="mrf0_k3,mrf0_k7d3,mrf0_k11d5")
    ap.add_argument("--precision", type=int, default=1)
    ap.add_argument("--chain", type=int, default=1, help="1: each launch reads the previous launch's output (same shapes only)")
    a = ap.parse_args()
    dev = "cuda:0"
    want = a.layers.split(",")
    plans = []
    for name, cin, cout, k, dil, T in LAYERS:
        if name not in want:
            continue
        w = torch.randn(cout, cin, k) / (cin * k) ** 0.5
        b = torch.randn(cout) * 0.1
        plan = ops.ConvPlan(w, b, dilation=dil, padding=dil * (k - 1) // 2, pre_act=1, pre_slope=0.1, precision=a.precision)
        bufs = [torch.randn(a.batch, cin, T, device=dev) * 0.5 for _ in range(3)]
        plans.append((name, plan, bufs, 2.0 * a.batch * cout * cin * k * T))
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plans] for _ in range(a.rounds)]
    for r in range(-5, a.rounds):
        for i, (name, plan, bufs, fl) in enumerate(plans):
            x, res, out = bufs[r % 3], bufs[(r + 1) % 3], bufs[(r + 2) % 3]
            if r >= 0:
                ev[r][i][0].record()
            plan(x, res, out=out)
            if r >= 0:
                ev[r][i][1].record()
    torch.cuda.synchronize()
    for i, (name, plan, bufs, fl) in enumerate(plans):
        ts = sorted(ev[r][i][0].elapsed_time(ev[r][i][1]) for r in range(a.rounds))
        med = ts[len(ts) // 2]
        print(json.dumps({"layer": name, "median_ms": round(med, 4), "min_ms": round(ts[0], 4), "tflops_median": round(fl / med / 1e9, 1)}))


if __name__ == "__main__":
    main()
