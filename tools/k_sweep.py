#!/usr/bin/env python3
"""Fixed cost of a layer launch: the wide plain-conv kernel (conv_split16, k in {7, 9, 11}) timed at B = 64 on the MRF shapes of
stages 0-2 for every tap count; a line fit t = a + b k separates the per-launch cost that does not scale with K (slab staging of
the first chunk, residual / output epilogue, launch tail) from the MFMA time.   python tools/k_sweep.py [--batch 64]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda:0"
    for C, T in ((256, 1280), (128, 5120), (64, 20480)):
        x = torch.randn(a.batch, C, T, device=dev)
        res = torch.randn(a.batch, C, T, device=dev)
        out = torch.empty(a.batch, C, T, device=dev)
        pts = []
        for k in (7, 9, 11):
            for with_res in (False, True):
                w = torch.randn(C, C, k) / (C * k) ** 0.5
                plan = ops.ConvPlan(w, torch.randn(C) * 0.1, dilation=1, padding=(k - 1) // 2, pre_act=1, pre_slope=0.1)
                for _ in range(3):
                    plan(x, res if with_res else None, out=out)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    plan(x, res if with_res else None, out=out)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / a.iters * 1e3
                pts.append((k, with_res, us))
                print(json.dumps({"C": C, "T": T, "k": k, "residual": with_res, "us": round(us, 1),
                                  "tflops": round(2.0 * a.batch * C * C * k * T / us / 1e6, 1)}), flush=True)
        for with_res in (False, True):
            p = [(k, us) for k, r, us in pts if r == with_res]
            b = (p[-1][1] - p[0][1]) / (p[-1][0] - p[0][0])
            a0 = p[0][1] - b * p[0][0]
            print(json.dumps({"C": C, "residual": with_res, "fit_us": {"per_launch_fixed": round(a0, 1), "per_tap": round(b, 2)},
                              "fixed_share_at_k11": round(a0 / (a0 + 11 * b), 3)}), flush=True)


if __name__ == "__main__":
    main()
