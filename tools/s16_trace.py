#!/usr/bin/env python3
"""Chunk timeline of workgroup 0 of a conv_split16 launch (experiment build with -DS16_TRACE):
    tools/build_exp.sh s16trace -DS16_TRACE
    PARROT_HIP_LIB=build_exp/libparrot_s16trace.so python tools/s16_trace.py [--layer ffn1_k9] [--batch 1] [--tile 2]
prints, per wave, the shader-clock deltas between the marks of conv_split16.h and the effective shader clock."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import _lib, ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from microbench_conv import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="ffn1_k9")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--tile", type=int, default=-1)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    name, cin, cout, k, dil, T = next(l for l in LAYERS if l[0] == a.layer)
    dev = "cuda:0"
    w = torch.randn(cout, cin, k) / (cin * k) ** 0.5
    b = torch.randn(cout) * 0.1
    x = torch.randn(a.batch, cin, T, device=dev)
    res = torch.randn(a.batch, cout, T, device=dev)
    out = torch.empty(a.batch, cout, T, device=dev)
    plan = ops.ConvPlan(w, b, dilation=dil, padding=dil * (k - 1) // 2, pre_act=1, pre_slope=0.1, tile_cfg=a.tile)
    for _ in range(3):
        plan(x, res, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        plan(x, res, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name} B={a.batch} tile={a.tile}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per launch (back to back)")
    fn = _lib.lib().parrot_debug_s16_trace
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * (4 * 64))()
    assert fn(buf) == 0
    for wv in range(4):
        t = [buf[wv * 64 + i] for i in range(64)]
        n = max(i for i in range(62) if t[i]) + 1 if any(t[:62]) else 0
        if n < 2:
            continue
        d = [t[i] - t[i - 1] for i in range(1, n)]
        real_us = (t[63] - t[62]) / 100.0
        print(f"wave {wv}: total {t[n - 1] - t[0]} clk in {real_us:.2f} us -> {(t[n - 1] - t[0]) / max(real_us, 1e-9) / 1e3:.2f} GHz; "
              f"prologue {d[0]}, chunks {d[1:-1]}, epilogue {d[-1]}")


if __name__ == "__main__":
    main()
