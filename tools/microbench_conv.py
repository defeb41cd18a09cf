#!/usr/bin/env python3
"""GPU micro-benchmark of the conv implicit-GEMM kernel on the layer shapes of the hot path, per tile
configuration.  Prints one JSON line per (layer, tile).   python tools/microbench_conv.py [--batch 16]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import ops  # noqa: E402

LAYERS = [  # name, cin, cout, k, dil, T per utterance
    ("mrf0_k3", 256, 256, 3, 1, 1280), ("mrf0_k7d3", 256, 256, 7, 3, 1280), ("mrf0_k11d5", 256, 256, 11, 5, 1280),
    ("mrf1_k3", 128, 128, 3, 1, 5120), ("mrf1_k11d5", 128, 128, 11, 5, 5120),
    ("mrf2_k3", 64, 64, 3, 1, 20480), ("mrf2_k11d5", 64, 64, 11, 5, 20480),
    ("mrf3_k3", 32, 32, 3, 1, 40960), ("mrf3_k11d5", 32, 32, 11, 5, 40960),
    ("mrf4_k3", 16, 16, 3, 1, 81920), ("mrf4_k11d5", 16, 16, 11, 5, 81920),
    ("conv_pre", 256, 512, 7, 1, 256), ("ffn1_k9", 256, 1024, 9, 1, 256), ("ffn2_k1", 1024, 256, 1, 1, 256),
    ("qkv_k1", 256, 768, 1, 1, 256), ("head_k1", 256, 1000, 1, 1, 256), ("conv_post", 16, 1, 7, 1, 81920),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--tiles", type=str, default="-1,0,1,2,3")
    ap.add_argument("--layers", type=str, default="", help="comma separated layer names (default: all)")
    ap.add_argument("--precision", type=int, default=-1, help="-1 library default, 0 exact fp32 MFMA, 1 split-bf16")
    ap.add_argument("--constant", action="store_true", help="all-ones inputs and weights (operand toggling / clock experiments)")
    a = ap.parse_args()
    dev = "cuda:0"
    want = set(a.layers.split(",")) if a.layers else None
    for name, cin, cout, k, dil, T in LAYERS:
        if want is not None and name not in want:
            continue
        w = torch.randn(cout, cin, k) / (cin * k) ** 0.5
        b = torch.randn(cout) * 0.1
        x = torch.randn(a.batch, cin, T, device=dev)
        res = torch.randn(a.batch, cout, T, device=dev)
        out = torch.empty(a.batch, cout, T, device=dev)
        if a.constant:
            w, b = torch.full_like(w, 0.01), torch.zeros_like(b)
            x, res = torch.ones_like(x), torch.ones_like(res)
        for tile in [int(t) for t in a.tiles.split(",")]:
            try:
                plan = ops.ConvPlan(w, b, dilation=dil, padding=dil * (k - 1) // 2, pre_act=1, pre_slope=0.1, tile_cfg=tile, precision=a.precision)
            except Exception as e:  # noqa: BLE001
                print(json.dumps({"layer": name, "tile": tile, "error": str(e)}))
                continue
            for _ in range(2):
                plan(x, res, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                plan(x, res, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            fl = 2.0 * a.batch * cout * cin * k * T
            by = 4.0 * a.batch * T * (cin + 2 * cout)
            print(json.dumps({"layer": name, "tile": tile, "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 2),
                              "alg_GBps": round(by / ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
