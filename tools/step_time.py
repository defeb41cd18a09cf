#!/usr/bin/env python3
"""Wall time per step of the full pipeline (bench.py's models and inputs) WITHOUT the per-launch HIP events of bench.py's kernel
table: same-box A/B runs of schedules and switches.

    python tools/step_time.py [--batch 64] [--steps 20] [--prof]     -> one JSON line per setting"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from parrot_tts_amd import _lib, ops, synth  # noqa: E402
from parrot_tts_amd.pipeline import SynthesisPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--src-len", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--prof", action="store_true", help="also time with bench.py's per-launch HIP events switched on")
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ops.set_default_precision(ops.PREC_NAMES[a.precision])
    vocab, n_spk = 300, 10
    cfg, h, tsd, vsd, parrot, gen = bench.build_models(dev, vocab, n_spk)
    batch = {k: v.to(dev) for k, v in synth.synth_tte_batch(a.batch, a.src_len, vocab, n_spk, seed=0).items()}
    lib = _lib.lib()
    for _ in (0,):
        for prof in ([False, True] if a.prof else [False]):
            pipe = SynthesisPipeline(parrot, gen)
            for _ in range(a.warmup):
                pipe(batch)
            torch.cuda.synchronize()
            if prof:
                lib.parrot_prof_begin()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                pipe(batch)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            if prof:
                import ctypes as C
                names = bench.tile_names(a.precision)
                buf = (C.c_double * (4 * len(names)))()
                lib.parrot_prof_end(buf, len(names))
            print(json.dumps({"batch": a.batch, "per_launch_events": prof,
                              "ms_per_step": round(ms, 4)}), flush=True)


if __name__ == "__main__":
    main()
