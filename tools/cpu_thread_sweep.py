#!/usr/bin/env python3
"""Thread sweep of the CPU baseline (the oracle, full pipeline) on the GPU box's host: justifies the thread count bench.py uses.
    python tools/cpu_thread_sweep.py [--batch 8] [--threads 16,32,64,128]   -> one JSON line per thread count"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import parrot_oracle as O  # noqa: E402  (the measured thing here IS the CPU baseline)
from parrot_tts_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--threads", default="16,32,64,128")
    a = ap.parse_args()
    cfg, h = synth.default_tte_config(), synth.default_voc_config()
    vocab, n_spk = 300, 10
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=42, forced_duration=4)
    vsd = O.fold_weight_norm(synth.synth_voc_state_dict(h, seed=1234, scale=1.0))
    batch = synth.synth_tte_batch(a.batch, 64, vocab, n_spk, seed=0)

    def run():
        with torch.no_grad():
            r = O.tte_forward(tsd, cfg, batch)
            return O.code_generator_forward(vsd, h, torch.argmax(r["logits"], -1), batch["speaker"].reshape(-1, 1))

    for nt in [int(v) for v in a.threads.split(",")]:
        torch.set_num_threads(nt)
        run()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            y = run()
            ts.append(time.perf_counter() - t0)
        n = y.shape[0] * y.shape[-1]
        print(json.dumps({"probe": "cpu_thread_sweep", "batch": a.batch, "threads": nt, "logical_cpus": os.cpu_count(), "s_per_pass": min(ts),
                          "samples_per_s": n / min(ts)}), flush=True)


if __name__ == "__main__":
    main()
