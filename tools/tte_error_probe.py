#!/usr/bin/env python3
"""Localise the TTE's numerical error: per-stage activations of the HIP path (Parrot.forward_stages) and of the fp32 CPU
oracle, both against an fp64 run of the oracle, at the bench shape.  (Test tooling: imports oracle/.)
    python tools/tte_error_probe.py [--batch 8] [--precision f16x3]      env PARROT_TTE_MERGE=0/1"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import parrot_oracle as O  # noqa: E402
from parrot_tts_amd import ops, synth  # noqa: E402
from parrot_tts_amd.tte import Parrot  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--src-len", type=int, default=64)
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    torch.set_num_threads(min(64, os.cpu_count() or 1))
    ops.set_default_precision(ops.PREC_NAMES[a.precision])
    cfg = synth.default_tte_config()
    vocab, n_spk = 300, 10
    tmp = tempfile.mkdtemp()
    cfg["path"]["root_path"] = tmp
    json.dump({f"s{i}": i for i in range(n_spk)}, open(os.path.join(tmp, "speakers.json"), "w"))
    sd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=42, forced_duration=4)
    batch = synth.synth_tte_batch(a.batch, a.src_len, vocab, n_spk, seed=0)
    with torch.no_grad():
        r32 = O.tte_forward(sd, cfg, batch, return_stages=True)
        sd64 = {k: v.double() for k, v in sd.items()}
        r64 = O.tte_forward(sd64, cfg, batch, return_stages=True)
    m = Parrot(cfg, vocab, 0)
    m.load_state_dict(sd)
    m = m.eval().to("cuda:0")
    g = m.forward_stages({k: v.to("cuda:0") for k, v in batch.items()})
    rows = []
    for k in list(r64["stages"]) + ["logits"]:
        ref = r64["stages"][k] if k != "logits" else r64["logits"]
        o32 = r32["stages"][k] if k != "logits" else r32["logits"]
        hip = (g["stages"][k] if k != "logits" else g["logits"]).cpu()
        rows.append({"stage": k, "scale": float(ref.abs().max()), "hip_vs_fp64": float((hip.double() - ref).abs().max()),
                     "cpu_fp32_vs_fp64": float((o32.double() - ref).abs().max()), "hip_vs_cpu_fp32": float((hip - o32).abs().max())})
    for r in rows:
        print(json.dumps(r))
    print(json.dumps({"precision": a.precision, "merge": os.environ.get("PARROT_TTE_MERGE", "1"), "log_dur_err": float((g["log_dur"].cpu().double() - r64["log_dur"]).abs().max())}))


if __name__ == "__main__":
    main()
