#!/usr/bin/env python3
"""Phase timeline of ONE workgroup of the whole-MRF launch of the 32-channel stage (experiment build: tools/build_exp.sh trace
-DRBS_TRACE; PARROT_HIP_LIB=build_exp/libparrot_trace.so python tools/rbs_trace.py): shader clocks between the stamps of
csrc/resblock_split.h, per conv, next to what the instruction counts predict (M = MFMAs x 32 clocks, V = VALU instructions of the
conversion at their measured pipe occupancy)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from parrot_tts_amd import _lib, ops, synth  # noqa: E402

NS = 128


def labels():
    out = ["start", "loaded", "conv0_in_converted", "barrier"]
    for br, k in enumerate((3, 7, 11)):
        if br > 0:
            out += [f"k{k}:switch_loads", f"k{k}:switch_barrier", f"k{k}:x_converted", f"k{k}:barrier"]
        for pair in range(3):
            out += [f"k{k}:p{pair}:c1:A", f"k{k}:p{pair}:c1:B", f"k{k}:p{pair}:c1:C", f"k{k}:p{pair}:c1:D", f"k{k}:p{pair}:c1:E", f"k{k}:p{pair}:c1:F"]
            out += [f"k{k}:p{pair}:c2:A", f"k{k}:p{pair}:c2:B"]
            if pair < 2:
                out += [f"k{k}:p{pair}:c2:C", f"k{k}:p{pair}:c2:D", f"k{k}:p{pair}:c2:E", f"k{k}:p{pair}:c2:F"]
    out.append("compute_done")
    return out


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ops.set_default_precision(ops.PREC_NAMES["f16x3"])
    cfg, h, tsd, vsd, parrot, gen = bench.build_models(dev, 300, 10)
    vb = {k: v.to(dev) for k, v in synth.synth_voc_batch(64, 256, h, seed=0).items()}
    for _ in range(3):
        gen(code=vb["code"], spkr=vb["spkr"])
    torch.cuda.synchronize()
    raw = C.CDLL(_lib.LIB_PATH)
    buf = (C.c_ulonglong * (8 * NS))()
    assert raw.parrot_debug_rbs_trace(buf, 8 * NS) == 0
    lab = labels()
    n = len(lab)
    waves = [[buf[w * NS + i] for i in range(n)] for w in range(8)]
    t0 = min(w[0] for w in waves)
    print(json.dumps({"stamps": n, "total_clocks_per_wave": [w[n - 1] - w[0] for w in waves]}))
    # per conv: conv clocks (A->B), inner conversion (B->C), barrier wait (C->D), outer conversion (D->E), barrier wait (E->F), mean over waves
    idx = {name: i for i, name in enumerate(lab)}
    rows = []
    for k, mf in ((3, 54), (7, 126), (11, 198)):
        acc = {"conv": [], "inner": [], "wait1": [], "outer": [], "wait2": [], "init": []}
        for pair in range(3):
            for c in ("c1", "c2"):
                base = f"k{k}:p{pair}:{c}:"
                if base + "F" not in idx:
                    for w in waves:
                        acc["conv"].append(w[idx[base + "B"]] - w[idx[base + "A"]])
                    continue
                for w in waves:
                    acc["conv"].append(w[idx[base + "B"]] - w[idx[base + "A"]])
                    acc["inner"].append(w[idx[base + "C"]] - w[idx[base + "B"]])
                    acc["wait1"].append(w[idx[base + "D"]] - w[idx[base + "C"]])
                    acc["outer"].append(w[idx[base + "E"]] - w[idx[base + "D"]])
                    acc["wait2"].append(w[idx[base + "F"]] - w[idx[base + "E"]])
        m = {a: (sum(v) / len(v) if v else 0.0) for a, v in acc.items()}
        rows.append({"k": k, "mfma_per_conv_per_wave": mf, "M_clocks_own": mf * 32, "conv_phase": round(m["conv"]), "inner_conversion": round(m["inner"]),
                     "barrier_wait_1": round(m["wait1"]), "outer_conversion": round(m["outer"]), "barrier_wait_2": round(m["wait2"]),
                     "per_conv_total": round(m["conv"] + m["inner"] + m["wait1"] + m["outer"] + m["wait2"])})
    for r in rows:
        print(json.dumps(r))
    w0 = waves[0]
    print(json.dumps({"wave0_prologue": {lab[i]: w0[i] - w0[0] for i in range(4)},
                      "wave0_branch_switch_k7": {lab[i]: w0[i] - w0[idx['k7:switch_loads'] - 1] for i in range(idx['k7:switch_loads'], idx['k7:switch_loads'] + 4)},
                      "start_skew_between_waves": max(w[0] for w in waves) - t0}))


if __name__ == "__main__":
    main()
