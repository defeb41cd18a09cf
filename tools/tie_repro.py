"""Repro loop for the box-dependent tie-guard count: the body of
tests/test_gpu_round3.py::test_tie_guard_reevaluates_low_margin_positions_in_fp64, N times in this process with a NEW handle each
time, one JSON line per deviation (L, lens, per-row duration sums, guard statistics, status flag, precision, margins)."""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parrot_tts_amd import _lib, synth  # noqa: E402
from parrot_tts_amd.ops import dptr, stream_ptr  # noqa: E402
from parrot_tts_amd.tte import Parrot  # noqa: E402

DEV = "cuda:0"


def main(n):
    tmp = tempfile.mkdtemp()
    cfg = synth.small_tte_config()
    cfg["path"]["root_path"] = tmp
    with open(os.path.join(tmp, "speakers.json"), "w") as f:
        json.dump({"a": 0, "b": 1}, f)
    vocab, n_spk = 30, 2
    sd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=8)
    hw, hb = sd["head.weight"].clone(), sd["head.bias"].clone()
    hw[:] = hw * 0.01
    hw[17] = hw[5] = torch.randn_like(hw[5])
    hb[:] = -50.0
    hb[17] = hb[5] = 3.0
    sd["head.weight"], sd["head.bias"] = hw, hb
    batch = synth.synth_tte_batch(3, 11, vocab, n_spk, seed=2, ragged=True)
    gb = {k: v.to(DEV) for k, v in batch.items()}
    bad = 0
    first = None
    for it in range(n):
        m = Parrot(cfg, vocab, 0)
        m.load_state_dict(sd)
        m = m.eval().to(DEV)
        r = m.infer_dense(gb)
        gs = m.guard_stats()
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        _lib.check(_lib.lib().parrot_tte_status_peek_async(m._handle, dptr(flag), stream_ptr(torch.device(DEV))))
        B, L = r["ids"].shape
        rec = {"it": it, "L": L, "lens": r["lens"].tolist(), "dur": r["dur"].cpu().tolist(), "log_dur": r["log_dur"].cpu().tolist(),
               "guard": gs, "flag": int(flag.cpu())}
        if first is None:
            first = rec
            print(json.dumps({"first": rec}), flush=True)
        if gs["n_guarded"] != B * L or rec["lens"] != first["lens"] or rec["log_dur"] != first["log_dur"] or gs["min_margin"] != 0.0:
            bad += 1
            lg = m.forward(gb, inference=True)[0].cpu()
            top2 = lg.topk(2, -1).values
            rec["margins"] = (top2[..., 0] - top2[..., 1]).tolist()
            rec["nonfinite_logits"] = int((~torch.isfinite(lg)).sum())
            # where does it start?  every stage against the oracle (max abs error per stage, and at the deviating positions)
            from oracle import parrot_oracle as O
            with torch.no_grad():
                ref = O.tte_forward(sd, cfg, batch, return_stages=True)
            st = m.forward_stages(gb)
            mg = torch.tensor(rec["margins"])
            pos = torch.nonzero(mg != 0)
            rec["bad_pos"] = pos.tolist()
            rec["stage_err"] = {}
            for k, v in st["stages"].items():
                e = (v.cpu() - ref["stages"][k]).abs()
                rec["stage_err"][k] = {"max": float(e.max()), "argmax": [int(i) for i in torch.nonzero(e == e.max())[0]]}
                if k.startswith("dec"):
                    rec["stage_err"][k]["at_bad"] = [float(e[b, t].max()) for b, t in pos.tolist()]
            le = (st["logits"].cpu() - ref["logits"]).abs()
            rec["logit_err_max"] = float(le.max())
            rec["logit_err_at_bad"] = [[float(le[b, t, 5]), float(le[b, t, 17]), float(st["logits"][b, t, 5]), float(st["logits"][b, t, 17]), float(ref["logits"][b, t, 5])] for b, t in pos.tolist()]
            rec["ptrs"] = {k: hex(v.data_ptr()) for k, v in r.items() if torch.is_tensor(v) and v.is_cuda}
            del rec["margins"], rec["dur"], rec["log_dur"]
            print(json.dumps({"deviation": rec}), flush=True)
        del m
    print(json.dumps({"runs": n, "deviations": bad}), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 50)
