"""GPU tests of the round-6 changes: degenerate rows of a row-exact batch, the prefix-mask check, the poison mode of the library,
the whole-MRF launch against per-branch launches, graph replay across streams / threads / evictions."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import parrot_oracle as O  # noqa: E402
from parrot_tts_amd import synth  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parrot(cfg, vocab, n_spk, sd, tmp_path):
    from parrot_tts_amd.tte import Parrot
    cfg = synth.clone_config(cfg)
    cfg["path"]["root_path"] = str(tmp_path)
    with open(os.path.join(str(tmp_path), "speakers.json"), "w") as f:
        json.dump({f"s{i}": i for i in range(n_spk)}, f)
    m = Parrot(cfg, vocab, 0)
    m.load_state_dict(sd)
    return m.eval().to(DEV)


def _gen(h, sd):
    g = CodeGenerator(AttrDict(h))
    g.load_state_dict(sd)
    return g.eval().to(DEV)


def test_row_exact_batch_with_an_empty_row(tmp_path):
    """a row that expands to nothing -- here a row of NO tokens -- used to mask every key of its own softmax
    (0 / 0), and its NaN raised the non-finite flag for the whole batch.  Now: that row emits [], every other row equals its
    single-utterance run id for id, and no flag is raised."""
    cfg, vocab, n_spk, B, S = synth.small_tte_config(), 40, 3, 6, 17
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=77)
    model = _parrot(cfg, vocab, n_spk, tsd, tmp_path)
    batch = synth.synth_tte_batch(B, S, vocab, n_spk, seed=5, ragged=True)
    batch["src_mask"][2, :] = False  # row 2: all padding
    batch["phones"][2, :] = 0
    gb = {k: v.to(DEV) for k, v in batch.items()}
    got = model.infer(gb, row_exact=True)  # (infer() runs check_outputs: a non-finite logit anywhere would raise here)
    assert got[2] == []
    assert model.precision_in_use == "f16x3", "no range fallback was triggered by the empty row"
    for b in (0, 1, 3, 4, 5):
        n = int(batch["src_mask"][b].sum())
        one = {"phones": batch["phones"][b:b + 1, :n].clone(), "src_mask": batch["src_mask"][b:b + 1, :n].clone(),
               "speaker": batch["speaker"][b:b + 1].clone()}
        alone = model.infer({k: v.to(DEV) for k, v in one.items()})[0]
        assert got[b] == alone, f"row {b} changed by the empty row beside it"
        with torch.no_grad():
            ref = O.tte_forward(tsd, cfg, one)
        top2 = torch.topk(ref["logits"], 2, dim=-1).values
        frac = torch.exp(ref["log_dur"][0]) - 1.0
        if bool(((top2[..., 0] - top2[..., 1]) > 1e-4).all()) and bool(((frac - torch.floor(frac) - 0.5).abs() > 1e-4).all()):
            assert got[b] == O.tte_infer(tsd, cfg, one)[0]


def test_row_exact_refuses_a_mask_that_is_not_a_prefix(tmp_path):
    """row_exact derives the row's length from the mask; a pad inside an utterance has no single-utterance reading."""
    cfg, vocab, n_spk = synth.small_tte_config(), 40, 3
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=77)
    model = _parrot(cfg, vocab, n_spk, tsd, tmp_path)
    batch = synth.synth_tte_batch(4, 12, vocab, n_spk, seed=5, ragged=False)
    batch["src_mask"][1, 3] = False
    gb = {k: v.to(DEV) for k, v in batch.items()}
    with pytest.raises(ValueError):
        model.infer(gb, row_exact=True)
    assert len(model.infer(gb)) == 4  # the padded-batch mode takes any mask, like the reference


def _run_py(code, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


_POISON_PROBE = r"""
import ctypes as C, torch
from parrot_tts_amd import _lib
from parrot_tts_amd.ops import dptr, stream_ptr
lib = _lib.lib()
ws = torch.zeros(lib.parrot_length_regulator_workspace_bytes(1, 2, 4, 3), dtype=torch.uint8, device="cuda:0")
seq = torch.ones(1, 2, 4, device="cuda:0"); dur = torch.tensor([[1, 2]], device="cuda:0")
out = torch.zeros(1, 3, 4, device="cuda:0"); mask = torch.zeros(1, 3, dtype=torch.uint8, device="cuda:0"); lens = torch.zeros(1, dtype=torch.int32, device="cuda:0")
_lib.check(lib.parrot_length_regulator(dptr(seq), dptr(dur), 1, 2, 4, 3, dptr(out), dptr(mask), dptr(lens), dptr(ws), ws.numel(), stream_ptr(torch.device("cuda:0"))))
torch.cuda.synchronize()
tail = ws[-64:].view(torch.int32)  # the arena's alignment slack: no kernel writes it
print(int(out.sum()), hex(int(tail[-1]) & 0xffffffff))
"""


def test_poison_mode_fills_what_the_caller_hands_over():
    """PARROT_POISON_WS (tests only): the library fills workspaces / outputs with NaN, inf or 0x7f bytes at the top of every compute
    entry point, so that a kernel reading a byte nobody wrote fails the parity suite deterministically.  Here: the unwritten tail of
    a workspace carries the pattern after a call, and the result is what it is without the mode.  (The whole -m gpu suite ran green
    under all three patterns: profiles/r06a_poison_*.log.)"""
    plain = _run_py(_POISON_PROBE, {"PARROT_POISON_WS": "0"}).split()
    assert plain[-2:] == ["12", "0x0"]
    for mode, word in (("nan", "0x7fc00000"), ("inf", "0x7f800000"), ("7f", "0x7f7f7f7f")):
        got = _run_py(_POISON_PROBE, {"PARROT_POISON_WS": mode}).split()
        assert got[-2:] == ["12", word], (mode, got)


_MRF_PROBE = r"""
import hashlib, torch
from parrot_tts_amd import synth
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator
h = synth.default_voc_config()
sd = synth.synth_voc_state_dict(h, seed=1234)
g = CodeGenerator(AttrDict(h)); g.load_state_dict(sd); g = g.eval().to("cuda:0")
b = synth.synth_voc_batch(8, 256, h, seed=3)
code, spkr = b["code"].to("cuda:0"), b["spkr"].to("cuda:0")
lens = torch.tensor([256, 200, 256, 131, 256, 77, 256, 256], dtype=torch.int32, device="cuda:0")
y = g(code=code, spkr=spkr, unit_lens=lens)
torch.cuda.synchronize()
print(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
for r in (1, 3):
    n = int(lens[r])
    y1 = g(code=code[r:r + 1, :n].contiguous(), spkr=spkr[r:r + 1])
    torch.cuda.synchronize()
    print(bool(torch.equal(y1[0, 0], y[r, 0, : y1.shape[-1]])))
"""


def test_whole_mrf_launch_equals_per_branch_launches_bit_for_bit():
    """which kernel family evaluates the 32-channel MRF depends on the launch size (whole-MRF launch at B x tiles >= 2 x
    CUs).  That is only legal because both give every output sample the same bits: a batch large enough for the whole-MRF launch
    (B = 8 x 256 units, ragged) hashes the same as PARROT_MRF_FUSED=0, and its rows equal their own B = 1 runs (which take the
    per-branch path)."""
    fused = _run_py(_MRF_PROBE, {"PARROT_MRF_FUSED": "1"}).split()
    plain = _run_py(_MRF_PROBE, {"PARROT_MRF_FUSED": "0"}).split()
    assert fused[-3] == plain[-3], "whole-MRF launch and per-branch launches differ"
    assert fused[-2:] == ["True", "True"] and plain[-2:] == ["True", "True"]


def test_graph_replay_from_two_streams_and_two_threads():
    """graph replay is default-on for small forwards and the staging buffers are the handle's.  Replays of one
    shape issued from two streams, and from two host threads, give what the direct path gives, every time."""
    import threading
    h = synth.small_voc_config()
    sd = synth.synth_voc_state_dict(h, seed=5)
    g = _gen(h, sd)
    batches = [synth.synth_voc_batch(2, 40, h, seed=s) for s in (1, 2, 3, 4)]
    dev = [{k: v.to(DEV) for k, v in b.items()} for b in batches]
    os.environ["PARROT_VOC_GRAPH"] = "1"
    direct = []
    for d in dev:
        g3 = _gen(h, sd)
        direct.append(g3(code=d["code"], spkr=d["spkr"]).clone())
        del g3
    torch.cuda.synchronize()
    for _ in range(4):  # warm the graph of this shape on g
        g(code=dev[0]["code"], spkr=dev[0]["spkr"])
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    outs = [None] * 4
    for rep in range(5):
        for i, st in ((0, s1), (1, s2), (2, s1), (3, s2)):
            with torch.cuda.stream(st):
                outs[i] = g(code=dev[i]["code"], spkr=dev[i]["spkr"])
        torch.cuda.synchronize()
        for i in range(4):
            assert torch.equal(outs[i], direct[i]), (rep, i)

    errs = []

    def worker(i):
        try:
            st = torch.cuda.Stream(DEV)
            for _ in range(10):
                with torch.cuda.stream(st):
                    y = g(code=dev[i]["code"], spkr=dev[i]["spkr"])
                st.synchronize()
                if not torch.equal(y, direct[i]):
                    errs.append(i)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


def test_graph_replay_on_ragged_batches_after_eviction():
    """more shapes than the graph cache holds, ragged `unit_lens`, then back to the first shape: every forward
    equals the direct (PARROT_VOC_GRAPH=0-equivalent: a fresh handle's first call) result bit for bit."""
    h = synth.small_voc_config()
    sd = synth.synth_voc_state_dict(h, seed=5)
    g = _gen(h, sd)
    shapes = [(2, 24 + 4 * i) for i in range(12)]  # more shapes than MAX_GRAPHS / MAX_SHAPES hold
    want = {}
    for B, U in shapes:
        b = synth.synth_voc_batch(B, U, h, seed=U)
        lens = torch.tensor([U, max(1, U - 7)], dtype=torch.int32)
        fresh = _gen(h, sd)
        want[(B, U)] = (b, lens, fresh(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV), unit_lens=lens.to(DEV)).clone())
        del fresh
    torch.cuda.synchronize()
    for sweep in range(2):
        for B, U in shapes + shapes[:2]:
            b, lens, ref = want[(B, U)]
            for _ in range(4):  # (the fourth sighting replays a captured graph)
                y = g(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV), unit_lens=lens.to(DEV))
            torch.cuda.synchronize()
            assert torch.equal(y, ref), (sweep, B, U)


@pytest.mark.parametrize("prec", ["f16x3", "bf16"])
def test_operand_planes_do_not_change_a_bit(prec):
    """Operand planes (csrc/conv_split16.h: the first conv of a layer-by-layer ResBlock pair writes its output as the second conv's
    ready-made MFMA operand -- leaky ReLU, scale, 16-bit split applied once in the producer's epilogue -- and, in the single-piece
    bf16 mode of BASELINE configs[2], every pair also writes its output's plane beside the fp32 residual; the consumers fetch their
    slabs with `buffer_load_dwordx4 ... lds`, no VALU).  The operands are bit for bit what the consumer's own conversion would have
    produced, so the waveforms of dense, ragged and tiny launches (large, 64-row and small-tile instantiations) hash the same with
    the planes on and off, in the parity-grade scheme (planes off by default there) and in bf16 (on by default)."""
    def hashes(planes):
        e = dict(os.environ)
        e.update({"PARROT_PLANES": planes, "PARROT_PRECISION": prec})
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wav_hash.py"), "--quick"], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout
    off, on = hashes("0"), hashes("1")
    assert len(off.strip().splitlines()) == 3
    assert on == off


def test_row_exact_at_the_baseline_shape_equals_64_single_utterance_oracle_runs(tmp_path):
    """the row-exact mode at the BASELINE shape -- B = 64 utterances padded to S = 64, full-size TTE -- against 64
    separate B = 1 runs of the oracle (what the reference driver computes, inference.py:34): ids equal id for id on every row whose
    margins allow a verdict (top-2 logit margin and rounding distance of the durations above 1e-4); at most three rows may fall
    below them."""
    cfg, vocab, n_spk, B, S = synth.default_tte_config(), 120, 10, 64, 64
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=77)
    # durations around 4 per token, as at the BASELINE shape (S = 64 -> L ~ 256): rows of 68 ... 297 units here
    tsd["duration_predictor.proj.weight"] = tsd["duration_predictor.proj.weight"] * 0.3
    tsd["duration_predictor.proj.bias"] = torch.full((1,), 1.6)
    model = _parrot(cfg, vocab, n_spk, tsd, tmp_path)
    batch = synth.synth_tte_batch(B, S, vocab, n_spk, seed=9, ragged=True)
    gb = {k: v.to(DEV) for k, v in batch.items()}
    got = model.infer(gb, row_exact=True)
    n_checked = 0
    for b in range(B):
        n = int(batch["src_lens"][b])
        one = {"phones": batch["phones"][b:b + 1, :n].clone(), "src_mask": batch["src_mask"][b:b + 1, :n].clone(),
               "speaker": batch["speaker"][b:b + 1].clone()}
        with torch.no_grad():
            ref = O.tte_forward(tsd, cfg, one)
        frac = torch.exp(ref["log_dur"][0]) - 1.0
        top2 = torch.topk(ref["logits"], 2, dim=-1).values
        if bool(((frac - torch.floor(frac) - 0.5).abs() > 1e-4).all()) and bool(((top2[..., 0] - top2[..., 1]) > 1e-4).all()):
            ref_ids = torch.argmax(ref["logits"], -1)[0][ref["tgt_mask"][0]].tolist()
            assert got[b] == ref_ids, f"row {b}: ids differ from the reference's single-utterance run"
            n_checked += 1
    assert n_checked >= B - 3, n_checked  # (62 of 64 rows on these seeds: ~170 positions x 1000 codes per row, two rows hold a near-tie)


@pytest.mark.parametrize("stage", [-1, 0, 2, 4])
def test_pipelined_schedule_with_a_stage_start_equals_sequential_calls(tmp_path, stage):
    """`SynthesisPipeline.submit` starts the next batch's TTE when the previous batch's vocoder reaches MRF stage
    PARROT_PIPE_STAGE (`parrot_voc_wait_stage`; default 2, -1: at once).  Whatever the stage, each batch's result equals what
    `__call__` returns for it -- on shapes large enough for direct (non-graph) forwards, where the stage events are recorded."""
    from parrot_tts_amd.pipeline import SynthesisPipeline
    cfg, h = synth.small_tte_config(), synth.small_voc_config()
    vocab, n_spk = 30, 2
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=31, forced_duration=4)  # (L = 4 S < max_len = 400 of the small config)
    vsd = synth.synth_voc_state_dict(h, seed=32)
    pipe = SynthesisPipeline(_parrot(cfg, vocab, n_spk, tsd, tmp_path), _gen(h, vsd))
    pipe.pipe_stage = stage
    batches = [{k: v.to(DEV) for k, v in synth.synth_tte_batch(B, S, vocab, n_spk, seed=60 + i, ragged=True).items()}
               for i, (B, S) in enumerate([(64, 40), (56, 44), (64, 40), (60, 48), (64, 40)])]
    want = []
    for b in batches:
        r = pipe(b)
        want.append({k: r[k].clone() for k in ("wav", "ids", "n_samples")})
    assert any(w["ids"].numel() > 8192 for w in want), "the batches must take the direct (non-graph) vocoder path"
    got = []
    for b in batches:
        out = pipe.submit(b)
        if out is not None:
            got.append(out)
    got.append(pipe.flush())
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert torch.equal(g["ids"], w["ids"]) and torch.equal(g["n_samples"].cpu(), w["n_samples"].cpu())
        for row in range(w["wav"].shape[0]):
            n = int(w["n_samples"][row])
            assert torch.equal(g["wav"][row, :, :n], w["wav"][row, :, :n])


def test_wait_stage_validates_its_arguments_and_is_a_no_op_before_the_first_forward():
    """`parrot_voc_wait_stage` (ABI v7): a stage outside 0 .. n_stages - 1 is PARROT_E_INVALID; before the handle's first forward
    there is nothing to wait for and the stream stays usable."""
    from parrot_tts_amd import _lib
    h = synth.small_voc_config()
    g = _gen(h, synth.synth_voc_state_dict(h, seed=5))
    b = synth.synth_voc_batch(2, 24, h, seed=1)
    g(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV))  # builds the handle
    st = torch.cuda.Stream(DEV)
    lib = _lib.lib()
    n_stages = len(h["upsample_rates"])
    import ctypes as C
    assert lib.parrot_voc_wait_stage(g._handle, -1, C.c_void_p(st.cuda_stream)) == -1
    assert lib.parrot_voc_wait_stage(g._handle, n_stages, C.c_void_p(st.cuda_stream)) == -1
    fresh = _gen(h, synth.synth_voc_state_dict(h, seed=5))
    fresh.wait_stage(0, st)  # no handle yet: returns
    for i in range(n_stages):
        g.wait_stage(i, st)
    with torch.cuda.stream(st):
        y = g(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV))
    st.synchronize()
    assert bool(torch.isfinite(y).all())
