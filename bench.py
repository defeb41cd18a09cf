#!/usr/bin/env python3
"""bench.py -- Parrot-TTS synthesis throughput on MI355X (BASELINE.json metric: audio samples/s + RTF
for batch-64 x 256-unit inputs).

One "step" = one pass of the hot path over one batch of synthetic input per GPU:
    TTE (S=64 tokens -> forced duration 4 -> L=256 units) -> HiFi-GAN generator (256 units -> 81 920 samples)
with inputs already resident in HBM.  N>1: one process per GPU (torchrun contract), utterances
sharded by batch row (weak scaling, per-GPU batch fixed), one RCCL gather of the waveforms to rank 0
per step.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from parrot_tts_amd import _lib, synth  # noqa: E402
from parrot_tts_amd import dist as pdist  # noqa: E402
from parrot_tts_amd.pipeline import SynthesisPipeline  # noqa: E402
from parrot_tts_amd.tte import Parrot  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector peak
HBM_PEAK_GBS = 8000.0
SAMPLE_RATE = 16000             # reference utils/vocoder/config.json:32 (the metric's "22.05 kHz" does not match the reference)
# rows of parrot_prof_end, in the library's order; SCH = the split scheme of the run (conv_split.h)
TILE_NAMES = ["conv_mfma_kernel<2,2,2,2,16,3>", "conv_mfma_kernel<1,4,2,2,16,3>", "conv_mfma_kernel<1,4,1,4,16,2>",
              "conv_mfma_kernel<2,2,2,2,32,3>", "conv_mfma_kernel<2,2,2,1,16,3>", "conv_mfma_kernel<1,4,1,2,16,4>",
              "conv_mfma16_kernel<8,2>", "conv_split_kernel<SCH,2,2,2,2,2>", "conv_split_kernel<SCH,1,4,2,2,2>", "resblock_fused16_kernel",
              "conv_split_kernel<SCH,4,1,1,2,3>",  # the 128 x 64 tile (2 x 2 waves for the non-default schemes)
              "conv_split_kernel<SCH,1,4,1,4,2>", "resblock_split_kernel<SCH,2>", "resblock16_split_kernel<SCH>",
              "conv1_valu_kernel", "convt_valu_kernel<16,4,2,1>", "conv_split16_kernel<SCH,2,2,4,4>",
              "conv_split16_kernel<SCH,2,2,2,4>",  # the 64-row tile
              "(unused)", "conv_split16_kernel<SCH,2,2,4,5>",
              "resblock_split_kernel<SCH,4>", "resblock_split_kernel<SCH,8>", "resblock_split_kernel<SCH,16>",
              "resblock_split_kernel<SCH,2,8,MRF>"]  # the whole-MRF launch of the 32-channel stage
SCHEMES = {"f16x3": ("SchF16x3", 3), "bf16x6": ("SchBf16x6", 6), "bf16": ("SchBf16", 1), "f16": ("SchF16", 1), "f32": ("-", 1)}
MFMA16_PEAK_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak; a split scheme spends 3 (f16x3) or 6 (bf16x6) MFMA FMAs per algorithmic fp32 FMA


def tile_names(precision):
    return [n.replace("SCH", SCHEMES[precision][0]) for n in TILE_NAMES]


def pmc_profile(kernel: str):
    """Counters of `kernel` from the newest committed PMC summary (profiles/rNN_pmc_traffic.json, tools/make_pmc_traffic.py):
    rocprofv3 cannot run inside the timed region, so the separately collected passes of the same command on the same BUILD are
    attached here -- HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes, calibrated), the MFMA-busy fraction (SQ pass:
    SQ_VALU_MFMA_BUSY_CYCLES / 4 SQ_BUSY_CU_CYCLES) -- with the file and the box they were measured on (nulls when no summary
    covers this kernel build)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    out = {"traffic": None, "mfma_busy": None, "source": None, "box": None}
    if not files:
        return out
    try:
        d = json.load(open(files[-1]))
        e = d.get(kernel, {})
        out.update(traffic=e.get("hbm_bytes_per_launch"), mfma_busy=e.get("mfma_busy"), source=os.path.basename(files[-1]), box=d.get("_box"))
    except (OSError, ValueError):
        pass
    return out


def algorithmic_macs(cfg, h, S, L):
    """Multiply-accumulates of ONE utterance of the path (real taps only): TTE at S tokens -> L units, HiFi-GAN at L units."""
    tr, dp = cfg["transformer"], cfg["duration_predictor"]
    D, F = tr["d_model"], tr["conv_n_filter"]
    k1, k2 = tr["conv_kernel_sizes"]

    def fft(T, layers):  # qkv + in_proj + out_proj + wo (8 D^2), QK^T + AV (2 T D), FFN
        return layers * T * (8 * D * D + 2 * T * D + F * D * k1 + D * F * k2)

    NF, dk = dp["n_filter"], dp["kernel_size"]
    tte = fft(S, tr["encoder"]["n_layer"]) + fft(L, tr["decoder"]["n_layer"]) + S * (NF * D * dk + NF * NF * dk + NF) \
        + L * D * cfg["preprocess"]["hubert_codes"]
    C = h["upsample_initial_channel"]
    T = L
    voc = C * h.get("model_in_dim", h["embedding_dim"] * (2 if h.get("multispkr") else 1)) * 7 * T
    n_conv = 2 if str(h.get("resblock", "1")) == "1" else 1
    for u, k in zip(h["upsample_rates"], h["upsample_kernel_sizes"]):
        voc += C * (C // 2) * k * T
        C, T = C // 2, T * u
        for rk, dil in zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"]):
            voc += C * C * rk * T * n_conv * len(dil[: 3 if n_conv == 2 else 2])
    voc += C * 7 * T
    return tte, voc


def build_models(dev, vocab=300, n_spk=10):
    cfg, h = synth.default_tte_config(), synth.default_voc_config()
    tmp = tempfile.mkdtemp()
    cfg["path"]["root_path"] = tmp
    with open(os.path.join(tmp, "speakers.json"), "w") as f:
        json.dump({f"spk{i}": i for i in range(n_spk)}, f)
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=42, forced_duration=4)
    vsd = synth.synth_voc_state_dict(h, seed=1234, scale=1.0)
    parrot = Parrot(cfg, vocab, 0)
    parrot.load_state_dict(tsd)
    gen = CodeGenerator(AttrDict(h))
    gen.load_state_dict(vsd)
    gen.eval()
    gen.remove_weight_norm()  # as reference utils/vocoder/inference.py:136-137
    return cfg, h, tsd, vsd, parrot.eval().to(dev), gen.to(dev)


def physical_cores():
    """Physical cores of this host (unique (package, core) pairs), falling back to the logical count."""
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip() and phys is not None and core is not None:
                seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_quota():
    """CPUs this process may actually use: the scheduler affinity, cut to the cgroup CPU quota when one is set
    (cgroup v2 cpu.max, v1 cfs_quota_us / cfs_period_us).  The GPU boxes show 256 logical CPUs under a 16-CPU quota:
    threads beyond the quota are throttled, not run (profiles/r03_cpu_thread_sweep.jsonl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, h, tsd, vsd, B, S, vocab, n_spk):
    """The oracle (CPU restatement, parity-pinned to the reference) timed on this host's cores on a bounded
    sample of the same workload: B utterances of the full pipeline."""
    from oracle import parrot_oracle as O
    # every CPU the process is allowed (affinity and cgroup quota), up to 64: beyond ~64 threads these layer sizes only
    # oversubscribe (256 threads: 100 s / pass), and threads beyond a cgroup quota are throttled
    cores = min(physical_cores(), cpu_quota(), 64)
    torch.set_num_threads(cores)
    batch = synth.synth_tte_batch(B, S, vocab, n_spk, seed=0)
    folded = O.fold_weight_norm(vsd)

    def run():
        with torch.no_grad():
            r = O.tte_forward(tsd, cfg, batch)
            ids = torch.argmax(r["logits"], -1)
            return O.code_generator_forward(folded, h, ids, batch["speaker"].reshape(-1, 1))

    run()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        y = run()
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    n = y.shape[0] * y.shape[-1]
    return {"value": n / best, "unit": "samples/s", "cores": cores, "host_physical_cores": physical_cores(), "host_logical_cpus": os.cpu_count(),
            "host_cpu_quota": cpu_quota(),
            "kind": "port",
            "sample": f"oracle full pipeline (TTE S={S}->L={y.shape[-1] // 320} + HiFi-GAN), batch {B}, fp32, torch-CPU {cores} threads, "
                      f"best of 2 after 1 warm-up ({best:.2f} s/pass)",
            "rtf": best / (n / SAMPLE_RATE)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU")
    ap.add_argument("--src-len", type=int, default=64)
    ap.add_argument("--workload", choices=["full", "vocoder"], default="full")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra exact-fp32 pass")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("PARROT_BENCH_OVERLAP", "1")),
                    help="1 (default since round 6): the headline run uses the two-stage pipeline across steps (TTE of batch i on a side HIP "
                         "stream beside the vocoder of batch i-1: the throughput schedule of SynthesisPipeline.submit / flush, every batch "
                         "submitted inside the timed region finishes inside it); 0: one batch at a time.  The other schedule's rate is "
                         "reported next to the headline either way (`sequential_steps` / `pipelined_steps`)")
    ap.add_argument("--precision", choices=["f32", "bf16x6", "f16x3", "bf16", "f16"], default=os.environ.get("PARROT_BENCH_PRECISION", "f16x3"),
                    help="product evaluation of the conv kernels for layers with >= 16 channels (fp32 data either way): f16x3 (default), "
                         "bf16x6 and f32 are parity-grade; bf16 / f16 are the single-MFMA reduced-precision operating point")
    a = ap.parse_args()

    # one GPU per rank (checked before RCCL is brought up): the only exception is the 2-ranks-on-one-GPU test of the N > 1 path,
    # where gloo carries the collective
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world > torch.cuda.device_count() and os.environ.get("PARROT_DIST_BACKEND") != "gloo":
        raise SystemExit(f"bench.py: WORLD_SIZE={env_world} ranks but only {torch.cuda.device_count()} visible GPU(s); one process per GPU "
                         "(set PARROT_DIST_BACKEND=gloo to share a device in tests)")
    rank, world, local = pdist.init_from_env("nccl")
    if world != a.gpus:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)
    dev = pdist.local_device(local)
    torch.cuda.set_device(dev)
    vocab, n_spk = 300, 10
    from parrot_tts_amd import ops
    lib = _lib.lib()
    B, S = a.batch, a.src_len

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(precision, steps, warmup, overlap_steps=False, workload=None, B=B, tte_precision=None, profile=True):
        """Build both models under `precision`, then W untimed + K timed steps.  Returns (max-over-ranks seconds,
        per-kernel profile rows, samples per step over all ranks, model pieces for the CPU baseline)."""
        workload = workload or a.workload
        ops.set_default_precision(ops.PREC_NAMES[precision])
        cfg, h, tsd, vsd, parrot, gen = build_models(dev, vocab, n_spk)
        if tte_precision is not None:  # (BASELINE configs[2]: fp32-class TTE in front of the reduced-precision vocoder)
            parrot._precision_override = ops.PREC_NAMES[tte_precision]
        pipe = SynthesisPipeline(parrot, gen)
        batch = {k: v.to(dev) for k, v in synth.synth_tte_batch(B, S, vocab, n_spk, seed=rank).items()}
        vb = {k: v.to(dev) for k, v in synth.synth_voc_batch(B, 4 * S, h, seed=rank).items()}

        overlap = overlap_steps and workload == "full"

        gather_ev = []  # (start, stop) CUDA events around every gather: the collective's share of a step, reported as gather_ms

        gather_stream = torch.cuda.Stream(device=dev) if world > 1 else None

        def gather(wav):
            # the collective runs on its OWN stream behind the vocoder that produced `wav`: the next batch's TTE and vocoder do not
            # queue behind it (xGMI traffic beside compute); the closing fence of the region waits for every stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            gather_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(gather_stream):
                e0.record()
                r = pdist.gather_waveforms(wav, dst=0, equal_shapes=True)  # forced durations: every shard is (B, 1, 320 L)
                e1.record()
            wav.record_stream(gather_stream)
            gather_ev.append((e0, e1))
            return r

        def finish(out):
            if out is None:
                return None
            return gather(out["wav"]) if world > 1 else out["wav"]

        def step():
            if overlap:  # TTE of this batch beside the vocoder of the previous one; results one call late
                return finish(pipe.submit(batch))
            wav = pipe(batch)["wav"] if workload == "full" else gen(code=vb["code"], spkr=vb["spkr"])
            return gather(wav) if world > 1 else wav

        def drain():  # every submitted batch is finished inside the region that submitted it
            return finish(pipe.flush()) if overlap else None

        names = tile_names(precision)

        def read_rows(n_steps):
            prof = (C.c_double * (4 * len(names)))()
            _lib.check(lib.parrot_prof_end(prof, len(names)))
            out = []
            for i, nm in enumerate(names):
                n, tms, fl, by = prof[4 * i: 4 * i + 4]
                if n > 0:
                    out.append({"kernel": nm, "row": i, "launches_per_step": n / n_steps, "avg_us": tms / n * 1e3, "ms_per_step": tms / n_steps,
                                "tflops": fl / tms / 1e9, "alg_gbs": by / tms / 1e6})
            return out

        # W untimed warm-up steps.  The last two of them (one more step when W < 3) run with HIP events around EVERY conv launch:
        # the per-kernel table (`all_conv_kernels`) and the choice of the dominant kernel.  The timed region then carries events
        # around the dominant kernel's launches only -- `roofline.achieved` is measured there, live, as the contract asks -- because
        # a pair of event records around all ~130 launches of a step is itself 0.6 ms of a B = 64 step (3 %) and 0.4 ms of a 2 ms
        # single-utterance step: measurement overhead, not work of the path (tools/step_time.py --prof shows the A/B).
        # (`profile=False`: no events at all -- latency rows such as the single utterance, whose ~130 launches take 7-40 us each)
        n_table = (2 if warmup >= 3 else 1) if profile else 0
        wav = None
        for _ in range(max(warmup - n_table, 1 if warmup else 0)):
            wav = step()
        wav = drain() if overlap and warmup else wav
        fence()
        table = []
        if profile:
            lib.parrot_prof_begin()
            for _ in range(n_table):
                wav = step()
            wav = drain() if overlap else wav
            fence()
            table = read_rows(n_table)
            table.sort(key=lambda r: -r["ms_per_step"])
        dom_row = table[0]["row"] if table else 0
        gather_ev.clear()
        if profile:
            _lib.check(lib.parrot_prof_begin_row(dom_row))
        # (HIP events around every batch, >= 10 runs, median -- next to the contract's mean over the K steps between fences)
        step_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for i in range(steps):
            step_ev[i][0].record()
            out = step()
            step_ev[i][1].record()
            wav = out if out is not None else wav
        out = drain()
        wav = out if out is not None else wav
        fence()
        elapsed = time.perf_counter() - t0
        run.step_ms = sorted(a_.elapsed_time(b_) for a_, b_ in step_ev)
        # a run that fell back to another precision (range-safe fallback of the shims) is not the run that was asked for
        in_use = {"tte": parrot.precision_in_use if workload == "full" else precision, "vocoder": gen.precision_in_use}
        if in_use["vocoder"] not in (None, precision) or in_use["tte"] not in (None, tte_precision or precision):
            raise SystemExit(f"bench.py: asked for precision {precision} but the handles ran as {in_use} (non-finite output -> fallback)")
        timed = read_rows(steps) if profile else []  # the dominant kernel's launches of the timed region
        g_ms = sum(a_.elapsed_time(b_) for a_, b_ in gather_ev) / max(steps, 1)
        t = torch.tensor([elapsed, g_ms], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        run.gather_ms = float(t[1].item())
        rows = []
        for r in table:
            r = dict(r)
            r["measured_in"] = "untimed warm-up steps with events around every launch"
            if timed and r["row"] == timed[0]["row"]:
                r.update(timed[0])
                r["measured_in"] = "the timed region (events around this kernel's launches only)"
            rows.append(r)
        # per row: fraction of the scheme's MFMA roof and of the 8 TB/s HBM peak (both from the ALGORITHMIC work of the row's
        # launches), and which roof binds it: the arithmetic intensity of the row against the ridge of its pipe.  `why` adds what the
        # counters / traces / probes of DESIGN.md section 7 say about the distance to that roof.
        n_mf = SCHEMES[precision][1]
        for r in rows:
            k = r["kernel"]
            split_k = "_split" in k
            roof_tf = (MFMA16_PEAK_TFLOPS / n_mf) if split_k else FP32_MFMA_PEAK_TFLOPS
            r["frac_of_mfma_roof"] = r["tflops"] / roof_tf
            r["frac_of_hbm_peak"] = r["alg_gbs"] / HBM_PEAK_GBS
            ai = (r["tflops"] * 1e12) / max(r["alg_gbs"] * 1e9, 1.0)      # FLOP per algorithmic byte
            ridge = roof_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
            r["arithmetic_intensity"], r["ridge"] = ai, ridge
            r["bound"] = "hbm" if ai < ridge else "mfma"
            r["frac_of_binding_roof"] = r["frac_of_hbm_peak"] if ai < ridge else r["frac_of_mfma_roof"]
            r["hbm"] = {"achieved": r["alg_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s (algorithmic)", "frac": r["frac_of_hbm_peak"]}
            prof = pmc_profile(k)
            r["mfma_busy"], r["traffic"], r["counters_source"] = prof["mfma_busy"], prof["traffic"], prof["source"]
            if "valu_kernel" in k:
                r["why"] = "fp32 FMA streaming kernel; 6.3 TB/s achievable"
            elif k.startswith("resblock"):
                r["why"] = ("operand conversion (scale / leaky ReLU / fp16 hi-lo split / LDS store of every conv input) is VALU work of the order "
                            "of the conv's MFMAs, and on gfx950 VALU does not hide under MFMA: tools/probes/interleave.hip measures "
                            "16 + ~4 clocks per VALU per v_mfma_f32_16x16x32_f16 within a wave, +3 per VALU beside 32x32x16 (profiles/r05a_interleave.jsonl)")
            elif k.startswith("conv_split16"):
                r["why"] = "power-limited clock (bare MFMA stream: roofline.ceiling_probe_tflops); short-K layers add exposed prologue / epilogue"
            elif "4,1,1,2,3" in k:
                r["why"] = "latency: ~27 launches of 13-50 us (1x1 convs of the TTE, K = 256 ... 1024)"
            elif k.startswith("conv_split"):
                r["why"] = "27-33 % zero polyphase taps (transposed convs)"
            else:
                r["why"] = "exact fp32 MFMA"
        rows.sort(key=lambda r: -r["ms_per_step"])
        rows.sort(key=lambda r: 0 if (timed and r["row"] == timed[0]["row"]) else 1)  # (stable: the timed-region row leads)
        n_samples = world * B * (wav.shape[-1] if wav is not None else 4 * S * 320)
        del pipe, parrot, gen
        return float(t[0].item()), rows, n_samples, (cfg, h, tsd, vsd)

    def long_form(precision, steps, B=8, U=1500, chunk=256):
        ops.set_default_precision(ops.PREC_NAMES[precision])
        cfg, h, tsd, vsd, parrot, gen = build_models(dev, vocab, n_spk)
        vb = {k: v.to(dev) for k, v in synth.synth_voc_batch(B, U, h, seed=3).items()}
        out = {}
        halo = gen.receptive_units(dev)
        for name, fn in (("whole", lambda: gen(code=vb["code"], spkr=vb["spkr"])),
                         ("chunk_streamed_256", lambda: gen.forward_chunked(chunk_units=chunk, code=vb["code"], spkr=vb["spkr"])),
                         ("chunk_streamed_512", lambda: gen.forward_chunked(chunk_units=512, code=vb["code"], spkr=vb["spkr"])),
                         ("chunk_streamed_768", lambda: gen.forward_chunked(chunk_units=768, code=vb["code"], spkr=vb["spkr"]))):
            fn()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                w = fn()
            fence()
            ms = (time.perf_counter() - t0) / steps * 1e3
            out[name] = {"value": B * w.shape[-1] / (ms / 1e3), "unit": "samples/s", "ms_per_step": ms}
        # units computed per utterance incl. the halo recompute (every interior chunk boundary costs the receptive field twice): the
        # floor of a chunked forward that stays EQUAL to the whole-utterance one, as a ratio to `whole`
        for c_ in (256, 512, 768):
            nb = (U + c_ - 1) // c_ - 1
            out["chunk_streamed_%d" % c_]["units_computed_ratio"] = (U + 2 * halo * nb) / U
            out["chunk_streamed_%d" % c_]["vs_whole"] = out["chunk_streamed_%d" % c_]["ms_per_step"] / out["whole"]["ms_per_step"]
        out["workload"] = ("HiFi-GAN generator, batch %d x %d units (30 s utterances), whole and in 256- / 512- / 768-unit chunks with the generator's "
                           "receptive field (%d units) of real context on both sides (BASELINE configs[4]); two chunks in flight" % (B, U, halo))
        del parrot, gen
        return out

    def product_error_vs_fp64():
        """What the `dtype` label means numerically: one wide MRF layer (256 -> 256 channels, k = 7, dilation 3) evaluated by the
        run's scheme and by the exact fp32 MFMA kernel, both against an fp64 evaluation of the same layer (torch CPU)."""
        import torch.nn.functional as F
        g = torch.Generator().manual_seed(11)
        w = torch.randn(256, 256, 7, generator=g) / (256 * 7) ** 0.5
        bias = torch.randn(256, generator=g) * 0.1
        x = torch.randn(2, 256, 640, generator=g)
        ref = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), bias.double(), dilation=3, padding=9)
        out = {"layer": "Conv1d(256, 256, k=7, dilation=3) on lrelu(x), B=2, T=640; max |y - y_fp64| / max |y_fp64|"}
        for key, prec_ in ((a.precision, ops.PREC_NAMES[a.precision]), ("f32_exact_mfma", ops.PREC_F32)):
            plan = ops.ConvPlan(w, bias, dilation=3, padding=9, pre_act=ops.PRE_LRELU, pre_slope=0.1, precision=prec_)
            y = plan(x.to(dev)).cpu().double()
            out[key] = float((y - ref).abs().max() / ref.abs().max())
        return out

    def driver_e2e(precision, n_items=256, n_single=32):
        """The shipped vocoder driver end to end (parrot_tts_amd/cli/voc_infer.run_batched): `n_items` utterances of 128-256
        units -> length-bucketed padded batches -> int16 -> pinned host -> peak-normalise -> WAV files on disk, against the same
        driver fed one utterance per batch (what reference utils/vocoder/inference.py:146-175 does)."""
        import numpy as np
        import shutil
        from parrot_tts_amd.cli.voc_infer import run_batched
        ops.set_default_precision(ops.PREC_NAMES[precision])
        cfg, h, tsd, vsd, parrot, gen = build_models(dev, vocab, n_spk)
        rng = np.random.Generator(np.random.PCG64(7))
        tmp = tempfile.mkdtemp()
        rows = [(rng.integers(0, h["num_embeddings"], int(rng.integers(128, 257))).astype(np.int64), int(rng.integers(0, 10)),
                 os.path.join(tmp, f"utt{i:04d}_gen.wav")) for i in range(n_items)]
        out = {}
        try:
            run_batched(gen, rows[:64], dev, SAMPLE_RATE)  # warm-up: workspace, handle, page cache
            for key, sel, mr in (("batched", rows, 64), ("one_utterance_per_batch", rows[:n_single], 1)):
                fence()
                t0 = time.perf_counter()
                nw = run_batched(gen, sel, dev, SAMPLE_RATE, max_rows=mr)
                fence()
                dt = time.perf_counter() - t0
                out[key] = {"utterances_per_s": nw / dt, "samples_per_s": sum(r[0].size for r in sel) * 320 / dt, "items": nw, "seconds": dt}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        out["speedup"] = out["batched"]["utterances_per_s"] / out["one_utterance_per_batch"]["utterances_per_s"]
        out["workload"] = ("vocoder driver incl. host post-processing and WAV writing: %d utterances of 128-256 units, length-bucketed "
                           "batches of <= 64 rows vs one utterance per batch (first %d items)" % (n_items, n_single))
        del parrot, gen
        return out

    def b1_split(precision, steps=100):
        """The single utterance's two halves on their own: the TTE alone (encode, the length round trip, decode:
        ~95 launches) and the vocoder alone on the ids it produced (replayed as a HIP graph from the fourth call on)."""
        ops.set_default_precision(ops.PREC_NAMES[precision])
        cfg, h, tsd, vsd, parrot, gen = build_models(dev, vocab, n_spk)
        batch = {k: v.to(dev) for k, v in synth.synth_tte_batch(1, S, vocab, n_spk, seed=0).items()}
        r = parrot.infer_dense(batch)
        ids, spk = r["ids"], batch["speaker"].reshape(-1, 1)
        out = {}
        for name, fn in (("tte_ms", lambda: parrot.infer_dense(batch)), ("vocoder_ms", lambda: gen(code=ids, spkr=spk, unit_lens=r["emitted_dev"]))):
            for _ in range(6):
                fn()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            fence()
            out[name] = (time.perf_counter() - t0) / steps * 1e3
        del parrot, gen
        return out

    elapsed, rows, n_samples_step, pieces = run(a.precision, a.steps, a.warmup, overlap_steps=bool(a.overlap))
    gather_ms = run.gather_ms
    head_step_ms = list(run.step_ms)

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        value = n_samples_step / (ms / 1e3)
        dom = rows[0]
        prof = pmc_profile(dom["kernel"])
        traffic, traffic_src = prof["traffic"], prof["source"]
        cfg_, h_ = pieces[0], pieces[1]
        tte_macs, voc_macs = algorithmic_macs(cfg_, h_, S, 4 * S)
        step_flops = 2.0 * world * B * ((tte_macs if a.workload == "full" else 0) + voc_macs)
        split = "_split" in dom["kernel"] or "resblock" in dom["kernel"]
        n_mfma = SCHEMES[a.precision][1]
        peak = MFMA16_PEAK_TFLOPS / n_mfma if split else FP32_MFMA_PEAK_TFLOPS
        # what a bare fp16 MFMA stream sustains on THIS box under its power limit (random operands; the constant-operand figure
        # shows the data dependence), per algorithmic fp32 FMA of the scheme: the practical ceiling next to the 2500 TF spec peak
        ceil = {}
        if split:
            for key, shape, const in (("f16_16x16x32_random", 1, 0), ("f16_32x32x16_random", 0, 0), ("f16_16x16x32_constant", 1, 1)):
                v_ = C.c_double()
                _lib.check(lib.parrot_debug_mfma_ceiling(shape, const, C.byref(v_)))
                ceil[key] = v_.value
        roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                "ceiling_probe_tflops": ({"mfma_stream": ceil, "per_fp32_fma": ceil["f16_16x16x32_random"] / n_mfma,
                                          "frac_of_ceiling": dom["tflops"] / (ceil["f16_16x16x32_random"] / n_mfma),
                                          "note": "parrot_debug_mfma_ceiling: bare v_mfma_f32_16x16x32_f16 stream, 2 waves/SIMD, measured in this run"}
                                         if ceil else None),
                "peak_note": (f"dense 16-bit MFMA peak 2500 TF / {n_mfma} MFMA(s) per algorithmic fp32 FMA ({a.precision}); the kernels are "
                              "clock/power-limited on real data (DESIGN.md section 7)" if split else "fp32 MFMA peak"),
                "frac": dom["tflops"] / peak, "traffic": traffic, "traffic_source": traffic_src, "traffic_box": prof["box"],
                "mfma_busy": prof["mfma_busy"], "avg_launch_us": dom["avg_us"],
                # the WHOLE step against the same roof: algorithmic FLOPs of the path (TTE + generator, real taps) / step time / peak
                "step_tflops": step_flops / (ms / 1e3) / 1e12 / world, "step_frac": step_flops / (ms / 1e3) / 1e12 / world / peak,
                "step_algorithmic_tflop": step_flops / 1e12,
                "launches_per_step": dom["launches_per_step"], "alg_GBps": dom["alg_gbs"], "all_conv_kernels": rows}
        res = {
            "metric": "audio samples/sec (16 kHz; see config.note) + RTF, 256-unit batch-64 per GPU",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms,
            # HIP-event time of every step of the timed region (rank 0): median / min / max next to the mean above
            "ms_per_step_median": head_step_ms[len(head_step_ms) // 2] if head_step_ms else None,
            "ms_per_step_min": head_step_ms[0] if head_step_ms else None, "ms_per_step_max": head_step_ms[-1] if head_step_ms else None,
            "timing": "value / ms_per_step: wall clock over the K steps between barrier + synchronize fences (max over ranks); "
                      "ms_per_step_median: median of the K per-step HIP-event times" +
                      ("; pipelined schedule: a step's event pair brackets what that step enqueues on its own stream, i.e. the vocoder of "
                       "the PREVIOUS batch (the first step of the region carries none: ms_per_step_min; the last batch's vocoder runs in "
                       "the closing flush, inside the fences)" if (a.overlap and a.workload == "full") else ""),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16x3": "f32 (f16x3 split products)", "bf16x6": "f32 (bf16x6 split products)", "f32": "f32", "bf16": "bf16 products, f32 accumulate",
                      "f16": "f16 products, f32 accumulate"}[a.precision],
            "data": "synthetic",
            "precision": {"f16x3": "fp32 data/accumulate; products of layers with >=16 channels evaluated as 3 fp16 MFMAs on 2-way fp16 splits "
                                   "of both (power-of-two pre-scaled) fp32 operands: fp32-class error, same parity tolerances; other layers exact fp32 MFMA",
                          "bf16x6": "fp32 data/accumulate; products of layers with >=16 channels evaluated as 6 bf16 MFMAs on 3-way bf16 splits of "
                                    "both fp32 operands (fp32-class error, same parity tolerances); other layers exact fp32 MFMA",
                          "f32": "exact fp32 MFMA everywhere",
                          "bf16": "REDUCED precision: operands rounded once to bf16, one MFMA per product group, fp32 accumulate / residual stream",
                          "f16": "REDUCED precision: operands rounded once to fp16, one MFMA per product group, fp32 accumulate / residual stream",
                          }[a.precision],
            "rtf": (ms / 1e3) / (n_samples_step / SAMPLE_RATE),
            "config": {"workload": ("full TTE(S=%d)->length-regulator(L=%d)->HiFi-GAN(%d samples/utt)" % (S, 4 * S, 4 * S * 320))
                       if a.workload == "full" else "HiFi-GAN generator only, %d units" % (4 * S),
                       "per_gpu_batch": B, "global_batch": world * B, "units_per_utt": 4 * S, "sample_rate": SAMPLE_RATE,
                       "parallelism": f"dp{world} (batch shard, RCCL waveform gather)",
                       "weights": "seeded synthetic, reference checkpoint layouts (TTE seed 42 forced duration 4; vocoder seed 1234)",
                       "note": "reference vocoder is 16 kHz / 320 samples per unit (utils/vocoder/config.json:24,32), not 22.05 kHz; "
                               "BASELINE configs[2] says single-speaker: this run uses the 10-speaker TTE (speaker_emb present, one extra "
                               "add kernel per step, parrot.py:98-99) and the multispkr vocoder -- a superset of the single-speaker work"},
            "err_vs_fp64": product_error_vs_fp64(),
            "roofline": roof,
            # the waveform gather's share of a step (max over ranks, CUDA events around the collective; 0 at N = 1) and what the
            # process group really is, so a scaling curve can be decomposed into compute and collective
            "gather_ms": gather_ms, "dist": pdist.dist_info(),
        }
    if rank == 0:
        res["schedule"] = ("two-stage pipeline across steps: the TTE of batch i runs on a side HIP stream beside the vocoder of batch "
                           "i-1 (SynthesisPipeline.submit); every batch submitted inside the timed region is finished inside it"
                           if (a.overlap and a.workload == "full") else "one batch at a time")
    if world == 1 and a.workload == "full" and not a.no_alt:
        # the same steps under the other schedule, for reference (kernel timings of the pipelined schedule include the
        # side stream's interference, so `roofline` is only clean with one batch at a time: the default headline)
        e1, _, n1, _ = run(a.precision, a.steps, max(a.warmup, 3), overlap_steps=not a.overlap)
        ms1 = e1 / a.steps * 1e3
        res["sequential_steps" if a.overlap else "pipelined_steps"] = {
            "value": n1 / (ms1 / 1e3), "unit": "samples/s", "ms_per_step": ms1, "steps": a.steps,
            "schedule": "one batch at a time" if a.overlap else "TTE of batch i on a side HIP stream beside the vocoder of batch i-1 "
                        "(SynthesisPipeline.submit / flush); all submitted batches finish inside the timed region"}
    if world == 1 and a.workload == "full" and not a.no_alt:
        # BASELINE.json configs[1]: HiFi-GAN generator only, batch 32 x 256 units -- in the run's precision and, as the
        # config is worded ("fp32"), with every product on the exact fp32 MFMA
        for key, prec_ in (("vocoder_only_b32", a.precision), ("vocoder_only_b32_exact_fp32", "f32")):
            e3, rows3, n3, _ = run(prec_, a.steps if prec_ != "f32" else min(a.steps, 5), max(a.warmup, 2), workload="vocoder", B=32)
            ms3 = e3 / (a.steps if prec_ != "f32" else min(a.steps, 5)) * 1e3
            res[key] = {"value": n3 / (ms3 / 1e3), "unit": "samples/s", "ms_per_step": ms3, "precision": prec_,
                        "workload": "HiFi-GAN generator only, batch 32 x %d units (BASELINE configs[1])" % (4 * S),
                        "dominant_kernel": rows3[0]["kernel"], "dominant_tflops": rows3[0]["tflops"]}
        # BASELINE.json configs[0] shape: ONE utterance end to end (latency-bound: ~130 launches)
        e4, _, n4, _ = run(a.precision, max(a.steps, 50), max(a.warmup, 3), B=1, profile=False, overlap_steps=False)
        ms4 = e4 / max(a.steps, 50) * 1e3
        res["single_utterance_b1"] = {"value": n4 / (ms4 / 1e3), "unit": "samples/s", "ms_per_step": ms4, "rtf": (ms4 / 1e3) / (n4 / SAMPLE_RATE),
                                      "workload": "full pipeline, ONE utterance (S=%d -> %d units), BASELINE configs[0] shape" % (S, 4 * S),
                                      "schedule": "one utterance at a time (latency: no cross-batch pipeline)", "split": b1_split(a.precision)}
        # BASELINE.json configs[4]: long-form 30 s utterances, batch 8 x 1500 units, chunk-streamed vocoder (256-unit chunks)
        res["long_form_b8_u1500"] = long_form(a.precision, min(a.steps, 5))
        # the shipped vocoder driver, batched vs one utterance per launch chain (host post-processing included)
        res["driver_e2e"] = driver_e2e(a.precision)
    if world == 1 and a.precision != "f32" and not a.no_alt:
        # the same workload with every product on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32), for reference
        e2, rows2, n2, _ = run("f32", min(a.steps, 5), 1)
        ms2 = e2 / min(a.steps, 5) * 1e3
        res["exact_fp32_mfma"] = {"value": n2 / (ms2 / 1e3), "unit": "samples/s", "ms_per_step": ms2, "steps": min(a.steps, 5),
                                  "dominant_kernel": rows2[0]["kernel"], "dominant_tflops": rows2[0]["tflops"],
                                  "frac_of_fp32_mfma_peak": rows2[0]["tflops"] / FP32_MFMA_PEAK_TFLOPS}
    if world == 1 and a.precision in ("f16x3", "bf16x6", "f32") and not a.no_alt:
        # BASELINE.json configs[2] names a bf16 vocoder: the reduced-precision operating point as a COMPANION line (never the
        # headline): fp32 TTE kernels' products and the vocoder's on ONE bf16 MFMA, fp32 accumulate / residual stream.
        # SNR vs the fp32 reference waveform: tests/test_gpu_baseline_shapes.py (>= 35.9 dB, the reference under autocast).
        for key, prec_ in (("bf16_vocoder", "bf16"), ("f16_vocoder", "f16")):
            # as configs[2] is worded: the TTE stays fp32-class (f16x3 products, parity-grade ids), only the vocoder drops to one MFMA
            e5, rows5, n5, _ = run(prec_, a.steps, max(a.warmup, 2), tte_precision="f16x3", overlap_steps=bool(a.overlap))
            ms5 = e5 / a.steps * 1e3
            d5 = rows5[0]  # (the timed-region row: stages 0-1 on the 128 x 160 tile -- vocoder launches only)
            mf, hb = d5["tflops"] / MFMA16_PEAK_TFLOPS, d5["alg_gbs"] / HBM_PEAK_GBS
            res[key] = {"value": n5 / (ms5 / 1e3), "unit": "samples/s", "ms_per_step": ms5, "precision": prec_, "tte_precision": "f16x3",
                        "schedule": "pipelined across steps" if a.overlap else "one batch at a time",
                        "note": "reduced precision (single %s MFMA per product group in the vocoder, fp32 accumulate and fp32 residual stream; the "
                                "layer-by-layer convs of stages 0-2 hand their activations over as pre-activated 16-bit operand planes written by "
                                "the producer's epilogue and fetched global -> LDS without VALU work (csrc/conv_split16.h); TTE in the parity-grade "
                                "f16x3 scheme): NOT parity-grade, companion to the headline" % prec_[:4],
                        "dominant_kernel": d5["kernel"], "dominant_tflops": d5["tflops"],
                        "roofline": {"kernel": d5["kernel"], "ms_per_step": d5["ms_per_step"], "launches_per_step": d5["launches_per_step"],
                                     "mfma": {"achieved": d5["tflops"], "peak": MFMA16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mf},
                                     "hbm": {"achieved": d5["alg_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s (algorithmic, fp32-sized activations)", "frac": hb},
                                     "bound": ("hbm" if hb > mf else "mfma") + ": neither roof is reached -- with the conversion gone from five of the "
                                              "six convs of a ResBlock (operand planes) a 32-channel chunk is bound by its weight fragments through "
                                              "the vector L1 (352 KB per chunk and CU at 64 B/clk = 5.5 k clocks against 7 k clocks of MFMA) and its "
                                              "LDS fragment reads, not by either roof (DESIGN.md section 7, round 6)"}}
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            cfg, h, tsd, vsd = pieces
            res["cpu_baseline"] = cpu_baseline(cfg, h, tsd, vsd, a.cpu_batch, S, vocab, n_spk)
            # the demo-style single utterance (BASELINE configs[0]); B=64 on the CPU would take ~45 s per pass and is
            # not run: B=8 above is already at the CPU's throughput plateau
            res["cpu_baseline_b1"] = cpu_baseline(cfg, h, tsd, vsd, 1, S, vocab, n_spk)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
