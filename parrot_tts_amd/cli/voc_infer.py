#!/usr/bin/env python3
"""Counterpart of reference utils/vocoder/inference.py (lines 146-175, 178-265): unit manifest -> WAV files.

    python -m parrot_tts_amd.cli.voc_infer --checkpoint_file CKPT_OR_DIR --config utils/vocoder/config.json \
        --input_code_file predictions.txt --output_dir out --vc

Same arguments as the reference driver (--code_file 'name|u1 u2 ...' lists, --input_code_file manifests, --vc, --parts,
--pad, -n; the unused --f0-stats / --random-speakers / --unseen-f0 / --debug are accepted and ignored) plus --config
(the reference hard-codes utils/vocoder/config.json, inference.py:95) and --all_speakers_batch.

One process per GPU (torchrun sets RANK/WORLD_SIZE; a plain `python -m ...` run is a single rank): manifest items are
sharded round-robin over ranks -- the reference's Pool(8)+Queue of GPU ids (inference.py:201-205,255) without shared
state.  A rank's items are vocoded as LENGTH-BUCKETED PADDED BATCHES (--batch_rows / --batch_units) with per-row unit counts:
every layer zero-pads at each row's own end, so each row equals the reference's B=1 call bit for bit, while the launches
run in the throughput regime instead of one utterance at a time.  With --vc and a multi-speaker model every item is
synthesised under all ten speakers of the fixed table (inference.py:159-170): ten rows of the batch.  The int16 PCM of a
batch is copied to pinned host memory on a side stream and normalised / written by worker threads while the next batch
is being synthesised.  Post-processing as the reference: x*32768 -> int16 (C cast) -> float32 -> peak-normalise -> scipy
WAV at h.sampling_rate; items whose ground-truth wav exists are trimmed to it and get a `_gt.wav` beside them
(dataset.py:226-229, inference.py:172-175).

Where this driver deliberately differs from the reference's:
  * the reference only synthesises under --vc (inference.py:157: without it nothing but `_gt.wav` is written); here a
    multi-speaker model without --vc synthesises each item under its OWN speaker (speaker parsed from the file name,
    id from the fixed table), a single-speaker model writes `<name>_gen.wav`;
  * a missing ground-truth wav is not an error (the units are then used untrimmed);
  * `-n -1` = all items is the default (reference: 10, counted after a random shuffle of the items)."""
import argparse
import json
import os
from pathlib import Path

import numpy as np
import torch
from scipy.io.wavfile import write

from .. import dist as pdist
from ..checkpoint import load_generator
from ..data import VOCODER_SPEAKERS, CodeDataset, parse_manifest, parse_speaker, peak_normalize
from ..ops import wav_to_int16
from ..vocoder import AttrDict


def build_dataset(a, h):
    """reference inference.py:112-127: --code_file gives (name | units) pairs, otherwise a manifest through CodeDataset."""
    if a.code_file is not None:
        items = []
        for line in open(a.code_file):
            if line.strip():
                name, units = line.strip().split("|")
                items.append(({"code": np.asarray([int(v) for v in units.split(" ")], dtype=np.int64)}, None, name, None))
        return items
    return CodeDataset(parse_manifest(a.input_code_file), -1, h.code_hop_size, sampling_rate=h.sampling_rate,
                       multispkr=h.get("multispkr", None), pad=a.pad)


def plan_batches(lengths, max_rows, max_units):
    """Length-bucketed batches over row indices: rows sorted by unit count (longest first, so the first batch sizes the
    workspace), cut where another row would exceed ``max_rows`` rows or ``max_units`` padded units (rows x longest row).
    Every row lands in exactly one batch; a row longer than ``max_units`` gets a batch of its own."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    batches, cur = [], []
    for i in order:
        longest = lengths[cur[0]] if cur else lengths[i]
        if cur and (len(cur) + 1 > max_rows or (len(cur) + 1) * longest > max_units):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def run_batched(gen, rows, dev, sampling_rate, max_rows=64, max_units=16384, write_wavs=True):
    """Vocode ``rows`` = [(units int64 array, speaker id or None, output path)] as padded, length-bucketed batches with per-row
    ``unit_lens`` -- every layer zero-pads at each row's own end, so a row equals the reference's B = 1 call of that utterance
    (utils/vocoder/inference.py:149-170) -- and write one WAV per row, post-processed as the reference does (x * 32768 -> int16
    -> float32 -> peak-normalise).  The int16 batch goes to pinned host memory on a side stream and is normalised / written by
    a worker thread while the next batch is being synthesised.  Returns the number of WAVs written."""
    from concurrent.futures import ThreadPoolExecutor
    if not rows:
        return 0
    hop = gen.upsample_factor
    multi = bool(gen.multispkr)
    lengths = [int(r[0].size) for r in rows]
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream(dev)

    def finish(pcm_host, ready, idx, lens):
        ready.synchronize()  # the D2H copy of this batch (a CUDA event: no device-wide sync)
        pcm = pcm_host.numpy()
        for r, i in enumerate(idx):
            audio = peak_normalize(pcm[r, : gen.out_samples(lens[r])].astype(np.float32))
            if write_wavs:
                write(rows[i][2], sampling_rate, audio)
        return len(idx)

    done = 0
    with ThreadPoolExecutor(max_workers=2) as pool:
        futures = []
        for idx in plan_batches(lengths, max_rows, max_units):
            lens = [lengths[i] for i in idx]
            U = max(lens)
            code_h = torch.zeros((len(idx), U), dtype=torch.int64).pin_memory()
            for r, i in enumerate(idx):
                code_h[r, : lens[r]] = torch.from_numpy(rows[i][0])
            code = code_h.to(dev, non_blocking=True)
            spk = torch.tensor([[rows[i][1]] for i in idx], device=dev) if multi else None
            wav = gen(code=code, spkr=spk, unit_lens=torch.tensor(lens, dtype=torch.int32, device=dev))
            pcm = wav_to_int16(wav.squeeze(1))
            ev = main.record_event()
            pcm_host = torch.empty(pcm.shape, dtype=torch.int16).pin_memory()
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(ev)
                pcm_host.copy_(pcm, non_blocking=True)
                pcm.record_stream(copy_stream)
                ready = copy_stream.record_event()
            futures.append(pool.submit(finish, pcm_host, ready, idx, lens))
        for f in futures:
            done += f.result()
    return done


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--code_file", default=None)
    ap.add_argument("--input_code_file", default="runs/vocoder/val.txt")
    ap.add_argument("--output_dir", default="generated_files")
    ap.add_argument("--checkpoint_file", required=True)
    ap.add_argument("--config", default="utils/vocoder/config.json")
    ap.add_argument("--vc", action="store_true")
    ap.add_argument("--pad", default=None, type=int)
    ap.add_argument("--parts", action="store_true")
    ap.add_argument("-n", type=int, default=-1, help="number of items (default: all)")
    ap.add_argument("--batch_rows", type=int, default=64, help="rows (item x speaker) per vocoder batch")
    ap.add_argument("--batch_units", type=int, default=16384, help="padded units (rows x longest row) per vocoder batch")
    for ignored in ("--f0-stats", "--unseen-f0"):
        ap.add_argument(ignored, type=Path, default=None, help="accepted for command-line compatibility; unused by the reference's path too")
    for ignored in ("--random-speakers", "--debug"):
        ap.add_argument(ignored, action="store_true", help="accepted for command-line compatibility")
    a = ap.parse_args(argv)
    rank, world, local = pdist.init_from_env()
    dev = pdist.local_device(local)
    with open(a.config) as f:
        h = AttrDict(json.load(f))
    gen = load_generator(h, a.checkpoint_file, dev)
    dataset = build_dataset(a, h)
    n = len(dataset) if a.n < 0 else min(a.n, len(dataset))
    os.makedirs(a.output_dir, exist_ok=True)
    multi = bool(h.get("multispkr"))
    # 1) this rank's rows: one row per (item, speaker) -- the --vc fan-out of an item stays inside the batch
    rows = []  # (units, speaker id or None, output path)
    for item in range(rank, n, world):
        feats, gt_audio, filename, _ = dataset[item]
        name = "_".join(Path(filename).parts[-3:])[:-4] if a.parts else Path(filename).stem
        code = np.asarray(feats["code"], dtype=np.int64).reshape(-1)
        if code.size == 0:
            continue
        if gt_audio is not None:  # inference.py:172-175
            gt = peak_normalize(gt_audio.squeeze().numpy().astype(np.float32))
            write(os.path.join(a.output_dir, name + "_gt.wav"), h.sampling_rate, gt)
        if multi and a.vc:
            spk_names = list(VOCODER_SPEAKERS)
        elif multi:
            spk_names = [parse_speaker(filename, h["multispkr"])]
        else:
            spk_names = [None]
        for s_ in spk_names:
            rows.append((code, VOCODER_SPEAKERS[s_] if s_ is not None else None,
                         os.path.join(a.output_dir, f"{name}_{s_}_gen.wav" if s_ is not None else f"{name}_gen.wav")))
    n_wavs = run_batched(gen, rows, dev, h.sampling_rate, a.batch_rows, a.batch_units)
    gen.check_inputs()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(f"synthesised {n} items ({n_wavs} wavs on rank 0) into {a.output_dir}")


if __name__ == "__main__":
    main()
