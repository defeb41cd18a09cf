#!/usr/bin/env python3
"""Counterpart of reference utils/vocoder/inference.py (lines 146-175, 178-265): unit manifest -> WAV files.

    python -m parrot_tts_amd.cli.voc_infer --checkpoint_file CKPT_OR_DIR --config utils/vocoder/config.json \
        --input_code_file predictions.txt --output_dir out --vc

Same arguments as the reference driver (--code_file 'name|u1 u2 ...' lists, --input_code_file manifests, --vc, --parts,
--pad, -n; the unused --f0-stats / --random-speakers / --unseen-f0 / --debug are accepted and ignored) plus --config
(the reference hard-codes utils/vocoder/config.json, inference.py:95) and --all_speakers_batch.

One process per GPU (torchrun sets RANK/WORLD_SIZE; a plain `python -m ...` run is a single rank): manifest items are
sharded round-robin over ranks -- the reference's Pool(8)+Queue of GPU ids (inference.py:201-205,255) without shared
state.  With --vc and a multi-speaker model every item is synthesised under all ten speakers of the fixed table
(inference.py:159-170) as ONE batch of 10 rows; batch rows are independent in the vocoder, so each row equals the
reference's B=1 call.  Post-processing as the reference: x*32768 -> int16 (C cast) -> float32 -> peak-normalise -> scipy
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
    for item in range(rank, n, world):
        feats, gt_audio, filename, _ = dataset[item]
        name = "_".join(Path(filename).parts[-3:])[:-4] if a.parts else Path(filename).stem
        code = torch.from_numpy(np.asarray(feats["code"], dtype=np.int64)).to(dev).unsqueeze(0)
        if code.shape[1] == 0:
            continue
        if multi and a.vc:
            spk_names = list(VOCODER_SPEAKERS)
            spk = torch.tensor([[VOCODER_SPEAKERS[s]] for s in spk_names], device=dev)
            wav = gen(code=code.expand(len(spk_names), -1).contiguous(), spkr=spk)
        elif multi:
            spk_names = [parse_speaker(filename, h["multispkr"])]
            spk = torch.tensor([[VOCODER_SPEAKERS[spk_names[0]]]], device=dev)
            wav = gen(code=code, spkr=spk)
        else:
            spk_names = [None]
            wav = gen(code=code)
        pcm = wav_to_int16(wav.squeeze(1)).cpu().numpy()
        for row, s in zip(pcm, spk_names):
            audio = peak_normalize(row.astype(np.float32))
            write(os.path.join(a.output_dir, f"{name}_{s}_gen.wav" if s is not None else f"{name}_gen.wav"), h.sampling_rate, audio)
        if gt_audio is not None:  # inference.py:172-175
            gt = peak_normalize(gt_audio.squeeze().numpy().astype(np.float32))
            write(os.path.join(a.output_dir, name + "_gt.wav"), h.sampling_rate, gt)
    gen.check_inputs()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(f"synthesised {n} items into {a.output_dir}")


if __name__ == "__main__":
    main()
