#!/usr/bin/env python3
"""Counterpart of reference utils/vocoder/inference.py (lines 146-175, 178-265): unit manifest -> WAV files.

    python -m parrot_tts_amd.cli.voc_infer --checkpoint_file CKPT_OR_DIR --config utils/vocoder/config.json \
        --input_code_file predictions.txt --output_dir out --vc

One process per GPU (torchrun sets RANK/WORLD_SIZE; a plain `python -m ...` run is a single rank): manifest
items are sharded round-robin over ranks -- the reference's Pool(8)+Queue of GPU ids (inference.py:201-205,255)
without shared state.  With --vc and a multi-speaker model every item is synthesised under all ten speakers
of the fixed table (inference.py:159-170) as ONE batch of 10 rows; batch rows are independent in the vocoder,
so each row equals the reference's B=1 call.  Post-processing as the reference: x*32768 -> int16 (C cast) ->
float32 -> peak-normalise -> scipy WAV at h.sampling_rate."""
import argparse
import json
import os
from pathlib import Path

import numpy as np
import torch
from scipy.io.wavfile import write

from .. import dist as pdist
from ..checkpoint import load_generator
from ..data import VOCODER_SPEAKERS, parse_manifest, parse_speaker, peak_normalize
from ..ops import wav_to_int16
from ..vocoder import AttrDict


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--input_code_file", required=True)
    ap.add_argument("--output_dir", default="generated_files")
    ap.add_argument("--checkpoint_file", required=True)
    ap.add_argument("--config", default="utils/vocoder/config.json")
    ap.add_argument("--vc", action="store_true")
    ap.add_argument("-n", type=int, default=-1, help="number of items (default: all)")
    ap.add_argument("--parts", action="store_true")
    a = ap.parse_args(argv)
    rank, world, local = pdist.init_from_env()
    dev = pdist.local_device(local)
    with open(a.config) as f:
        h = AttrDict(json.load(f))
    gen = load_generator(h, a.checkpoint_file, dev)
    files, codes = parse_manifest(a.input_code_file)
    n = len(codes) if a.n < 0 else min(a.n, len(codes))
    os.makedirs(a.output_dir, exist_ok=True)
    multi = bool(h.get("multispkr"))
    for item in range(rank, n, world):
        name = "_".join(Path(files[item]).parts[-3:])[:-4] if a.parts else Path(files[item]).stem
        code = torch.from_numpy(codes[item]).to(dev).unsqueeze(0)
        if multi and a.vc:
            spk_names = list(VOCODER_SPEAKERS)
            spk = torch.tensor([[VOCODER_SPEAKERS[s]] for s in spk_names], device=dev)
            wav = gen(code=code.expand(len(spk_names), -1).contiguous(), spkr=spk)
        elif multi:
            spk_names = [parse_speaker(files[item], h["multispkr"])]
            spk = torch.tensor([[VOCODER_SPEAKERS[spk_names[0]]]], device=dev)
            wav = gen(code=code, spkr=spk)
        else:
            spk_names = ["gen"]
            wav = gen(code=code)
        pcm = wav_to_int16(wav.squeeze(1)).cpu().numpy()
        for row, s in zip(pcm, spk_names):
            audio = peak_normalize(row.astype(np.float32))
            write(os.path.join(a.output_dir, f"{name}_{s}_gen.wav"), h.sampling_rate, audio)
    gen.check_inputs()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(f"synthesised {n} items into {a.output_dir}")


if __name__ == "__main__":
    main()
