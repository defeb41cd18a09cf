#!/usr/bin/env python3
"""SHA-256 of the vocoder output for a few launch shapes (dense and ragged batches): two builds / switch settings that claim the
same arithmetic must print the same lines.   PARROT_PLANES=0 python tools/wav_hash.py ; PARROT_PLANES=1 python tools/wav_hash.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import synth  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402


def main():
    h = synth.default_voc_config()
    g = CodeGenerator(AttrDict(h))
    g.load_state_dict(synth.synth_voc_state_dict(h, seed=1234, scale=1.0))
    g = g.eval().to("cuda:0")
    shapes = [(64, 256, False), (64, 256, True), (32, 256, False), (16, 200, True), (8, 1500, True), (3, 77, True), (1, 256, False)]
    if "--quick" in sys.argv:
        shapes = [(64, 256, True), (16, 200, True), (2, 77, True)]
    for B, U, ragged in shapes:
        vb = synth.synth_voc_batch(B, U, h, seed=B + U)
        lens = None
        if ragged:
            gen = torch.Generator().manual_seed(B * 1000 + U)
            lens = torch.randint(max(1, U // 3), U + 1, (B,), generator=gen)
            lens[0] = U
        with torch.no_grad():
            y = g(code=vb["code"].to("cuda:0"), spkr=vb["spkr"].to("cuda:0"), unit_lens=lens)
        if lens is not None:  # (samples past a row's end are unspecified)
            hop = g.upsample_factor
            y = y.clone()
            for b in range(B):
                y[b, :, int(lens[b]) * hop:] = 0
        print(B, U, "ragged" if ragged else "dense", hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:24], flush=True)


if __name__ == "__main__":
    main()
