#!/usr/bin/env python3
"""Counterpart of reference inference.py (lines 25-72): read <root_path>/val.txt, run the TTE on the
GPU, write <root_path>/predictions.txt in the same one-python-dict-per-line format.

    python -m parrot_tts_amd.cli.tte_infer --config utils/TTE/TTE_config.yaml --checkpoint_pth CKPT [--batch_size 1] [--row_exact]

``--batch_size 1`` (default) reproduces the reference exactly (it never batches, inference.py:34).  Larger
batches are padded like ``ParrotDataset.collate_fn`` does and, because of reference quirks Q1/Q2/Q7
(pe[T] indexed by the padded length, conv leakage across pads), produce what the REFERENCE would produce for
that same padded batch -- not what it produces utterance by utterance.  ``--row_exact`` removes that difference at
any batch size: every row is evaluated as the reference evaluates that utterance alone (per-row pe[S_b] / pe[L_b],
per-row conv padding and key masks, exactly L_b units), so ``--batch_size 64 --row_exact`` writes the
``predictions.txt`` that ``--batch_size 1`` writes.  The `duration` field is the length of the
utterance's wav in seconds (librosa in the reference, inference.py:62-63; scipy here) when that file exists, otherwise
the duration of the emitted units (n_units / --units_per_second)."""
import argparse
import os

import torch
import yaml

from ..checkpoint import LitParrot
from ..data import ParrotDataset, format_dict_line, load_wav_int16_scale


def wav_seconds(path, fallback: float) -> float:
    if os.path.isfile(path):
        audio, sr = load_wav_int16_scale(path)
        return float(len(audio)) / float(sr)
    return fallback


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--checkpoint_pth", type=str, required=True)
    ap.add_argument("--device", type=str, default="cuda:0")
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--units_per_second", type=float, default=50.0)
    ap.add_argument("--row_exact", action="store_true",
                    help="evaluate every row of a padded batch as the reference evaluates that utterance alone (its driver's result)")
    a = ap.parse_args(argv)
    cfg = yaml.load(open(a.config, "r"), Loader=yaml.FullLoader)
    ds = ParrotDataset("val", data_config=cfg)
    model = LitParrot.load_from_checkpoint(a.checkpoint_pth, weights_only=True).to(a.device)
    audio_dir = cfg["path"].get("wav_path", "")
    order = sorted(range(len(ds)), key=lambda i: len(ds.data_list[i]["characters"].split(" "))) if a.batch_size > 1 else list(range(len(ds)))
    results = {}
    with torch.no_grad():
        for s in range(0, len(order), a.batch_size):
            idx = order[s: s + a.batch_size]
            batch = ds.collate_fn([ds[i] for i in idx])
            gpu = {k: (v.to(a.device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
            rows = model.infer(gpu, row_exact=a.row_exact)
            for i, name, codes in zip(idx, batch["ids"], rows):
                speaker = "_".join(name.split("_")[:2])
                wav = os.path.join(audio_dir, speaker, "wavs", name + ".wav")
                results[i] = {"audio": wav, "hubert": " ".join(map(str, codes)), "duration": wav_seconds(wav, len(codes) / a.units_per_second)}
    out = os.path.join(cfg["path"]["root_path"], "predictions.txt")
    with open(out, "w") as f:
        for i in range(len(ds)):
            f.write(format_dict_line(results[i]))
    print(f"wrote {len(results)} predictions to {out}")


if __name__ == "__main__":
    main()
