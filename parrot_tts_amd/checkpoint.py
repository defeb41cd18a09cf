"""Checkpoint formats of the reference, read with plain torch.load (no lightning needed).

* TTE: Lightning ``.ckpt`` = {'state_dict': {'parrot.<key>': tensor}, 'hyper_parameters': {data_config,
  src_vocab_size, src_pad_idx}, ...} (reference inference.py:10-18,43; train.py:61,144-151).
* vocoder: ``g_%08d`` = {'generator': state_dict} with weight_g / weight_v keys
  (reference utils/vocoder/train.py:183-186; inference.py:104-109 picks the lexicographically last)."""
from __future__ import annotations

import glob
import os
from typing import Optional

import torch
import torch.nn as nn

from .tte import Parrot
from .vocoder import AttrDict, CodeGenerator


class LitParrot(nn.Module):
    """Stand-in for the reference's LightningModule wrapper (inference.py:10-23): holds ``self.parrot``,
    ``infer`` switches to eval mode first, ``load_from_checkpoint`` understands Lightning's layout."""

    def __init__(self, data_config, src_vocab_size, src_pad_idx):
        super().__init__()
        self.hparams = AttrDict(data_config=data_config, src_vocab_size=src_vocab_size, src_pad_idx=src_pad_idx)
        self.parrot = Parrot(data_config, src_vocab_size, src_pad_idx)

    def infer(self, batch, row_exact: bool = False):
        self.eval()
        return self.parrot.infer(batch, row_exact=row_exact)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location="cpu", weights_only: bool = True, **overrides):
        ckpt = torch.load(checkpoint_path, map_location=map_location, weights_only=weights_only)
        hp = dict(ckpt.get("hyper_parameters", {}))
        hp.update(overrides)
        missing = [k for k in ("data_config", "src_vocab_size", "src_pad_idx") if k not in hp]
        if missing:
            raise KeyError(f"checkpoint lacks hyper_parameters {missing}; pass them as keyword arguments")
        model = cls(hp["data_config"], hp["src_vocab_size"], hp["src_pad_idx"])
        model.load_state_dict(ckpt["state_dict"], strict=True)
        return model


def save_lightning_style(path, parrot_state_dict, data_config, src_vocab_size, src_pad_idx) -> None:
    """Write a checkpoint in the layout ``LitParrot.load_from_checkpoint`` (and Lightning) reads."""
    torch.save({"state_dict": {"parrot." + k: v for k, v in parrot_state_dict.items()},
                "hyper_parameters": {"data_config": data_config, "src_vocab_size": src_vocab_size, "src_pad_idx": src_pad_idx}}, path)


def scan_checkpoint(cp_dir, prefix: str, pattern: str = "????????") -> Optional[str]:
    """reference utils/vocoder/utils.py:62-67 (``prefix + '????????'``, the training scripts' rule).  The inference driver
    carries its own copy with ``prefix + '*'`` (utils/vocoder/inference.py:57-62): pass ``pattern='*'`` for that one."""
    found = sorted(glob.glob(os.path.join(cp_dir, prefix + pattern)))
    return found[-1] if found else None


def load_generator(h, checkpoint: str, device) -> CodeGenerator:
    """reference utils/vocoder/inference.py:103-109,136-137: build, load {'generator': sd}, eval, fold weight norm."""
    path = scan_checkpoint(checkpoint, "g_", "*") if os.path.isdir(checkpoint) else checkpoint  # the driver's own glob (inference.py:57-62)
    if not path or not os.path.isfile(path):
        raise FileNotFoundError(f"no generator checkpoint at {checkpoint}")
    g = CodeGenerator(h if isinstance(h, AttrDict) else AttrDict(h))
    g.load_state_dict(torch.load(path, map_location="cpu", weights_only=True)["generator"])
    g.eval()
    g.remove_weight_norm()
    return g.to(device)
