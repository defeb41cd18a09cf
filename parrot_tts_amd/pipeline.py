"""Text-units pipeline: TTE (Parrot) -> unit ids -> HiFi-GAN vocoder (CodeGenerator), device resident.

Counterpart of running reference inference.py and then utils/vocoder/inference.py, without the
predictions.txt round trip: the dense (B, L) unit ids of ``Parrot.infer_dense`` feed the generator
directly.  Utterances are independent, so multi-GPU operation is a plain batch shard
(``shard_batch``) with one final waveform gather (``gather_waveforms``) -- the data-parallel
scheme of reference utils/vocoder/inference.py:201-205,255 with RCCL instead of a process pool."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .tte import Parrot
from .vocoder import CodeGenerator


class SynthesisPipeline:
    def __init__(self, parrot: Parrot, generator: CodeGenerator):
        self.parrot, self.generator = parrot, generator

    @torch.no_grad()
    def __call__(self, batch: Dict[str, torch.Tensor], spkr: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """batch: collated TTE batch (phones, src_mask, speaker).  ``spkr`` (B,1): vocoder speaker ids
        (defaults to the TTE speaker ids).  Returns wav (B,1,hop*L), n_samples (B,) = hop*emitted ids
        per row (rows shorter than L emit len+1 ids, reference quirk Q2), ids, tgt_mask.  wav[b, :, n_samples[b]:] is
        unspecified (padding)."""
        r = self.parrot.infer_dense(batch)
        ids = r["ids"]
        if spkr is None and self.generator.multispkr:
            spkr = batch["speaker"].reshape(-1, 1)
        hop = self.generator.upsample_factor
        L = ids.shape[1]
        emitted = torch.clamp(r["lens"].to(torch.int64) + 1, max=L)  # ids per row as Parrot.infer returns them (Q2)
        # each row is vocoded with its own sequence end, i.e. exactly as the reference would vocode that row's ids alone
        wav = self.generator(code=ids, spkr=spkr, unit_lens=emitted)
        return {"wav": wav, "n_samples": emitted * hop, "ids": ids, "tgt_mask": r["tgt_mask"], "lens": r["lens"]}
