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
    """``row_exact`` (default False): the TTE evaluates every row as the reference evaluates that utterance ALONE (its drivers run
    batch_size = 1, inference.py:34) instead of the reference's padded-batch result, which depends on the batch composition (quirk
    Q7) -- see ``Parrot.infer``.  The vocoder always treats rows independently (``unit_lens``)."""

    def __init__(self, parrot: Parrot, generator: CodeGenerator, row_exact: bool = False):
        self.parrot, self.generator = parrot, generator
        self.row_exact = bool(row_exact)
        self._side: Optional[torch.cuda.Stream] = None
        self._pending: Optional[dict] = None
        # `submit`: the next batch's TTE starts when the previous batch's vocoder reaches this MRF stage (PARROT_PIPE_STAGE; -1: at once)
        import os
        self.pipe_stage = int(os.environ.get("PARROT_PIPE_STAGE", "2"))

    def _emitted(self, r, L):
        """ids per row as ``Parrot.infer`` returns them, on the host and on the device: len + 1 clamped to L in the padded-batch
        mode (quirk Q2), exactly len in the row-exact mode.  (The device copy comes from the lengths the encoder left there --
        ``Parrot._run`` forms it before the decoder is enqueued: handing the host tensor over would be a pageable host-to-device
        copy, which blocks the host until the decoder has drained -- 0.2-0.3 ms of idle GPU per batch.)"""
        if r.get("row_exact"):
            return r["lens"].to(torch.int64), r["emitted_dev"]
        return torch.clamp(r["lens"].to(torch.int64) + 1, max=L), r["emitted_dev"]

    @torch.no_grad()
    def __call__(self, batch: Dict[str, torch.Tensor], spkr: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """batch: collated TTE batch (phones, src_mask, speaker).  ``spkr`` (B,1): vocoder speaker ids
        (defaults to the TTE speaker ids).  Returns wav (B,1,hop*L), n_samples (B,) = hop*emitted ids
        per row (rows shorter than L emit len+1 ids, reference quirk Q2), ids, tgt_mask.  wav[b, :, n_samples[b]:] is
        unspecified (padding)."""
        # the vocoder's device flag of the PREVIOUS call (bad unit id, non-finite waveform: an activation beyond the fp16 split
        # scheme's range) is read with the TTE's length transfer: a checkpoint that leaves the range fails loudly, by default,
        # one call late and at no extra synchronisation (`check()` covers the last call)
        r = self.parrot.infer_dense(batch, status_hooks=(self.generator._status_hook,), row_exact=self.row_exact)
        ids = r["ids"]
        if spkr is None and self.generator.multispkr:
            spkr = batch["speaker"].reshape(-1, 1)
        emitted, emitted_dev = self._emitted(r, ids.shape[1])
        # each row is vocoded with its own sequence end, i.e. exactly as the reference would vocode that row's ids alone
        wav = self.generator(code=ids, spkr=spkr, unit_lens=emitted_dev)
        return {"wav": wav, "n_samples": self.generator.out_samples(emitted), "ids": ids, "tgt_mask": r["tgt_mask"], "lens": r["lens"]}

    def check(self) -> None:
        """Synchronise and raise what the device flagged in the calls so far (IndexError: bad ids; FloatingPointError: non-finite
        logits / waveform).  `__call__` reports the previous call's flags by itself; this covers the last one."""
        self.parrot.check_outputs()
        self.generator.check_inputs()

    # ---- two-stage software pipeline across batches --------------------------------------------------------------
    # The TTE is ~13 % of a step but made of small, latency-bound launches (64-512 workgroups, one host round trip for
    # the expanded length); the vocoder is 100+ chip-filling launches.  `submit` runs the TTE of batch i on a side HIP
    # stream while the vocoder of batch i-1 -- enqueued first, on the caller's stream -- owns the chip, so the TTE's
    # bubbles are filled by vocoder workgroups.  Results come back one call late, in order; `flush` drains.
    @torch.no_grad()
    def submit(self, batch: Dict[str, torch.Tensor], spkr: Optional[torch.Tensor] = None) -> Optional[Dict[str, torch.Tensor]]:
        """Enqueue the vocoder of the previously submitted batch, run this batch's TTE beside it, return the previous
        batch's result (None on the first call).  Same kernels and results as ``__call__``."""
        dev = batch["phones"].device
        main = torch.cuda.current_stream(dev)
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        inputs_ready = main.record_event()  # recorded BEFORE the previous vocoder is enqueued behind it
        done = self._vocode_pending(main)
        with torch.cuda.stream(self._side):
            self._side.wait_event(inputs_ready)
            # ... and beside WHICH part of that vocoder (profiles/r06j_pipe_stage_ab.txt, B = 64, one box; one batch at a time: 18.41 ms):
            # at once 17.63 ms, with the stage 0 / 1 layer convs 19 % slower per launch while the TTE shares their CUs; from stage 2 on
            # 17.65-17.74 ms with those launches undisturbed (228 us, as alone); from stage 3 18.03, from stage 4 18.37 -- too late: the
            # TTE's own critical path (encoder, the length round trip, decoder) no longer fits beside what is left of the vocoder
            if done is not None and self.pipe_stage >= 0:
                self.generator.wait_stage(self.pipe_stage, self._side)
            # (no vocoder status hook here: this TTE runs on a side stream BESIDE the previous batch's vocoder, and the hook's
            #  read-and-clear of the flag would race with that forward; `check()` covers the pipelined schedule)
            r = self.parrot.infer_dense(batch, row_exact=self.row_exact)  # (its host sync for L waits on the side stream only)
            ids = r["ids"]
            if spkr is None and self.generator.multispkr:
                spkr = batch["speaker"].reshape(-1, 1)
            emitted, emitted_dev = self._emitted(r, ids.shape[1])
            self._pending = {"ids": ids, "spkr": spkr, "emitted": emitted, "emitted_dev": emitted_dev, "tgt_mask": r["tgt_mask"],
                             "lens": r["lens"], "event": self._side.record_event()}
        return done

    @torch.no_grad()
    def flush(self) -> Optional[Dict[str, torch.Tensor]]:
        """Vocode the last submitted batch (on the current stream) and return its result."""
        return self._vocode_pending(torch.cuda.current_stream()) if self._pending is not None else None

    def _vocode_pending(self, main: "torch.cuda.Stream") -> Optional[Dict[str, torch.Tensor]]:
        p, self._pending = self._pending, None
        if p is None:
            return None
        main.wait_event(p["event"])
        for t in (p["ids"], p["emitted_dev"], p["tgt_mask"]) + ((p["spkr"],) if p["spkr"] is not None else ()):
            t.record_stream(main)  # allocated on the side stream, consumed here
        wav = self.generator(code=p["ids"], spkr=p["spkr"], unit_lens=p["emitted_dev"])
        return {"wav": wav, "n_samples": self.generator.out_samples(p["emitted"]), "ids": p["ids"], "tgt_mask": p["tgt_mask"],
                "lens": p["lens"]}
