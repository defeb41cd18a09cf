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


def _default_row_groups():
    """PARROT_ROW_GROUPS: a group count ("1", the default: off) or explicit group sizes for one batch size ("16+48": used when they
    add up to the batch, an even split into as many groups otherwise)."""
    import os
    v = os.environ.get("PARROT_ROW_GROUPS", "1")
    if "+" in v:
        return [max(1, int(x)) for x in v.split("+")]
    return max(1, int(v))


class SynthesisPipeline:
    """``row_groups`` (default 1 = off, ``PARROT_ROW_GROUPS``): one batch can run as a software pipeline over groups of its rows --
    the TTE decoder of row group g + 1 on a side HIP stream beside the vocoder of group g (parrot_tte_decode_rows).  Results do not
    depend on the grouping: the encoder, the duration predictor and the expanded length L (pe[L] is indexed by the batch-max
    length, quirk Q7) are the whole batch's, and every decoder / vocoder kernel works row by row
    (tests/test_gpu_round4.py::test_row_groups_leave_every_output_bit_unchanged).  Measured at B = 64 x 256 units it does NOT pay:
    18.6 ms whole, 19.1 ms in two groups, 21.2 ms in three (tools/step_time.py) -- a 32-row decoder is hardly shorter than a 64-row
    one (its launches are latency-bound) and two 32-row vocoder passes cost 0.3 ms more than one 64-row pass, which is more than
    the hidden half decoder saves.  Kept as an option for pipelines whose vocoder batch is capped anyway.  Batches with fewer than
    16 rows per group run whole."""

    def __init__(self, parrot: Parrot, generator: CodeGenerator, row_groups: Optional[int] = None):
        self.parrot, self.generator = parrot, generator
        self.row_groups = _default_row_groups() if row_groups is None else row_groups
        self._side: Optional[torch.cuda.Stream] = None
        self._pending: Optional[dict] = None

    def _groups(self, B: int):
        rg = self.row_groups
        if isinstance(rg, (list, tuple)):
            if sum(rg) == B:
                out, b0 = [], 0
                for n in rg:
                    out.append((b0, int(n)))
                    b0 += int(n)
                return out
            rg = len(rg)
        g = min(max(1, int(rg)), max(1, B // 16))
        base, rem = divmod(B, g)
        out, b0 = [], 0
        for i in range(g):
            n = base + (1 if i < rem else 0)
            out.append((b0, n))
            b0 += n
        return out

    @torch.no_grad()
    def _call_grouped(self, batch, spkr, groups) -> Dict[str, torch.Tensor]:
        par, gen = self.parrot, self.generator
        dev = batch["phones"].device
        main = torch.cuda.current_stream(dev)
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        side = self._side
        ctx = par._encode(batch, status_hooks=(gen._status_hook,))  # whole batch, main stream; host sync for L
        B, L = ctx["B"], ctx["L"]
        ids = torch.empty((B, L), dtype=torch.int64, device=dev)
        tgt = torch.empty((B, L), dtype=torch.uint8, device=dev)
        if spkr is None and gen.multispkr:
            spkr = batch["speaker"].reshape(-1, 1)
        emitted = torch.clamp(ctx["lens"].to(torch.int64) + 1, max=L)  # (host) ids per row as Parrot.infer returns them (Q2)
        # the same on the device, from the lengths the encoder left there: a host-to-device copy of `emitted` would be a pageable
        # transfer, i.e. the host would wait for the whole decoder before it could enqueue the first vocoder kernel
        emitted_dev = torch.clamp(ctx["lens_dev"] + 1, max=L)
        wav = torch.empty((B, 1, gen.out_samples(L)), dtype=torch.float32, device=dev)
        side.wait_stream(main)  # the encoder's state, ids / tgt allocations
        for t in (ids, tgt):
            t.record_stream(side)

        def decode(g):
            b0, n = groups[g]
            with torch.cuda.stream(side):  # every group's decoder, in row order, on the side stream
                par._decode(ctx, ids, tgt, None, b0, n)
                return side.record_event()

        # enqueue order = the order the GPU should see the work in: decoder of group 0, then vocoder of group g right before the
        # decoder of group g + 2 ... so that the first vocoder kernels are in their queue when the first decoder finishes
        ev = decode(0)
        for g, (b0, n) in enumerate(groups):
            nxt = decode(g + 1) if g + 1 < len(groups) else None
            main.wait_event(ev)
            gen(code=ids[b0:b0 + n], spkr=None if spkr is None else spkr[b0:b0 + n], unit_lens=emitted_dev[b0:b0 + n], out=wav[b0:b0 + n])
            ev = nxt
        return {"wav": wav, "n_samples": gen.out_samples(emitted), "ids": ids, "tgt_mask": tgt.bool(), "lens": ctx["lens"]}

    @torch.no_grad()
    def __call__(self, batch: Dict[str, torch.Tensor], spkr: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """batch: collated TTE batch (phones, src_mask, speaker).  ``spkr`` (B,1): vocoder speaker ids
        (defaults to the TTE speaker ids).  Returns wav (B,1,hop*L), n_samples (B,) = hop*emitted ids
        per row (rows shorter than L emit len+1 ids, reference quirk Q2), ids, tgt_mask.  wav[b, :, n_samples[b]:] is
        unspecified (padding)."""
        groups = self._groups(batch["phones"].shape[0])
        # (the first forward of a handle runs whole: the shims take one synchronous look at the range flags there)
        if len(groups) > 1 and not self.parrot._probe_pending and self.parrot._handle is not None and self.generator._handle is not None \
                and not getattr(self.generator, "_probe_pending", False):
            return self._call_grouped(batch, spkr, groups)
        # the vocoder's device flag of the PREVIOUS call (bad unit id, non-finite waveform: an activation beyond the fp16 split
        # scheme's range) is read with the TTE's length transfer: a checkpoint that leaves the range fails loudly, by default,
        # one call late and at no extra synchronisation (`check()` covers the last call)
        r = self.parrot.infer_dense(batch, status_hooks=(self.generator._status_hook,))
        ids = r["ids"]
        if spkr is None and self.generator.multispkr:
            spkr = batch["speaker"].reshape(-1, 1)
        hop = self.generator.upsample_factor
        L = ids.shape[1]
        emitted = torch.clamp(r["lens"].to(torch.int64) + 1, max=L)  # ids per row as Parrot.infer returns them (Q2)
        # each row is vocoded with its own sequence end, i.e. exactly as the reference would vocode that row's ids alone
        # (the lengths are clamped on the DEVICE, from the copy the encoder left there: handing the host tensor over would be a
        #  pageable host-to-device copy, which blocks the host until the decoder has drained -- 0.2-0.3 ms of idle GPU per batch)
        wav = self.generator(code=ids, spkr=spkr, unit_lens=torch.clamp(r["lens_dev"] + 1, max=L))
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
            # (no vocoder status hook here: this TTE runs on a side stream BESIDE the previous batch's vocoder, and the hook's
            #  read-and-clear of the flag would race with that forward; `check()` covers the pipelined schedule)
            r = self.parrot.infer_dense(batch)  # (its host sync for L waits on the side stream only)
            ids = r["ids"]
            if spkr is None and self.generator.multispkr:
                spkr = batch["speaker"].reshape(-1, 1)
            emitted = torch.clamp(r["lens"].to(torch.int64) + 1, max=ids.shape[1])
            emitted_dev = torch.clamp(r["lens_dev"] + 1, max=ids.shape[1])
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
