"""Drop-in ``Parrot`` for reference modules/parrot.py:12-120 backed by libparrot_hip.so.

Same constructor ``Parrot(data_config, src_vocab_size, src_pad_idx)`` (reads
``<root_path>/speakers.json`` like the reference, parrot.py:24-26), same ``state_dict`` keys
(SURVEY 8b), same ``infer(batch) -> List[List[int]]`` / ``forward(batch, inference=True)``.
The module only holds parameters; the arithmetic runs in the HIP library.  Training forward
(``inference=False``) is out of scope and raises."""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib
from .ops import PREC_BF16X6, PREC_F16X3, PREC_STR, dptr, param_fingerprint, range_fallback_default, require_cuda, stream_ptr
from .synth import sinusoid_table



def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False) -> None:
    """Register ``tensor`` under a dotted state_dict key, creating plain container modules on the way
    (numeric path parts are fine: nn.Module accepts '0' as a child name, as ModuleList does)."""
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, nn.Module())
        m = m._modules[p]
    if buffer:
        m.register_buffer(parts[-1], tensor)
    else:
        m.register_parameter(parts[-1], nn.Parameter(tensor))


def parrot_param_shapes(cfg: dict, vocab: int, n_speaker: int) -> Dict[str, tuple]:
    """state_dict key -> shape for the reference ``Parrot`` ([verified] there)."""
    tr, dp = cfg["transformer"], cfg["duration_predictor"]
    D, F = tr["d_model"], tr["conv_n_filter"]
    k1, k2 = tr["conv_kernel_sizes"]
    NF, DK = dp["n_filter"], dp["kernel_size"]
    sh = {"tok_emb.weight": (vocab, D)}
    if n_speaker > 1:
        sh["speaker_emb.weight"] = (n_speaker, D)
    sh["duration_predictor.layers.0.conv.weight"] = (NF, D, DK)
    sh["duration_predictor.layers.0.conv.bias"] = (NF,)
    sh["duration_predictor.layers.4.conv.weight"] = (NF, NF, DK)
    sh["duration_predictor.layers.4.conv.bias"] = (NF,)
    for i in (2, 6):
        sh[f"duration_predictor.layers.{i}.weight"] = (NF,)
        sh[f"duration_predictor.layers.{i}.bias"] = (NF,)
    sh["duration_predictor.proj.weight"] = (1, NF)
    sh["duration_predictor.proj.bias"] = (1,)
    for side in ("encoder", "decoder"):
        for n in range(tr[side]["n_layer"]):
            p = f"{side}_layers.{n}."
            sh[p + "attention.qkv.weight"] = (3 * D, D)
            sh[p + "attention.mha.in_proj_weight"] = (3 * D, D)
            sh[p + "attention.mha.out_proj.weight"] = (D, D)
            sh[p + "attention.wo.weight"] = (D, D)
            sh[p + "convlayer.conv1.weight"] = (F, D, k1)
            sh[p + "convlayer.conv1.bias"] = (F,)
            sh[p + "convlayer.conv2.weight"] = (D, F, k2)
            sh[p + "convlayer.conv2.bias"] = (D,)
            for nm in ("attn_norm", "conv_norm"):
                sh[p + nm + ".weight"] = (D,)
                sh[p + nm + ".bias"] = (D,)
    sh["head.weight"] = (cfg["preprocess"]["hubert_codes"], D)
    sh["head.bias"] = (cfg["preprocess"]["hubert_codes"],)
    return sh


class Parrot(nn.Module):
    def __init__(self, data_config, src_vocab_size, src_pad_idx):
        super().__init__()
        tr = data_config["transformer"]
        self.max_len, self.d_model = tr["max_len"], tr["d_model"]
        self.data_config = data_config
        self.src_vocab_size, self.src_pad_idx = int(src_vocab_size), src_pad_idx
        if self.d_model % tr["encoder"]["n_head"] or self.d_model % tr["decoder"]["n_head"]:
            raise AssertionError("d_model % n_head != 0")  # reference modules/fft.py:44
        spk_path = os.path.join(data_config["path"]["root_path"], "speakers.json")
        with open(spk_path, "r") as f:
            self.n_speaker = len(json.load(f))
        _attach(self, "pos_emb.pe", sinusoid_table(self.max_len, self.d_model), buffer=True)
        gen = torch.Generator().manual_seed(0)
        for key, shape in parrot_param_shapes(data_config, self.src_vocab_size, self.n_speaker).items():
            if key.endswith("norm.weight") or key in ("duration_predictor.layers.2.weight", "duration_predictor.layers.6.weight"):
                t = torch.ones(shape)
            elif key.endswith("bias"):
                t = torch.zeros(shape)
            else:
                fan_in = 1
                for s in shape[1:]:
                    fan_in *= s
                t = torch.randn(shape, generator=gen) / max(fan_in, 1) ** 0.5
            _attach(self, key, t)
        self._handle: Optional[C.c_void_p] = None
        self._handle_device = None
        self._handle_fp = None
        # range-safe fallback (see CodeGenerator): the first decode of every new handle is checked synchronously; non-finite logits
        # under the default fp16x3 scheme rebuild the handle in bf16x6 and re-run the batch, with a warning
        self.range_fallback = range_fallback_default()
        self._precision_override: Optional[int] = None
        self._merge_override: Optional[bool] = None  # None: library default (merged); tests build unmerged handles beside it
        self._probe_pending = False
        # torch's load_state_dict recurses through _load_from_state_dict and never calls a CHILD's load_state_dict
        # override, so a reload through any wrapper (LitParrot, nn.Sequential ...) is caught here: the post hook runs
        # for every module of the tree, and _current_handle() also compares the parameters' version fingerprint.
        self._fell_back = False  # the override above was set by the range-safe fallback (not by the caller)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._on_load())

    # ---- parameter bookkeeping ----------------------------------------------------------------
    def _on_load(self):
        """New weights: drop the packed handle -- and a bf16x6 override that the range-safe fallback set for the OLD weights (the new
        checkpoint gets the default scheme and its own first-forward probe)."""
        self._invalidate()
        if self._fell_back:
            self._precision_override, self._fell_back = None, False

    def _invalidate(self):
        if self._handle is not None:
            _lib.lib().parrot_tte_destroy(self._handle)
        self._handle = None

    def _current_handle(self, dev):
        """The packed-weight handle for ``dev``, rebuilt when any parameter was replaced, moved or written in place."""
        fp = param_fingerprint(self)
        if self._handle is None or self._handle_device != dev or self._handle_fp != fp:
            self._invalidate()
            self._build(dev)
            self._handle_fp = fp
        return self._handle

    def _apply(self, fn, recurse=True):
        self._invalidate()
        return super()._apply(fn, recurse)

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass

    # ---- HIP handle ---------------------------------------------------------------------------
    def _build(self, device):
        cfgd = self.data_config
        tr, dp = cfgd["transformer"], cfgd["duration_predictor"]
        sd = {k: v.detach().to("cpu", torch.float32).contiguous() for k, v in self.state_dict().items()}
        P = lambda k: _lib.fptr(sd[k])  # noqa: E731
        cfg = _lib.TteCfg(tr["d_model"], tr["conv_n_filter"], tr["conv_kernel_sizes"][0], tr["conv_kernel_sizes"][1], tr["max_len"],
                          tr["encoder"]["n_layer"], tr["encoder"]["n_head"], tr["decoder"]["n_layer"], tr["decoder"]["n_head"],
                          dp["n_filter"], dp["kernel_size"], self.src_vocab_size, self.n_speaker, cfgd["preprocess"]["hubert_codes"])

        def fft(prefix):
            return _lib.FftWeights(P(prefix + "attention.qkv.weight"), P(prefix + "attention.mha.in_proj_weight"),
                                   P(prefix + "attention.mha.out_proj.weight"), P(prefix + "attention.wo.weight"),
                                   P(prefix + "convlayer.conv1.weight"), P(prefix + "convlayer.conv1.bias"),
                                   P(prefix + "convlayer.conv2.weight"), P(prefix + "convlayer.conv2.bias"),
                                   P(prefix + "attn_norm.weight"), P(prefix + "attn_norm.bias"),
                                   P(prefix + "conv_norm.weight"), P(prefix + "conv_norm.bias"))

        enc = (_lib.FftWeights * max(1, cfg.enc_layers))(*[fft(f"encoder_layers.{n}.") for n in range(cfg.enc_layers)])
        dec = (_lib.FftWeights * max(1, cfg.dec_layers))(*[fft(f"decoder_layers.{n}.") for n in range(cfg.dec_layers)])
        w = _lib.TteWeights()
        w.pe, w.tok_emb = P("pos_emb.pe"), P("tok_emb.weight")
        w.speaker_emb = P("speaker_emb.weight") if self.n_speaker > 1 else None
        d = "duration_predictor."
        w.dp_conv0_w, w.dp_conv0_b = P(d + "layers.0.conv.weight"), P(d + "layers.0.conv.bias")
        w.dp_ln0_w, w.dp_ln0_b = P(d + "layers.2.weight"), P(d + "layers.2.bias")
        w.dp_conv1_w, w.dp_conv1_b = P(d + "layers.4.conv.weight"), P(d + "layers.4.conv.bias")
        w.dp_ln1_w, w.dp_ln1_b = P(d + "layers.6.weight"), P(d + "layers.6.bias")
        w.dp_proj_w, w.dp_proj_b = P(d + "proj.weight"), P(d + "proj.bias")
        w.enc, w.dec = C.cast(enc, C.POINTER(_lib.FftWeights)), C.cast(dec, C.POINTER(_lib.FftWeights))
        w.head_w, w.head_b = P("head.weight"), P("head.bias")
        hdl = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().parrot_tte_create_ex(C.byref(hdl), C.byref(cfg), C.byref(w),
                                                       -1 if self._precision_override is None else self._precision_override,
                                                       -1 if self._merge_override is None else int(self._merge_override)))
        self._handle, self._handle_device = hdl, device
        self._probe_pending = True

    @property
    def precision_in_use(self) -> Optional[str]:
        """Precision of the live handle ("f16x3", "bf16x6", ...; None before the first forward)."""
        if self._handle is None:
            return None
        return PREC_STR.get(int(_lib.lib().parrot_tte_precision(self._handle)))

    def _fall_back(self, why: str) -> bool:
        import warnings
        cur = None if self._handle is None else int(_lib.lib().parrot_tte_precision(self._handle))
        if not self.range_fallback or cur != PREC_F16X3:
            return False
        warnings.warn(f"parrot_tts_amd TTE: {why}: an activation left the fp16x3 scheme's range (|x| < 8190); rebuilding the handle in "
                      "bf16x6 (fp32's range) for this and all later batches.", RuntimeWarning, stacklevel=3)
        self._precision_override = PREC_BF16X6
        self._fell_back = True
        self._invalidate()
        return True

    # ---- forward ------------------------------------------------------------------------------
    @staticmethod
    def _raise_status(code: int, who: str):
        """Device status codes (csrc/kernels_misc.h) -> the exceptions the reference raises / this package promises."""
        if code == 0:
            return
        if code == 5:
            raise FloatingPointError(f"{who}: non-finite output (waveform sample / logits) -- an activation left the range of the fp16 split "
                                     "scheme (|x| < 8190); use PARROT_PRECISION=bf16x6 or f32 for this checkpoint")
        raise IndexError(f"{who}: embedding index out of range (code {code})")

    @torch.no_grad()
    def _encode(self, batch, status_hooks=(), row_exact: bool = False) -> dict:
        """Phase 1 (encoder, duration predictor) and the ONE device-to-host transfer of the path (the expanded lengths: the
        reference's own host sync, duration.py:10).  ``status_hooks``: callables ``hook(dst_ptr, stream_ptr) -> name`` that enqueue a
        copy of another handle's device status flag (the vocoder's, see SynthesisPipeline): they ride on that transfer.
        ``row_exact``: every row is evaluated as the reference evaluates that utterance ALONE (see ``infer``).
        Returns the context ``_decode`` continues from."""
        phones = batch["phones"]
        require_cuda(phones, "batch['phones']")
        dev = phones.device
        self._current_handle(dev)
        lib = _lib.lib()
        phones = phones.to(torch.int64).contiguous()
        B, S = phones.shape
        src_mask = batch["src_mask"].to(dev)
        valid = src_mask.to(torch.uint8).contiguous()
        src_len = None
        if row_exact:  # real tokens per row; the key mask is that prefix (collate pads on the right, modules/data.py:97-104)
            src_len = valid.sum(1, dtype=torch.int32).contiguous()
            prefix = torch.arange(S, device=dev)[None, :] < src_len[:, None]
            # a mask that is not a right-padded prefix (a pad-index token inside an utterance, left padding) has no "row alone"
            # reading: the count would drop real trailing tokens silently.  Checked on the device; the verdict rides on the length
            # transfer below (no extra sync) and raises there.
            not_prefix = (prefix != valid.bool()).any().to(torch.int32).reshape(1)
            # (a row of no tokens at all keeps key 0 attendable -- a softmax over no keys is 0 / 0 -- and expands to nothing: the
            #  duration kernel masks by src_len, not by this key mask)
            valid = (torch.arange(S, device=dev)[None, :] < src_len.clamp(min=1)[:, None]).to(torch.uint8).contiguous()
        speaker = None
        if self.n_speaker > 1:
            speaker = batch["speaker"].to(dev, torch.int64).contiguous()
        log_dur = torch.empty((B, S), dtype=torch.float32, device=dev)
        dur = torch.empty((B, S), dtype=torch.int64, device=dev)
        # expanded lengths + this handle's status flag + one slot per hook: ONE device-to-host transfer fetches them all
        status = torch.zeros((B + 2 + len(status_hooks),), dtype=torch.int32, device=dev)  # (last slot: the prefix check above)
        if row_exact:
            status[-1:] = not_prefix
        lens = status[:B]
        state = torch.empty(lib.parrot_tte_state_bytes(self._handle, B, S), dtype=torch.uint8, device=dev)
        st = stream_ptr(dev)
        with torch.cuda.device(dev):
            try:
                ws = torch.empty(lib.parrot_tte_workspace_bytes(self._handle, B, S, 0), dtype=torch.uint8, device=dev)
                _lib.check(lib.parrot_tte_encode(self._handle, dptr(phones), dptr(valid), dptr(speaker) if speaker is not None else None,
                                                 dptr(src_len) if src_len is not None else None, B, S, dptr(log_dur), dptr(dur), dptr(lens),
                                                 dptr(state), state.numel(), dptr(ws), ws.numel(), st))
                _lib.check(lib.parrot_tte_status_async(self._handle, status.data_ptr() + 4 * B, st))
                hooked = [hook(status.data_ptr() + 4 * (B + 1 + i), st) for i, hook in enumerate(status_hooks)]
                hooked = [h if isinstance(h, tuple) else (h, None) for h in hooked]  # (name, on_nonfinite callback or None)
                status_h = status.cpu()  # the one host sync the reference also has (duration.py:10)
            except _lib.ParrotHipError as e:
                self._reraise(e)
        lens_h = status_h[:B]
        if int(status_h[-1]):
            raise ValueError("row_exact=True needs src_mask to be a right-padded prefix per row (modules/data.py:97-104 pads on the right); "
                             "this mask has a pad inside an utterance or left padding -- run the padded-batch mode (row_exact=False) instead")
        # bad phone / speaker ids of THIS encode (the reference's Embedding IndexError), non-finite logits of the previous
        # decode, and whatever the hooks watch (the previous vocoder forward): raised here, by default, at no extra sync
        if int(status_h[B]) == 5:
            self._fall_back("a previous decode produced non-finite logits")  # (later batches run in bf16x6; this one is reported)
        self._raise_status(int(status_h[B]), "tte")
        for i, (nm, on_nonfinite) in enumerate(hooked):
            if int(status_h[B + 1 + i]) == 5 and on_nonfinite is not None:
                on_nonfinite()
            self._raise_status(int(status_h[B + 1 + i]), nm)
        return {"B": B, "S": S, "L": int(lens_h.max()), "dev": dev, "state": state, "log_dur": log_dur, "dur": dur, "lens": lens_h,
                "lens_dev": lens, "src_mask": src_mask, "handle": self._handle, "row_exact": bool(row_exact)}

    @staticmethod
    def _reraise(e):
        if e.code == -2:  # PARROT_E_RANGE <-> the reference's IndexError (pe[T], Embedding)
            raise IndexError(str(e)) from None
        if e.code == -6:  # PARROT_E_NONFINITE (flag raised by an earlier decode)
            raise FloatingPointError(str(e)) from None
        raise e

    @torch.no_grad()
    def _decode(self, ctx: dict, ids: torch.Tensor, tgt: torch.Tensor, logits: Optional[torch.Tensor]) -> None:
        """Phase 2 of the encoded batch, on the CURRENT stream: length regulator, decoder, head, argmax (+ tie guard) into
        ``ids`` / ``tgt`` (/ ``logits``).  The scratch buffer is allocated on the current stream."""
        lib = _lib.lib()
        B, S, L, dev = ctx["B"], ctx["S"], ctx["L"], ctx["dev"]
        ws = torch.empty(lib.parrot_tte_workspace_bytes(self._handle, B, S, L), dtype=torch.uint8, device=dev)
        state = ctx["state"]
        state.record_stream(torch.cuda.current_stream(dev))
        with torch.cuda.device(dev):
            try:
                _lib.check(lib.parrot_tte_decode(self._handle, B, S, L, 1 if ctx["row_exact"] else 0, dptr(ids), dptr(tgt),
                                                 dptr(logits) if logits is not None else None, dptr(state), state.numel(),
                                                 dptr(ws), ws.numel(), stream_ptr(dev)))
            except _lib.ParrotHipError as e:
                self._reraise(e)

    @torch.no_grad()
    def _run(self, batch, want_logits: bool, status_hooks=(), row_exact: bool = False):
        ctx = self._encode(batch, status_hooks, row_exact=row_exact)
        lib = _lib.lib()
        B, L, dev = ctx["B"], ctx["L"], ctx["dev"]
        ids = torch.empty((B, L), dtype=torch.int64, device=dev)
        tgt = torch.empty((B, L), dtype=torch.uint8, device=dev)
        logits = torch.empty((B, L, lib_n_codes(self)), dtype=torch.float32, device=dev) if want_logits else None
        # ids per row as `infer` returns them (len + 1 clamped to L: quirk Q2; exactly len in the row-exact mode), on the device for a
        # vocoder that follows: computed HERE, in the shadow of the length sync, so that no small kernel sits between the decoder's
        # last launch and the vocoder's first
        emitted_dev = ctx["lens_dev"] if row_exact else torch.clamp(ctx["lens_dev"] + 1, max=L)
        self._decode(ctx, ids, tgt, logits)
        if self._probe_pending:  # first decode of this handle: one synchronous look at the device flag
            self._probe_pending = False
            if self.range_fallback and int(lib.parrot_tte_precision(self._handle)) == PREC_F16X3:
                flag = torch.zeros(1, dtype=torch.int32, device=dev)
                with torch.cuda.device(dev):
                    st = stream_ptr(dev)
                    _lib.check(lib.parrot_tte_status_peek_async(self._handle, dptr(flag), st))
                    if int(flag.cpu()) == 5:
                        _lib.check(lib.parrot_tte_status_async(self._handle, dptr(flag), st))  # handled here: clear it
                        if self._fall_back("the first decode of this handle produced non-finite logits"):
                            return self._run(batch, want_logits, status_hooks=(), row_exact=row_exact)
        tgt_mask = tgt.bool()
        if row_exact:  # a row alone is the longest of its batch: exactly `lens` ids, no extra frame (the device mask keeps one key
            tgt_mask = torch.arange(L, device=dev)[None, :] < ctx["lens_dev"][:, None]  # valid for rows of length 0)
        return {"ids": ids, "tgt_mask": tgt_mask, "log_dur": ctx["log_dur"], "dur": ctx["dur"], "lens": ctx["lens"], "logits": logits,
                "src_mask": ctx["src_mask"], "lens_dev": ctx["lens_dev"], "row_exact": bool(row_exact), "emitted_dev": emitted_dev}

    @torch.no_grad()
    def forward_stages(self, batch) -> dict:
        """Tests / error localisation: one inference forward that also returns the activation after every stage in the
        reference's (B, T, D) layout, keyed like the oracle's ``return_stages`` ("emb", "enc0".., "enc_out", "dec_in",
        "dec0"..) -- plus everything ``forward`` returns."""
        dev = batch["phones"].device
        self._current_handle(dev)
        lib = _lib.lib()
        tr = self.data_config["transformer"]
        ne, nd = tr["encoder"]["n_layer"], tr["decoder"]["n_layer"]
        B, S = batch["phones"].shape
        D = self.d_model
        enc = [torch.empty((B, D, S), dtype=torch.float32, device=dev) for _ in range(ne + 2)]
        enc_ptrs = (C.c_void_p * (ne + 2))(*[C.c_void_p(t.data_ptr()) for t in enc])
        _lib.check(lib.parrot_tte_debug_stages(self._handle, enc_ptrs, None))
        try:
            L = int(self._run(batch, want_logits=False)["lens"].max())  # pass 1: encoder stages (+ the expanded length)
            dec = [torch.empty((B, D, L), dtype=torch.float32, device=dev) for _ in range(nd + 1)]
            dec_ptrs = (C.c_void_p * (nd + 1))(*[C.c_void_p(t.data_ptr()) for t in dec])
            _lib.check(lib.parrot_tte_debug_stages(self._handle, enc_ptrs, dec_ptrs))
            r = self._run(batch, want_logits=True)
            torch.cuda.synchronize(dev)
        finally:
            _lib.check(lib.parrot_tte_debug_stages(self._handle, None, None))
        names = ["emb"] + [f"enc{n}" for n in range(ne)] + ["enc_out"]
        st = {k: t.transpose(1, 2).contiguous() for k, t in zip(names, enc)}
        st.update({k: t.transpose(1, 2).contiguous() for k, t in zip(["dec_in"] + [f"dec{n}" for n in range(nd)], dec)})
        r["stages"] = st
        return r

    def guard_stats(self) -> dict:
        """Tie-guard statistics of the last decode (synchronises): positions whose top-2 logit margin was below the guard
        (PARROT_TIE_GUARD, default 1e-4) and had their head re-evaluated in fp64, the smallest margin of the call, and how many
        ids the re-evaluation changed."""
        import struct
        dev = self._handle_device
        buf = torch.zeros(3, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().parrot_tte_guard_stats_async(self._handle, buf.data_ptr(), stream_ptr(dev)))
        n, changed, bits = (int(v) for v in buf.cpu())
        return {"n_guarded": n, "min_margin": struct.unpack("f", struct.pack("i", bits))[0], "ids_changed": changed,
                "precision_in_use": self.precision_in_use}

    def guard_logits(self, max_n: int = 256):
        """Tests / parity reports: the refined logits of the guarded positions of the last decode (fp64 re-evaluation of the last
        decoder block's conv2 + bias + residual and of the head, rounded to fp32) -> (logits (n, V), positions (n, 2) as (b, t))."""
        dev = self._handle_device
        n = min(self.guard_stats()["n_guarded"], max_n, 256)
        V = lib_n_codes(self)
        lg = torch.empty((max(n, 1), V), dtype=torch.float32, device=dev)
        pos = torch.empty((max(n, 1), 2), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().parrot_tte_guard_logits(self._handle, dptr(lg), dptr(pos), max(n, 1), stream_ptr(dev)))
        torch.cuda.synchronize(dev)
        return lg[:n].cpu(), pos[:n].cpu()

    def check_outputs(self) -> None:
        """Synchronise and raise FloatingPointError if the last decode produced NaN / inf logits (an activation beyond the
        fp16 split scheme's range; use PARROT_PRECISION=bf16x6 for such a checkpoint), IndexError for a bad id."""
        if self._handle is not None:
            try:
                _lib.check(_lib.lib().parrot_tte_check(self._handle, stream_ptr(self._handle_device)))
            except _lib.ParrotHipError as e:
                if e.code == -2:
                    raise IndexError(str(e)) from None
                if e.code == -6:
                    raise FloatingPointError(str(e)) from None
                raise

    def forward(self, batch, inference=False):
        if inference is not True:
            raise NotImplementedError("parrot_tts_amd.Parrot implements the inference path only (training is out of scope)")
        r = self._run(batch, want_logits=True)
        return (r["logits"], batch["src_mask"], r["tgt_mask"], r["log_dur"])

    def infer(self, batch, row_exact: bool = False) -> List[List[int]]:
        """``Parrot.infer`` of the reference (modules/parrot.py:112-120).  Default: the reference's result for THIS padded batch
        (quirk Q7: it depends on the batch composition).  ``row_exact=True``: every row as the reference evaluates that utterance
        alone -- what its driver, which runs batch_size = 1 (inference.py:34), writes to predictions.txt -- at batched speed."""
        assert self.training == False  # noqa: E712  (reference modules/parrot.py:113)
        r = self._run(batch, want_logits=False, row_exact=row_exact)
        self.check_outputs()  # (infer() synchronises anyway to hand python lists back)
        ids, msk = r["ids"].cpu(), r["tgt_mask"].cpu()
        return [c[m].numpy().tolist() for c, m in zip(ids, msk)]

    def infer_dense(self, batch, status_hooks=(), row_exact: bool = False) -> dict:
        """Batched, device-resident result (ids (B,L), tgt_mask, lens) for pipelines that feed the vocoder
        directly instead of going through Python lists."""
        assert self.training == False  # noqa: E712
        return self._run(batch, want_logits=False, status_hooks=status_hooks, row_exact=row_exact)


def lib_n_codes(m: "Parrot") -> int:
    return int(m.data_config["preprocess"]["hubert_codes"])
