"""Drop-in ``CodeGenerator`` for reference utils/vocoder/models.py:122-169 backed by libparrot_hip.so.

Same constructor (``CodeGenerator(h)`` with the AttrDict / dict of utils/vocoder/config.json), same
``state_dict`` keys (weight-norm ``weight_g``/``weight_v`` as training checkpoints carry them, or
plain ``weight`` after ``remove_weight_norm()``), same call ``generator(code=LongTensor(B,U),
spkr=LongTensor(B,1)) -> FloatTensor(B,1,U*prod(upsample_rates))``.  The module only *holds*
parameters; all arithmetic runs in the HIP library (no torch compute, no CPU fallback)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from .ops import PREC_BF16X6, PREC_F16X3, PREC_STR, dptr, param_fingerprint, range_fallback_default, require_cuda, stream_ptr


class AttrDict(dict):
    """Dict with attribute access -- what reference utils/vocoder/utils.py:77-80 provides."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


MAX_WAV_VALUE = 32768.0  # reference utils/vocoder/dataset.py:22
_CHECK_FINITE = os.environ.get("PARROT_CHECK_FINITE", "0") not in ("", "0")  # optional guard (costs a sync per forward)


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """reference utils/vocoder/utils.py:44-45"""
    return int((kernel_size * dilation - dilation) / 2)


class _WNConv(nn.Module):
    """Parameter holder for one weight-normed Conv1d / ConvTranspose1d (keys: bias, weight_g, weight_v
    -- or bias, weight once weight norm has been removed).  g has shape (dim0,1,1) for both kinds."""

    def __init__(self, shape, n_bias, std=0.01):
        super().__init__()
        v = torch.randn(shape) * std
        self.weight_g = nn.Parameter(v.flatten(1).norm(dim=1).reshape(-1, 1, 1))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(n_bias))

    def folded(self) -> torch.Tensor:
        """w = v * (g / ||v||), norm over all dims but 0, on the CPU so the fold is bit-identical to what
        torch.nn.utils.weight_norm computes in the reference's CPU path."""
        if "weight" in self._parameters:
            return self.weight.detach().to("cpu", torch.float32).contiguous()
        return torch._weight_norm(self.weight_v.detach().to("cpu", torch.float32),
                                  self.weight_g.detach().to("cpu", torch.float32), 0).contiguous()

    def remove_weight_norm(self):
        if "weight" in self._parameters:
            raise ValueError("weight_norm of this layer was already removed")  # torch raises ValueError too
        w = self.folded().to(self.weight_v.device)
        del self._parameters["weight_g"], self._parameters["weight_v"]
        self.weight = nn.Parameter(w)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # accept either representation regardless of which one this module currently holds
        has_plain = prefix + "weight" in state_dict
        has_wn = prefix + "weight_g" in state_dict and prefix + "weight_v" in state_dict
        holds_plain = "weight" in self._parameters
        if has_plain and not holds_plain:
            dev = self.weight_v.device
            del self._parameters["weight_g"], self._parameters["weight_v"]
            self.weight = nn.Parameter(torch.empty_like(state_dict[prefix + "weight"], device=dev))
        elif has_wn and holds_plain:
            dev = self.weight.device
            del self._parameters["weight"]
            self.weight_g = nn.Parameter(torch.empty_like(state_dict[prefix + "weight_g"], device=dev))
            self.weight_v = nn.Parameter(torch.empty_like(state_dict[prefix + "weight_v"], device=dev))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


class _ResBlock(nn.Module):
    def __init__(self, kind: str, ch: int, k: int, n_dil: int):
        super().__init__()
        self.kind = kind
        if kind == "1":
            self.convs1 = nn.ModuleList([_WNConv((ch, ch, k), ch) for _ in range(n_dil)])
            self.convs2 = nn.ModuleList([_WNConv((ch, ch, k), ch) for _ in range(n_dil)])
        else:
            self.convs = nn.ModuleList([_WNConv((ch, ch, k), ch) for _ in range(n_dil)])

    def ordered(self):
        if self.kind == "1":
            out = []
            for a, b in zip(self.convs1, self.convs2):
                out += [a, b]
            return out
        return list(self.convs)


class CodeGenerator(nn.Module):
    def __init__(self, h):
        super().__init__()
        self.h = h
        get = h.get if hasattr(h, "get") else (lambda k, d=None: getattr(h, k, d))
        self._kind = "1" if str(get("resblock")) == "1" else "2"  # models.py:77: anything but '1' is ResBlock2
        self._rates = list(get("upsample_rates"))
        self._up_k = list(get("upsample_kernel_sizes"))
        self._rb_k = list(get("resblock_kernel_sizes"))
        self._rb_d = [list(d) for d in get("resblock_dilation_sizes")]
        self._c0 = int(get("upsample_initial_channel"))
        self._in_dim = int(get("model_in_dim", 128))
        self._emb_dim = int(get("embedding_dim"))
        self._n_emb = int(get("num_embeddings"))
        self.f0 = get("f0", None)
        self.multispkr = get("multispkr", None)
        # (the reference stores h.f0 and never reads it again: forward() skips the `f0` keyword, models.py:163-164)
        # ResBlock1 reads dilation[0], [1], [2] and ResBlock2 dilation[0], [1] -- literally (models.py:17-22,51-54): longer lists
        # are cut there, shorter ones raise IndexError, so lists of different lengths per kernel size are fine
        self._n_dil = 3 if self._kind == "1" else 2
        if len(self._rb_d) < len(self._rb_k):
            raise IndexError("resblock_dilation_sizes has fewer entries than resblock_kernel_sizes")  # (zip() would silently drop kernels)
        for d in self._rb_d[:len(self._rb_k)]:
            if len(d) < self._n_dil:
                raise IndexError("tuple index out of range")  # what `dilation[%d]` raises in the reference
        self._rb_d = [d[:self._n_dil] for d in self._rb_d[:len(self._rb_k)]]
        self.num_kernels = len(self._rb_k)
        self.num_upsamples = len(self._rates)

        self.conv_pre = _WNConv((self._c0, self._in_dim, 7), self._c0)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        for i, (u, k) in enumerate(zip(self._rates, self._up_k)):
            cin, cout = self._c0 // (2 ** i), self._c0 // (2 ** (i + 1))
            self.ups.append(_WNConv((cin, cout, k), cout))
        for i in range(len(self._rates)):
            ch = self._c0 // (2 ** (i + 1))
            for k in self._rb_k:
                self.resblocks.append(_ResBlock(self._kind, ch, k, self._n_dil))
        self.conv_post = _WNConv((1, self._c0 // (2 ** len(self._rates)), 7), 1)
        self.dict = nn.Embedding(self._n_emb, self._emb_dim)
        if self.multispkr:
            self.spkr = nn.Embedding(10, self._emb_dim)

        self._handle: Optional[C.c_void_p] = None
        self._handle_device = None
        self._handle_fp = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        # range-safe fallback: the default fp16x3 scheme needs |activation| < 8190, which no checkpoint format promises.  The FIRST
        # forward of every new handle is checked synchronously (one sync per handle lifetime); a non-finite result rebuilds the
        # handle in bf16x6 (fp32's range), re-runs the batch and warns.  Later forwards are covered by the asynchronous flag.
        self.range_fallback = range_fallback_default()
        self._precision_override: Optional[int] = None  # PARROT_PREC_* this module's handles are created with (None: library default)
        self._probe_pending = False
        # a reload through a parent module never reaches a child's load_state_dict override: invalidate from the post
        # hook torch runs for every module of the tree, and compare the parameter fingerprint before each forward
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
            _lib.lib().parrot_voc_destroy(self._handle)
        self._handle, self._ws = None, {}

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

    def remove_weight_norm(self):
        """reference models.py:113-119 -- afterwards state_dict() has plain ``weight`` keys."""
        self._invalidate()
        for m in self._wn_layers():
            m.remove_weight_norm()

    def _wn_layers(self):
        out = [self.conv_pre] + list(self.ups)
        for rb in self.resblocks:
            out += rb.ordered()
        return out + [self.conv_post]

    def __del__(self):
        try:
            self._invalidate()
        except Exception:
            pass

    # ---- HIP handle ---------------------------------------------------------------------------
    def _build(self, device, fused=None, debug_handle=False):
        if self._in_dim < self._emb_dim * (2 if self.multispkr else 1):
            raise ValueError("model_in_dim is smaller than embedding_dim * (1 + multispkr)")
        cfg = _lib.VocCfg()
        cfg.num_embeddings, cfg.embedding_dim = self._n_emb, self._emb_dim
        cfg.multispkr, cfg.n_spkr = int(bool(self.multispkr)), 10
        cfg.model_in_dim, cfg.upsample_initial_channel = self._in_dim, self._c0
        cfg.n_stages = len(self._rates)
        for i, (u, k) in enumerate(zip(self._rates, self._up_k)):
            cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i] = u, k
        cfg.n_kernels, cfg.n_dil = len(self._rb_k), self._n_dil
        for j, k in enumerate(self._rb_k):
            cfg.resblock_kernel_sizes[j] = k
            for m, d in enumerate(self._rb_d[j]):
                cfg.resblock_dilation_sizes[j][m] = d
        cfg.resblock_type = 1 if self._kind == "1" else 2

        keep = []  # host tensors must outlive parrot_voc_create

        def host(t):
            t = t.detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            return _lib.fptr(t)

        w = _lib.VocWeights()
        w.dict = host(self.dict.weight)
        w.spkr = host(self.spkr.weight) if self.multispkr else None
        w.conv_pre_w, w.conv_pre_b = host(self.conv_pre.folded()), host(self.conv_pre.bias)
        for i, up in enumerate(self.ups):
            w.ups_w[i], w.ups_b[i] = host(up.folded()), host(up.bias)
        rb_layers = [m for rb in self.resblocks for m in rb.ordered()]
        arr_t = _lib.c_float_p * len(rb_layers)
        rb_w, rb_b = arr_t(*[host(m.folded()) for m in rb_layers]), arr_t(*[host(m.bias) for m in rb_layers])
        w.rb_w, w.rb_b = C.cast(rb_w, C.POINTER(_lib.c_float_p)), C.cast(rb_b, C.POINTER(_lib.c_float_p))
        w.n_rb = len(rb_layers)
        w.conv_post_w, w.conv_post_b = host(self.conv_post.folded()), host(self.conv_post.bias)
        hdl = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().parrot_voc_create_ex(C.byref(hdl), C.byref(cfg), C.byref(w),
                                                       -1 if self._precision_override is None else self._precision_override,
                                                       -1 if fused is None else int(fused)))
        if debug_handle:
            return hdl
        self._handle, self._handle_device = hdl, device
        self._probe_pending = True

    @property
    def precision_in_use(self) -> Optional[str]:
        """Precision of the live handle ("f16x3", "bf16x6", ...; None before the first forward): "bf16x6" after the range-safe
        fallback replaced a default-precision handle."""
        if self._handle is None:
            return None
        return PREC_STR.get(int(_lib.lib().parrot_voc_precision(self._handle)))

    def _first_forward_overflowed(self, dev) -> bool:
        """After the FIRST forward of a handle: one synchronous look at the device flag; True = the handle was replaced by a
        bf16x6 one and the caller re-runs its batch.  (A bad-id flag stays set for check_inputs() / the status hook.)"""
        if not self._probe_pending:
            return False
        self._probe_pending = False
        if not self.range_fallback:
            return False
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        lib = _lib.lib()
        with torch.cuda.device(dev):
            _lib.check(lib.parrot_voc_status_peek_async(self._handle, dptr(flag), stream_ptr(dev)))
        if int(flag.cpu()) != 5 or int(lib.parrot_voc_precision(self._handle)) != PREC_F16X3:
            return False  # (a bad-id flag, or a non-finite result with nothing to fall back to, stays set for check_inputs() / the hook)
        with torch.cuda.device(dev):
            _lib.check(lib.parrot_voc_status_async(self._handle, dptr(flag), stream_ptr(dev)))  # handled here: clear it
        return self._fall_back("the first forward of this handle produced a non-finite waveform")

    def _fall_back(self, why: str) -> bool:
        """Switch this module to bf16x6 handles after a non-finite result under an fp16 scheme.  Returns False when there is
        nothing to fall back to (already fp32-range, or the fallback is switched off)."""
        import warnings
        cur = None if self._handle is None else int(_lib.lib().parrot_voc_precision(self._handle))
        if not self.range_fallback or cur not in (PREC_F16X3,):
            return False
        warnings.warn(f"parrot_tts_amd vocoder: {why}: an activation left the fp16x3 scheme's range (|x| < 8190); rebuilding the handle "
                      "in bf16x6 (fp32's range, ~1.5x slower) for this and all later batches. CodeGenerator.activation_headroom() "
                      "shows the per-stage maxima.", RuntimeWarning, stacklevel=3)
        self._precision_override = PREC_BF16X6
        self._fell_back = True
        self._invalidate()
        return True

    @property
    def upsample_factor(self) -> int:
        f = 1
        for u in self._rates:
            f *= u
        return f

    def out_samples(self, units):
        """Waveform samples of an utterance of ``units`` units (int or integer tensor): the ConvTranspose1d length chain
        ``(T - 1) u - 2 ((k - u) // 2) + k`` of models.py:80-83 -- ``units * upsample_factor`` for every shipped config, one
        sample more per stage with odd ``k - u``; 0 for an empty row."""
        T = units
        for u, k in zip(self._rates, self._up_k):
            T = (T - 1) * u - 2 * ((k - u) // 2) + k
        if isinstance(units, torch.Tensor):
            return torch.where(units > 0, T, torch.zeros_like(T))
        return T if units > 0 else 0

    @property
    def _constant_hop(self) -> bool:
        return all((k - u) % 2 == 0 for u, k in zip(self._rates, self._up_k))

    # ---- forward ------------------------------------------------------------------------------
    def _checked_inputs(self, kwargs, unit_lens):
        """Validate what the C ABI takes as raw device pointers (shared by ``forward`` and the native chunked path): ``code``
        (B, U) int64 on the GPU, ``spkr`` int64 with B entries (multi-speaker models), ``unit_lens`` with B entries.
        Returns contiguous (code, spkr or None, unit_lens as int32 or None)."""
        code = kwargs["code"]
        require_cuda(code, "code")
        if code.dim() != 2 or code.dtype != torch.int64:
            raise ValueError("code must be a LongTensor of shape (B, U)")
        dev = code.device
        spkr = None
        if self.multispkr:
            spkr = kwargs["spkr"].to(dev).reshape(-1).contiguous()
            if spkr.dtype != torch.int64 or spkr.numel() != code.shape[0]:
                raise ValueError("spkr must be a LongTensor of shape (B, 1)")
        lens32 = None
        if unit_lens is not None:
            lens32 = unit_lens.to(dev, torch.int32).contiguous()
            if lens32.numel() != code.shape[0]:
                raise ValueError("unit_lens must have one entry per batch row")
        return code.contiguous(), spkr, lens32

    @torch.no_grad()
    def forward(self, stages: Optional[dict] = None, unit_lens: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                **kwargs) -> torch.Tensor:
        """``generator(code=..., spkr=...)`` as in the reference.  ``unit_lens`` (B,) int: real units per row of a padded
        (ragged) batch -- every layer then zero-pads at each row's own end, so ``wav[b, :, :unit_lens[b]*hop]`` equals the
        reference run of that utterance alone (the reference vocoder driver is B=1 only).  ``out``: a contiguous float32
        (B, 1, samples) tensor to write the waveform into (row groups of one batch fill slices of one tensor)."""
        code, spkr, lens32 = self._checked_inputs(kwargs, unit_lens)
        dev = code.device
        self._current_handle(dev)
        B, U = code.shape
        # extra conditioning keywords (models.py:162-167): everything but code / spkr / f0 (f0 is skipped by the reference
        # too) is upsampled to U frames and concatenated, in keyword order, behind the embeddings
        feats, n_feat = None, 0
        extra = [self._upsample(v.to(dev, torch.float32), U) for k, v in kwargs.items() if k not in ("code", "spkr", "f0")]
        if extra:
            feats = torch.cat(extra, dim=1).contiguous()
            if feats.shape[0] != B:
                raise ValueError("conditioning features must have one row per batch element")
            n_feat = feats.shape[1]
        lib = _lib.lib()
        key = (B, U)
        if key not in self._ws:
            self._ws = {key: torch.empty(lib.parrot_voc_workspace_bytes(self._handle, B, U), dtype=torch.uint8, device=dev)}
        ws = self._ws[key]
        if out is not None:
            if out.shape != (B, 1, self.out_samples(U)) or out.dtype != torch.float32 or out.device != dev or not out.is_contiguous():
                raise ValueError("out must be a contiguous float32 (B, 1, out_samples(U)) tensor on the inputs' device")
            wav = out
        else:
            wav = torch.empty((B, 1, self.out_samples(U)), dtype=torch.float32, device=dev)
        stage_ptrs = None
        if stages is not None:  # tests: capture conv_pre / ups_i / mrf_i activations
            names = ["conv_pre"]
            T, bufs = U, []
            bufs.append(torch.empty((B, self._c0, U), dtype=torch.float32, device=dev))
            for i, (u, k) in enumerate(zip(self._rates, self._up_k)):
                T = (T - 1) * u - 2 * ((k - u) // 2) + k
                ch = self._c0 // (2 ** (i + 1))
                names += [f"ups{i}", f"mrf{i}"]
                bufs += [torch.empty((B, ch, T), dtype=torch.float32, device=dev) for _ in range(2)]
            stage_ptrs = (C.c_void_p * len(bufs))(*[C.c_void_p(b.data_ptr()) for b in bufs])
            stages.update(dict(zip(names, bufs)))
        with torch.cuda.device(dev):
            _lib.check(lib.parrot_voc_forward_feats(self._handle, dptr(code), dptr(spkr), dptr(feats), n_feat, dptr(lens32), B, U,
                                                    dptr(wav), stage_ptrs, dptr(ws), ws.numel(), stream_ptr(dev)))
        if self._first_forward_overflowed(dev):
            return self.forward(stages=stages, unit_lens=unit_lens, out=out, **kwargs)
        if _CHECK_FINITE and not bool(torch.isfinite(wav).all()):
            # the default fp16x3 scheme needs |activation| < 8190 (include/parrot_hip.h): beyond that the output is inf/NaN
            raise FloatingPointError("non-finite waveform: an activation left the fp16 split scheme's range; "
                                     "use PARROT_PRECISION=bf16x6 (fp32's range) for this checkpoint")
        return wav

    def receptive_units(self, device=None) -> int:
        """Receptive field of the whole generator in units, either side of an output frame, computed by the library from
        the configuration (interval propagation through conv_post, the MRF stages, the transposed convs and conv_pre): 21 for
        the shipped config (SURVEY's +-6378 samples leaves out the reach of the transposed convs).  An interior chunk
        computed with this much real context equals the whole-utterance forward."""
        dev = device or self._handle_device or next(self.parameters()).device
        self._current_handle(dev)
        return int(_lib.lib().parrot_voc_receptive_units(self._handle))

    def wait_stage(self, stage: int, stream: "torch.cuda.Stream") -> None:
        """Make ``stream`` wait until the most recently enqueued forward has reached MRF stage ``stage`` (see
        ``parrot_voc_wait_stage``).  No-op before the first forward."""
        if self._handle is not None and 0 <= stage < len(self._rates):
            _lib.check(_lib.lib().parrot_voc_wait_stage(self._handle, int(stage), C.c_void_p(stream.cuda_stream)))

    def _status_hook(self, dst_ptr: int, stream: int):
        """Enqueue a copy of this handle's device status flag (bad ids / non-finite waveform of EARLIER forwards) to ``dst_ptr``
        without synchronising: `Parrot._run` fetches it together with the expanded lengths.  Returns (name, on_nonfinite): the
        callback switches this module to bf16x6 handles before the FloatingPointError is raised, so a retry succeeds."""
        if self._handle is not None:
            _lib.check(_lib.lib().parrot_voc_status_async(self._handle, dst_ptr, stream))
        return "vocoder (previous forward)", lambda: self._fall_back("a previous forward produced a non-finite waveform")

    @torch.no_grad()
    def activation_headroom(self, **kwargs) -> dict:
        """Debug aid for the fp16x3 range (|conv input| < 8190): runs ONE forward of these inputs on a temporary layer-by-layer
        handle in this module's precision (bf16x6 when the default would overflow) that records max |input element| of every conv,
        grouped as conv_pre / stage i / conv_post.  Returns {"max_abs": {...}, "headroom": {group: 8190 / max}, "limit": 8190.0}."""
        code, spkr, lens32 = self._checked_inputs(kwargs, None)
        dev = code.device
        keep, self._precision_override = self._precision_override, (self._precision_override if self._precision_override is not None else PREC_BF16X6)
        try:
            hdl = self._build(dev, fused=0, debug_handle=True)
        finally:
            self._precision_override = keep
        lib = _lib.lib()
        try:
            B, U = code.shape
            n_groups = len(self._rates) + 2
            amax = torch.zeros(n_groups, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.parrot_voc_workspace_bytes(hdl, B, U), dtype=torch.uint8, device=dev)
            wav = torch.empty((B, 1, self.out_samples(U)), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.parrot_voc_debug_absmax(hdl, dptr(amax)))
                _lib.check(lib.parrot_voc_forward(hdl, dptr(code), dptr(spkr), None, B, U, dptr(wav), None, dptr(ws), ws.numel(), stream_ptr(dev)))
                _lib.check(lib.parrot_voc_debug_absmax(hdl, None))
            vals = amax.cpu().tolist()
        finally:
            lib.parrot_voc_destroy(hdl)
        names = ["conv_pre"] + [f"stage{i}" for i in range(len(self._rates))] + ["conv_post"]
        return {"max_abs": dict(zip(names, vals)), "headroom": {n: (8190.0 / v if v > 0 else float("inf")) for n, v in zip(names, vals)},
                "limit": 8190.0}

    @torch.no_grad()
    def stream(self, chunk_units: int = 256, halo_units: Optional[int] = None, unit_lens: Optional[torch.Tensor] = None, **kwargs):
        """Chunk-streamed synthesis (BASELINE config 5, long-form utterances): yields ``(first_sample, wav_chunk)`` for
        consecutive chunks of ``chunk_units`` units.  Each chunk is vocoded with ``halo_units`` (default: the receptive
        field) of real context on both sides and only its own samples are kept, so the concatenation equals the
        whole-utterance ``forward`` to fp32 round-off; at the true sequence edges the chunk contains the edge itself, i.e.
        the per-layer zero padding is applied where the reference applies it.  First audio after one chunk instead of the
        whole utterance; peak workspace is that of ``chunk_units + 2 * halo_units`` units."""
        code = kwargs["code"]
        if chunk_units <= 0:
            raise ValueError("chunk_units must be positive")
        if not self._constant_hop:
            raise NotImplementedError("chunk streaming needs a constant samples-per-unit hop: a stage of this config has odd upsample_kernel_size - upsample_rate")
        halo = self.receptive_units(code.device) if halo_units is None else int(halo_units)
        U, hop = code.shape[1], self.upsample_factor
        for start in range(0, U, chunk_units):
            stop = min(U, start + chunk_units)
            lo, hi = max(0, start - halo), min(U, stop + halo)
            kw = dict(kwargs)
            kw["code"] = code[:, lo:hi]
            lens = None
            if unit_lens is not None:
                lens = torch.clamp(unit_lens.to(code.device, torch.int64) - lo, min=0, max=hi - lo)
            wav = self.forward(unit_lens=lens, **kw)
            yield start * hop, wav[:, :, (start - lo) * hop:(stop - lo) * hop]

    @torch.no_grad()
    def forward_chunked(self, chunk_units: int = 256, halo_units: Optional[int] = None, unit_lens: Optional[torch.Tensor] = None,
                        **kwargs) -> torch.Tensor:
        """``forward`` in chunks: same result, bounded activation memory.  Without extra conditioning keywords the chunk loop
        runs inside the library (``parrot_voc_forward_chunked``); otherwise it is assembled from ``stream`` chunks."""
        code = kwargs["code"]
        if not self._constant_hop:
            raise NotImplementedError("chunk streaming needs a constant samples-per-unit hop: a stage of this config has odd upsample_kernel_size - upsample_rate")
        if not [k for k in kwargs if k not in ("code", "spkr", "f0")] and self._in_dim == self._emb_dim * (2 if self.multispkr else 1):
            code, spkr, lens32 = self._checked_inputs(kwargs, unit_lens)  # (raw pointers cross the ABI below)
            dev = code.device
            self._current_handle(dev)
            B, U = code.shape
            halo = -1 if halo_units is None else int(halo_units)
            lib = _lib.lib()
            key = ("chunked", B, int(chunk_units), halo)
            if key not in self._ws:
                self._ws = {key: torch.empty(lib.parrot_voc_chunked_workspace_bytes(self._handle, B, int(chunk_units), halo), dtype=torch.uint8, device=dev)}
            ws = self._ws[key]
            wav = torch.empty((B, 1, U * self.upsample_factor), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.parrot_voc_forward_chunked(self._handle, dptr(code), dptr(spkr), dptr(lens32), B, U, int(chunk_units), halo,
                                                          dptr(wav), dptr(ws), ws.numel(), stream_ptr(dev)))
            if self._first_forward_overflowed(dev):
                return self.forward_chunked(chunk_units=chunk_units, halo_units=halo_units, unit_lens=unit_lens, **kwargs)
            return wav
        out = torch.empty((code.shape[0], 1, code.shape[1] * self.upsample_factor), dtype=torch.float32, device=code.device)
        for first, wav in self.stream(chunk_units, halo_units, unit_lens=unit_lens, **kwargs):
            out[:, :, first:first + wav.shape[-1]] = wav
        return out

    @staticmethod
    def _upsample(signal: torch.Tensor, max_frames: int) -> torch.Tensor:
        """`CodeGenerator._upsample` (models.py:132-151): (B,C,T') / (B,C) / flat signal -> (B,C,max_frames) by repeating each
        frame max_frames // T' times; a remainder raises NotImplementedError like the reference."""
        if signal.dim() == 2:
            signal = signal.unsqueeze(2)
        elif signal.dim() != 3:
            signal = signal.reshape(-1, 1, 1)
        cond = signal.shape[2]
        rep = max_frames // cond if cond else 0
        if rep == 0 or (max_frames - cond * rep) // rep > 0:
            raise NotImplementedError("Padding condition signal - misalignment between condition features.")
        if cond * rep != max_frames:  # the reference gets past its remainder check here and dies in .view()
            raise RuntimeError(f"shape '[{signal.shape[0]}, {signal.shape[1]}, {max_frames}]' is invalid for input of size "
                               f"{signal.shape[0] * signal.shape[1] * cond * rep}")
        return signal.repeat_interleave(rep, dim=2)

    def check_inputs(self) -> None:
        """Synchronise and raise IndexError if a unit / speaker id of an earlier forward was out of range (what nn.Embedding
        raises eagerly in the reference), FloatingPointError if a waveform sample came out non-finite (an activation beyond
        the fp16 split scheme's range: use PARROT_PRECISION=bf16x6 for such a checkpoint)."""
        if self._handle is not None:
            try:
                _lib.check(_lib.lib().parrot_voc_check(self._handle, stream_ptr(self._handle_device)))
            except _lib.ParrotHipError as e:
                if e.code == -2:
                    raise IndexError(str(e)) from None
                if e.code == -6:
                    self._fall_back("an earlier forward produced a non-finite waveform")  # (later forwards run in bf16x6)
                    raise FloatingPointError(str(e)) from None
                raise


def generate(h, generator: CodeGenerator, code: dict):
    """Counterpart of reference utils/vocoder/inference.py:65-74: returns (int16 numpy audio, rtf).
    Unlike the reference the wall time is taken after a stream sync, so the RTF is meaningful on a GPU."""
    import time
    dev = code["code"].device
    torch.cuda.synchronize(dev)
    start = time.time()
    y = generator(**code)
    audio16 = _wav16(y.squeeze())
    torch.cuda.synchronize(dev)
    sr = h["sampling_rate"] if isinstance(h, dict) else h.sampling_rate
    rtf = (time.time() - start) / (y.shape[-1] / sr)
    generator.check_inputs()  # the stream is idle already: a bad id / a non-finite sample fails here, loudly, by default
    return audio16.cpu().numpy(), rtf


def _wav16(x):
    from .ops import wav_to_int16
    return wav_to_int16(x)
