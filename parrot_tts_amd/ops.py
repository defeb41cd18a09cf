"""Thin tensor-level wrappers over the single-op C-ABI entry points (conv plan, int16 cast, selftest).

PyTorch is plumbing only: it owns device memory and the stream; all arithmetic happens in
libparrot_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

PRE_NONE, PRE_LRELU = 0, 1
ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2
EPI_STORE, EPI_ADD, EPI_ADD_DIV = 0, 1, 2
PREC_F32, PREC_BF16X6, PREC_F16X3, PREC_BF16, PREC_F16 = 0, 1, 2, 3, 4
PREC_NAMES = {"f32": PREC_F32, "bf16x6": PREC_BF16X6, "f16x3": PREC_F16X3, "bf16": PREC_BF16, "f16": PREC_F16}
PREC_DEFAULT = PREC_F16X3
PREC_STR = {v: k for k, v in PREC_NAMES.items()}


def set_default_precision(prec: int) -> None:
    """Library-wide default for handles created afterwards (include/parrot_hip.h PARROT_PREC_*): PREC_F32 exact fp32
    MFMA, PREC_F16X3 (default) / PREC_BF16X6 split schemes with fp32-class error, PREC_BF16 / PREC_F16 single-MFMA
    reduced precision."""
    _lib.check(_lib.lib().parrot_set_default_precision(int(prec)))


def set_fused_resblocks(mode: int) -> None:
    """ResBlocks through the fused LDS-resident kernels: 0 off, 1 every eligible stage, 2 all but the exact-fp32
    32-channel kernel (library default).  Applies to handles created afterwards."""
    _lib.check(_lib.lib().parrot_set_fused_resblocks(int(mode)))


def set_tte_merge(on: bool) -> None:
    """FFT blocks: evaluate the back-to-back bias-free projections (quirk Q3) as their fp64-formed product (True, library
    default) or one after the other as the reference does.  Applies to handles created afterwards."""
    _lib.check(_lib.lib().parrot_set_tte_merge(int(bool(on))))


def range_fallback_default() -> bool:
    """PARROT_RANGE_FALLBACK (default on): a handle whose first forward reports a non-finite result under the default fp16x3
    scheme (|activation| >= 8190) is rebuilt in bf16x6 (fp32's range) and the batch re-run, with a warning."""
    import os
    return os.environ.get("PARROT_RANGE_FALLBACK", "1") not in ("", "0")


def stream_ptr(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def require_cuda(t: torch.Tensor, name: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"parrot_tts_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU fallback")


def param_fingerprint(module) -> tuple:
    """Identity + version of every parameter / buffer of ``module``: changes whenever a tensor is replaced, moved or
    written in place (``load_state_dict`` through ANY parent module copies in place and bumps ``_version``).  The HIP
    handle caches packed copies of the weights, so the shims compare this before every forward and rebuild on mismatch."""
    # (a direct walk over the modules' own dicts: `module.parameters()` -- a generator over named_modules with name strings and a
    #  memo set -- cost 390-460 us per call on the full-size generator, on the host's critical path between the TTE's last launch and
    #  the vocoder's first at small batches; this is 130 us and sees the same tensors)
    out = []
    stack = [module]
    while stack:
        m = stack.pop()
        for t in m._parameters.values():
            if t is not None:
                out.append((id(t), t.data_ptr(), t._version))
        for t in m._buffers.values():
            if t is not None:
                out.append((id(t), t.data_ptr(), t._version))
        for c in m._modules.values():
            if c is not None:
                stack.append(c)
    return tuple(out)


def selftest() -> None:
    _lib.check(_lib.lib().parrot_selftest(stream_ptr()))


class ConvPlan:
    """One Conv1d / ConvTranspose1d layer packed for the MFMA implicit-GEMM kernel.

    ``weight`` / ``bias`` are CPU fp32 tensors in torch layout ((C_out,C_in,k), or (C_in,C_out,k) when
    ``transposed``).  ``__call__`` runs y = act(conv(pre(x)) + bias) (+ res) on (B, C, T) CUDA tensors."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, dilation: int = 1, padding: int = 0,
                 transposed: bool = False, stride: int = 1, pre_act: int = PRE_NONE, pre_slope: float = 0.0,
                 act: int = ACT_NONE, tile_cfg: int = -1, precision: int = -1):
        w = weight.detach().to("cpu", torch.float32).contiguous()
        b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
        c_in, c_out = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        self.c_in, self.c_out = int(c_in), int(c_out)
        d = _lib.ConvDesc(self.c_in, self.c_out, int(w.shape[2]), int(dilation), int(padding), int(bool(transposed)), int(stride),
                          int(pre_act), float(pre_slope), int(act), int(tile_cfg), int(precision))
        self._h = C.c_void_p()
        _lib.check(_lib.lib().parrot_conv_create(C.byref(self._h), C.byref(d), _lib.fptr(w), None if b is None else _lib.fptr(b)))

    def out_len(self, t_in: int) -> int:
        return int(_lib.lib().parrot_conv_out_len(self._h, int(t_in)))

    def __call__(self, x: torch.Tensor, res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                 epilogue: int = EPI_STORE, div: float = 1.0) -> torch.Tensor:
        require_cuda(x, "x")
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3 and x.shape[1] == self.c_in
        B, _, T = x.shape
        t_out = self.out_len(T)
        if out is None:
            assert epilogue == EPI_STORE
            out = torch.empty((B, self.c_out, t_out), device=x.device, dtype=torch.float32)
        assert out.is_contiguous() and tuple(out.shape) == (B, self.c_out, t_out)
        if res is not None:
            assert res.is_contiguous() and res.shape == out.shape
        _lib.check(_lib.lib().parrot_conv_run(self._h, dptr(x), dptr(res), dptr(out), B, T, int(epilogue), float(div), stream_ptr(x.device)))
        return out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:  # (module globals are already torn down at interpreter exit)
            try:
                _lib.lib().parrot_conv_destroy(h)
            except Exception:
                pass


def wav_to_int16(wav: torch.Tensor) -> torch.Tensor:
    """``(audio * 32768).astype('int16')`` of reference utils/vocoder/inference.py:71-73, on the GPU."""
    require_cuda(wav, "wav")
    w = wav.contiguous()
    out = torch.empty(w.shape, device=w.device, dtype=torch.int16)
    _lib.check(_lib.lib().parrot_wav_to_int16(dptr(w), dptr(out), w.numel(), stream_ptr(w.device)))
    return out


def length_regulator(seq: torch.Tensor, dur: torch.Tensor):
    """``length_regulator`` of reference modules/duration.py:6-24 on the HIP kernel the TTE decoder uses: seq (B,S,D) f32,
    dur (B,S) int64 -> (expanded (B,L,D) zero-padded to L = max row sum, tgt_mask (B,L) bool with the ``ids <= len`` rule
    of modules/data.py:18, out_lens list).  L is read back to the host like the reference does (duration.py:10)."""
    require_cuda(seq, "seq")
    seq = seq.to(torch.float32).contiguous()
    dur = dur.to(seq.device, torch.int64).contiguous()
    B, S, D = seq.shape
    L = int(torch.clamp(dur, min=0).sum(dim=1).max())  # the reference's own host sync
    if L <= 0:
        raise ValueError("length_regulator: every duration is zero (the reference fails downstream on an empty sequence)")
    lib = _lib.lib()
    ws = torch.empty(lib.parrot_length_regulator_workspace_bytes(B, S, D, L), dtype=torch.uint8, device=seq.device)
    out = torch.empty((B, L, D), dtype=torch.float32, device=seq.device)
    mask = torch.empty((B, L), dtype=torch.uint8, device=seq.device)
    lens = torch.empty((B,), dtype=torch.int32, device=seq.device)
    with torch.cuda.device(seq.device):
        _lib.check(lib.parrot_length_regulator(dptr(seq), dptr(dur), B, S, D, L, dptr(out), dptr(mask), dptr(lens), dptr(ws), ws.numel(),
                                               stream_ptr(seq.device)))
    return out, mask.bool(), lens.cpu().tolist()
