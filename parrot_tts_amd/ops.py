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
PREC_F32, PREC_BF16X6 = 0, 1


def set_default_precision(prec: int) -> None:
    """Library-wide default for handles created afterwards (PREC_F32 exact fp32 MFMA, PREC_BF16X6 split-bf16)."""
    _lib.check(_lib.lib().parrot_set_default_precision(int(prec)))


def set_fused_resblocks(mode: int) -> None:
    """ResBlocks through the fused LDS-resident kernels (csrc/resblock_fused.h): 0 off, 1 for 16- and 32-channel
    stages, 2 for 16-channel stages only (library default)."""
    _lib.check(_lib.lib().parrot_set_fused_resblocks(int(mode)))


def stream_ptr(device=None) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def require_cuda(t: torch.Tensor, name: str) -> None:
    if t.device.type != "cuda":
        raise RuntimeError(f"parrot_tts_amd: `{name}` must live on the GPU (got {t.device}); there is no CPU fallback")


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
