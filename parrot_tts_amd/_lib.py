"""ctypes binding of libparrot_hip.so (include/parrot_hip.h; test / profiling entry points: include/parrot_hip_debug.h).  Fails loudly when the HIP library
is missing or does not load: there is NO CPU / eager fallback in this package."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PARROT_HIP_LIB: kernel-experiment builds of the same library (tools/); the default is the in-tree build
LIB_PATH = os.environ.get("PARROT_HIP_LIB") or os.path.join(HERE, "libparrot_hip.so")

MAX_STAGES, MAX_KERNELS, MAX_DIL = 8, 4, 4
ABI_VERSION = 7  # PARROT_ABI_VERSION of include/parrot_hip.h
c_float_p = C.POINTER(C.c_float)


class ParrotHipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libparrot_hip error {code}: {msg}")
        self.code = code


class ConvDesc(C.Structure):
    _fields_ = [("c_in", C.c_int32), ("c_out", C.c_int32), ("k", C.c_int32), ("dilation", C.c_int32),
                ("padding", C.c_int32), ("transposed", C.c_int32), ("stride", C.c_int32), ("pre_act", C.c_int32),
                ("pre_slope", C.c_float), ("act", C.c_int32), ("tile_cfg", C.c_int32), ("precision", C.c_int32)]


class VocCfg(C.Structure):
    _fields_ = [("num_embeddings", C.c_int32), ("embedding_dim", C.c_int32), ("multispkr", C.c_int32),
                ("n_spkr", C.c_int32), ("model_in_dim", C.c_int32), ("upsample_initial_channel", C.c_int32),
                ("n_stages", C.c_int32), ("upsample_rates", C.c_int32 * MAX_STAGES),
                ("upsample_kernel_sizes", C.c_int32 * MAX_STAGES), ("n_kernels", C.c_int32),
                ("resblock_kernel_sizes", C.c_int32 * MAX_KERNELS), ("n_dil", C.c_int32),
                ("resblock_dilation_sizes", (C.c_int32 * MAX_DIL) * MAX_KERNELS), ("resblock_type", C.c_int32)]


class VocWeights(C.Structure):
    _fields_ = [("dict", c_float_p), ("spkr", c_float_p), ("conv_pre_w", c_float_p), ("conv_pre_b", c_float_p),
                ("ups_w", c_float_p * MAX_STAGES), ("ups_b", c_float_p * MAX_STAGES),
                ("rb_w", C.POINTER(c_float_p)), ("rb_b", C.POINTER(c_float_p)), ("n_rb", C.c_int32),
                ("conv_post_w", c_float_p), ("conv_post_b", c_float_p)]


class TteCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_filter_ffn", "ffn_k1", "ffn_k2", "max_len", "enc_layers",
                                         "enc_heads", "dec_layers", "dec_heads", "dp_filter", "dp_kernel", "vocab",
                                         "n_speaker", "n_codes")]


class FftWeights(C.Structure):
    _fields_ = [(n, c_float_p) for n in ("qkv", "in_proj", "out_proj", "wo", "conv1_w", "conv1_b", "conv2_w",
                                         "conv2_b", "attn_norm_w", "attn_norm_b", "conv_norm_w", "conv_norm_b")]


class TteWeights(C.Structure):
    _fields_ = [("pe", c_float_p), ("tok_emb", c_float_p), ("speaker_emb", c_float_p),
                ("dp_conv0_w", c_float_p), ("dp_conv0_b", c_float_p), ("dp_ln0_w", c_float_p), ("dp_ln0_b", c_float_p),
                ("dp_conv1_w", c_float_p), ("dp_conv1_b", c_float_p), ("dp_ln1_w", c_float_p), ("dp_ln1_b", c_float_p),
                ("dp_proj_w", c_float_p), ("dp_proj_b", c_float_p),
                ("enc", C.POINTER(FftWeights)), ("dec", C.POINTER(FftWeights)),
                ("head_w", c_float_p), ("head_b", c_float_p)]


# name -> (restype, argtypes); every symbol include/parrot_hip.h and include/parrot_hip_debug.h declare
vp, i32, sz, f32 = C.c_void_p, C.c_int32, C.c_size_t, C.c_float
SIGNATURES = {
    "parrot_abi_version": (C.c_int, []),
    "parrot_last_error": (C.c_char_p, []),
    "parrot_selftest": (C.c_int, [vp]),
    "parrot_set_default_precision": (C.c_int, [i32]),
    "parrot_set_fused_resblocks": (C.c_int, [i32]),
    "parrot_set_tte_merge": (C.c_int, [i32]),
    "parrot_conv_create": (C.c_int, [C.POINTER(vp), C.POINTER(ConvDesc), c_float_p, c_float_p]),
    "parrot_conv_destroy": (None, [vp]),
    "parrot_conv_run": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "parrot_conv_out_len": (C.c_int, [vp, i32]),
    "parrot_conv_num_tile_cfgs": (C.c_int, []),
    "parrot_debug_copy": (C.c_int, [vp, vp, sz, vp]),
    "parrot_debug_mfma_ceiling": (C.c_int, [i32, i32, C.POINTER(C.c_double)]),
    "parrot_prof_begin": (C.c_int, []),
    "parrot_prof_begin_row": (C.c_int, [i32]),
    "parrot_prof_end": (C.c_int, [C.POINTER(C.c_double), i32]),
    "parrot_voc_create": (C.c_int, [C.POINTER(vp), C.POINTER(VocCfg), C.POINTER(VocWeights)]),
    "parrot_voc_create_ex": (C.c_int, [C.POINTER(vp), C.POINTER(VocCfg), C.POINTER(VocWeights), i32, i32]),
    "parrot_voc_destroy": (None, [vp]),
    "parrot_voc_precision": (C.c_int, [vp]),
    "parrot_voc_debug_absmax": (C.c_int, [vp, vp]),
    "parrot_voc_workspace_bytes": (sz, [vp, i32, i32]),
    "parrot_voc_forward": (C.c_int, [vp, vp, vp, vp, i32, i32, vp, C.POINTER(vp), vp, sz, vp]),
    "parrot_voc_forward_feats": (C.c_int, [vp, vp, vp, vp, i32, vp, i32, i32, vp, C.POINTER(vp), vp, sz, vp]),
    "parrot_voc_chunked_workspace_bytes": (sz, [vp, i32, i32, i32]),
    "parrot_voc_forward_chunked": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "parrot_voc_check": (C.c_int, [vp, vp]),
    "parrot_voc_status_async": (C.c_int, [vp, vp, vp]),
    "parrot_voc_status_peek_async": (C.c_int, [vp, vp, vp]),
    "parrot_voc_receptive_units": (C.c_int, [vp]),
    "parrot_voc_wait_stage": (C.c_int, [vp, C.c_int32, vp]),
    "parrot_voc_out_len": (C.c_int64, [vp, i32]),
    "parrot_wav_to_int16": (C.c_int, [vp, vp, sz, vp]),
    "parrot_tte_create": (C.c_int, [C.POINTER(vp), C.POINTER(TteCfg), C.POINTER(TteWeights)]),
    "parrot_tte_create_ex": (C.c_int, [C.POINTER(vp), C.POINTER(TteCfg), C.POINTER(TteWeights), i32, i32]),
    "parrot_tte_destroy": (None, [vp]),
    "parrot_tte_precision": (C.c_int, [vp]),
    "parrot_tte_guard_logits": (C.c_int, [vp, vp, vp, i32, vp]),
    "parrot_tte_state_bytes": (sz, [vp, i32, i32]),
    "parrot_tte_workspace_bytes": (sz, [vp, i32, i32, i32]),
    "parrot_tte_encode": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, sz, vp, sz, vp]),
    "parrot_tte_decode": (C.c_int, [vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp, sz, vp]),
    "parrot_tte_check": (C.c_int, [vp, vp]),
    "parrot_tte_status_async": (C.c_int, [vp, vp, vp]),
    "parrot_tte_status_peek_async": (C.c_int, [vp, vp, vp]),
    "parrot_tte_guard_stats_async": (C.c_int, [vp, vp, vp]),
    "parrot_tte_debug_stages": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "parrot_length_regulator_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "parrot_length_regulator": (C.c_int, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
}

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the bound library; raises if it is absent -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run `python -m parrot_tts_amd.build` (hipcc, gfx950). "
                              "parrot_tts_amd has no CPU fallback.")
        # torch first, always: it brings its own HIP runtime, and the library must bind to THAT copy -- loaded the other
        # way round (e.g. __graft_entry__.build() before the first `import torch`) the process ends up with two
        # runtimes and the second one sees no device
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if handle.parrot_abi_version() != ABI_VERSION:
            raise ImportError("libparrot_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(code: int) -> None:
    if code != 0:
        raise ParrotHipError(code, lib().parrot_last_error().decode())


def fptr(t):
    """Host fp32 pointer of a contiguous CPU float tensor (caller keeps `t` alive)."""
    assert t.device.type == "cpu" and t.dtype.is_floating_point and t.is_contiguous()
    return C.cast(t.data_ptr(), c_float_p)
