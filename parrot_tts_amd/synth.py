"""Synthetic configs, checkpoints and inputs in the reference's exact on-disk layouts.

No reference checkpoint is available offline (demo.ipynb cell 4 downloads them), so every
golden vector, parity test and bench run uses seeded synthetic weights laid out exactly like
the reference's ``state_dict``s:

* TTE  : keys of ``Parrot.state_dict()``            (reference modules/parrot.py:13-65)
* vocoder: keys of ``CodeGenerator.state_dict()`` with weight-norm attached
           (``weight_g``/``weight_v``; reference utils/vocoder/models.py:69-130)

Weights come from ``numpy.random.Generator(PCG64(seed))`` so they regenerate bit-identically
on the GPU box (same image, same numpy); ``state_digest`` lets tests prove that.
"""
from __future__ import annotations

import copy
import hashlib
import math
from typing import Dict, Optional

import numpy as np
import torch


# --------------------------------------------------------------------------------------
# configs (same keys/values as utils/TTE/TTE_config.yaml and utils/vocoder/config.json)
# --------------------------------------------------------------------------------------
def default_tte_config(root_path: str = "runs/TTE") -> dict:
    """Model-relevant part of utils/TTE/TTE_config.yaml:1-30."""
    return {
        "path": {"root_path": root_path, "alignment_path": "runs/aligner"},
        "preprocess": {"val_size": 100, "hubert_codes": 1000, "speaker": "_"},
        "transformer": {
            "encoder": {"n_layer": 4, "n_head": 2, "dropout_p": 0.1},
            "decoder": {"n_layer": 4, "n_head": 2, "dropout_p": 0.1},
            "d_model": 256,
            "conv_n_filter": 1024,
            "conv_kernel_sizes": [9, 1],
            "max_len": 3500,
        },
        "duration_predictor": {"n_filter": 256, "kernel_size": 3, "dropout_p": 0.5},
    }


def default_voc_config() -> dict:
    """Model-relevant part of utils/vocoder/config.json:5-32."""
    return {
        "resblock": "1",
        "seed": 1234,
        "upsample_rates": [5, 4, 4, 2, 2],
        "upsample_kernel_sizes": [11, 8, 8, 4, 4],
        "upsample_initial_channel": 512,
        "resblock_kernel_sizes": [3, 7, 11],
        "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        "num_embeddings": 1000,
        "embedding_dim": 128,
        "model_in_dim": 256,
        "code_hop_size": 320,
        "multispkr": "_",
        "sampling_rate": 16000,
    }


def small_tte_config(root_path: str = "runs/TTE") -> dict:
    """A reduced TTE (same graph, smaller dims) for fast CPU goldens / smoke tests."""
    cfg = default_tte_config(root_path)
    cfg["transformer"].update(d_model=64, conv_n_filter=128, max_len=400)
    cfg["transformer"]["encoder"]["n_layer"] = 2
    cfg["transformer"]["decoder"]["n_layer"] = 2
    cfg["duration_predictor"]["n_filter"] = 64
    cfg["preprocess"]["hubert_codes"] = 100
    return cfg


def small_voc_config() -> dict:
    """A reduced generator (same graph: 5 stages, 3x3 MRF) for fast CPU goldens / smoke tests."""
    h = default_voc_config()
    h.update(upsample_initial_channel=64, embedding_dim=16, model_in_dim=32, num_embeddings=100)
    return h


def corner_voc_config() -> dict:
    """Config corners the reference accepts: odd ``upsample_kernel_size - upsample_rate`` (ConvTranspose1d then
    yields T u + 1 samples, reference utils/vocoder/models.py:80-83) and dilation lists of different lengths per kernel size
    (ResBlock1 reads the first three entries only, models.py:17-22).  Golden: tests/golden/voc_small_corners.npz."""
    h = small_voc_config()
    h["upsample_rates"], h["upsample_kernel_sizes"] = [4, 2, 2], [9, 4, 5]       # k - u = 5, 2, 3: two odd stages
    h["resblock_dilation_sizes"] = [[1, 2, 3], [1, 3, 5, 7], [2, 1, 4, 9, 11]]    # entries beyond the third are never read
    return h


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------
def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _normal(rng, shape, std):
    # .clone(): torch-owned (64-byte aligned) storage, like nn.Parameter copies in the reference --
    # CPU BLAS picks alignment-dependent code paths, so this keeps oracle == reference bit-exact.
    return torch.from_numpy((rng.standard_normal(size=shape) * std).astype(np.float32)).clone()


def sinusoid_table(max_len: int, d_model: int) -> torch.Tensor:
    """The (max_len, d_model) buffer ``pos_emb.pe`` (reference modules/fft.py:21-38).

    Real checkpoints carry this buffer in their state_dict; synthetic ones rebuild it with
    the same formula (sin on even columns, cos on odd, 10000^(-2i/d))."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float) * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position.float() * div_term)
    pe[:, 1::2] = torch.cos(position.float() * div_term)
    return pe


def patch_pe_rows(sd: Dict[str, torch.Tensor], idx, rows) -> None:
    """Overwrite rows ``idx`` of ``pos_emb.pe`` with the exact values a golden run used (only rows S and L
    are ever read: reference quirk Q1)."""
    for i, r in zip(idx, rows):
        sd["pos_emb.pe"][int(i)] = torch.as_tensor(r)


def state_digest(sd: Dict[str, torch.Tensor]) -> str:
    """sha256 over (key, shape, raw bytes) of every tensor, in sorted key order.

    ``pos_emb.pe`` is skipped: it is rebuilt with torch.sin/cos/exp, whose last bit depends on the
    host CPU's vector ISA; goldens carry the two rows they used instead (``patch_pe_rows``)."""
    hsh = hashlib.sha256()
    for k in sorted(sd):
        if k == "pos_emb.pe":
            continue
        t = sd[k].detach().cpu().contiguous()
        hsh.update(k.encode())
        hsh.update(str(tuple(t.shape)).encode())
        hsh.update(t.numpy().tobytes())
    return hsh.hexdigest()


# --------------------------------------------------------------------------------------
# TTE
# --------------------------------------------------------------------------------------
def synth_tte_state_dict(cfg: dict, src_vocab_size: int, n_speaker: int, seed: int = 42,
                         forced_duration: Optional[int] = None, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded weights with the key set of ``Parrot.state_dict()``.

    ``forced_duration=d`` zeroes ``duration_predictor.proj.weight`` and sets its bias to
    ``ln(d+1)`` so every valid token gets duration ``d`` (L = d*S exactly).
    """
    rng = _rng(seed)
    tr = cfg["transformer"]
    D, F = tr["d_model"], tr["conv_n_filter"]
    k1, k2 = tr["conv_kernel_sizes"]
    dp = cfg["duration_predictor"]
    NF, DK = dp["n_filter"], dp["kernel_size"]
    V = cfg["preprocess"]["hubert_codes"]
    sd: Dict[str, torch.Tensor] = {}
    sd["pos_emb.pe"] = sinusoid_table(tr["max_len"], D)
    sd["tok_emb.weight"] = _normal(rng, (src_vocab_size, D), 1.0)
    if n_speaker > 1:
        sd["speaker_emb.weight"] = _normal(rng, (n_speaker, D), 0.5)
    for idx, (cin, cout) in ((0, (D, NF)), (4, (NF, NF))):
        sd[f"duration_predictor.layers.{idx}.conv.weight"] = _normal(rng, (cout, cin, DK), gain / math.sqrt(cin * DK))
        sd[f"duration_predictor.layers.{idx}.conv.bias"] = _normal(rng, (cout,), 0.05)
    for idx in (2, 6):
        sd[f"duration_predictor.layers.{idx}.weight"] = 1.0 + _normal(rng, (NF,), 0.1)
        sd[f"duration_predictor.layers.{idx}.bias"] = _normal(rng, (NF,), 0.05)
    sd["duration_predictor.proj.weight"] = _normal(rng, (1, NF), 1.0 / math.sqrt(NF))
    sd["duration_predictor.proj.bias"] = torch.full((1,), 1.0)
    for side in ("encoder", "decoder"):
        for n in range(tr[side]["n_layer"]):
            p = f"{side}_layers.{n}."
            sd[p + "attention.qkv.weight"] = _normal(rng, (3 * D, D), gain / math.sqrt(D))
            sd[p + "attention.mha.in_proj_weight"] = _normal(rng, (3 * D, D), gain / math.sqrt(D))
            sd[p + "attention.mha.out_proj.weight"] = _normal(rng, (D, D), gain / math.sqrt(D))
            sd[p + "attention.wo.weight"] = _normal(rng, (D, D), gain / math.sqrt(D))
            sd[p + "convlayer.conv1.weight"] = _normal(rng, (F, D, k1), gain / math.sqrt(D * k1))
            sd[p + "convlayer.conv1.bias"] = _normal(rng, (F,), 0.05)
            sd[p + "convlayer.conv2.weight"] = _normal(rng, (D, F, k2), gain / math.sqrt(F * k2))
            sd[p + "convlayer.conv2.bias"] = _normal(rng, (D,), 0.05)
            for nm in ("attn_norm", "conv_norm"):
                sd[p + nm + ".weight"] = 1.0 + _normal(rng, (D,), 0.1)
                sd[p + nm + ".bias"] = _normal(rng, (D,), 0.05)
    sd["head.weight"] = _normal(rng, (V, D), 1.0 / math.sqrt(D))
    sd["head.bias"] = _normal(rng, (V,), 0.05)
    if forced_duration is not None:
        sd["duration_predictor.proj.weight"] = torch.zeros(1, NF)
        sd["duration_predictor.proj.bias"] = torch.full((1,), math.log(forced_duration + 1.0))
    return sd


def synth_tte_batch(B: int, S: int, vocab: int, n_speaker: int, seed: int = 0, ragged: bool = False) -> dict:
    """A collated inference batch like ``ParrotDataset.collate_fn`` makes (modules/data.py:102-120):
    ``phones`` (B,S) int64 right-padded with 0, ``src_mask`` (B,S) bool True=valid, ``speaker`` (B,)."""
    rng = _rng(seed)
    phones = rng.integers(2, vocab, size=(B, S), dtype=np.int64)
    lens = np.full((B,), S, dtype=np.int64)
    if ragged and B > 1:
        lens = rng.integers(max(1, S // 2), S + 1, size=(B,), dtype=np.int64)
        lens[0] = S  # pad_sequence pads to the longest row
        for b in range(B):
            phones[b, lens[b]:] = 0
    speaker = rng.integers(0, max(1, n_speaker), size=(B,), dtype=np.int64)
    phones_t = torch.from_numpy(phones)
    return {"phones": phones_t, "src_mask": phones_t != 0, "speaker": torch.from_numpy(speaker),
            "src_lens": torch.from_numpy(lens)}


# --------------------------------------------------------------------------------------
# vocoder
# --------------------------------------------------------------------------------------
def voc_layer_shapes(h: dict):
    """(name, kind, cin, cout, k, stride) for every weight-normed layer, in state_dict order."""
    c0 = h["upsample_initial_channel"]
    out = [("conv_pre", "conv", h.get("model_in_dim", 128), c0, 7, 1)]
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        out.append((f"ups.{i}", "convT", c0 // (2 ** i), c0 // (2 ** (i + 1)), k, u))
    nk = len(h["resblock_kernel_sizes"])
    for i in range(len(h["upsample_rates"])):
        ch = c0 // (2 ** (i + 1))
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            n_conv = 3 if str(h["resblock"]) == "1" else 2  # what the reference constructors build (models.py:17-22,51-54), whatever the list length
            for m in range(n_conv):
                if str(h["resblock"]) == "1":
                    out.append((f"resblocks.{i * nk + j}.convs1.{m}", "conv", ch, ch, k, 1))
                    out.append((f"resblocks.{i * nk + j}.convs2.{m}", "conv", ch, ch, k, 1))
                else:
                    out.append((f"resblocks.{i * nk + j}.convs.{m}", "conv", ch, ch, k, 1))
    out.append(("conv_post", "conv", c0 // (2 ** len(h["upsample_rates"])), 1, 7, 1))
    return out


def synth_voc_state_dict(h: dict, seed: int = 1234, scale: float = 1.0, jitter_g: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded generator checkpoint body (what sits under ``['generator']`` in ``g_%08d`` files,
    reference utils/vocoder/train.py:183-186) with weight-norm attached.

    Recipe from SURVEY §8c: ``weight_v ~ N(0, scale/sqrt(fan_in))`` with fan_in = C_in*k
    (Conv1d) or C_in*k/u (ConvTranspose1d); ``weight_g = ||v|| * U(1,1.3)``; ``bias ~ N(0,0.02)``.
    Conv1d ``weight_v`` is (C_out,C_in,k) with g per out-channel; ConvTranspose1d ``weight_v`` is
    (C_in,C_out,k) with g per *in*-channel (both: norm over dims != 0)."""
    rng = _rng(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, kind, cin, cout, k, u in voc_layer_shapes(h):
        if kind == "conv":
            v = _normal(rng, (cout, cin, k), scale / math.sqrt(cin * k))
        else:
            v = _normal(rng, (cin, cout, k), scale / math.sqrt(cin * k / u))
        norm = v.flatten(1).norm(dim=1).reshape(-1, 1, 1)
        g = norm.clone()
        if jitter_g:
            g = g * torch.from_numpy(rng.uniform(1.0, 1.3, size=tuple(g.shape)).astype(np.float32))
        sd[name + ".bias"] = _normal(rng, (cout,), 0.02)
        sd[name + ".weight_g"] = g
        sd[name + ".weight_v"] = v
    sd["dict.weight"] = _normal(rng, (h["num_embeddings"], h["embedding_dim"]), 1.0)
    if h.get("multispkr"):
        sd["spkr.weight"] = _normal(rng, (10, h["embedding_dim"]), 1.0)
    return sd


def synth_voc_batch(B: int, U: int, h: dict, seed: int = 0) -> dict:
    """``code`` (B,U) int64 in [0,num_embeddings), ``spkr`` (B,1) int64 in [0,10)."""
    rng = _rng(seed)
    code = torch.from_numpy(rng.integers(0, h["num_embeddings"], size=(B, U), dtype=np.int64))
    spkr = torch.from_numpy(rng.integers(0, 10, size=(B, 1), dtype=np.int64))
    return {"code": code, "spkr": spkr}


def clone_config(cfg: dict) -> dict:
    return copy.deepcopy(cfg)
