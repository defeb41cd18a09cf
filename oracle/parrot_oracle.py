"""CPU oracle: a functional fp32 restatement of the Parrot-TTS synthesis hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``parrot_tts_amd/`` may import this module; the
only legal users are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` (where it is the thing *compared against / timed beside*, never the product).

Parity pinning: the reference has no tests and no golden files, so this oracle is
pinned against outputs of the reference itself: ``tools/make_goldens.py`` imports the
reference modules from /root/reference in the build container, runs them on seeded synthetic
checkpoints (``parrot_tts_amd.synth``) and stores inputs+outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` demands bit-equality (max-abs diff 0.0) between this file
and those vectors.  It deliberately uses the same ATen CPU ops in the same order as the
reference so that equality is exact, including the quirks Q1-Q7 of SURVEY §8a.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1  # utils/vocoder/models.py:10


# ======================================================================================
# TTE  (modules/fft.py, modules/duration.py, modules/data.py, modules/parrot.py)
# ======================================================================================
def pos_emb(pe: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """modules/fft.py:17-19 -- adds the single row ``pe[T]`` to every position (quirk Q1)."""
    return pe[x.size(1)] + x


def mha_math_path(q, k, v, in_proj_weight, out_proj_weight, n_head, key_padding_mask):
    """torch.nn.MultiheadAttention (bias=False, batch_first=True, eval) as reached from
    modules/fft.py:56: distinct q/k/v tensors + default need_weights=True -> the explicit
    math path of F.multi_head_attention_forward (quirk Q3)."""
    B, T, E = q.shape
    hd = E // n_head
    # batch_first -> (T,B,E)
    q, k, v = (t.transpose(1, 0) for t in (q, k, v))
    w_q, w_k, w_v = in_proj_weight.chunk(3)
    # .contiguous(): the reference's weights are nn.Parameters (requires_grad), for which ATen's
    # matmul folds (T,B,E)x(E,E) into ONE mm over a contiguous copy of the left operand; a plain
    # tensor weight would take the broadcast-bmm route instead (different summation order, ~1e-6).
    q, k, v = F.linear(q.contiguous(), w_q), F.linear(k.contiguous(), w_k), F.linear(v.contiguous(), w_v)
    q = q.view(T, B * n_head, hd).transpose(0, 1)
    k = k.view(T, B * n_head, hd).transpose(0, 1)
    v = v.view(T, B * n_head, hd).transpose(0, 1)
    attn_mask = None
    if key_padding_mask is not None:
        m = torch.zeros_like(key_padding_mask, dtype=q.dtype).masked_fill_(key_padding_mask, float("-inf"))
        attn_mask = m.view(B, 1, 1, T).expand(-1, n_head, -1, -1).reshape(B * n_head, 1, T)
    q_scaled = q * math.sqrt(1.0 / float(hd))
    if attn_mask is not None:
        w = torch.baddbmm(attn_mask, q_scaled, k.transpose(-2, -1))
    else:
        w = torch.bmm(q_scaled, k.transpose(-2, -1))
    w = F.softmax(w, dim=-1)
    o = torch.bmm(w, v)
    o = o.transpose(0, 1).contiguous().view(T * B, E)
    o = F.linear(o, out_proj_weight).view(T, B, E)
    return o.transpose(1, 0)


def attention(sd, p, x, n_head, key_padding_mask):
    """modules/fft.py:53-59 -- qkv Linear, torch MHA, wo Linear (all bias-free)."""
    D = x.shape[-1]
    q, k, v = F.linear(x, sd[p + "qkv.weight"]).split(D, dim=2)
    y = mha_math_path(q, k, v, sd[p + "mha.in_proj_weight"], sd[p + "mha.out_proj.weight"], n_head, key_padding_mask)
    return F.linear(y.contiguous(), sd[p + "wo.weight"])  # same fold as in mha_math_path


def conv_layer(sd, p, x, kernel_sizes):
    """modules/fft.py:78-82 -- transpose, Conv1d(k1,'same'), ReLU, Conv1d(k2,'same'), transpose."""
    o = x.transpose(1, 2)
    o = F.conv1d(o, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=(kernel_sizes[0] - 1) // 2)
    o = F.conv1d(F.relu(o), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=(kernel_sizes[1] - 1) // 2)
    return o.transpose(1, 2)


def fft_block(sd, p, x, n_head, kernel_sizes, key_padding_mask):
    """modules/fft.py:94-100 -- pre-LN attention + pre-LN conv FFN, both residual."""
    D = x.shape[-1]
    h = x + attention(sd, p + "attention.", F.layer_norm(x, (D,), sd[p + "attn_norm.weight"], sd[p + "attn_norm.bias"]),
                      n_head, key_padding_mask)
    return h + conv_layer(sd, p + "convlayer.", F.layer_norm(h, (D,), sd[p + "conv_norm.weight"], sd[p + "conv_norm.bias"]),
                          kernel_sizes)


def duration_predictor(sd, x, mask, kernel_size):
    """modules/duration.py:41-48 (+ Conv wrapper :74-79).  Second conv pads 1 regardless of k (Q4)."""
    p = "duration_predictor."
    NF = sd[p + "layers.0.conv.weight"].shape[0]

    def conv(idx, t, pad):
        t = t.contiguous().transpose(1, 2)
        t = F.conv1d(t, sd[p + f"layers.{idx}.conv.weight"], sd[p + f"layers.{idx}.conv.bias"], padding=pad)
        return t.contiguous().transpose(1, 2)

    o = conv(0, x, (kernel_size - 1) // 2)
    o = F.layer_norm(F.relu(o), (NF,), sd[p + "layers.2.weight"], sd[p + "layers.2.bias"])
    o = conv(4, o, 1)
    o = F.layer_norm(F.relu(o), (NF,), sd[p + "layers.6.weight"], sd[p + "layers.6.bias"])
    o = F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"]).squeeze(-1)
    if mask is not None:
        o = o.masked_fill(mask, 0.0)
    return o


def get_mask_from_lengths(lengths: List[int], max_len: int) -> torch.Tensor:
    """modules/data.py:8-20 -- ``ids <= len`` (one extra True per short row, quirk Q2)."""
    lengths_t = torch.tensor(lengths)
    ids = torch.arange(0, max_len).unsqueeze(0).expand(lengths_t.shape[0], -1)
    return ids <= lengths_t.unsqueeze(1).expand(-1, max_len)


def length_regulator(batch_seq, batch_dur):
    """modules/duration.py:6-24 -- per-row repeat_interleave, zero right-pad to max sum, Q2 mask."""
    expanded, out_lens = [], []
    max_len = int(batch_dur.sum(dim=1).max())
    for seq, dur in zip(batch_seq, batch_dur):
        seq = seq.repeat_interleave(dur, dim=0)
        out_lens.append(seq.shape[0])
        expanded.append(F.pad(seq, (0, 0, 0, max_len - seq.shape[0]), "constant", 0.0))
    return torch.stack(expanded), get_mask_from_lengths(out_lens, max_len), out_lens


def durations_from_log(log_dur: torch.Tensor) -> torch.Tensor:
    """modules/parrot.py:82-86 -- clamp(round(exp(ld) - 1), min=0).long(); round = half-to-even."""
    return torch.clamp(torch.round(torch.exp(log_dur) - 1), min=0).long()


def tte_forward(sd: Dict[str, torch.Tensor], cfg: dict, batch: dict, return_stages: bool = False):
    """modules/parrot.py:90-110 with inference=True.  ``sd`` uses Parrot.state_dict() keys.

    Returns dict(logits (B,L,V), tgt_mask (B,L) bool, log_dur (B,S), dur (B,S) i64, lens list)."""
    tr = cfg["transformer"]
    ks = tr["conv_kernel_sizes"]
    src_kpm = ~batch["src_mask"]
    out = F.embedding(batch["phones"], sd["tok_emb.weight"])
    out = pos_emb(sd["pos_emb.pe"], out)
    stages = {"emb": out}
    for n in range(tr["encoder"]["n_layer"]):
        out = fft_block(sd, f"encoder_layers.{n}.", out, tr["encoder"]["n_head"], ks, src_kpm)
        stages[f"enc{n}"] = out
    if "speaker_emb.weight" in sd:
        out = out + F.embedding(batch["speaker"], sd["speaker_emb.weight"]).unsqueeze(1)
    stages["enc_out"] = out
    log_dur = duration_predictor(sd, out, src_kpm, cfg["duration_predictor"]["kernel_size"])
    dur = durations_from_log(log_dur)
    out, tgt_mask, lens = length_regulator(out, dur)
    out = pos_emb(sd["pos_emb.pe"], out)
    stages["dec_in"] = out
    for n in range(tr["decoder"]["n_layer"]):
        out = fft_block(sd, f"decoder_layers.{n}.", out, tr["decoder"]["n_head"], ks, ~tgt_mask)
        stages[f"dec{n}"] = out
    logits = F.linear(out, sd["head.weight"], sd["head.bias"])
    res = {"logits": logits, "tgt_mask": tgt_mask, "log_dur": log_dur, "dur": dur, "lens": lens}
    if return_stages:
        res["stages"] = stages
    return res


def tte_infer(sd, cfg, batch) -> List[List[int]]:
    """modules/parrot.py:112-120 -- argmax + per-row mask select (rows emit len+1 ids, Q2)."""
    r = tte_forward(sd, cfg, batch)
    codes = torch.argmax(r["logits"], dim=-1)
    return [c[m].numpy().tolist() for c, m in zip(codes, r["tgt_mask"])]


# ======================================================================================
# vocoder  (utils/vocoder/models.py:13-169, utils/vocoder/utils.py:44-45)
# ======================================================================================
def get_padding(kernel_size: int, dilation: int = 1) -> int:
    """utils/vocoder/utils.py:44-45."""
    return int((kernel_size * dilation - dilation) / 2)


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Old-style ``torch.nn.utils.weight_norm`` (dim=0): w = v * (g / ||v||) with the norm over
    all dims but 0 (utils/vocoder/models.py:7,17-28,75,81,91; ``remove_weight_norm`` :113-119).
    Keys ``X.weight_g``/``X.weight_v`` become ``X.weight``; everything else passes through."""
    out = {}
    for k, t in sd.items():
        if k.endswith(".weight_g"):
            base = k[: -len(".weight_g")]
            out[base + ".weight"] = torch._weight_norm(sd[base + ".weight_v"], t, 0)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = t
    return out


def resblock1(w, p, x, k, dilations):
    """utils/vocoder/models.py:31-38; the constructor (:17-22) reads dilation[0], [1], [2] literally: exactly three pairs,
    IndexError for a shorter list, further entries ignored."""
    for m, d in enumerate((dilations[0], dilations[1], dilations[2])):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w[p + f"convs1.{m}.weight"], w[p + f"convs1.{m}.bias"], padding=get_padding(k, d), dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, w[p + f"convs2.{m}.weight"], w[p + f"convs2.{m}.bias"], padding=get_padding(k, 1))
        x = xt + x
    return x


def resblock2(w, p, x, k, dilations):
    """utils/vocoder/models.py:58-62; the constructor (:51-54) reads dilation[0], [1]: exactly two convs."""
    for m, d in enumerate((dilations[0], dilations[1])):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, w[p + f"convs.{m}.weight"], w[p + f"convs.{m}.bias"], padding=get_padding(k, d), dilation=d)
        x = xt + x
    return x


def generator_forward(w, h, x, stages: Optional[dict] = None):
    """utils/vocoder/models.py:95-111.  ``w`` = weight-norm-folded weights."""
    nk = len(h["resblock_kernel_sizes"])
    rb = resblock1 if str(h["resblock"]) == "1" else resblock2
    x = F.conv1d(x, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    if stages is not None:
        stages["conv_pre"] = x
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if stages is not None:
            stages[f"ups{i}"] = x
        xs = None
        for j in range(nk):
            r = rb(w, f"resblocks.{i * nk + j}.", x, h["resblock_kernel_sizes"][j], h["resblock_dilation_sizes"][j])
            if xs is None:
                xs = r
            else:
                xs += r
        x = xs / nk
        if stages is not None:
            stages[f"mrf{i}"] = x
    x = F.leaky_relu(x)  # default slope 0.01 (quirk Q5, models.py:107)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    return torch.tanh(x)


def upsample_condition(signal, max_frames: int):
    """utils/vocoder/models.py:132-151 (`CodeGenerator._upsample`): a (B,C,T') / (B,C) / flat conditioning signal is
    repeated max_frames // T' times per frame; a remainder raises like the reference."""
    if signal.dim() == 2:
        signal = signal.unsqueeze(2)
    elif signal.dim() != 3:
        signal = signal.view(-1, 1, 1)
    bsz, ch, cond = signal.shape
    rep = max_frames // cond
    if (max_frames - cond * rep) // rep > 0:
        raise NotImplementedError("Padding condition signal - misalignment between condition features.")
    return signal.unsqueeze(3).repeat(1, 1, 1, rep).view(bsz, ch, max_frames)


def code_generator_forward(sd, h, code, spkr=None, stages: Optional[dict] = None, feats: Optional[dict] = None):
    """utils/vocoder/models.py:153-169 (+ _upsample :132-151): unit embedding (B,U,E)->(B,E,U),
    speaker embedding broadcast over time, channel concat, then every EXTRA keyword tensor of the call (``feats``, in
    keyword order; the reference skips 'spkr', 'code' and 'f0', :162-167) upsampled and concatenated, Generator.forward.
    ``sd`` may carry weight_g/weight_v or plain weight."""
    w = fold_weight_norm(sd)
    x = F.embedding(code, w["dict.weight"]).transpose(1, 2)
    if h.get("multispkr"):
        s = F.embedding(spkr, w["spkr.weight"]).transpose(1, 2)  # (B,E,1)
        bsz, ch, cond = s.shape
        s = s.unsqueeze(3).repeat(1, 1, 1, x.shape[-1] // cond).view(bsz, ch, x.shape[-1])
        x = torch.cat([x, s], dim=1)
    for name, feat in (feats or {}).items():
        if name in ("spkr", "code", "f0"):
            continue
        x = torch.cat([x, upsample_condition(feat, x.shape[-1])], dim=1)
    if stages is not None:
        stages["embed"] = x
    return generator_forward(w, h, x, stages)


def to_int16(audio: torch.Tensor):
    """utils/vocoder/inference.py:71-73 -- x*32768 then numpy astype('int16') (C cast)."""
    return (audio * 32768.0).cpu().numpy().astype("int16")
