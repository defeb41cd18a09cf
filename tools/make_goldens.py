#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE implementation (imported from
/root/reference, CPU fp32) on seeded synthetic checkpoints.

Runs only in the build container (the GPU box has no /root/reference).  Nothing from the
reference is copied: the fixtures hold plain input/output arrays plus a sha256 digest of the
synthetic state_dict so tests can prove they regenerated the same weights.

Also cross-checks oracle/parrot_oracle.py against the reference while it is at hand and
prints the max-abs differences (expected: 0.0 everywhere).

    python tools/make_goldens.py            # writes tests/golden/*.npz
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from parrot_tts_amd import synth  # noqa: E402
from oracle import parrot_oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_num_threads(8)


def _import_reference():
    # vocoder dir first: its bare `utils.py` must win over the namespace package /root/reference/utils
    sys.path.insert(0, os.path.join(REF, "utils", "vocoder"))
    import models as ref_voc_models  # noqa
    import utils as ref_voc_utils  # noqa
    sys.path.insert(1, REF)
    from modules.parrot import Parrot  # noqa
    from modules.fft import FFTBlock  # noqa
    from modules.duration import DurationPredictor, length_regulator  # noqa
    return ref_voc_models, ref_voc_utils, Parrot, FFTBlock, DurationPredictor, length_regulator


ref_voc_models, ref_voc_utils, RefParrot, RefFFTBlock, RefDurationPredictor, ref_length_regulator = _import_reference()


def build_ref_parrot(cfg, vocab, n_spk, sd):
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "speakers.json"), "w") as f:
        json.dump({f"spk{i}": i for i in range(n_spk)}, f)
    cfg = synth.clone_config(cfg)
    cfg["path"]["root_path"] = tmp
    m = RefParrot(cfg, vocab, 0)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


def top2(logits):
    v, i = torch.topk(logits, 2, dim=-1)
    return i[..., 0], (v[..., 0] - v[..., 1])


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max()) if a.numel() else 0.0


def tte_case(name, cfg, vocab, n_spk, B, S, seed_w, seed_in, ragged, forced=None, keep_logits_rows=1, gain=1.0):
    sd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=seed_w, forced_duration=forced, gain=gain)
    batch = synth.synth_tte_batch(B, S, vocab, n_spk, seed=seed_in, ragged=ragged)
    model = build_ref_parrot(cfg, vocab, n_spk, sd)
    with torch.no_grad():
        logits, _, tgt_mask, log_dur = model(batch, inference=True)
        ids_ragged = model.infer(batch)
    ids, margin = top2(logits)
    dur = torch.clamp(torch.round(torch.exp(log_dur) - 1), min=0).long()
    frac = torch.exp(log_dur) - 1
    half_dist = ((frac - torch.floor(frac)) - 0.5).abs()
    # oracle cross-check
    with torch.no_grad():
        o = O.tte_forward(sd, cfg, batch)
        o_ids = O.tte_infer(sd, cfg, batch)
    print(f"[{name}] L={logits.shape[1]} oracle-vs-ref: logits {maxdiff(o['logits'], logits):.3g} "
          f"log_dur {maxdiff(o['log_dur'], log_dur):.3g} mask_eq {bool((o['tgt_mask'] == tgt_mask).all())} "
          f"ids_eq {o_ids == ids_ragged}  min-margin {float(margin[tgt_mask].min()):.3g} "
          f"min-half-dist {float(half_dist[batch['src_mask']].min()):.3g}")
    rag = np.full((B, logits.shape[1] + 1), -1, dtype=np.int64)
    for b, r in enumerate(ids_ragged):
        rag[b, : len(r)] = r
    np.savez_compressed(
        os.path.join(GOLD, name + ".npz"),
        digest=np.array(synth.state_digest(sd)),
        meta=np.array(json.dumps(dict(vocab=vocab, n_spk=n_spk, B=B, S=S, seed_w=seed_w, seed_in=seed_in,
                                      ragged=ragged, forced=forced, gain=gain))),
        phones=batch["phones"].numpy(), src_mask=batch["src_mask"].numpy(), speaker=batch["speaker"].numpy(),
        log_dur=log_dur.numpy(), dur=dur.numpy(), half_dist=half_dist.numpy(),
        tgt_mask=tgt_mask.numpy(), ids=ids.numpy(), margin=margin.numpy(), ids_ragged=rag,
        ids_ragged_len=np.array([len(r) for r in ids_ragged]),
        logits_head=logits[:keep_logits_rows].numpy(),
        pe_idx=np.array([S, logits.shape[1]]), pe_rows=sd["pos_emb.pe"][[S, logits.shape[1]]].numpy(),
    )


def block_cases():
    torch.manual_seed(0)
    cfg = synth.default_tte_config()
    D = 256
    sd_full = synth.synth_tte_state_dict(cfg, 50, 2, seed=7)
    # one FFTBlock with key-padding mask
    blk = RefFFTBlock(D, 2, 1024, [9, 1], 0.1).eval()
    p = "decoder_layers.1."
    blk.load_state_dict({k[len(p):]: v for k, v in sd_full.items() if k.startswith(p)})
    rng = np.random.Generator(np.random.PCG64(11))
    x = torch.from_numpy(rng.standard_normal((2, 37, D)).astype(np.float32))
    kpm = torch.zeros(2, 37, dtype=torch.bool)
    kpm[1, 29:] = True
    with torch.no_grad():
        y = blk(x, key_padding_mask=kpm)
        yo = O.fft_block(sd_full, p, x, 2, [9, 1], kpm)
    print(f"[fftblock] oracle-vs-ref {maxdiff(y, yo):.3g}")
    # duration predictor
    dp = RefDurationPredictor(D, 256, 3, 0.5).eval()
    p2 = "duration_predictor."
    dp.load_state_dict({k[len(p2):]: v for k, v in sd_full.items() if k.startswith(p2)})
    with torch.no_grad():
        ld = dp(x, kpm)
        ldo = O.duration_predictor(sd_full, x, kpm, 3)
    print(f"[durpred] oracle-vs-ref {maxdiff(ld, ldo):.3g}")
    # length regulator, adversarial durations: zeros, an all-zero row, one long token
    seq = torch.from_numpy(rng.standard_normal((4, 6, 8)).astype(np.float32))
    dur = torch.tensor([[0, 2, 0, 1, 3, 0], [0, 0, 0, 0, 0, 0], [9, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1]])
    with torch.no_grad():
        ex, tm = ref_length_regulator(seq, dur)
        exo, tmo, _ = O.length_regulator(seq, dur)
    print(f"[lenreg] oracle-vs-ref {maxdiff(ex, exo):.3g} mask_eq {bool((tm == tmo).all())}")
    np.savez_compressed(os.path.join(GOLD, "tte_blocks.npz"), digest=np.array(synth.state_digest(sd_full)),
                        x=x.numpy(), kpm=kpm.numpy(), fft_out=y.numpy(), log_dur=ld.numpy(),
                        lr_seq=seq.numpy(), lr_dur=dur.numpy(), lr_out=ex.numpy(), lr_mask=tm.numpy())


def build_ref_codegen(h, sd, remove_wn):
    g = ref_voc_models.CodeGenerator(ref_voc_utils.AttrDict(h))
    g.load_state_dict(sd)
    g.eval()
    if remove_wn:
        g.remove_weight_norm()
    return g


def voc_case(name, h, B, U, seed_w, seed_in, scale, with_stages=False, fp64=False):
    sd = synth.synth_voc_state_dict(h, seed=seed_w, scale=scale)
    batch = synth.synth_voc_batch(B, U, h, seed=seed_in)
    g_wn = build_ref_codegen(h, sd, remove_wn=False)
    stages = {}
    hooks = []
    if with_stages:
        hooks.append(g_wn.conv_pre.register_forward_hook(lambda m, i, o: stages.__setitem__("conv_pre", o.detach().clone())))
        for i, up in enumerate(g_wn.ups):
            hooks.append(up.register_forward_hook(lambda m, inp, o, i=i: stages.__setitem__(f"ups{i}", o.detach().clone())))
        # MRF output of stage i == input of ups[i+1] before leaky_relu / input of final leaky_relu:
        # capture through forward-pre hooks is awkward (functional lrelu) -> recompute from resblocks
    with torch.no_grad():
        y_wn = g_wn(**batch)
        if with_stages:
            nk = len(h["resblock_kernel_sizes"])
            for i in range(len(h["upsample_rates"])):
                xs = None
                for j in range(nk):
                    r = g_wn.resblocks[i * nk + j](stages[f"ups{i}"])
                    xs = r if xs is None else xs + r
                stages[f"mrf{i}"] = xs / nk
        g_plain = build_ref_codegen(h, sd, remove_wn=True)
        y_plain = g_plain(**batch)
        ost = {}
        y_o = O.code_generator_forward(sd, h, batch["code"], batch["spkr"], stages=ost)
    for hk in hooks:
        hk.remove()
    msg = f"[{name}] |y|max {float(y_wn.abs().max()):.3f} rms {float(y_wn.pow(2).mean().sqrt()):.3f} " \
          f"wn-vs-plain {maxdiff(y_wn, y_plain):.3g} oracle-vs-ref {maxdiff(y_o, y_wn):.3g}"
    extra = {}
    if with_stages:
        msg += " stages " + " ".join(f"{k}:{maxdiff(ost[k], v):.2g}" for k, v in stages.items())
        extra = {"stage_" + k: v.numpy() for k, v in stages.items()}
    if fp64:
        g64 = build_ref_codegen(h, sd, remove_wn=True).double()
        with torch.no_grad():
            y64 = g64(**batch)
        msg += f" fp32-vs-fp64 {maxdiff(y_wn, y64):.3g}"
        extra["wav_fp64"] = y64.float().numpy()
    print(msg)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), digest=np.array(synth.state_digest(sd)),
                        meta=np.array(json.dumps(dict(B=B, U=U, seed_w=seed_w, seed_in=seed_in, scale=scale))),
                        code=batch["code"].numpy(), spkr=batch["spkr"].numpy(), wav=y_wn.numpy(),
                        wav_int16=O.to_int16(y_wn.squeeze(1)), **extra)


def voc_feats_case(name="voc_small_feats"):
    """CodeGenerator.forward with extra conditioning keywords (models.py:162-167): a (B,2,U/2) stream and a (B,1) global
    value are upsampled and concatenated after the unit / speaker embeddings; `f0` is passed and must be ignored."""
    h = synth.clone_config(synth.small_voc_config())
    h["model_in_dim"] = h["model_in_dim"] + 3
    B, U = 2, 20
    sd = synth.synth_voc_state_dict(h, seed=13)
    batch = synth.synth_voc_batch(B, U, h, seed=8)
    g = torch.Generator().manual_seed(99)
    energy = torch.randn(B, 2, U // 2, generator=g)
    style = torch.randn(B, 1, generator=g)
    f0 = torch.randn(B, 1, U, generator=g)
    ref = build_ref_codegen(h, sd, remove_wn=False)
    with torch.no_grad():
        y = ref(code=batch["code"], spkr=batch["spkr"], f0=f0, energy=energy, style=style)
        y_o = O.code_generator_forward(sd, h, batch["code"], batch["spkr"], feats={"f0": f0, "energy": energy, "style": style})
    print(f"[{name}] |y|max {float(y.abs().max()):.3f} oracle-vs-ref {maxdiff(y_o, y):.3g}")
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), digest=np.array(synth.state_digest(sd)),
                        meta=np.array(json.dumps(dict(B=B, U=U, seed_w=13, seed_in=8, extra_channels=3))),
                        code=batch["code"].numpy(), spkr=batch["spkr"].numpy(), energy=energy.numpy(), style=style.numpy(),
                        f0=f0.numpy(), wav=y.numpy())


def main():
    if "--only-feats" in sys.argv:  # added later: leaves the other (bit-pinned) goldens untouched
        return voc_feats_case()
    if "--only-corners" in sys.argv:  # round 3
        return voc_case("voc_small_corners", synth.corner_voc_config(), B=3, U=11, seed_w=13, seed_in=8, scale=1.0, with_stages=True)
    full_t, small_t = synth.default_tte_config(), synth.small_tte_config()
    # 1. TTE-small: ragged pads, 2 speakers (full-size model)
    tte_case("tte_full_ragged", full_t, vocab=60, n_spk=2, B=3, S=23, seed_w=0, seed_in=1, ragged=True)
    # 2. TTE bench shape subset: forced durations 4 -> L=256
    tte_case("tte_full_forced", full_t, vocab=300, n_spk=10, B=4, S=64, seed_w=42, seed_in=0, ragged=False, forced=4)
    # single-speaker (no speaker_emb key) + reduced model: full logits kept
    tte_case("tte_small_ragged", small_t, vocab=40, n_spk=1, B=4, S=17, seed_w=3, seed_in=4, ragged=True, keep_logits_rows=4)
    tte_case("tte_small_multi", small_t, vocab=40, n_spk=3, B=2, S=12, seed_w=5, seed_in=6, ragged=True, keep_logits_rows=2)
    # 3. per-block
    block_cases()
    # 4. vocoder
    full_v, small_v = synth.default_voc_config(), synth.small_voc_config()
    voc_case("voc_full_stages", full_v, B=1, U=12, seed_w=1234, seed_in=0, scale=1.0, with_stages=True)
    voc_case("voc_full_u40", full_v, B=2, U=40, seed_w=1234, seed_in=2, scale=1.0)
    voc_case("voc_full_u256", full_v, B=1, U=256, seed_w=1234, seed_in=3, scale=1.0, fp64=True)
    voc_case("voc_full_u40_hot", full_v, B=2, U=40, seed_w=1234, seed_in=2, scale=1.2)
    voc_case("voc_small", small_v, B=3, U=25, seed_w=9, seed_in=5, scale=1.0, with_stages=True)
    single = synth.clone_config(small_v)
    single["multispkr"] = None
    single["model_in_dim"] = single["embedding_dim"]
    voc_case("voc_small_singlespk", single, B=2, U=9, seed_w=10, seed_in=6, scale=1.0)
    rb2 = synth.clone_config(small_v)
    rb2["resblock"] = "2"
    rb2["resblock_dilation_sizes"] = [[1, 3], [1, 3], [1, 3]]
    voc_case("voc_small_resblock2", rb2, B=2, U=9, seed_w=11, seed_in=7, scale=1.0)
    voc_feats_case()


if __name__ == "__main__":
    main()
