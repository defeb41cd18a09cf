"""CPU: host-side logic around the hot path -- data formats, masks, checkpoint layouts, drop-in import
names, batch sharding -- and that the product path refuses to run without the GPU (no CPU fallback)."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import parrot_oracle as O
from parrot_tts_amd import checkpoint, data, dist as pdist, synth
from parrot_tts_amd.tte import Parrot, parrot_param_shapes
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator, get_padding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tte_dir(tmp_path, n_spk=2):
    d = tmp_path / "tte"
    d.mkdir()
    (d / "speakers.json").write_text(json.dumps({f"bho_{'fm'[i % 2]}{i}": i for i in range(n_spk)}))
    cfg = synth.small_tte_config(str(d))
    cfg["path"]["alignment_path"] = str(d)
    return d, cfg


def test_masks_match_oracle_quirk_q2():
    lens = [0, 3, 7, 7]
    assert torch.equal(data.get_mask_from_lengths(lens, 7), O.get_mask_from_lengths(lens, 7))
    m = data.get_mask_from_lengths(lens, 7)
    assert m.sum(1).tolist() == [1, 4, 7, 7]  # len+1 True per short row, capped at max_len


def test_tokenizer_dataset_collate(tmp_path):
    d, cfg = _tte_dir(tmp_path)
    with open(d / "symbols.pkl", "wb") as f:
        pickle.dump(["a", " ", "b", "c", " "], f)
    tok = data.DFATokenizer(d)
    assert tok.symbols == ["<pad>", "<sep>", "a", "sil", "b", "c", " "] and tok.pad_idx == 0 and tok.sep_idx == 1
    recs = [{"audio": "/x/bho_f0_001.wav", "speaker": "bho_f0", "characters": "a sil b", "hubert": "5 6 7 8", "duration": "1 2 1"},
            {"audio": "/x/bho_m1_002.wav", "speaker": "bho_m1", "characters": "c a", "hubert": "1 2", "duration": "1 1"}]
    (d / "val.txt").write_text("".join(data.format_dict_line(r) for r in recs))
    ds = data.ParrotDataset("val", cfg)
    assert len(ds) == 2 and ds.src_vocab_size == 7 and ds.code_pad_idx == cfg["preprocess"]["hubert_codes"]
    b = ds.collate_fn([ds[0], ds[1]])
    assert b["phones"].tolist() == [[2, 3, 4], [5, 2, 0]]
    assert b["src_mask"].tolist() == [[True, True, True], [True, True, False]]
    assert b["speaker"].tolist() == [0, 1] and b["ids"] == ["bho_f0_001", "bho_m1_002"]
    assert b["codes"][1].tolist() == [1, 2, 100, 100] and b["tgt_mask"][1].tolist() == [True, True, False, False]
    with open(d / "symbols.pkl", "wb") as f:
        pickle.dump({"x": 0, "y": 1}, f)
    assert data.DFATokenizer(d).symbols == ["<pad>", "<sep>", "x", "y"]


def test_prediction_line_format_and_manifest(tmp_path):
    rec = {"audio": "/w/en_f_12.wav", "hubert": "3 1 4 1 5", "duration": 0.1}
    line = data.format_dict_line(rec)
    assert line == "{'audio': '/w/en_f_12.wav', 'hubert': '3 1 4 1 5', 'duration': 0.1}\n"  # str(dict), inference.py:72
    assert data.parse_dict_line(line) == rec
    assert data.parse_dict_line(line.replace("'", '"')) == rec  # the JSON spelling modules/data.py:76 produces
    p = tmp_path / "pred.txt"
    p.write_text(line + "/plain/path.wav\n")
    files, codes = data.parse_manifest(p)
    assert [str(f) for f in files] == ["/w/en_f_12.wav", "/plain/path.wav"] and codes[0].tolist() == [3, 1, 4, 1, 5]
    assert data.parse_speaker(files[0], "_") == "en_f" and data.parse_speaker(files[0], "single") == "A"
    assert data.VOCODER_SPEAKERS["en_f"] == 2 and len(data.VOCODER_SPEAKERS) == 10
    x = np.array([0.0, -4.0, 2.0], dtype=np.float32)
    assert np.array_equal(data.peak_normalize(x), np.array([0.0, -1.0, 0.5], dtype=np.float32))
    assert np.array_equal(data.peak_normalize(np.zeros(3, np.float32)), np.zeros(3, np.float32))
    assert get_padding(11, 5) == 25 and get_padding(8, 1) == 3  # int((k*d - d)/2)


def test_parrot_state_dict_layout_and_lightning_checkpoint(tmp_path):
    d, cfg = _tte_dir(tmp_path, n_spk=3)
    sd = synth.synth_tte_state_dict(cfg, 40, 3, seed=5)
    m = Parrot(cfg, 40, 0)
    assert set(m.state_dict()) == set(sd) and all(m.state_dict()[k].shape == v.shape for k, v in sd.items())
    assert set(parrot_param_shapes(cfg, 40, 3)) | {"pos_emb.pe"} == set(sd)
    single = Parrot(synth.small_tte_config(str(_one_speaker_dir(tmp_path))), 40, 0)
    assert "speaker_emb.weight" not in single.state_dict()  # only when speakers.json has > 1 entry (parrot.py:28-32)
    ck = tmp_path / "parrot_model-step=1.ckpt"
    checkpoint.save_lightning_style(ck, sd, cfg, 40, 0)
    lit = checkpoint.LitParrot.load_from_checkpoint(str(ck), weights_only=True)
    assert all(torch.equal(lit.parrot.state_dict()[k], v) for k, v in sd.items())
    assert all(k.startswith("parrot.") for k in lit.state_dict())
    with pytest.raises(RuntimeError):  # strict load: a generator checkpoint is not a TTE checkpoint
        torch.save({"state_dict": {"parrot.bogus": torch.zeros(1)}, "hyper_parameters": dict(lit.hparams)}, ck)
        checkpoint.LitParrot.load_from_checkpoint(str(ck))


def _one_speaker_dir(tmp_path):
    d = tmp_path / "one"
    d.mkdir(exist_ok=True)
    (d / "speakers.json").write_text(json.dumps({"A": 0}))
    return d


def test_generator_state_dict_layouts(tmp_path):
    h = synth.small_voc_config()
    sd = synth.synth_voc_state_dict(h, seed=3)
    g = CodeGenerator(AttrDict(h))
    assert set(g.state_dict()) == set(sd)
    g.load_state_dict(sd)
    folded = O.fold_weight_norm(sd)
    for name, layer in [("conv_pre", g.conv_pre), ("ups.2", g.ups[2]), ("resblocks.7.convs2.1", g.resblocks[7].convs2[1]), ("conv_post", g.conv_post)]:
        assert torch.equal(layer.folded(), folded[name + ".weight"]), name  # same fold as torch.nn.utils.weight_norm
    g.remove_weight_norm()
    assert set(g.state_dict()) == set(folded)
    assert all(torch.equal(g.state_dict()[k], folded[k]) for k in folded)
    with pytest.raises(ValueError):
        g.remove_weight_norm()
    g.load_state_dict(sd)          # a weight_g / weight_v checkpoint loads into a folded module ...
    assert set(g.state_dict()) == set(sd)
    g.load_state_dict(folded)      # ... and the other way round
    assert set(g.state_dict()) == set(folded)
    torch.save({"generator": sd}, tmp_path / "g_00000010")
    torch.save({"generator": sd}, tmp_path / "g_00000002")
    assert checkpoint.scan_checkpoint(str(tmp_path), "g_").endswith("g_00000010")
    assert CodeGenerator(AttrDict(dict(h, f0=True))).f0 is True  # stored and, like the reference's forward, never used


def test_weight_reload_through_a_wrapper_invalidates_the_cached_handle(tmp_path):
    """torch's load_state_dict never calls a CHILD module's load_state_dict override, so the packed-weight handle must be
    invalidated from hooks torch always runs + a parameter-version fingerprint: reload through LitParrot /
    nn.Sequential, in-place parameter writes and .to() all change what the next forward must pack."""
    from parrot_tts_amd.ops import param_fingerprint
    d, cfg = _tte_dir(tmp_path)
    lit = checkpoint.LitParrot(cfg, 30, 0)
    calls = []
    lit.parrot._invalidate = lambda: calls.append("tte")
    fp0 = param_fingerprint(lit.parrot)
    lit.load_state_dict({"parrot." + k: v for k, v in synth.synth_tte_state_dict(cfg, 30, 2, seed=9).items()})
    assert calls == ["tte"] and param_fingerprint(lit.parrot) != fp0
    fp1 = param_fingerprint(lit.parrot)
    with torch.no_grad():
        lit.parrot.state_dict()["head.bias"].add_(1.0)  # in-place edit of one parameter
    assert param_fingerprint(lit.parrot) != fp1
    h = synth.small_voc_config()
    g = CodeGenerator(AttrDict(h))
    seq = torch.nn.Sequential(g)
    g._invalidate = lambda: calls.append("voc")
    fp0 = param_fingerprint(g)
    seq.load_state_dict({"0." + k: v for k, v in synth.synth_voc_state_dict(h, seed=4).items()})
    assert calls[-1] == "voc" and param_fingerprint(g) != fp0
    fp1 = param_fingerprint(g)
    g.conv_pre.remove_weight_norm()  # parameter objects replaced
    assert param_fingerprint(g) != fp1


def test_no_cpu_fallback_and_oracle_is_not_imported_by_the_product(tmp_path):
    d, cfg = _tte_dir(tmp_path)
    m = Parrot(cfg, 30, 0).eval()
    batch = synth.synth_tte_batch(1, 4, 30, 2, seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.infer(batch)
    g = CodeGenerator(AttrDict(synth.small_voc_config())).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g(code=torch.zeros(1, 3, dtype=torch.int64), spkr=torch.zeros(1, 1, dtype=torch.int64))
    m.train()
    with pytest.raises(AssertionError):
        m.infer(batch)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "parrot_tts_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "parrot_oracle" not in src, f


def test_dropin_import_names():
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from modules import ParrotDataset, Parrot, ModelLoss;"
            "from utils import AttrDict; from utils.vocoder.models import CodeGenerator;"
            "from utils.vocoder.dataset import MAX_WAV_VALUE, parse_manifest;"
            "sys.path.insert(0, %r); import importlib; [sys.modules.pop(k) for k in list(sys.modules) if k == 'utils' or k.startswith('utils.')];"
            "import models, utils, dataset; assert models.CodeGenerator is CodeGenerator and utils.AttrDict is AttrDict and dataset.MAX_WAV_VALUE == 32768.0;"
            "print('ok')") % (ROOT, os.path.join(ROOT, "parrot_tts_amd", "dropin"), os.path.join(ROOT, "parrot_tts_amd", "dropin", "utils", "vocoder"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_shard_rows_partition():
    for n in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            parts = [pdist.shard_rows(n, r, world) for r in range(world)]
            flat = [i for s in parts for i in range(s.start, s.stop)]
            assert flat == list(range(n))
            sizes = [s.stop - s.start for s in parts]
            assert max(sizes) - min(sizes) <= 1


def test_profile_kernel_names_fold_to_bench_rows():
    """tools/make_pmc_traffic.norm maps rocprofv3's instantiation names onto the rows bench.py reports (TILE_NAMES), so
    `roofline.traffic` finds the dominant kernel's PMC entry."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("make_pmc_traffic", os.path.join(root, "tools", "make_pmc_traffic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import bench
    cases = {"conv_split_kernel<parrot::SchF16x3, 2, 2, 2, 2, 2, 11, 1>": "conv_split_kernel<SchF16x3,2,2,2,2,2>",
             "conv_split_kernel<parrot::SchF16x3, 4, 1, 1, 2, 3, 1, 4, true>": "conv_split_kernel<SchF16x3,4,1,1,2,3>",
             "conv_split_kernel<parrot::SchF16x3, 1, 4, 1, 4, 2, 3, 1>": "conv_split_kernel<SchF16x3,1,4,1,4,2>",
             "resblock_split_kernel<parrot::SchF16x3, 4, 0, false>": "resblock_split_kernel<SchF16x3,4>",
             "resblock_split_kernel<parrot::SchF16x3, 16, 0, false>": "resblock_split_kernel<SchF16x3,16>",
             "resblock_split_kernel<parrot::SchF16x3, 2, 8, true>": "resblock_split_kernel<SchF16x3,2,8,MRF>",
             "conv_split16_kernel<parrot::SchF16x3, 2, 2, 4, 5, 11, 2>": "conv_split16_kernel<SchF16x3,2,2,4,5>",
             "resblock16_split_kernel<parrot::SchF16x3>": "resblock16_split_kernel<SchF16x3>",
             "conv1_valu_kernel<7>": "conv1_valu_kernel", "conv1_valu7_vec_kernel": "conv1_valu_kernel", "convt_valu_kernel<16, 4, 2, 1>": "convt_valu_kernel<16,4,2,1>",
             "conv_mfma_kernel<2, 2, 2, 2, 16, 3>": "conv_mfma_kernel<2,2,2,2,16,3>"}
    for raw, want in cases.items():
        assert mod.norm(raw) == want
        assert want in bench.tile_names("f16x3")


def test_resblock_dilation_lists_follow_the_reference_constructors():
    """ResBlock1 reads dilation[0..2], ResBlock2 dilation[0..1] -- literally (reference utils/vocoder/models.py:17-22,51-54):
    longer lists are cut, shorter ones raise IndexError, lists of different lengths per kernel size are accepted."""
    import pytest
    from parrot_tts_amd import synth
    from parrot_tts_amd.vocoder import AttrDict, CodeGenerator
    h = synth.corner_voc_config()
    g = CodeGenerator(AttrDict(h))
    keys = set(g.state_dict())
    assert set(synth.synth_voc_state_dict(h, seed=1)) == keys
    assert "resblocks.1.convs1.2.weight_v" in keys and "resblocks.1.convs1.3.weight_v" not in keys
    h["resblock_dilation_sizes"] = [[1, 3], [1, 3, 5], [1, 3, 5]]
    with pytest.raises(IndexError):
        CodeGenerator(AttrDict(h))
    h["resblock"] = "2"
    g2 = CodeGenerator(AttrDict(h))
    assert "resblocks.2.convs.1.bias" in g2.state_dict() and "resblocks.2.convs.2.bias" not in g2.state_dict()
    # out_samples: the ConvTranspose1d length chain (T u + 1 per odd stage), 0 for an empty row
    g3 = CodeGenerator(AttrDict(synth.corner_voc_config()))
    assert g3.out_samples(11) == ((11 * 4 + 1) * 2) * 2 + 1 and g3.out_samples(0) == 0 and g3.upsample_factor == 16
    import torch
    assert g3.out_samples(torch.tensor([0, 1, 11])).tolist() == [0, ((1 * 4 + 1) * 2) * 2 + 1, 181]
    assert CodeGenerator(AttrDict(synth.small_voc_config())).out_samples(7) == 7 * 320


def test_duration_predictor_kernel_other_than_3_fails_like_the_reference():
    """duration.py:30-34 hard-codes padding=1 for the second conv: with kernel_size != 3 its output has S - k + 3 frames and the
    reference's own `masked_fill(mask)` (duration.py:45-46, always reached from Parrot.forward) raises RuntimeError.  The oracle
    restates that; the HIP handle refuses such a config at create (a RuntimeError subclass) instead of inventing a result."""
    import pytest
    import torch
    from oracle import parrot_oracle as O
    from parrot_tts_amd import synth
    cfg = synth.small_tte_config()
    cfg["duration_predictor"]["kernel_size"] = 5
    sd = synth.synth_tte_state_dict(cfg, 20, 1, seed=2)
    batch = synth.synth_tte_batch(2, 9, 20, 1, seed=1, ragged=True)
    with pytest.raises(RuntimeError), torch.no_grad():
        O.tte_forward(sd, cfg, batch)
