"""GPU: the driver counterparts end to end on a temp tree -- `voc_infer --vc` writes ten WAVs per item whose PCM equals ten
B=1 `generate()` calls (reference utils/vocoder/inference.py:157-175), `tte_infer` writes predictions.txt in the
reference's line format with the units `Parrot.infer` / the oracle give (reference inference.py:25-72)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch
import yaml
from scipy.io import wavfile

pytestmark = pytest.mark.gpu

from oracle import parrot_oracle as O  # noqa: E402
from parrot_tts_amd import checkpoint, data, synth  # noqa: E402
from parrot_tts_amd.cli import tte_infer, voc_infer  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator, generate  # noqa: E402

DEV = "cuda:0"


def test_voc_infer_vc_writes_ten_wavs_equal_to_single_calls(tmp_path):
    h = synth.small_voc_config()
    h["sampling_rate"] = 16000
    cfg = tmp_path / "config.json"
    cfg.write_text(json.dumps(h))
    ck = tmp_path / "ck"
    ck.mkdir()
    vsd = synth.synth_voc_state_dict(h, seed=12)
    torch.save({"generator": vsd}, ck / "g_00000003")
    rng = np.random.Generator(np.random.PCG64(2))
    (tmp_path / "wavs").mkdir()
    gt = tmp_path / "wavs" / "hi_f_0001.wav"
    wavfile.write(str(gt), 16000, (rng.standard_normal(320 * 9 + 11) * 2000).astype(np.int16))  # 9 units of ground truth
    recs = [{"audio": str(gt), "hubert": " ".join(map(str, rng.integers(0, 100, 14))), "duration": 0.2},
            {"audio": str(tmp_path / "wavs" / "gu_m_0002.wav"), "hubert": " ".join(map(str, rng.integers(0, 100, 6))), "duration": 0.1}]
    man = tmp_path / "predictions.txt"
    man.write_text("".join(data.format_dict_line(r) for r in recs))
    out = tmp_path / "out"
    voc_infer.main(["--checkpoint_file", str(ck), "--config", str(cfg), "--input_code_file", str(man), "--output_dir", str(out), "--vc"])
    files = sorted(os.listdir(out))
    assert len(files) == 2 * 10 + 1 and "hi_f_0001_gt.wav" in files
    g = CodeGenerator(AttrDict(h))
    g.load_state_dict(vsd)
    g = g.eval().to(DEV)
    for rec, n_units in zip(recs, (9, 6)):  # item 0 is trimmed to its 9 units of audio (dataset.py:226-229)
        units = [int(v) for v in rec["hubert"].split(" ")][:n_units]
        code = torch.tensor([units], device=DEV)
        stem = os.path.splitext(os.path.basename(rec["audio"]))[0]
        for name, sid in data.VOCODER_SPEAKERS.items():
            sr, got = wavfile.read(str(out / f"{stem}_{name}_gen.wav"))
            pcm, _ = generate(h, g, {"code": code, "spkr": torch.tensor([[sid]], device=DEV)})   # reference inference.py:65-74
            want = data.peak_normalize(pcm.astype(np.float32))
            assert sr == 16000 and got.dtype == np.float32 and got.shape == (n_units * 320,)
            assert np.array_equal(got, want), (stem, name)
    sr, gtw = wavfile.read(str(out / "hi_f_0001_gt.wav"))
    assert gtw.shape == (9 * 320,) and abs(float(np.abs(gtw).max()) - 1.0) < 1e-6
    # without --vc: each item under its own speaker (extension; the reference writes nothing there)
    out2 = tmp_path / "out2"
    voc_infer.main(["--checkpoint_file", str(ck / "g_00000003"), "--config", str(cfg), "--input_code_file", str(man), "--output_dir", str(out2), "-n", "1"])
    assert sorted(os.listdir(out2)) == ["hi_f_0001_gt.wav", "hi_f_0001_hi_f_gen.wav"]


def test_voc_infer_batched_driver_writes_byte_identical_wavs(tmp_path):
    """The length-bucketed batched driver against the same driver fed one row per batch: 37 ragged items x
    (--vc) ten speakers through padded batches with per-row unit counts must give byte-identical WAV files."""
    h = synth.small_voc_config()
    h["sampling_rate"] = 16000
    cfg = tmp_path / "config.json"
    cfg.write_text(json.dumps(h))
    vsd = synth.synth_voc_state_dict(h, seed=31)
    torch.save({"generator": vsd}, tmp_path / "g_00000001")
    rng = np.random.Generator(np.random.PCG64(5))
    recs = [{"audio": str(tmp_path / "wavs" / f"hi_f_{i:04d}.wav"), "hubert": " ".join(map(str, rng.integers(0, 100, int(rng.integers(1, 60))))),
             "duration": 0.1} for i in range(37)]
    man = tmp_path / "predictions.txt"
    man.write_text("".join(data.format_dict_line(r) for r in recs))
    outs = {}
    for tag, extra in (("batched", ["--batch_rows", "48", "--batch_units", "1500"]), ("single", ["--batch_rows", "1"])):
        out = tmp_path / tag
        voc_infer.main(["--checkpoint_file", str(tmp_path / "g_00000001"), "--config", str(cfg), "--input_code_file", str(man),
                        "--output_dir", str(out), "--vc"] + extra)
        outs[tag] = out
    files = sorted(os.listdir(outs["batched"]))
    assert files == sorted(os.listdir(outs["single"])) and len(files) == 370
    for f in files:
        assert (outs["batched"] / f).read_bytes() == (outs["single"] / f).read_bytes(), f
    # batches really were padded and ragged
    lens = [len(r["hubert"].split(" ")) for r in recs for _ in range(10)]
    plan = voc_infer.plan_batches(lens, 48, 1500)
    assert sorted(i for b in plan for i in b) == list(range(370)) and max(len(b) for b in plan) > 10
    assert all(len(b) <= 48 and len(b) * max(lens[i] for i in b) <= 1500 for b in plan)


def test_tte_infer_writes_predictions_in_the_reference_format(tmp_path):
    root = tmp_path / "tte"
    root.mkdir()
    speakers = {"bho_f": 0, "en_m": 1}
    (root / "speakers.json").write_text(json.dumps(speakers))
    symbols = ["a", " ", "b", "c", "d"]
    with open(root / "symbols.pkl", "wb") as f:
        pickle.dump(symbols, f)
    cfg = synth.small_tte_config(str(root))
    cfg["path"]["alignment_path"] = str(root)
    cfg["path"]["wav_path"] = str(tmp_path / "audio")
    recs = [{"audio": "/x/bho_f_001.wav", "speaker": "bho_f", "characters": "a sil b c", "hubert": "1 2", "duration": "1 1 1 1"},
            {"audio": "/x/en_m_002.wav", "speaker": "en_m", "characters": "d a", "hubert": "3", "duration": "1 1"},
            {"audio": "/x/bho_f_003.wav", "speaker": "bho_f", "characters": "c c sil a b d", "hubert": "3", "duration": "1 1 1 1 1 1"}]
    (root / "val.txt").write_text("".join(data.format_dict_line(r) for r in recs))
    (tmp_path / "audio" / "en_m" / "wavs").mkdir(parents=True)
    wavfile.write(str(tmp_path / "audio" / "en_m" / "wavs" / "en_m_002.wav"), 16000, np.zeros(8000, dtype=np.int16))
    vocab = len(symbols) + 2
    sd = synth.synth_tte_state_dict(cfg, vocab, 2, seed=17)
    ck = tmp_path / "parrot.ckpt"
    checkpoint.save_lightning_style(ck, sd, cfg, vocab, 0)
    ycfg = tmp_path / "cfg.yaml"
    ycfg.write_text(yaml.safe_dump(cfg))
    tte_infer.main(["--config", str(ycfg), "--checkpoint_pth", str(ck), "--device", DEV])
    lines = (root / "predictions.txt").read_text().splitlines()
    assert len(lines) == 3
    ds = data.ParrotDataset("val", cfg)
    for i, line in enumerate(lines):
        rec = data.parse_dict_line(line)
        assert list(rec) == ["audio", "hubert", "duration"]                       # inference.py:64-69 key order
        stem = os.path.splitext(os.path.basename(recs[i]["audio"]))[0]
        assert rec["audio"] == os.path.join(cfg["path"]["wav_path"], recs[i]["speaker"], "wavs", stem + ".wav")
        batch = ds.collate_fn([ds[i]])                                            # batch_size 1, like the reference
        with torch.no_grad():
            want = O.tte_infer(sd, cfg, {"phones": batch["phones"], "src_mask": batch["src_mask"], "speaker": batch["speaker"]})[0]
        assert [int(v) for v in rec["hubert"].split(" ")] == want
        assert rec["duration"] == (0.5 if i == 1 else len(want) / 50.0)           # the wav's length when it exists
        assert line == str(rec)                                                   # str(dict), one per line
    files, codes = data.parse_manifest(root / "predictions.txt")                  # ... and the vocoder driver reads it back
    assert len(codes) == 3 and codes[0].dtype == np.int64
