"""CPU: the driver counterparts' host logic (parrot_tts_amd/cli/*, reference inference.py:25-72 and
utils/vocoder/inference.py:112-175): manifest / code-file parsing, CodeDataset trimming and speaker table, wav I/O helpers,
argument compatibility -- and that without a GPU the drivers fail loudly instead of falling back."""
import json
import os

import numpy as np
import pytest
import torch
from scipy.io import wavfile

from parrot_tts_amd import checkpoint, data, synth
from parrot_tts_amd.cli import tte_infer, voc_infer
from parrot_tts_amd.vocoder import AttrDict


def _manifest(tmp_path, with_wav=("bho_f_0001",), n_units=(30, 12, 7)):
    names = ["bho_f_0001", "en_m_0002", "kn_f_0003"]
    lines = []
    rng = np.random.Generator(np.random.PCG64(1))
    for nm, n in zip(names, n_units):
        wav = tmp_path / "wavs" / (nm + ".wav")
        if nm in with_wav:
            wav.parent.mkdir(exist_ok=True)
            wavfile.write(str(wav), 16000, (rng.standard_normal(320 * 20 + 57) * 3000).astype(np.int16))  # 20 units + a tail
        lines.append(data.format_dict_line({"audio": str(wav), "hubert": " ".join(str(int(v)) for v in rng.integers(0, 100, n)), "duration": 0.1}))
    p = tmp_path / "predictions.txt"
    p.write_text("".join(lines))
    return p, names


def test_code_dataset_trims_to_ground_truth_audio_and_builds_the_sorted_speaker_table(tmp_path):
    p, names = _manifest(tmp_path)
    ds = data.CodeDataset(data.parse_manifest(p), -1, 320, multispkr="_")
    assert len(ds) == 3 and ds.id_to_spkr == ["bho_f", "en_m", "kn_f"] and ds.spkr_to_id["kn_f"] == 2  # dataset.py:171-178
    feats, audio, filename, mel = ds[0]
    assert feats["code"].shape == (20,) and feats["spkr"].tolist() == [0] and mel is None   # min(len(audio)//320, 30) = 20
    assert audio.shape == (1, 20 * 320) and abs(float(audio.abs().max()) - 0.95) < 1e-6       # normalize(audio) * 0.95
    feats, audio, filename, _ = ds[1]
    assert feats["code"].shape == (12,) and audio is None and filename.endswith("en_m_0002.wav")  # no wav: untrimmed
    with pytest.raises(NotImplementedError):
        data.CodeDataset(data.parse_manifest(p), 8960, 320)
    with pytest.raises(NotImplementedError):
        data.mel_spectrogram(torch.zeros(1, 100), 1024, 80, 16000, 256, 1024, 0, 8000)
    audio, sr = data.load_wav_int16_scale(tmp_path / "wavs" / "bho_f_0001.wav")
    assert sr == 16000 and audio.dtype == np.float64 and np.abs(audio).max() > 100


def test_voc_infer_arguments_and_code_file(tmp_path):
    cf = tmp_path / "codes.txt"
    cf.write_text("utt_a|1 2 3 4\nutt_b|9 8\n\n")
    ns = type("A", (), dict(code_file=str(cf), input_code_file=None, pad=None))
    items = voc_infer.build_dataset(ns, AttrDict(synth.small_voc_config()))
    assert [i[2] for i in items] == ["utt_a", "utt_b"] and items[0][0]["code"].tolist() == [1, 2, 3, 4] and items[1][1] is None
    h = synth.small_voc_config()
    cfg = tmp_path / "config.json"
    cfg.write_text(json.dumps(h))
    torch.save({"generator": synth.synth_voc_state_dict(h, seed=3)}, tmp_path / "g_00000007")
    torch.save({"generator": synth.synth_voc_state_dict(h, seed=4)}, tmp_path / "g_latest")
    assert checkpoint.scan_checkpoint(str(tmp_path), "g_").endswith("g_00000007")            # utils.py:62-67 rule
    assert checkpoint.scan_checkpoint(str(tmp_path), "g_", "*").endswith("g_latest")         # the driver's own rule (inference.py:57-62)
    if not torch.cuda.is_available():
        # every reference flag parses; without a GPU the driver dies loudly (no CPU fallback)
        with pytest.raises(Exception):
            voc_infer.main(["--checkpoint_file", str(tmp_path), "--config", str(cfg), "--code_file", str(cf), "--output_dir", str(tmp_path / "o"),
                            "--vc", "--parts", "--pad", "320", "--debug", "--random-speakers", "-n", "1"])
    with pytest.raises(SystemExit):
        voc_infer.main(["--output_dir", "x"])  # --checkpoint_file is required, as in the reference


def test_tte_infer_duration_field(tmp_path):
    wav = tmp_path / "a.wav"
    wavfile.write(str(wav), 16000, np.zeros(24000, dtype=np.int16))
    assert tte_infer.wav_seconds(str(wav), 9.0) == 1.5          # librosa.get_duration of the 16 kHz file (inference.py:62-63)
    assert tte_infer.wav_seconds(str(tmp_path / "missing.wav"), 2.5) == 2.5
    with pytest.raises(SystemExit):
        tte_infer.main([])


def test_plan_batches_covers_every_row_once_within_the_limits():
    from parrot_tts_amd.cli.voc_infer import plan_batches
    import random
    rnd = random.Random(3)
    for _ in range(50):
        lens = [rnd.randint(1, 400) for _ in range(rnd.randint(1, 90))]
        mr, mu = rnd.randint(1, 20), rnd.randint(50, 3000)
        plan = plan_batches(lens, mr, mu)
        assert sorted(i for b in plan for i in b) == list(range(len(lens)))
        for b in plan:
            longest = max(lens[i] for i in b)
            assert len(b) <= mr and (len(b) == 1 or len(b) * longest <= mu)
            assert lens[b[0]] == longest  # longest first: the first row sizes the padded batch
    assert plan_batches([], 4, 100) == []
