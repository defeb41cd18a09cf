"""Host-side input formats either side of the hot path (reference modules/data.py, utils/vocoder/dataset.py):
the TTE's val.txt / predictions.txt line format, the DFA tokenizer, batch collation and masks, the
vocoder manifest and the speaker-from-filename rule.  Pure Python/torch-CPU plumbing: no arithmetic of
the synthesis path happens here."""
from __future__ import annotations

import ast
import json
import pickle
from pathlib import Path
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

PAD, SEP = "<pad>", "<sep>"
# utils/vocoder/inference.py:159 -- the fixed speaker table of the released multi-speaker vocoder
VOCODER_SPEAKERS = {"bho_f": 0, "bho_m": 1, "en_f": 2, "en_m": 3, "gu_f": 4, "gu_m": 5, "hi_f": 6, "hi_m": 7, "kn_f": 8, "kn_m": 9}


def get_mask_from_lengths(lengths: Sequence[int], max_len: int = None, device=None) -> torch.Tensor:
    """reference modules/data.py:8-20: True where ``position <= length`` -- note ``<=`` (quirk Q2)."""
    ln = torch.as_tensor(list(lengths))
    if max_len is None:
        max_len = int(ln.max())
    pos = torch.arange(max_len)
    if device is not None:
        ln, pos = ln.to(device), pos.to(device)
    return pos[None, :] <= ln[:, None]


def get_mask_from_batch(batch: torch.Tensor, pad_idx: int) -> torch.Tensor:
    """reference modules/data.py:22-23"""
    return batch != pad_idx


class DFATokenizer:
    """reference modules/data.py:28-61: symbols.pkl (list, or dict whose keys are the symbols) ->
    ['<pad>', '<sep>'] + symbols, with the blank ' ' renamed to 'sil'."""
    pad, sep = PAD, SEP

    def __init__(self, alignment_path):
        with open(Path(alignment_path) / "symbols.pkl", "rb") as f:
            loaded = pickle.load(f)
        if isinstance(loaded, dict):
            loaded = list(loaded.keys())
        elif not isinstance(loaded, list):
            raise TypeError(f"symbols.pkl holds a {type(loaded).__name__}; expected list or dict")
        self.symbols = [PAD, SEP] + self._rename_first_blank(loaded)
        self.stoi = {s: i for i, s in enumerate(self.symbols)}
        self.itos = dict(enumerate(self.symbols))
        self.pad_idx, self.sep_idx = self.stoi[PAD], self.stoi[SEP]

    @staticmethod
    def _rename_first_blank(symbols: List[str]) -> List[str]:
        # the reference replaces only the FIRST ' ' (list.index); later blanks stay as they are
        out = list(symbols)
        if " " in out:
            out[out.index(" ")] = "sil"
        return out

    def __len__(self):
        return len(self.symbols)

    def tokenize(self, phoneme_seq: Sequence[str]) -> List[int]:
        return [self.stoi[s] for s in phoneme_seq]


def parse_dict_line(line: str) -> dict:
    """One record of train.txt / val.txt / predictions.txt: a python-dict repr per line
    (written by reference inference.py:70-72; read as JSON after quote replacement at modules/data.py:76-77
    and with eval at utils/vocoder/dataset.py:113).  ast.literal_eval accepts both spellings safely."""
    return ast.literal_eval(line.strip())


def format_dict_line(d: dict) -> str:
    """``str(dict) + '\\n'`` exactly as reference inference.py:72 writes predictions.txt."""
    return str(d) + "\n"


class ParrotDataset(torch.utils.data.Dataset):
    """reference modules/data.py:63-120 (same files, same item / batch dictionaries)."""

    def __init__(self, split: str, data_config: dict):
        self.root_dir = Path(data_config["path"]["root_path"])
        self.tokenizer = DFATokenizer(Path(data_config["path"]["alignment_path"]))
        self.src_vocab_size, self.src_pad_idx = len(self.tokenizer), self.tokenizer.pad_idx
        self.code_pad_idx = data_config["preprocess"]["hubert_codes"]
        with open(self.root_dir / f"{split}.txt") as f:
            self.data_list = [parse_dict_line(l) for l in f if l.strip()]
        with open(self.root_dir / "speakers.json") as f:
            self.speaker_map = json.load(f)

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, idx) -> dict:
        d = self.data_list[idx]
        return {"id": Path(d["audio"]).stem, "speaker": self.speaker_map[d["speaker"]],
                "phones": self.tokenizer.tokenize(d["characters"].split(" ")),
                "codes": [int(i) for i in d["hubert"].split(" ")], "duration": [int(i) for i in d["duration"].split(" ")]}

    def collate_fn(self, items: List[dict]) -> Dict[str, torch.Tensor]:
        pad = torch.nn.utils.rnn.pad_sequence
        as_long = lambda key: [torch.tensor(d[key], dtype=torch.long) for d in items]  # noqa: E731
        out = {"ids": [d["id"] for d in items], "speaker": torch.tensor([d["speaker"] for d in items], dtype=torch.long)}
        out["phones"] = pad(as_long("phones"), batch_first=True, padding_value=self.src_pad_idx)
        out["codes"] = pad(as_long("codes"), batch_first=True, padding_value=self.code_pad_idx)
        out["duration"] = pad(as_long("duration"), batch_first=True)
        out["src_mask"] = get_mask_from_batch(out["phones"], self.src_pad_idx)
        out["tgt_mask"] = get_mask_from_batch(out["codes"], self.code_pad_idx)
        return out


def parse_manifest(manifest) -> Tuple[List[Path], List[np.ndarray]]:
    """reference utils/vocoder/dataset.py:107-123: dict lines carry 'audio' + 'hubert' (space separated units);
    any other line is a bare audio path."""
    audio_files, codes = [], []
    with open(manifest) as f:
        for line in f:
            if not line.strip():
                continue
            if line[0] == "{":
                rec = parse_dict_line(line)
                codes.append(np.asarray([int(x) for x in rec["hubert"].split(" ")], dtype=np.int64))
                audio_files.append(Path(rec["audio"]))
            else:
                audio_files.append(Path(line.strip()))
    return audio_files, codes


def parse_speaker(path, method: str) -> str:
    """reference utils/vocoder/dataset.py:133-142"""
    path = Path(path)
    if method == "_":
        return "_".join(path.name.split("_")[:2])
    if method == "single":
        return "A"
    raise NotImplementedError(method)


def peak_normalize(audio: np.ndarray) -> np.ndarray:
    """librosa.util.normalize(x) with its defaults (norm=inf, axis=0) as used at
    utils/vocoder/inference.py:169: divide by max|x| unless that is below the dtype's tiny threshold."""
    mag = np.abs(audio).max() if audio.size else 0.0
    tiny = np.finfo(audio.dtype if np.issubdtype(audio.dtype, np.floating) else np.float32).tiny
    return audio if mag < tiny else audio / mag


def load_wav_int16_scale(path) -> Tuple[np.ndarray, int]:
    """A mono wav as floats on the int16 scale, like reference utils/vocoder/dataset.py:66-78 (soundfile int16 read;
    multi-channel files are averaged).  scipy reads PCM16 natively; float files are rescaled to that range."""
    from scipy.io import wavfile
    sr, data = wavfile.read(str(path))
    if data.dtype.kind == "f":
        data = data * 32768.0
    elif data.dtype != np.int16:
        data = data.astype(np.float64) / float(np.iinfo(data.dtype).max) * 32768.0
    data = np.asarray(data, dtype=np.float64)
    if data.ndim == 2:
        data = data.mean(axis=1)
    return data, int(sr)


class CodeDataset(torch.utils.data.Dataset):
    """Inference-side counterpart of reference utils/vocoder/dataset.py:146-251 as ``init_worker`` builds it
    (utils/vocoder/inference.py:122-127: segment_size = -1, i.e. whole utterances): ``(audio_files, codes)`` from
    ``parse_manifest``, the speaker table derived from the SORTED speaker names of the manifest (dataset.py:171-178),
    and per item ``(feats, gt_audio, filename, None)`` with ``feats = {'code': (U,) int64[, 'spkr': (1,) int64]}``.

    Like the reference, the units are trimmed to the ground-truth audio (``min(len(audio) // code_hop_size, len(code))``,
    dataset.py:226-229) WHEN that audio file exists; when it does not (synthesis from predicted units only) the item keeps
    all its units and ``gt_audio`` is None, where the reference would fail in ``load_audio``.  Training-only parts
    (random segment sampling, the mel target, f0) are not built: the 4th element is None."""

    def __init__(self, training_files, segment_size=-1, code_hop_size=320, *unused, sampling_rate=16000, multispkr=False, pad=None, **kw):
        self.audio_files, self.codes = training_files
        if segment_size not in (-1, None):
            raise NotImplementedError("parrot_tts_amd.CodeDataset covers inference (segment_size = -1) only")
        self.code_hop_size, self.sampling_rate = int(code_hop_size), sampling_rate
        self.multispkr, self.pad = multispkr, pad
        if self.multispkr:
            self.id_to_spkr = sorted({parse_speaker(f, self.multispkr) for f in self.audio_files})
            self.spkr_to_id = {k: v for v, k in enumerate(self.id_to_spkr)}

    def __len__(self):
        return len(self.audio_files)

    def _get_spkr(self, idx) -> np.ndarray:
        return np.asarray([self.spkr_to_id[parse_speaker(self.audio_files[idx], self.multispkr)]], dtype=np.int64)

    def __getitem__(self, index):
        filename = self.audio_files[index]
        code, audio = np.asarray(self.codes[index], dtype=np.int64), None
        if Path(filename).is_file():
            audio, _ = load_wav_int16_scale(filename)
            if self.pad:
                audio = np.pad(audio, (0, self.pad - (audio.shape[-1] % self.pad)), "constant", constant_values=0)
            audio = peak_normalize(audio / 32768.0) * 0.95          # dataset.py:217-218
            n = min(audio.shape[0] // self.code_hop_size, code.shape[0])
            code, audio = code[:n], torch.from_numpy(audio[: n * self.code_hop_size].astype(np.float32)).unsqueeze(0)
        feats = {"code": code}
        if self.multispkr:
            feats["spkr"] = self._get_spkr(index)
        return feats, audio, str(filename), None


def mel_spectrogram(*args, **kwargs):
    """reference utils/vocoder/dataset.py:44-64 is a training-loss component (and the ``_gt`` mel of the driver): out of scope."""
    raise NotImplementedError("parrot_tts_amd covers the synthesis path only; mel_spectrogram is a training component")
