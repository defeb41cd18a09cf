"""Text front-end of the demo path: raw text -> cleaned text -> character tokens -> the
TTE's `phones` batch, and the end-to-end text -> waveform helper of demo.ipynb.

Host-side string work only.  The cleaner for non-transliterated Indic text is a restatement of the reference's
dependency-free pipeline (utils/aligner/cleaners.py:204-233 with :169-202) and is pinned by golden vectors produced by
the reference's own functions (tools/make_text_goldens.py -> tests/golden/text_cleaners.json).  The English and the
transliterating pipelines of the reference need `unidecode` and `inflect`, which this image does not have: they are
provided only when those packages are importable and are NOT pinned here (stated, not silently approximated)."""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Optional, Sequence

import torch

# Hindi number words as spelled by the reference (utils/aligner/cleaners.py:39-45): units, tens and whole hundreds
_UNITS = ["शून्य", "एक", "दो", "तीन", "चार", "पाँच", "छह", "सात", "आठ", "नौ", "दस", "ग्यारह", "बारह", "तेरह", "चौदह", "पंद्रह", "सोलह",
          "सत्रह", "अठारह", "उन्नीस"]
_TENS = {20: "बीस", 30: "तीस", 40: "चालीस", 50: "पचास", 60: "साठ", 70: "सत्तर", 80: "अस्सी", 90: "नब्बे"}
_HUNDRED = "सौ"
_DEVANAGARI_DIGITS = "०१२३४५६७८९"
_DEV_NUMBER = re.compile("[" + _DEVANAGARI_DIGITS + "]+")
_SPACE_RUN = re.compile(r"\s+")
# characters the Indic pipeline deletes (utils/aligner/cleaners.py:212), ASCII digits (:208-209) and the stray
# control / C1 characters of its corpus (:224-230)
_DROP = set("—⇒'–…“”%+=[]^\\{}_`‘’@/->< ()*\":;!".replace(" ", "")) | set("0123456789") | set("\x92\xad\x10\x13\x14\x16\x91")


def collapse_whitespace(text: str) -> str:
    """Every run of whitespace becomes one blank (cleaners.py:97-98)."""
    return _SPACE_RUN.sub(" ", text)


def number_to_hindi(num: int) -> Optional[str]:
    """0..999 in the reference's wording (cleaners.py:169-184): tens and units are simply juxtaposed ("बीस एक"), hundreds
    are "<unit> सौ"; anything larger yields None, exactly as the reference falls through."""
    if num < 0 or num >= 1000:
        return None
    if num < 20:
        return _UNITS[num]
    if num < 100:
        tens, ones = num - num % 10, num % 10
        return _TENS[tens] if ones == 0 else _TENS[tens] + " " + _UNITS[ones]
    hundreds, rest = divmod(num, 100)
    head = _HUNDRED if hundreds == 1 else _UNITS[hundreds] + " " + _HUNDRED
    return head if rest == 0 else head + " " + number_to_hindi(rest)


def replace_devanagari_numbers(text: str) -> str:
    """Runs of Devanagari digits -> Hindi words (cleaners.py:186-202).  A number >= 1000 has no wording and the run is
    dropped (the reference's callback returns None there, which re.sub takes as an empty replacement)."""
    def words(m: "re.Match") -> str:
        value = int("".join(str(_DEVANAGARI_DIGITS.index(ch)) for ch in m.group(0)))
        return number_to_hindi(value) or ""
    return _DEV_NUMBER.sub(words, text)


def indic_cleaners(text: str) -> str:
    """`nonenglish_cleaners_no_transliteration` (cleaners.py:204-233): whitespace runs collapsed, ASCII digits and a fixed
    set of punctuation / control characters deleted, '|' (danda typed as a bar) -> '.', Devanagari numbers spelled out,
    then blanks squeezed and trimmed.  The deletions commute with one another, so they are done in one pass."""
    text = collapse_whitespace(text)
    text = "".join(ch for ch in text if ch not in _DROP).replace("|", ".")
    text = replace_devanagari_numbers(text)
    return " ".join(text.split())


def ascii_cleaners(text: str, english: bool) -> str:
    """The reference's `english_cleaners` / `nonenglish_cleaners` (cleaners.py:120-167) transliterate with `unidecode` and
    (English) spell numbers with `inflect`.  Neither package ships with this image, so this path is gated on them and
    is not covered by golden vectors."""
    try:
        from unidecode import unidecode  # noqa: F401
        if english:
            import inflect  # noqa: F401
    except ImportError as e:  # pragma: no cover - depends on the deployment image
        raise ImportError("the English / transliterating cleaners need the `unidecode` and `inflect` packages "
                          "(reference utils/aligner/cleaners.py:1-3); only indic_cleaners() is dependency-free") from e
    raise NotImplementedError("English / transliterating cleaners are not restated in this build (unpinned: their "
                              "dependencies are absent from the build image); clean the text upstream or use indic_cleaners()")


def text_to_characters(cleaned: str, symbols: Iterable[str]) -> List[str]:
    """demo.ipynb cell 9: keep the characters the aligner knows, blanks become the silence token."""
    known = set(symbols)
    return ["sil" if ch == " " else ch for ch in cleaned if ch in known]


def characters_to_batch(tokenizer, characters: Sequence[str], speaker: int = 0, device=None) -> Dict[str, torch.Tensor]:
    """demo.ipynb cell 11: one utterance -> the TTE batch dict (`phones`, `src_mask` True = valid, `speaker`)."""
    ids = torch.tensor([tokenizer.tokenize(list(characters))], dtype=torch.long)
    batch = {"phones": ids, "src_mask": ids != tokenizer.pad_idx, "speaker": torch.tensor([speaker], dtype=torch.long)}
    return {k: v.to(device) for k, v in batch.items()} if device is not None else batch


@torch.no_grad()
def synthesize_text(text: str, parrot, generator, tokenizer, symbols: Iterable[str], speaker: int = 0,
                    vocoder_speakers: Sequence[int] = (0,), cleaner=indic_cleaners):
    """demo.ipynb cells 9-13 on the GPU path: clean -> characters -> TTE unit ids -> HiFi-GAN, the notebook's loop over the
    vocoder's speakers run as ONE batch.  Returns (unit ids, waveforms (len(vocoder_speakers), 1, 320 * n_units))."""
    dev = next(generator.parameters()).device
    chars = text_to_characters(cleaner(text), symbols)
    if not chars:
        raise ValueError("no known characters left after cleaning")
    units = parrot.infer(characters_to_batch(tokenizer, chars, speaker, dev))[0]
    n = len(vocoder_speakers)
    code = torch.tensor([units] * n, dtype=torch.long, device=dev)
    spkr = torch.tensor(list(vocoder_speakers), dtype=torch.long, device=dev).reshape(n, 1)
    return units, generator(code=code, spkr=spkr)
