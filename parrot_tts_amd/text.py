"""Text front-end of the demo path: raw text -> cleaned text -> character tokens -> the
TTE's `phones` batch, and the end-to-end text -> waveform helper of demo.ipynb.

Host-side string work only.  The cleaner for non-transliterated Indic text is a restatement of the reference's
dependency-free pipeline (utils/aligner/cleaners.py:204-233 with :169-202) and is pinned by golden vectors produced by
the reference's own functions (tools/make_text_goldens.py -> tests/golden/text_cleaners.json).  The English and the
transliterating pipelines of the reference use `unidecode` and `inflect`, which this image does not have: they are
restated and pinned for the part of the input space where neither package can matter (ASCII text; digit-free for the
English one), use the packages when importable beyond that, and raise instead of approximating otherwise."""
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


# ---- English / transliterating pipelines (cleaners.py:46-166) ---------------------------------------------------------
# The reference transliterates with `unidecode` and spells numbers with `inflect`; neither ships with this image.  Both
# are only NEEDED for part of the input space: `unidecode` passes ASCII through unchanged (its documented contract), and
# every number rule of `normalize_numbers` starts from a digit.  So ASCII text is handled here without either package
# (pinned by goldens the reference's own functions produced on such inputs, tools/make_text_goldens.py), and anything
# beyond that uses the packages when they are importable and raises -- never approximates -- when they are not.
_ABBREVIATIONS = [(re.compile("\\b%s\\." % a, re.IGNORECASE), b) for a, b in [
    ("mrs", "misess"), ("&", "and"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
    ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"), ("hon", "honorable"),
    ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"), ("col", "colonel"), ("ft", "fort"),
    ("tts", "text to speech")]]  # cleaners.py:16-37, in the reference's order (so "mrs." is taken before "mr.")
_ENGLISH_DROP = ["#", "+", "\\", "_", "`", "@", "/", "-", "'", ">", "<", "(", ")", "*", '"', ":", ";", "!"]            # cleaners.py:129
_NONENGLISH_DROP = ["|", "%", "+", "=", "[", "]", "^", "\\", "{", "}", "_", "`", "‘", "’", "@", "/", "-", "'", ">", "<", "(", ")",
                    "*", '"', ":", ";", "!"]                                                                            # cleaners.py:153
_NON_PRINTABLE = re.compile(r"[^\x20-\x7E]")
_HAS_DIGIT = re.compile(r"[0-9]")


def expand_abbreviations(text: str) -> str:
    """cleaners.py:46-49"""
    for rx, rep in _ABBREVIATIONS:
        text = rx.sub(rep, text)
    return text


def _to_ascii(text: str) -> str:
    """`unidecode(text)` (cleaners.py:122,144): the identity on ASCII input; anything else needs the package."""
    if text.isascii():
        return text
    try:
        from unidecode import unidecode
    except ImportError as e:
        raise NotImplementedError("transliterating non-ASCII text needs the `unidecode` package (reference "
                                  "utils/aligner/cleaners.py:3); it is not installed and is not approximated here") from e
    return unidecode(text)


def _normalize_numbers(text: str) -> str:
    """`normalize_numbers` (cleaners.py:51-112).  Every rule needs a digit, so digit-free text passes through; with digits
    the wording comes from `inflect`, exactly as in the reference, or not at all."""
    if not _HAS_DIGIT.search(text):
        return text
    try:
        import inflect
    except ImportError as e:
        raise NotImplementedError("spelling out numbers needs the `inflect` package (reference utils/aligner/cleaners.py:2); "
                                  "it is not installed and is not approximated here") from e
    eng = inflect.engine()

    def dollars(m):
        parts = m.group(1).split(".")
        if len(parts) > 2:
            return m.group(1) + " dollars"
        d = int(parts[0]) if parts[0] else 0
        c = int(parts[1]) if len(parts) > 1 and parts[1] else 0
        if d and c:
            return "%s %s, %s %s" % (d, "dollar" if d == 1 else "dollars", c, "cent" if c == 1 else "cents")
        if d:
            return "%s %s" % (d, "dollar" if d == 1 else "dollars")
        if c:
            return "%s %s" % (c, "cent" if c == 1 else "cents")
        return "zero dollars"

    def number(m):
        num = int(m.group(0))
        if 1000 < num < 3000:
            if num == 2000:
                return "two thousand"
            if 2000 < num < 2010:
                return "two thousand " + eng.number_to_words(num % 100)
            if num % 100 == 0:
                return eng.number_to_words(num // 100) + " hundred"
            return eng.number_to_words(num, andword="", zero="oh", group=2).replace(", ", " ")
        return eng.number_to_words(num, andword="")

    text = re.sub(r"([0-9][0-9\,]+[0-9])", lambda m: m.group(1).replace(",", ""), text)
    text = re.sub(r"£([0-9\,]*[0-9]+)", r"\1 pounds", text)
    text = re.sub(r"\$([0-9\.\,]*[0-9]+)", dollars, text)
    text = re.sub(r"([0-9]+\.[0-9]+)", lambda m: m.group(1).replace(".", " point "), text)
    text = re.sub(r"[0-9]+(st|nd|rd|th)", lambda m: eng.number_to_words(m.group(0)), text)
    return re.sub(r"[0-9]+", number, text)


def english_cleaners(text: str) -> str:
    """`english_cleaners` (cleaners.py:120-140): transliterate, lowercase, spell numbers, expand abbreviations, squeeze blanks,
    delete non-printable ASCII and a fixed punctuation set, '&' -> 'and'.  Self-contained for ASCII text without digits."""
    text = _normalize_numbers(_to_ascii(text).lower())
    text = collapse_whitespace(expand_abbreviations(text))
    text = _NON_PRINTABLE.sub("", text)
    for ch in _ENGLISH_DROP:
        text = text.replace(ch, "")
    return text.replace("&", "and")


def nonenglish_cleaners(text: str) -> str:
    """`nonenglish_cleaners` (cleaners.py:142-166), the transliterating pipeline: lowercase, squeeze blanks, delete digits,
    non-printable ASCII and a fixed punctuation set, '&' -> 'and', trim.  Self-contained for ASCII text."""
    text = collapse_whitespace(_to_ascii(text).lower())
    for d in "0123456789":
        text = text.replace(d, "")
    text = _NON_PRINTABLE.sub("", text)
    for ch in _NONENGLISH_DROP:
        text = text.replace(ch, "")
    return " ".join(text.replace("&", "and").split())


def ascii_cleaners(text: str, english: bool) -> str:
    """Either of the two pipelines above (kept for callers of the round-1 name)."""
    return english_cleaners(text) if english else nonenglish_cleaners(text)


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
