#!/usr/bin/env python3
"""Golden vectors for the dependency-free text cleaners of the reference (utils/aligner/cleaners.py:120-233).

The reference module imports `inflect` and `unidecode` at the top; neither is installed here and neither is used by
`nonenglish_cleaners_no_transliteration`, `replace_devanagari_numbers`, `number_to_hindi`, `collapse_whitespace`.
This script executes the reference file WITHOUT those two imports (and the `_inflect = ...` line) and records the
outputs of the reference's own functions on a fixed list of inputs -> tests/golden/text_cleaners.json (data only).
Run in the build container (needs /root/reference): python tools/make_text_goldens.py"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/utils/aligner/cleaners.py"

CASES = [
    "इसके लिए संरक्षित खेती, मधुमक्खी पालन इत्यादि पर अगले पंचवर्षीय योजना में महत्त्व देना होगा",
    "  कई   खाली \t जगह\nऔर नई पंक्ति  ",
    "वर्ष २०२ में ४५ लोग, १०० रुपये और ९९९ किताबें | ० शून्य है",
    "अंक 2024 और 15 हटते हैं; (कोष्ठक) [वर्ग] {धनु} <कोण> \"उद्धरण\" 'एकल' — डैश – छोटा … दीर्घ",
    "प्रतिशत 50% + योग = बराबर ^ घात \\ उल्टा _ रेखा ` उच्चारण @ पर / भाग - घटा * गुणा : ; ! अंत",
    "“घुमावदार” ‘उद्धरण’ ⇒ तीर",
    "नियंत्रण\x92वर्ण\xadयहाँ\x10और\x13भी\x14हैं\x16तथा\x91अंत",
    "mixed English text, with ASCII: kept as is? Yes.",
    "२१ २५ ३० ४७ ५९ ६६ ७८ ८४ ९३ १०१ ११० २५० ३९९ ५०० ७४२ ९०९",
    "",
    "|||",
    "संख्या १००० बहुत बड़ी",   # >= 1000: number_to_hindi falls through (returns None) and re.sub drops the match
    "१२३४५",
]
NUMBERS = [0, 1, 7, 10, 11, 19, 20, 21, 35, 48, 50, 77, 99, 100, 101, 110, 119, 200, 342, 500, 999]
# ASCII inputs for the English / transliterating pipelines (cleaners.py:120-166).  On ASCII text `unidecode` is the identity
# (its documented contract) and, without digits, no rule of `normalize_numbers` fires, so the reference's own functions run
# here WITHOUT either package: `unidecode` is bound to a checked identity that refuses non-ASCII input and `_inflect` to an
# object that raises on any use -- if either mattered for a case, generating the goldens would fail.
ENGLISH_CASES = [
    "Dr. Smith & Mrs. O'Neil met Mr. Brown at St. Mary's.",
    "The  Hon. Rev. Lt. Col. Gen. Sgt. Capt. Maj. (ret.) spoke;  TTS. Ltd. & Co. Esq. Jr. Drs. Ft. Knox",
    "UPPER case, Mixed\tWhitespace\n and #hash +plus \\back _under `tick @at /slash -dash 'quote >gt <lt (p) *star \"dq\" :colon ;semi !bang",
    "mr smith without a dot, mrs.jones, dr.who and a st.bernard",
    "AT&T & friends --- it's <fine>",
    "tilde ~ caret ^ pipe | percent % equals = brackets [ ] braces { } question ? comma , period .",
    "",
    "   ",
]
NONENGLISH_CASES = ENGLISH_CASES + [
    "namaste duniya 123 times, room 4B | 50% [ok] {x} = y ^ z",
    "Digits 0123456789 vanish; 'quotes' and \"doubles\" too",
]


def load_reference_functions():
    src = open(REF, encoding="utf-8").read()
    src = re.sub(r"^import inflect\s*$", "", src, flags=re.M)
    src = re.sub(r"^from unidecode import unidecode\s*$", "", src, flags=re.M)
    src = re.sub(r"^_inflect = inflect\.engine\(\)\s*$", "", src, flags=re.M)
    class _NoInflect:
        def __getattr__(self, name):
            raise RuntimeError("inflect would be needed for this input: not a dependency-free case")

    def _ascii_identity(text):
        if not text.isascii():
            raise RuntimeError("unidecode would be needed for this input: not a dependency-free case")
        return text
    ns = {"unidecode": _ascii_identity, "_inflect": _NoInflect()}
    exec(compile(src, REF, "exec"), ns)
    return ns


def main():
    ns = load_reference_functions()
    out = {
        "generator": "tools/make_text_goldens.py (reference functions executed from " + REF + ")",
        "nonenglish_cleaners_no_transliteration": [[c, ns["nonenglish_cleaners_no_transliteration"](c)] for c in CASES],
        "replace_devanagari_numbers": [[c, ns["replace_devanagari_numbers"](c)] for c in CASES],
        "collapse_whitespace": [[c, ns["collapse_whitespace"](c)] for c in CASES],
        "number_to_hindi": [[n, ns["number_to_hindi"](n)] for n in NUMBERS],
        "english_cleaners": [[c, ns["english_cleaners"](c)] for c in ENGLISH_CASES],
        "nonenglish_cleaners": [[c, ns["nonenglish_cleaners"](c)] for c in NONENGLISH_CASES],
        "expand_abbreviations": [[c.lower(), ns["expand_abbreviations"](c.lower())] for c in ENGLISH_CASES],
    }
    path = os.path.join(ROOT, "tests", "golden", "text_cleaners.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=1)
    print("wrote", path, {k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
