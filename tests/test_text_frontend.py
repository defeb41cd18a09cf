"""Text front-end restatement (parrot_tts_amd/text.py) against golden vectors produced by the reference's own functions
(tools/make_text_goldens.py; utils/aligner/cleaners.py:97-98,169-233) -- CPU only."""
import json
import os

import pytest
import torch

from parrot_tts_amd import text as T

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "text_cleaners.json"), encoding="utf-8"))


@pytest.mark.parametrize("fn,key", [(T.indic_cleaners, "nonenglish_cleaners_no_transliteration"),
                                    (T.replace_devanagari_numbers, "replace_devanagari_numbers"),
                                    (T.collapse_whitespace, "collapse_whitespace"),
                                    (T.english_cleaners, "english_cleaners"),          # ASCII, digit-free inputs
                                    (T.nonenglish_cleaners, "nonenglish_cleaners"),    # ASCII inputs
                                    (T.expand_abbreviations, "expand_abbreviations")])
def test_cleaners_match_reference_outputs(fn, key):
    assert len(GOLD[key]) >= 8
    for src, want in GOLD[key]:
        assert fn(src) == want, (key, src)


def test_number_words_match_reference():
    for n, want in GOLD["number_to_hindi"]:
        assert T.number_to_hindi(n) == want
    assert T.number_to_hindi(1000) is None and T.number_to_hindi(12345) is None


def test_characters_and_batch_follow_the_demo_notebook():
    symbols = ["क", "ख", " ", "ा", "."]

    class Tok:  # DFATokenizer's interface (modules/data.py:28-61): <pad>, <sep>, then the symbols with ' ' renamed 'sil'
        pad_idx = 0
        stoi = {s: i for i, s in enumerate(["<pad>", "<sep>", "क", "ख", "sil", "ा", "."])}

        def tokenize(self, seq):
            return [self.stoi[s] for s in seq]

    chars = T.text_to_characters(T.indic_cleaners("का  खा! ग | २"), symbols)
    assert chars == ["क", "ा", "sil", "ख", "ा", "sil", "sil", ".", "sil"]  # unknown 'ग' and the number words drop out
    batch = T.characters_to_batch(Tok(), chars, speaker=3)
    assert batch["phones"].tolist() == [[2, 5, 4, 3, 5, 4, 4, 6, 4]]
    assert batch["src_mask"].dtype == torch.bool and bool(batch["src_mask"].all())
    assert batch["speaker"].tolist() == [3]


def test_english_path_is_gated_not_faked():
    """Beyond ASCII (needs `unidecode`) and, for English, digits (needs `inflect`) the cleaners use the packages when they
    are importable and otherwise raise -- they never approximate."""
    try:
        import inflect  # noqa: F401
        have_inflect = True
    except ImportError:
        have_inflect = False
    try:
        import unidecode  # noqa: F401
        have_unidecode = True
    except ImportError:
        have_unidecode = False
    if not have_inflect:
        with pytest.raises(NotImplementedError):
            T.english_cleaners("Dr. Smith paid $5")
    if not have_unidecode:
        with pytest.raises(NotImplementedError):
            T.nonenglish_cleaners("नमस्ते")
        with pytest.raises(NotImplementedError):
            T.english_cleaners("café")
    assert T.ascii_cleaners("Mr. X", english=True) == "mister x" and T.ascii_cleaners("A  1 b", english=False) == "a b"
