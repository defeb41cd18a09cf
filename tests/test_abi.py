"""CPU: the C-ABI library builds, loads, and exports every symbol include/parrot_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from parrot_tts_amd import build, _lib
    build.build()
    return _lib.lib()


def _declared():
    src = open(os.path.join(ROOT, "include", "parrot_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(parrot_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from parrot_tts_amd import _lib
    names = _declared()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in parrot_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_abi_basics(lib):
    assert lib.parrot_abi_version() == 1
    assert lib.parrot_conv_num_tile_cfgs() >= 3
    assert isinstance(lib.parrot_last_error(), bytes)
    # argument validation happens before any HIP call
    assert lib.parrot_conv_run(None, None, None, None, 1, 1, 0, 1.0, None) == -1
    assert b"null" in lib.parrot_last_error()


def test_struct_sizes_match_header():
    from parrot_tts_amd import _lib
    assert ctypes.sizeof(_lib.ConvDesc) == 12 * 4
    assert ctypes.sizeof(_lib.TteCfg) == 14 * 4
    n_int = 7 + 8 + 8 + 1 + 4 + 1 + 16 + 1
    assert ctypes.sizeof(_lib.VocCfg) == n_int * 4
