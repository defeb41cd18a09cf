"""CPU: the C-ABI library builds, loads, and exports every symbol include/parrot_hip.h (the product interface) and
include/parrot_hip_debug.h (test / profiling / probe entry points) declare (no compute calls -- there is no GPU here)."""
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


def _declared(header="parrot_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(parrot_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from parrot_tts_amd import _lib
    product, debug = _declared(), _declared("parrot_hip_debug.h")
    assert len(product) >= 20
    # the product header carries no debug / profiling / probe entry point
    assert not [n for n in product if "debug" in n or "prof" in n or "selftest" in n]
    assert not set(product) & set(debug)
    names = sorted(product + debug)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in parrot_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_abi_basics(lib):
    assert lib.parrot_abi_version() == 7
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


def test_header_is_plain_c_and_links_from_c(lib, tmp_path):
    """The boundary is a C ABI: include/parrot_hip.h compiles as C (gcc -std=c99 -pedantic) and a C program binds the
    entry points by name with dlopen -- no C++ or torch types involved."""
    import shutil
    import subprocess
    from parrot_tts_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text('''
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "parrot_hip.h"
int main(int argc, char** argv) {
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\\n", dlerror()); return 2; }
    int (*ver)(void) = (int (*)(void))dlsym(h, "parrot_abi_version");
    const char* (*err)(void) = (const char* (*)(void))dlsym(h, "parrot_last_error");
    int (*run)(parrot_conv_t*, const float*, const float*, float*, int32_t, int32_t, int32_t, float, void*) =
        (int (*)(parrot_conv_t*, const float*, const float*, float*, int32_t, int32_t, int32_t, float, void*))dlsym(h, "parrot_conv_run");
    if (!ver || !err || !run) return 3;
    if (ver() != PARROT_ABI_VERSION) return 4;
    if (run(NULL, NULL, NULL, NULL, 1, 1, 0, 1.0f, NULL) != PARROT_E_INVALID) return 5;
    if (!strstr(err(), "null")) return 6;
    printf("ok\\n");
    return 0;
}
''')
    hdr = tmp_path / "hdr.c"
    hdr.write_text('#include "parrot_hip.h"\n#include "parrot_hip_debug.h"\nint parrot_header_only_tu;\n')
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(hdr)], check=True)
    exe = tmp_path / "abi"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-ldl"], check=True)
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([str(exe), _lib.LIB_PATH], check=True, capture_output=True, text=True, env=env)
    assert out.stdout.strip() == "ok"
