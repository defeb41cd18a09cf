"""Build libparrot_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).  The kernel templates are
instantiated in several translation units (csrc/tu_*.hip + parrot_hip.hip) compiled in parallel, then linked."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libparrot_hip.so")
OBJ = os.path.join(HERE, "build")
SOURCES = ["parrot_hip.hip"] + sorted(f for f in os.listdir(CSRC) if f.startswith("tu_") and f.endswith(".hip"))
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "parrot_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    """Compile csrc/*.hip -> parrot_tts_amd/libparrot_hip.so.  Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
