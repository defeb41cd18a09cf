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
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", h) for h in ("parrot_hip.h", "parrot_hip_debug.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# Per-source flags.  tu_split16: LLVM's machine-sink pass moves the weight refills for the next chunk's first k-step -- which feed
# nothing but loop-carried registers -- out of the MFMA block into the loop latch, behind the slab conversion and the barrier, so
# that the first step of every chunk waits a full fetch latency (tools/isa_schedule.py shows it in the ISA).  Without the pass
# they stay where they are written: conv_split16 128 x 128: 6.88 -> 6.78 ms per step (mean of five same-box A/B pairs, the
# sign held in four, one tie: at the edge of the run-to-run noise), no spills, nothing else moves.
TU_FLAGS = {f: ["-mllvm", "-disable-machine-sink"] for f in ("tu_split16.hip", "tu_split16_wide.hip", "tu_split16_xpl.hip")}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, extra_flags=(), lib_path: str = None, obj_dir: str = None) -> str:
    """Compile csrc/*.hip -> parrot_tts_amd/libparrot_hip.so.  Returns the library path.
    `lib_path` / `obj_dir` / `extra_flags`: experiment builds of the same sources (tools/build_exp.sh -> build_exp/)."""
    if lib_path is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    LIB_OUT, OBJ_OUT = lib_path or LIB, obj_dir or OBJ
    os.makedirs(OBJ_OUT, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ_OUT, os.path.splitext(src)[0] + ".o")
        tu = [] if os.environ.get("PARROT_NO_TU_FLAGS") else TU_FLAGS.get(src, [])  # (A/B builds: tools/build_exp.sh)
        cmd = [hipcc] + FLAGS + tu + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, (os.cpu_count() or 2) - 1))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_OUT] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_OUT


if __name__ == "__main__":
    if "--exp" in sys.argv:  # python -m parrot_tts_amd.build --exp TAG -DFOO -DBAR ...  ->  build_exp/libparrot_TAG.so
        i = sys.argv.index("--exp")
        tag, flags = sys.argv[i + 1], sys.argv[i + 2:]
        root = os.path.dirname(HERE)
        print(build(force=True, verbose=False, extra_flags=flags, lib_path=os.path.join(root, "build_exp", f"libparrot_{tag}.so"),
                    obj_dir=os.path.join(root, "build_exp", "obj_" + tag)))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
