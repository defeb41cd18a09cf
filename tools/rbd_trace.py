#!/usr/bin/env python3
"""Phase timeline of one dual-window fused ResBlock workgroup (experiment build with -DRBD_TRACE):
    tools/build_exp.sh trace -DRBD_TRACE
    PARROT_HIP_LIB=build_exp/libparrot_trace.so PARROT_RB_DUAL=1 PARROT_RBD_TRACE_SEL=3203 python tools/rbd_trace.py
prints, per wave of the traced workgroup, the shader-clock deltas between the marks of resblock_dual.h."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import _lib, synth  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402


def main():
    B = int(os.environ.get("TRACE_B", "64"))
    h = synth.default_voc_config()
    g = CodeGenerator(AttrDict(h))
    g.load_state_dict(synth.synth_voc_state_dict(h, seed=1234, scale=1.0))
    g = g.eval().to("cuda:0")
    vb = synth.synth_voc_batch(B, 256, h, seed=0)
    for _ in range(3):
        g(code=vb["code"].to("cuda:0"), spkr=vb["spkr"].to("cuda:0"))
    torch.cuda.synchronize()
    lib = _lib.lib()
    fn = lib.parrot_debug_rbd_trace
    fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * (8 * 64))()
    assert fn(buf) == 0
    names = {0: "start", 1: "pre-wp0", 2: "wp0 done", 3: "bar"}
    for w in range(8):
        t = [buf[w * 64 + i] for i in range(64)]
        hw = t[63]
        simd, cu, wave_slot = (hw >> 4) & 3, (hw >> 8) & 15, hw & 15
        n = max(i for i in range(62) if t[i]) + 1 if any(t[:62]) else 0
        d = [t[i] - t[i - 1] for i in range(1, n)]
        print(f"wave {w} (win {w >> 2}) simd {simd} cu {cu} slot {wave_slot}: total {t[n - 1] - t[0] if n else 0}")
        print("   deltas:", " ".join(str(x) for x in d))
    # marks: 0 start | 1 pre-wp | 2 wp done | 3 barrier | then per pair: 4 conv1 done | 5 bar | 6 wp(h) done | 7 bar | 8 conv2 done | 9 bar | 10 wp(R) | 11 bar


if __name__ == "__main__":
    main()
