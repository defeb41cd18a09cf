#!/usr/bin/env python3
"""Probe: the vocoder of a B = 64 x 256 batch as ONE forward vs TWO concurrent forwards of 32 rows each on two HIP streams (two
handles, so that each lane owns its workspace).  The chunk-streamed path's two lanes beat the whole-utterance forward on long
utterances; does the same hold inside the BASELINE batch?      python tools/two_lane_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from parrot_tts_amd import synth  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402


def main():
    dev = "cuda:0"
    h = synth.default_voc_config()
    sd = synth.synth_voc_state_dict(h, seed=1234)
    gens = []
    for _ in range(2):
        g = CodeGenerator(AttrDict(h))
        g.load_state_dict(sd)
        gens.append(g.eval().to(dev))
    B, U = 64, 256
    vb = {k: v.to(dev) for k, v in synth.synth_voc_batch(B, U, h, seed=3).items()}
    halves = [{k: v[i * B // 2:(i + 1) * B // 2].contiguous() for k, v in vb.items()} for i in range(2)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

    def whole():
        return gens[0](code=vb["code"], spkr=vb["spkr"])

    def lanes():
        outs = []
        for g, hb, st in zip(gens, halves, streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(g(code=hb["code"], spkr=hb["spkr"]))
        for st in streams:
            torch.cuda.current_stream().wait_stream(st)
        return outs

    y = whole()
    o = lanes()
    torch.cuda.synchronize()
    print("equal", bool(torch.equal(torch.cat(o, 0), y)))
    for rep in range(3):
        for name, fn in (("one forward of 64 rows", whole), ("two concurrent forwards of 32 rows", lanes)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
