"""Does HIP-graph replay shorten the launch-bound parts at B = 1?  Times direct launches vs graph replay of parrot_tte_encode,
parrot_tte_decode and the vocoder forward with fixed buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from parrot_tts_amd import _lib, ops, synth
from parrot_tts_amd.ops import dptr, stream_ptr

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ops.set_default_precision(ops.PREC_NAMES["f16x3"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg, h, tsd, vsd, parrot, gen = bench.build_models(dev, 300, 10)
batch = {k: v.to(dev) for k, v in synth.synth_tte_batch(B, 64, 300, 10, seed=0).items()}
r = parrot.infer_dense(batch); r = parrot.infer_dense(batch)
gen(code=r["ids"], spkr=batch["speaker"].reshape(-1, 1)); gen(code=r["ids"], spkr=batch["speaker"].reshape(-1, 1))
torch.cuda.synchronize()
lib = _lib.lib()
H = parrot._handle
S, L = 64, r["ids"].shape[1]
phones = batch["phones"].contiguous(); valid = batch["src_mask"].to(torch.uint8).contiguous(); spk = batch["speaker"].contiguous()
log_dur = torch.empty((B, S), device=dev); dur = torch.empty((B, S), dtype=torch.int64, device=dev); lens = torch.zeros(B, dtype=torch.int32, device=dev)
state = torch.empty(lib.parrot_tte_state_bytes(H, B, S), dtype=torch.uint8, device=dev)
ws_e = torch.empty(lib.parrot_tte_workspace_bytes(H, B, S, 0), dtype=torch.uint8, device=dev)
ws_d = torch.empty(lib.parrot_tte_workspace_bytes(H, B, S, L), dtype=torch.uint8, device=dev)
ids = torch.empty((B, L), dtype=torch.int64, device=dev); tgt = torch.empty((B, L), dtype=torch.uint8, device=dev)
VH = gen._handle
ws_v = torch.empty(lib.parrot_voc_workspace_bytes(VH, B, L), dtype=torch.uint8, device=dev)
wav = torch.empty((B, 1, L * 320), device=dev)
spk2 = batch["speaker"].reshape(-1).contiguous()

def enc():
    _lib.check(lib.parrot_tte_encode(H, dptr(phones), dptr(valid), dptr(spk), None, B, S, dptr(log_dur), dptr(dur), dptr(lens), dptr(state), state.numel(), dptr(ws_e), ws_e.numel(), stream_ptr(dev)))
def dec():
    _lib.check(lib.parrot_tte_decode(H, B, S, L, 0, dptr(ids), dptr(tgt), None, dptr(state), state.numel(), dptr(ws_d), ws_d.numel(), stream_ptr(dev)))
def voc():
    _lib.check(lib.parrot_voc_forward(VH, dptr(ids), dptr(spk2), None, B, L, dptr(wav), None, dptr(ws_v), ws_v.numel(), stream_ptr(dev)))

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

for name, fn in (("encode", enc), ("decode", dec), ("vocoder", voc)):
    fn(); torch.cuda.synchronize()
    direct = timeit(fn)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        rep = timeit(g.replay)
        print(f"B={B} {name}: direct {direct:.1f} us, graph replay {rep:.1f} us", flush=True)
    except Exception as e:
        print(f"B={B} {name}: direct {direct:.1f} us, graph capture failed: {type(e).__name__}: {str(e)[:200]}", flush=True)
