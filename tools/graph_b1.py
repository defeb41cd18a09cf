"""One utterance: vocoder forward through the C ABI, direct launches vs the library's own graph cache (PARROT_VOC_GRAPH), on the
legacy default stream and on a torch side stream; and the whole pipeline at B = 1.  Run once with PARROT_VOC_GRAPH=0, once with 1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from parrot_tts_amd import _lib, ops, synth
from parrot_tts_amd.ops import dptr, stream_ptr
from parrot_tts_amd.pipeline import SynthesisPipeline

if os.environ.get("PARROT_FP_OLD") == "1":  # A/B: the round-4 parameter fingerprint (module.parameters() walk, ~400 us per forward)
    import parrot_tts_amd.tte as _t, parrot_tts_amd.vocoder as _v
    _old = lambda module: tuple((id(t), t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))  # noqa: E731
    ops.param_fingerprint = _t.param_fingerprint = _v.param_fingerprint = _old
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ops.set_default_precision(ops.PREC_NAMES["f16x3"])
cfg, h, tsd, vsd, parrot, gen = bench.build_models(dev, 300, 10)
batch = {k: v.to(dev) for k, v in synth.synth_tte_batch(1, 64, 300, 10, seed=0).items()}
r = parrot.infer_dense(batch)
gen(code=r["ids"], spkr=batch["speaker"].reshape(-1, 1))
torch.cuda.synchronize()
lib = _lib.lib()
VH = gen._handle
B, L = 1, r["ids"].shape[1]
ids = r["ids"].contiguous()
ws_v = torch.empty(lib.parrot_voc_workspace_bytes(VH, B, L), dtype=torch.uint8, device=dev)
wav = torch.empty((B, 1, L * 320), device=dev)
spk2 = batch["speaker"].reshape(-1).contiguous()


def voc():
    _lib.check(lib.parrot_voc_forward(VH, dptr(ids), dptr(spk2), None, B, L, dptr(wav), None, dptr(ws_v), ws_v.numel(), stream_ptr(dev)))


def timeit(fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, t_host


mode = os.environ.get("PARROT_VOC_GRAPH", "1")
t, th = timeit(voc)
print(f"graph={mode} vocoder on the default stream: {t:.1f} us per call (host enqueue {th:.1f})", flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    t, th = timeit(voc)
print(f"graph={mode} vocoder on a side stream: {t:.1f} us per call (host enqueue {th:.1f})", flush=True)
pipe = SynthesisPipeline(parrot, gen)
t, th = timeit(lambda: pipe(batch), 200)
print(f"graph={mode} whole pipeline B=1: {t:.1f} us per call", flush=True)
