#!/usr/bin/env python3
"""CPU emulation of the split product schemes on the whole generator (test tooling: imports oracle/): operands are split as
the kernels split them (fp16x3 with the 2^3 / per-layer power-of-two pre-scales, bf16x6, single bf16 / fp16), the piece
products are summed in fp64 and rounded to fp32 per layer -- the schemes' own error without any accumulation noise --
and compared with an fp64 run of the oracle.   python tools/split_numerics.py [units] [weight scale]"""
import sys, json, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import parrot_oracle as O
from parrot_tts_amd import synth
torch.set_num_threads(8)
h = synth.default_voc_config()
U = int(sys.argv[1]) if len(sys.argv) > 1 else 24
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sd = synth.synth_voc_state_dict(h, seed=1234, scale=scale)
b = synth.synth_voc_batch(1, U, h, seed=5)
sd64 = {k: v.double() for k, v in sd.items()}

def split_f16(t, sx):  # t fp32 -> pieces (as fp64) : x1 = f16(x*sx), x2 = f16(x*sx - x1)
    ts = t.float() * sx
    x1 = ts.half().float()
    x2 = (ts - x1).half().float()
    return x1.double() / sx, x2.double() / sx
def split_bf16(t, n):
    r = t.float(); out = []
    for _ in range(n):
        p = r.bfloat16().float(); out.append(p.double()); r = r - p
    return out

MODE = None
orig_c, orig_ct = F.conv1d, F.conv_transpose1d
def wscale(w):
    m = float(w.abs().max()); return 2.0 ** np.floor(np.log2(32768.0 / m)) if m > 0 else 1.0
def emu(fn, x, w, bias, **kw):
    if MODE is None or x.dtype == torch.float64:
        return fn(x, w, bias, **kw)
    if MODE == 'f16x3':
        x1, x2 = split_f16(x, 8.0); w1, w2 = split_f16(w, wscale(w))
        y = fn(x1, w1, None, **kw) + fn(x1, w2, None, **kw) + fn(x2, w1, None, **kw)
    elif MODE == 'f16x3_unscaled':
        x1, x2 = split_f16(x, 1.0); w1, w2 = split_f16(w, 1.0)
        y = fn(x1, w1, None, **kw) + fn(x1, w2, None, **kw) + fn(x2, w1, None, **kw)
    elif MODE == 'f16x4':
        x1, x2 = split_f16(x, 8.0); w1, w2 = split_f16(w, wscale(w))
        y = fn(x1, w1, None, **kw) + fn(x1, w2, None, **kw) + fn(x2, w1, None, **kw) + fn(x2, w2, None, **kw)
    elif MODE == 'bf16x6':
        x1, x2, x3 = split_bf16(x, 3); w1, w2, w3 = split_bf16(w, 3)
        y = fn(x1, w1, None, **kw) + fn(x1, w2, None, **kw) + fn(x2, w1, None, **kw) + fn(x2, w2, None, **kw) + fn(x1, w3, None, **kw) + fn(x3, w1, None, **kw)
    elif MODE == 'bf16x3':
        x1, x2 = split_bf16(x, 2); w1, w2 = split_bf16(w, 2)
        y = fn(x1, w1, None, **kw) + fn(x1, w2, None, **kw) + fn(x2, w1, None, **kw)
    elif MODE == 'bf16':
        (x1,) = split_bf16(x, 1); (w1,) = split_bf16(w, 1)
        y = fn(x1, w1, None, **kw)
    elif MODE == 'f16':
        y = fn(x.half().double(), w.half().double(), None, **kw)
    elif MODE == 'exact':
        y = fn(x.double(), w.double(), None, **kw)
    if bias is not None:
        y = y + bias.double().view(1, -1, 1)
    return y.float()
F.conv1d = lambda x, w, b=None, **kw: emu(orig_c, x, w, b, **kw)
F.conv_transpose1d = lambda x, w, b=None, **kw: emu(orig_ct, x, w, b, **kw)
with torch.no_grad():
    ref = O.code_generator_forward(sd64, h, b['code'], b['spkr'])
    print('ref max', float(ref.abs().max()), 'rms', float(ref.pow(2).mean().sqrt()))
    for m in [None, 'exact', 'bf16x6', 'f16x4', 'f16x3', 'f16x3_unscaled', 'bf16x3', 'f16', 'bf16']:
        MODE = m
        y = O.code_generator_forward(sd, h, b['code'], b['spkr']).double()
        e = (y - ref)
        snr = 10 * np.log10(float(ref.pow(2).sum() / e.pow(2).sum()))
        print(f'{str(m):16s} max-abs {float(e.abs().max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}  SNR {snr:.1f} dB')
