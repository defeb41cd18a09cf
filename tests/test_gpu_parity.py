"""GPU parity tests (run with -m gpu on the MI355X box): every call goes through the C ABI of
libparrot_hip.so (via the ctypes shims) and is compared with the CPU oracle / torch CPU ops on the
same seeded inputs, and with the reference-generated golden vectors.

Tolerances (fp32 path):
  * single conv layer      : |diff| <= 2e-5 * max|y|   (different summation order only)
  * vocoder waveform       : max-abs <= 5e-5 of full scale on the realistic-scale weights (the
                             reference's own fp32-vs-fp64 error there is 7e-6).  The "hot" golden (weight scale 1.2,
                             rms 0.98, tanh saturated everywhere) is a stress case with pre-tanh magnitudes ~1e2: the
                             reference's own fp32 result is 1.2e-4 off an fp64 run there, and the HIP waveform is held
                             to <= 3x that distance from the fp64 run (measured: 2.5e-4 from the golden).
  * TTE log-durations      : 2e-5 abs;  durations exact where |frac-0.5| > 1e-4
  * TTE unit ids           : bit-exact where the top-2 logit margin > 1e-4 (all goldens satisfy it)
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import parrot_oracle as O  # noqa: E402
from parrot_tts_amd import ops, synth  # noqa: E402
from parrot_tts_amd.tte import Parrot  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402

DEV = "cuda:0"
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


def _report(**kw):
    """Append measured errors to gpurun_out/parity_report.jsonl (scratch; summarised in DESIGN.md)."""
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _randn(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def test_mfma_layout_probe():
    ops.selftest()


@pytest.fixture(params=[0, 1, 2], ids=["layerwise", "fused_16_32", "fused_16"])
def fused(request):
    """Low-channel ResBlocks layer by layer, or through the fused LDS-resident kernels (16+32 channels / 16 only)."""
    ops.set_fused_resblocks(request.param)
    yield request.param
    ops.set_fused_resblocks(2)


@pytest.fixture(params=["f32", "bf16x6", "f16x3"])
def prec(request):
    """Run a model-level test under every parity-grade product-evaluation mode of the conv kernels (include/parrot_hip.h
    PARROT_PREC_*): exact fp32 MFMA, the split-bf16 scheme (6 bf16 MFMAs per product group) and the split-fp16 scheme
    (3 fp16 MFMAs; the library default) -- fp32 data and accumulation in all three.  All are held to the SAME tolerances."""
    ops.set_default_precision(ops.PREC_NAMES[request.param])
    yield request.param
    ops.set_default_precision(ops.PREC_DEFAULT)


# ----------------------------------------------------------------------------------------------
# single layers
# ----------------------------------------------------------------------------------------------
CONV_CASES = [
    # cin, cout, k, dil, B, T, pre, res, act
    (16, 16, 3, 1, 2, 700, True, True, 0),
    (16, 16, 11, 5, 1, 1500, True, True, 0),
    (32, 32, 7, 3, 2, 1000, True, True, 0),
    (64, 64, 11, 5, 2, 600, True, True, 0),
    (128, 128, 7, 1, 1, 300, True, False, 0),
    (256, 256, 3, 5, 2, 200, True, True, 0),
    (256, 256, 11, 3, 1, 130, True, True, 0),
    (256, 512, 7, 1, 2, 77, False, False, 0),     # conv_pre shape, ragged T
    (16, 1, 7, 1, 2, 999, True, False, 2),        # conv_post shape (tanh)
    (256, 1024, 9, 1, 2, 61, False, False, 1),    # FFN conv1 (relu), T < tile
    (1024, 256, 1, 1, 2, 61, False, True, 0),     # FFN conv2 (k=1, residual)
    (256, 768, 1, 1, 3, 23, False, False, 0),     # qkv linear
    (256, 1000, 1, 1, 2, 40, False, False, 0),    # head (M not a tile multiple)
    (256, 1, 1, 1, 3, 23, False, False, 0),       # duration proj
    (20, 24, 5, 2, 2, 50, True, True, 0),         # odd channel counts (zero-padded chunks / rows)
    (3, 5, 3, 1, 1, 1, False, False, 0),          # T = 1
    (16, 16, 7, 3, 3, 2000, True, True, 0),       # stage-4 shapes on the 16-row MFMA tile
    (40, 9, 5, 1, 2, 300, False, True, 1),        # 16-row tile, 3 channel chunks, relu + late residual
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5, 6])
def test_conv1d_matches_torch(case, tile):
    cin, cout, k, dil, B, T, pre, res, act = case
    rng = _rng(cin * 131 + cout * 7 + k + dil + T + tile)
    w = _randn(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
    b = _randn(rng, cout, scale=0.1)
    x = _randn(rng, B, cin, T)
    r = _randn(rng, B, cout, T) if res else None
    pad = dil * (k - 1) // 2
    xin = F.leaky_relu(x, 0.1) if pre else x
    y_ref = F.conv1d(xin, w, b, padding=pad, dilation=dil)
    if act == 1:
        y_ref = F.relu(y_ref)
    elif act == 2:
        y_ref = torch.tanh(y_ref)
    if res:
        y_ref = y_ref + r
    if tile == 6 and cout > 16:
        with pytest.raises(Exception):  # the 16-row MFMA tile only takes layers with <= 16 output channels
            ops.ConvPlan(w, b, dilation=dil, padding=pad, tile_cfg=tile)
        return
    plan = ops.ConvPlan(w, b, dilation=dil, padding=pad, pre_act=int(pre), pre_slope=0.1, act=act, tile_cfg=tile)
    y = plan(x.to(DEV), None if r is None else r.to(DEV)).cpu()
    assert y.shape == y_ref.shape
    tol = 2e-5 * max(1.0, float(y_ref.abs().max()))
    assert float((y - y_ref).abs().max()) <= tol


@pytest.mark.parametrize("scheme", ["bf16x6", "f16x3"])
@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[1] >= 32], ids=lambda c: "x".join(map(str, c)))
def test_conv1d_split_schemes_match_fp64(case, scheme):
    cin, cout, k, dil, B, T, pre, res, act = case
    rng = _rng(cin * 31 + cout * 17 + k + dil + T)
    w = _randn(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
    b = _randn(rng, cout, scale=0.1)
    x = _randn(rng, B, cin, T)
    r = _randn(rng, B, cout, T) if res else None
    pad = dil * (k - 1) // 2
    y_ref = F.conv1d((F.leaky_relu(x, 0.1) if pre else x).double(), w.double(), b.double(), padding=pad, dilation=dil)
    if act == 1:
        y_ref = F.relu(y_ref)
    if res:
        y_ref = y_ref + r.double()
    args = dict(dilation=dil, padding=pad, pre_act=int(pre), pre_slope=0.1, act=act)
    xd, rd = x.to(DEV), None if r is None else r.to(DEV)
    y6 = ops.ConvPlan(w, b, precision=ops.PREC_NAMES[scheme], **args)(xd, rd).cpu().double()
    y32 = ops.ConvPlan(w, b, precision=ops.PREC_F32, **args)(xd, rd).cpu().double()
    scale = max(1.0, float(y_ref.abs().max()))
    e6, e32 = float((y6 - y_ref).abs().max()) / scale, float((y32 - y_ref).abs().max()) / scale
    _report(test="conv_split_vs_fp64", scheme=scheme, case="x".join(map(str, case)), err_split=e6, err_f32_mfma=e32)
    assert e6 <= 2e-5
    assert e6 <= 4 * e32 + 1e-6, "a split evaluation must stay in the same error class as the exact fp32 kernel"


def test_conv1d_fuzz_wide_layers_all_kernel_variants():
    """Seeded random layer shapes through the default (fp16x3) plans: channel counts / tap counts / dilations that select
    conv_split16_kernel (k >= 7, C_in % 32 == 0, >= 64 rows: 128- and 64-row tiles, and their 64-column small-batch variants),
    conv_split_kernel (k = 3, 1, other channel counts) and the exact kernel (odd channel counts), with sequence lengths
    around every tile boundary, residual / ReLU / leaky-ReLU on and off -- against torch in fp64."""
    rng = _rng(2024)
    for case in range(48):
        cin = int(rng.choice([32, 64, 96, 128, 160, 256, 48, 20]))
        cout = int(rng.choice([64, 128, 192, 256, 72, 1024]))
        k = int(rng.choice([1, 3, 7, 9, 11]))
        dil = int(rng.choice([1, 3, 5])) if k > 1 else 1
        B = int(rng.integers(1, 4))
        T = int(rng.choice([1, 15, 63, 64, 65, 127, 128, 129, 191, 193, 255, 257, 300, 511, 640, 700]))
        pre, res, act = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), int(rng.choice([0, 0, 1]))
        w = _randn(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
        b = _randn(rng, cout, scale=0.1)
        x = _randn(rng, B, cin, T)
        r = _randn(rng, B, cout, T) if res else None
        pad = dil * (k - 1) // 2
        y_ref = F.conv1d((F.leaky_relu(x, 0.1) if pre else x).double(), w.double(), b.double(), padding=pad, dilation=dil)
        if act == 1:
            y_ref = F.relu(y_ref)
        if res:
            y_ref = y_ref + r.double()
        plan = ops.ConvPlan(w, b, dilation=dil, padding=pad, pre_act=int(pre), pre_slope=0.1, act=act, precision=ops.PREC_F16X3)
        y = plan(x.to(DEV), None if r is None else r.to(DEV)).cpu().double()
        err = float((y - y_ref).abs().max()) / max(1.0, float(y_ref.abs().max()))
        assert err <= 2e-5, (case, cin, cout, k, dil, B, T, pre, res, act, err)


def test_conv1d_fuzz_chip_filling_launches_take_the_wide_tiles():
    """The same check on launches of >= 256 workgroups, where conv_split16 picks between its 128 x 128, 128 x 160 and 64 x 128 tiles
    by rounds x width (parrot_hip.hip / conv_split16.h split16_wide_fits): sequence lengths that leave partial 128- and 160-column
    tiles, 64 / 128 / 256 rows, residual on and off -- against torch in fp64."""
    rng = _rng(4242)
    for case in range(10):
        cin = int(rng.choice([64, 128, 256]))
        cout = int(rng.choice([64, 128, 256]))
        k = int(rng.choice([7, 11]))
        dil = int(rng.choice([1, 3, 5]))
        B = int(rng.choice([24, 33, 48]))
        T = int(rng.choice([330, 641, 1250, 1285, 1601]))
        pre, res = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        w = _randn(rng, cout, cin, k, scale=1.0 / np.sqrt(cin * k))
        b = _randn(rng, cout, scale=0.1)
        x = _randn(rng, B, cin, T)
        r = _randn(rng, B, cout, T) if res else None
        pad = dil * (k - 1) // 2
        y_ref = F.conv1d((F.leaky_relu(x, 0.1) if pre else x).double(), w.double(), b.double(), padding=pad, dilation=dil)
        if res:
            y_ref = y_ref + r.double()
        plan = ops.ConvPlan(w, b, dilation=dil, padding=pad, pre_act=int(pre), pre_slope=0.1, precision=ops.PREC_F16X3)
        y = plan(x.to(DEV), None if r is None else r.to(DEV)).cpu().double()
        err = float((y - y_ref).abs().max()) / max(1.0, float(y_ref.abs().max()))
        assert err <= 2e-5, (case, cin, cout, k, dil, B, T, pre, res, err)


def test_split_f16_scaling_covers_small_and_large_operands_and_overflows_loudly():
    """The fp16 split pre-scales activations by 2^3 and the layer's weights by a power of two: full relative accuracy for
    tiny weights (1e-6 scale) and activations from 1e-3 to 1e3 in one tensor; beyond fp16's range (|x| >= 8190) the
    result is non-finite -- never a silently wrong finite number."""
    rng = _rng(77)
    cin, cout, k, T = 64, 64, 3, 512
    mag = torch.from_numpy(10.0 ** rng.uniform(-3, 3, size=(1, cin, T))).float()
    x = _randn(rng, 1, cin, T) * mag
    for wscale in (1e-6, 1.0, 300.0):
        w = _randn(rng, cout, cin, k, scale=wscale / np.sqrt(cin * k))
        y_ref = F.conv1d(x.double(), w.double(), None, padding=1)
        y = ops.ConvPlan(w, None, padding=1, precision=ops.PREC_F16X3)(x.to(DEV)).cpu().double()
        y32 = ops.ConvPlan(w, None, padding=1, precision=ops.PREC_F32)(x.to(DEV)).cpu().double()
        scale = float(y_ref.abs().max())
        e, e32 = float((y - y_ref).abs().max()) / scale, float((y32 - y_ref).abs().max()) / scale
        _report(test="f16x3_dynamic_range", wscale=wscale, err_f16x3=e, err_f32_mfma=e32)
        assert e <= 4 * e32 + 1e-6
    x[0, 3, 100] = 9000.0
    y = ops.ConvPlan(w, None, padding=1, precision=ops.PREC_F16X3)(x.to(DEV)).cpu()
    assert not bool(torch.isfinite(y[0, :, 99:102]).all())


def test_conv1d_epilogues_accumulate_like_mrf():
    rng = _rng(5)
    C, T, B = 32, 333, 2
    x = _randn(rng, B, C, T)
    ws = [_randn(rng, C, C, 3, scale=0.1) for _ in range(3)]
    bs = [_randn(rng, C, scale=0.1) for _ in range(3)]
    outs = [F.conv1d(x, w, b, padding=1) + x for w, b in zip(ws, bs)]
    ref = ((outs[0] + outs[1]) + outs[2]) / 3
    xs = torch.empty(B, C, T, device=DEV)
    xd = x.to(DEV)
    for j, (w, b) in enumerate(zip(ws, bs)):
        ops.ConvPlan(w, b, padding=1)(xd, xd, out=xs, epilogue=[ops.EPI_STORE, ops.EPI_ADD, ops.EPI_ADD_DIV][j], div=3.0)
    assert float((xs.cpu() - ref).abs().max()) <= 1e-5


CONVT_CASES = [(512, 256, 11, 5, 2, 40), (256, 128, 8, 4, 1, 90), (128, 64, 8, 4, 2, 33), (64, 32, 4, 2, 1, 257),
               (32, 16, 4, 2, 2, 513), (8, 4, 5, 2, 1, 19), (8, 4, 3, 3, 1, 10), (6, 6, 7, 2, 2, 1)]


@pytest.mark.parametrize("precision", [0, 1, 2])
@pytest.mark.parametrize("case", CONVT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_transpose1d_matches_torch(case, precision):
    cin, cout, k, u, B, T = case
    rng = _rng(cin + cout * 3 + k * 5 + u)
    w = _randn(rng, cin, cout, k, scale=1.0 / np.sqrt(cin * k / u))
    b = _randn(rng, cout, scale=0.1)
    x = _randn(rng, B, cin, T)
    p = (k - u) // 2
    y_ref = F.conv_transpose1d(F.leaky_relu(x, 0.1), w, b, stride=u, padding=p)
    plan = ops.ConvPlan(w, b, padding=p, transposed=True, stride=u, pre_act=1, pre_slope=0.1, precision=precision)
    y = plan(x.to(DEV)).cpu()
    assert y.shape == y_ref.shape
    assert float((y - y_ref).abs().max()) <= 2e-5 * max(1.0, float(y_ref.abs().max()))


def test_wav_to_int16_matches_numpy_cast():
    rng = _rng(3)
    w = torch.from_numpy(rng.uniform(-1, 1, size=(3, 4097)).astype(np.float32))
    w[0, :4] = torch.tensor([0.0, -1.0, 0.99999, -0.99999])
    got = ops.wav_to_int16(w.to(DEV)).cpu().numpy()
    assert np.array_equal(got, O.to_int16(w))


# ----------------------------------------------------------------------------------------------
# vocoder
# ----------------------------------------------------------------------------------------------
def _voc_cfg(name):
    if name.startswith("voc_full"):
        return synth.default_voc_config()
    if name == "voc_small_corners":  # odd k - u upsampling stages (T u + 1 samples), dilation lists of unequal length
        return synth.corner_voc_config()
    h = synth.small_voc_config()
    if name == "voc_small_singlespk":
        h["multispkr"] = None
        h["model_in_dim"] = h["embedding_dim"]
    if name == "voc_small_resblock2":
        h["resblock"] = "2"
        h["resblock_dilation_sizes"] = [[1, 3], [1, 3], [1, 3]]
    return h


def _gen(h, sd):
    g = CodeGenerator(AttrDict(h))
    g.load_state_dict(sd)
    return g.eval().to(DEV)


VOC_GOLDENS = ["voc_small", "voc_small_singlespk", "voc_small_resblock2", "voc_full_stages", "voc_full_u40", "voc_full_u40_hot",
               "voc_full_u256", "voc_small_corners"]


def _skip_duplicate_mode(prec, fused):
    """Fused modes 1 and 2 differ only in the exact-fp32 32-channel whole-block kernel: under the split schemes they run
    the same kernels, so that third of the matrix would be duplicate work."""
    if fused == 1 and prec != "f32":
        pytest.skip("fused modes 1 and 2 are the same kernels under the split schemes")


@pytest.mark.parametrize("name", VOC_GOLDENS)
def test_vocoder_matches_reference_golden(golden_dir, name, prec, fused):
    _skip_duplicate_mode(prec, fused)
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = json.loads(str(z["meta"]))
    h = _voc_cfg(name)
    sd = synth.synth_voc_state_dict(h, seed=m["seed_w"], scale=m["scale"])
    assert synth.state_digest(sd) == str(z["digest"])
    g = _gen(h, sd)
    code, spkr = torch.from_numpy(z["code"]).to(DEV), torch.from_numpy(z["spkr"]).to(DEV)
    st = {}
    y = g(code=code, spkr=spkr, stages=st)
    g.check_inputs()
    assert y.shape == z["wav"].shape
    for k in z.files:
        if k.startswith("stage_"):
            ref = z[k]
            got = st[k[6:]].cpu().numpy()
            assert np.abs(got - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max()), k
    err = float(np.abs(y.cpu().numpy() - z["wav"]).max())
    if name.endswith("_hot"):
        # Stress case (weights x 1.2, pre-tanh magnitudes ~1e2, rms 0.98): the reference's OWN fp32 result is 1.2e-4 away from
        # exact arithmetic here, so "distance to the fp32 golden" mostly measures the golden.  The yardstick is an fp64 run of
        # the oracle: the HIP waveform may be at most 3x as far from it as the reference's fp32 waveform is (and the distance to
        # the golden at most the sum of the two).
        with torch.no_grad():
            ref64 = O.code_generator_forward({k: v.double() for k, v in sd.items()}, h, torch.from_numpy(z["code"]), torch.from_numpy(z["spkr"])).numpy()
        ref_err = float(np.abs(z["wav"].astype(np.float64) - ref64).max())
        hip_err = float(np.abs(y.cpu().numpy().astype(np.float64) - ref64).max())
        _report(test="vocoder_golden", name=name, precision=prec, fused=int(fused), wav_max_abs_err=err, hip_vs_fp64=hip_err,
                reference_fp32_vs_fp64=ref_err)
        # numeric bounds (round 5): the reference's own fp32 run is 1.04e-4 from exact arithmetic on this case; the HIP path measures
        # 1.3-1.6e-4 (default scheme) ... 2.3e-4 (exact-fp32 MFMA: one product per rounding step), i.e. 1.2-2.2x -- spread evenly over
        # the stages (tests/test_gpu_round5.py::test_hot_golden_error_by_stage: 1.1-1.3x the reference's error after EVERY stage, no
        # single layer stands out) and amplified by pre-tanh magnitudes of ~120.
        assert ref_err <= 1.5e-4, ref_err
        assert hip_err <= 3.0e-4, f"HIP vs fp64 {hip_err} (reference fp32 vs fp64 {ref_err})"
        assert err <= 4.0e-4, f"waveform max-abs error {err} vs the fp32 golden (reference's own error {ref_err})"
    else:
        _report(test="vocoder_golden", name=name, precision=prec, fused=int(fused), wav_max_abs_err=err)
        assert err <= 5e-5, f"waveform max-abs error {err}"
    # int16 PCM as the reference driver emits it: allow +-1 LSB where the fp32 error straddles an integer
    pcm = ops.wav_to_int16(y.squeeze(1)).cpu().numpy().astype(np.int32)
    assert np.abs(pcm - z["wav_int16"].astype(np.int32)).max() <= (20 if name.endswith("_hot") else 2)
    # weight-norm removed checkpoint (plain `weight` keys) gives the same waveform bit-for-bit
    g.remove_weight_norm()
    assert not any(k.endswith("weight_g") for k in g.state_dict())
    y2 = g(code=code, spkr=spkr)
    assert torch.equal(y, y2)


def test_vocoder_matches_oracle_ragged_shapes(prec, fused):
    _skip_duplicate_mode(prec, fused)
    h = synth.small_voc_config()
    sd = synth.synth_voc_state_dict(h, seed=21, scale=1.0)
    g = _gen(h, sd)
    for B, U in [(1, 1), (2, 3), (5, 31), (1, 257)]:
        batch = synth.synth_voc_batch(B, U, h, seed=B * 100 + U)
        with torch.no_grad():
            ref = O.code_generator_forward(sd, h, batch["code"], batch["spkr"])
        y = g(code=batch["code"].to(DEV), spkr=batch["spkr"].to(DEV)).cpu()
        assert y.shape == ref.shape
        assert float((y - ref).abs().max()) <= 5e-5
    # batch rows are independent: row b of a batch equals the same utterance run alone
    batch = synth.synth_voc_batch(4, 20, h, seed=77)
    yb = g(code=batch["code"].to(DEV), spkr=batch["spkr"].to(DEV))
    y1 = g(code=batch["code"][2:3].to(DEV), spkr=batch["spkr"][2:3].to(DEV))
    assert torch.equal(yb[2:3], y1)


def test_vocoder_flags_a_non_finite_waveform():
    """fp16x3 needs |activation| < 8190: a checkpoint that blows through it must not produce silently wrong audio -- the
    waveform goes non-finite and check_inputs() raises; the bf16x6 mode (fp32's range) synthesises the same model fine."""
    h = synth.small_voc_config()
    sd = synth.synth_voc_state_dict(h, seed=3)
    sd["conv_pre.bias"] = sd["conv_pre.bias"] + 3.0e4     # drives the first activations far beyond the fp16 range
    code = torch.zeros(1, 8, dtype=torch.int64, device=DEV)
    spk = torch.zeros(1, 1, dtype=torch.int64, device=DEV)
    g = _gen(h, sd)
    g.range_fallback = False  # (the default would rebuild the handle in bf16x6 after this first forward: tests/test_gpu_round4.py)
    y = g(code=code, spkr=spk)
    assert not bool(torch.isfinite(y).all())
    with pytest.raises(FloatingPointError):
        g.check_inputs()
    ops.set_default_precision(ops.PREC_BF16X6)
    try:
        g2 = _gen(h, sd)
        y2 = g2(code=code, spkr=spk)
        g2.check_inputs()
        assert bool(torch.isfinite(y2).all())
    finally:
        ops.set_default_precision(ops.PREC_DEFAULT)


def test_vocoder_rejects_bad_ids_and_cpu_tensors():
    h = synth.small_voc_config()
    g = _gen(h, synth.synth_voc_state_dict(h, seed=1))
    with pytest.raises(RuntimeError):
        g(code=torch.zeros(1, 4, dtype=torch.int64), spkr=torch.zeros(1, 1, dtype=torch.int64))
    code = torch.full((1, 4), h["num_embeddings"], dtype=torch.int64, device=DEV)
    g(code=code, spkr=torch.zeros(1, 1, dtype=torch.int64, device=DEV))
    with pytest.raises(IndexError):
        g.check_inputs()


# ----------------------------------------------------------------------------------------------
# TTE
# ----------------------------------------------------------------------------------------------
TTE_CASES = {
    "tte_small_ragged": synth.small_tte_config,
    "tte_small_multi": synth.small_tte_config,
    "tte_full_ragged": synth.default_tte_config,
    "tte_full_forced": synth.default_tte_config,
}


def _parrot(cfg, vocab, n_spk, sd, tmp_path):
    cfg = synth.clone_config(cfg)
    cfg["path"]["root_path"] = str(tmp_path)
    with open(os.path.join(str(tmp_path), "speakers.json"), "w") as f:
        json.dump({f"s{i}": i for i in range(n_spk)}, f)
    m = Parrot(cfg, vocab, 0)
    m.load_state_dict(sd)
    return m.eval().to(DEV)


@pytest.mark.parametrize("name", list(TTE_CASES))
def test_tte_matches_reference_golden(golden_dir, tmp_path, name, prec):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    m = json.loads(str(z["meta"]))
    cfg = TTE_CASES[name]()
    sd = synth.synth_tte_state_dict(cfg, m["vocab"], m["n_spk"], seed=m["seed_w"], forced_duration=m["forced"], gain=m["gain"])
    assert synth.state_digest(sd) == str(z["digest"])
    synth.patch_pe_rows(sd, z["pe_idx"], z["pe_rows"])
    model = _parrot(cfg, m["vocab"], m["n_spk"], sd, tmp_path)
    batch = {"phones": torch.from_numpy(z["phones"]).to(DEV), "src_mask": torch.from_numpy(z["src_mask"]).to(DEV),
             "speaker": torch.from_numpy(z["speaker"]).to(DEV)}
    logits, _, tgt_mask, log_dur = model(batch, inference=True)
    n_ = z["logits_head"].shape[0]
    _report(test="tte_golden", name=name, precision=prec, log_dur_max_abs_err=float(np.abs(log_dur.cpu().numpy() - z["log_dur"]).max()),
            logits_max_abs_err=float(np.abs(logits[:n_].cpu().numpy() - z["logits_head"])[z["tgt_mask"][:n_]].max()),
            min_margin=float(z["margin"][z["tgt_mask"]].min()))
    assert np.abs(log_dur.cpu().numpy() - z["log_dur"]).max() <= 2e-5
    assert np.array_equal(tgt_mask.cpu().numpy(), z["tgt_mask"])
    r = model.infer_dense(batch)
    safe = z["half_dist"] > 1e-4
    assert np.array_equal(r["dur"].cpu().numpy()[safe], z["dur"][safe])
    assert np.array_equal(r["dur"].cpu().numpy(), z["dur"])  # all goldens are clear of rounding ties
    n = z["logits_head"].shape[0]
    msk = z["tgt_mask"][:n]
    assert np.abs(logits[:n].cpu().numpy() - z["logits_head"])[msk].max() <= 1e-4
    ids = r["ids"].cpu().numpy()
    decided = z["tgt_mask"] & (z["margin"] > 1e-4)
    assert np.array_equal(ids[decided], z["ids"][decided])
    assert decided[z["tgt_mask"]].mean() > 0.999
    rag = model.infer(batch)
    for b, row in enumerate(rag):
        ln = int(z["ids_ragged_len"][b])
        assert row == z["ids_ragged"][b, :ln].tolist()  # includes the Q2 extra id per short row


def _enc_out(sd, cfg, batch):
    tr = cfg["transformer"]
    out = O.pos_emb(sd["pos_emb.pe"], F.embedding(batch["phones"], sd["tok_emb.weight"]))
    for n in range(tr["encoder"]["n_layer"]):
        out = O.fft_block(sd, f"encoder_layers.{n}.", out, tr["encoder"]["n_head"], tr["conv_kernel_sizes"], ~batch["src_mask"])
    if "speaker_emb.weight" in sd:
        out = out + F.embedding(batch["speaker"], sd["speaker_emb.weight"]).unsqueeze(1)
    return out


def test_tte_matches_oracle_other_shapes(tmp_path, prec):
    cfg = synth.small_tte_config()
    for (B, S, n_spk, seed) in [(1, 1, 1, 0), (1, 2, 1, 4), (2, 5, 2, 1), (7, 33, 3, 2), (3, 70, 2, 3)]:
        d = tmp_path / f"c{seed}"
        d.mkdir()
        sd = synth.synth_tte_state_dict(cfg, 30, n_spk, seed=seed)
        model = _parrot(cfg, 30, n_spk, sd, d)
        batch = synth.synth_tte_batch(B, S, 30, n_spk, seed=seed + 10, ragged=True)
        gb = {k: v.to(DEV) for k, v in batch.items()}
        with torch.no_grad():
            enc_only = O.durations_from_log(O.duration_predictor(sd, _enc_out(sd, cfg, batch), ~batch["src_mask"], 3))
        if int(enc_only.sum(1).max()) == 0:
            # every duration rounds to 0: the reference dies inside Conv1d/MHA on an empty sequence;
            # the HIP path reports it as an error instead of launching empty grids
            with pytest.raises(Exception):
                model.infer(gb)
            continue
        with torch.no_grad():
            ref = O.tte_forward(sd, cfg, batch)
            ref_ids = O.tte_infer(sd, cfg, batch)
        logits, _, tgt_mask, log_dur = model(gb, inference=True)
        assert float((log_dur.cpu() - ref["log_dur"]).abs().max()) <= 2e-5
        assert torch.equal(tgt_mask.cpu(), ref["tgt_mask"])
        m = ref["tgt_mask"]
        assert float((logits.cpu() - ref["logits"])[m].abs().max()) <= 1e-4
        assert model.infer(gb) == ref_ids


def test_tte_flags_non_finite_logits(tmp_path):
    """An activation beyond the fp16 split range must not turn into silently wrong unit ids: infer() raises."""
    cfg = synth.small_tte_config()
    sd = synth.synth_tte_state_dict(cfg, 30, 1, seed=2, forced_duration=2)
    sd["decoder_layers.0.convlayer.conv1.bias"] = sd["decoder_layers.0.convlayer.conv1.bias"] + 5.0e4
    model = _parrot(cfg, 30, 1, sd, tmp_path)
    batch = {k: v.to(DEV) for k, v in synth.synth_tte_batch(2, 9, 30, 1, seed=1).items()}
    model.range_fallback = False  # (the fail-loud contract; by default the first decode falls back to bf16x6, checked below)
    with pytest.raises(FloatingPointError):
        model.infer(batch)
    m1 = _parrot(cfg, 30, 1, sd, tmp_path)
    with pytest.warns(RuntimeWarning, match="bf16x6"):
        rows = m1.infer(batch)  # range-safe fallback: the handle is rebuilt in bf16x6 and the batch re-run
    assert len(rows) == 2 and m1.precision_in_use == "bf16x6"
    ops.set_default_precision(ops.PREC_BF16X6)
    try:
        m2 = _parrot(cfg, 30, 1, sd, tmp_path)
        assert m2.infer(batch) == rows
    finally:
        ops.set_default_precision(ops.PREC_DEFAULT)


def test_tte_error_behaviour(tmp_path):
    cfg = synth.small_tte_config()
    sd = synth.synth_tte_state_dict(cfg, 30, 1, seed=0)
    model = _parrot(cfg, 30, 1, sd, tmp_path)
    batch = {k: v.to(DEV) for k, v in synth.synth_tte_batch(1, cfg["transformer"]["max_len"], 30, 1, seed=0).items()}
    with pytest.raises(IndexError):      # pe[T] out of range, reference modules/fft.py:18
        model.infer(batch)
    bad = {k: v.to(DEV) for k, v in synth.synth_tte_batch(1, 4, 30, 1, seed=0).items()}
    bad["phones"][0, 1] = 30
    with pytest.raises(IndexError):      # nn.Embedding IndexError in the reference
        model.infer(bad)
    model.train()
    with pytest.raises(AssertionError):  # reference modules/parrot.py:113
        model.infer(bad)
    with pytest.raises(NotImplementedError):
        model(bad, inference=False)


# ----------------------------------------------------------------------------------------------
# BASELINE.json sizes: size-independent properties (the CPU oracle is too slow to referee B=64 x 256 units)
# ----------------------------------------------------------------------------------------------
def test_full_size_pipeline_properties(tmp_path):
    """Batch 64 x 256 units, full-size models (BASELINE config 3 shapes): every row emits exactly 256 ids with
    forced durations; rows are independent in the vocoder (row b of the batch == the same utterance alone, bit for
    bit); the TTE row equals the same row inside a smaller batch of the SAME padded shape; a sampled row matches
    the CPU oracle within the stated tolerance; output is finite and inside (-1, 1)."""
    from parrot_tts_amd.pipeline import SynthesisPipeline
    cfg, h = synth.default_tte_config(), synth.default_voc_config()
    vocab, n_spk, B, S = 300, 10, 64, 64
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=42, forced_duration=4)
    vsd = synth.synth_voc_state_dict(h, seed=1234, scale=1.0)
    parrot = _parrot(cfg, vocab, n_spk, tsd, tmp_path)
    gen = _gen(h, vsd)
    batch = synth.synth_tte_batch(B, S, vocab, n_spk, seed=0)
    gb = {k: v.to(DEV) for k, v in batch.items()}
    out = SynthesisPipeline(parrot, gen)(gb)
    ids, wav = out["ids"], out["wav"]
    assert ids.shape == (B, 4 * S) and wav.shape == (B, 1, 4 * S * 320)
    assert bool(out["tgt_mask"].all()) and out["lens"].tolist() == [4 * S] * B
    assert int(out["n_samples"].min()) == 4 * S * 320
    assert bool(torch.isfinite(wav).all()) and float(wav.abs().max()) <= 1.0
    # TTE: rows 8..15 as their own batch of the same padded shape -> identical ids (no cross-row op anywhere)
    sub = {k: v[8:16].contiguous() for k, v in gb.items()}
    assert torch.equal(parrot.infer_dense(sub)["ids"], ids[8:16])
    # vocoder: row 5 alone == row 5 of the batch, bit for bit
    spk = gb["speaker"].reshape(-1, 1)
    assert torch.equal(gen(code=ids[5:6].contiguous(), spkr=spk[5:6].contiguous()), wav[5:6])
    # one row against the CPU oracle (TTE ids exact where the margin allows; vocoder waveform within tolerance)
    with torch.no_grad():
        one = {k: v[3:4] for k, v in batch.items()}
        ref = O.tte_forward(tsd, cfg, one)
        top2 = torch.topk(ref["logits"], 2, dim=-1).values
        decided = (top2[..., 0] - top2[..., 1]) > 1e-4
        # B=1 and B=64 share the padded shape (S=64, L=256), so the reference gives the same row either way
        assert torch.equal(ids[3:4].cpu()[decided], torch.argmax(ref["logits"], -1)[decided])
        ref_wav = O.code_generator_forward(vsd, h, ids[3:4].cpu(), batch["speaker"][3:4].reshape(-1, 1))
    err = float((wav[3:4].cpu() - ref_wav).abs().max())
    _report(test="full_size_row_vs_oracle", wav_max_abs_err=err, decided_frac=float(decided.float().mean()))
    assert err <= 5e-5


def test_long_form_and_odd_lengths(tmp_path):
    """BASELINE config 5 shape class: long sequences (no chunk streaming needed: the whole utterance fits the
    workspace) and lengths that are not multiples of any tile size, against the CPU oracle (reduced-width models so
    the oracle stays fast; the kernels and tilings are the same)."""
    h = synth.small_voc_config()
    vsd = synth.synth_voc_state_dict(h, seed=4)
    gen = _gen(h, vsd)
    for B, U in [(2, 1500), (1, 1237)]:
        batch = synth.synth_voc_batch(B, U, h, seed=U)
        with torch.no_grad():
            ref = O.code_generator_forward(vsd, h, batch["code"], batch["spkr"])
        y = gen(code=batch["code"].to(DEV), spkr=batch["spkr"].to(DEV)).cpu()
        assert y.shape == (B, 1, U * 320)
        assert float((y - ref).abs().max()) <= 5e-5
    cfg = synth.small_tte_config()
    cfg["transformer"]["max_len"] = 1700
    tsd = synth.synth_tte_state_dict(cfg, 40, 2, seed=8, forced_duration=4)
    model = _parrot(cfg, 40, 2, tsd, tmp_path)
    tb = synth.synth_tte_batch(2, 375, 40, 2, seed=9, ragged=True)   # L = 1500 for the full row
    with torch.no_grad():
        ref = O.tte_forward(tsd, cfg, tb)
        ref_rows = O.tte_infer(tsd, cfg, tb)
    assert ref["logits"].shape[1] == 1500
    got = model.infer({k: v.to(DEV) for k, v in tb.items()})
    top2 = torch.topk(ref["logits"], 2, dim=-1).values
    margin_ok = ((top2[..., 0] - top2[..., 1]) > 1e-4) | ~ref["tgt_mask"]
    if bool(margin_ok.all()):
        assert got == ref_rows
    else:  # compare only decided positions
        for g, r, ok, m in zip(got, ref_rows, margin_ok, ref["tgt_mask"]):
            okm = ok[m].tolist()
            assert len(g) == len(r) and all(a == b for a, b, o in zip(g, r, okm) if o)


@pytest.mark.parametrize("cfg_name", ["small", "full"])
def test_vocoder_ragged_batch_rows_equal_single_utterance_runs(cfg_name, prec, fused):
    """A padded (ragged) batch with per-row unit counts: every row must equal the reference's own run of that utterance
    ALONE (the reference vocoder driver is B=1), although the rows are padded with arbitrary codes."""
    _skip_duplicate_mode(prec, fused)
    h = synth.small_voc_config() if cfg_name == "small" else synth.default_voc_config()
    vsd = synth.synth_voc_state_dict(h, seed=31)
    gen = _gen(h, vsd)
    lens = [37, 11, 1, 24] if cfg_name == "small" else [21, 9]
    U = max(lens)
    batch = synth.synth_voc_batch(len(lens), U, h, seed=5)  # positions beyond lens[b] hold arbitrary (valid) codes
    y = gen(code=batch["code"].to(DEV), spkr=batch["spkr"].to(DEV), unit_lens=torch.tensor(lens)).cpu()
    hop = 320
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref = O.code_generator_forward(vsd, h, batch["code"][b:b + 1, :n], batch["spkr"][b:b + 1])
        err = float((y[b:b + 1, :, : n * hop] - ref).abs().max())
        assert err <= 5e-5, (b, n, err)
    # without unit_lens the padded tail leaks into the last ~20 units of a short row (that is the point of the option)
    y_plain = gen(code=batch["code"].to(DEV), spkr=batch["spkr"].to(DEV)).cpu()
    n = lens[1]
    assert float((y_plain[1, :, : n * hop] - y[1, :, : n * hop]).abs().max()) > 1e-3


def test_vocoder_fuzz_full_size_ragged_rows():
    """Seeded random (batch, units, per-row lengths) on the FULL-SIZE generator in the default mode: every row of the padded
    batch against the oracle run of that utterance alone -- window / halo / edge handling of all fused and layer kernels at
    lengths that fall anywhere relative to their tiles."""
    h = synth.default_voc_config()
    vsd = synth.synth_voc_state_dict(h, seed=1234)
    gen = _gen(h, vsd)
    rng = _rng(77)
    for case in range(5):
        B = int(rng.integers(1, 4))
        U = int(rng.choice([2, 7, 13, 26, 41, 58]))
        lens = [U] + [int(rng.integers(1, U + 1)) for _ in range(B - 1)]
        b = synth.synth_voc_batch(B, U, h, seed=100 + case)
        y = gen(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV), unit_lens=torch.tensor(lens)).cpu()
        for r, n in enumerate(lens):
            with torch.no_grad():
                ref = O.code_generator_forward(vsd, h, b["code"][r:r + 1, :n], b["spkr"][r:r + 1])
            err = float((y[r:r + 1, :, : n * 320] - ref).abs().max())
            assert err <= 5e-5, (case, B, U, lens, r, err)
    gen.check_inputs()


def test_chunk_streamed_vocoder_equals_whole_utterance(prec):
    """SURVEY section 4 item 6 / BASELINE config 5: U = 200 whole vs chunk-streamed (64-unit chunks, receptive-field
    halo) on the full-size generator; also ragged rows and a chunk size that does not divide U.  Interior chunks see
    +-20 units of real context, edge chunks contain the true edge, so the two must agree to fp32 round-off: 2.4e-7
    measured with the exact kernels; in the split-bf16 mode a short last chunk runs other tile variants (different
    summation order), which the network depth amplifies to <= 1e-5 -- the same class as the parity error itself."""
    h = synth.default_voc_config()
    vsd = synth.synth_voc_state_dict(h, seed=1234)
    gen = _gen(h, vsd)
    batch = synth.synth_voc_batch(2, 200, h, seed=5)
    code, spkr = batch["code"].to(DEV), batch["spkr"].to(DEV)
    whole = gen(code=code, spkr=spkr)
    for chunk in (64, 77):
        got = gen.forward_chunked(chunk_units=chunk, code=code, spkr=spkr)
        err = float((got - whole).abs().max())
        _report(test="voc_chunk_stream", chunk_units=chunk, precision=prec, max_abs_diff=err)
        assert err <= (2e-6 if prec == "f32" else 2e-5)
    firsts = [f for f, _ in gen.stream(64, code=code, spkr=spkr)]
    assert firsts == [0, 64 * 320, 128 * 320, 192 * 320]
    lens = torch.tensor([200, 131])
    whole_r = gen(code=code, spkr=spkr, unit_lens=lens.to(DEV))
    got_r = gen.forward_chunked(chunk_units=64, code=code, spkr=spkr, unit_lens=lens.to(DEV))
    for b in range(2):
        n = int(lens[b]) * 320
        assert float((got_r[b, :, :n] - whole_r[b, :, :n]).abs().max()) <= (2e-6 if prec == "f32" else 2e-5)
    # a halo shorter than the receptive field must NOT be exact (the test would otherwise prove nothing)
    short = gen.forward_chunked(chunk_units=64, halo_units=2, code=code, spkr=spkr)
    assert float((short - whole).abs().max()) > 1e-3


def test_text_to_waveform_helper_matches_manual_pipeline(tmp_path):
    """demo.ipynb cells 9-13 via parrot_tts_amd.text.synthesize_text: same ids as Parrot.infer on the hand-built batch,
    and the per-speaker fan-out run as one vocoder batch equals the per-speaker runs; ids also match the CPU oracle."""
    from parrot_tts_amd import text as T
    cfg, h = synth.small_tte_config(), synth.small_voc_config()
    symbols = ["क", "ख", "ग", " ", "ा", "ि", ".", "म", "न"]
    toks = ["<pad>", "<sep>"] + ["sil" if s == " " else s for s in symbols]

    class Tok:
        pad_idx = 0
        stoi = {s: i for i, s in enumerate(toks)}

        def tokenize(self, seq):
            return [self.stoi[s] for s in seq]

    vocab = len(toks)
    tsd = synth.synth_tte_state_dict(cfg, vocab, 2, seed=21, forced_duration=3)
    for k in list(tsd):  # the small vocoder knows 100 units: keep the head inside that range
        if k.endswith("head.weight") or k.endswith("head.bias"):
            tsd[k] = tsd[k].clone()
            tsd[k][100:] = -10.0 if k.endswith("bias") else 0.0
    vsd = synth.synth_voc_state_dict(h, seed=22)
    parrot = _parrot(cfg, vocab, 2, tsd, tmp_path)
    gen = _gen(h, vsd)
    units, wav = T.synthesize_text("का खिग! मन | ३", parrot, gen, Tok(), symbols, speaker=1, vocoder_speakers=(0, 3, 7))
    chars = T.text_to_characters(T.indic_cleaners("का खिग! मन | ३"), symbols)
    batch = T.characters_to_batch(Tok(), chars, speaker=1)
    assert units == parrot.infer({k: v.to(DEV) for k, v in batch.items()})[0]
    with torch.no_grad():
        assert units == O.tte_infer(tsd, cfg, batch)[0]
    assert wav.shape == (3, 1, len(units) * 320)
    code = torch.tensor([units], device=DEV)
    for i, s in enumerate((0, 3, 7)):
        one = gen(code=code, spkr=torch.tensor([[s]], device=DEV))
        assert float((one[0] - wav[i]).abs().max()) <= 2e-6


def test_pipelined_submit_equals_sequential_calls(tmp_path):
    """SynthesisPipeline.submit / flush (TTE of batch i on a side stream beside the vocoder of batch i-1) returns, one call
    late and in order, exactly what __call__ returns for each batch -- different shapes per batch on purpose."""
    from parrot_tts_amd.pipeline import SynthesisPipeline
    cfg, h = synth.small_tte_config(), synth.small_voc_config()
    vocab, n_spk = 30, 2
    tsd = synth.synth_tte_state_dict(cfg, vocab, n_spk, seed=31)
    for k in list(tsd):
        if k.endswith("head.weight") or k.endswith("head.bias"):
            tsd[k] = tsd[k].clone()
            tsd[k][100:] = -10.0 if k.endswith("bias") else 0.0
    vsd = synth.synth_voc_state_dict(h, seed=32)
    pipe = SynthesisPipeline(_parrot(cfg, vocab, n_spk, tsd, tmp_path), _gen(h, vsd))
    batches = [{k: v.to(DEV) for k, v in synth.synth_tte_batch(B, S, vocab, n_spk, seed=40 + i, ragged=True).items()}
               for i, (B, S) in enumerate([(3, 11), (2, 17), (4, 9), (3, 11)])]
    want = []
    for b in batches:
        r = pipe(b)
        want.append({k: r[k].clone() for k in ("wav", "ids", "n_samples")})
    got = []
    for b in batches:
        out = pipe.submit(b)
        if out is not None:
            got.append(out)
    got.append(pipe.flush())
    assert pipe.flush() is None and len(got) == len(want)
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g["ids"], w["ids"]) and torch.equal(g["n_samples"].cpu(), w["n_samples"].cpu())
        for row in range(w["wav"].shape[0]):
            n = int(w["n_samples"][row])
            assert torch.equal(g["wav"][row, :, :n], w["wav"][row, :, :n])


def test_vocoder_extra_conditioning_streams(golden_dir, prec):
    """Extra keyword tensors of CodeGenerator.forward (models.py:162-167) through parrot_voc_forward_feats, against the
    reference golden; `f0` is ignored as in the reference; a stream that does not tile U raises like the reference."""
    z = np.load(os.path.join(golden_dir, "voc_small_feats.npz"))
    m = json.loads(str(z["meta"]))
    h = synth.clone_config(synth.small_voc_config())
    h["model_in_dim"] += m["extra_channels"]
    sd = synth.synth_voc_state_dict(h, seed=m["seed_w"])
    assert synth.state_digest(sd) == str(z["digest"])
    g = _gen(h, sd)
    kw = dict(code=torch.from_numpy(z["code"]).to(DEV), spkr=torch.from_numpy(z["spkr"]).to(DEV), f0=torch.from_numpy(z["f0"]).to(DEV),
              energy=torch.from_numpy(z["energy"]).to(DEV), style=torch.from_numpy(z["style"]).to(DEV))
    y = g(**kw).cpu().numpy()
    err = float(np.abs(y - z["wav"]).max())
    _report(test="voc_extra_feats", precision=prec, max_abs_err=err)
    assert err <= 5e-5
    with pytest.raises(NotImplementedError):  # 20 = 2 * 7 + 6 (models.py:146-148)
        g(code=kw["code"], spkr=kw["spkr"], energy=torch.zeros(2, 2, 7, device=DEV), style=kw["style"])
    with pytest.raises(RuntimeError):         # 20 = 6 * 3 + 2: the reference fails in .view() there
        g(code=kw["code"], spkr=kw["spkr"], energy=torch.zeros(2, 2, 3, device=DEV), style=kw["style"])
    with pytest.raises(Exception):  # the model expects 3 extra channels
        g(code=kw["code"], spkr=kw["spkr"])
    g.check_inputs()


def test_full_vocoder_on_tiny_sequences(prec):
    """Sequences far shorter than any tile / fused window (1-7 units = 320-2240 samples) and ragged rows down to one unit,
    on the full-size generator: every layer's window, halo and edge handling with T < W, against the oracle."""
    h = synth.default_voc_config()
    vsd = synth.synth_voc_state_dict(h, seed=1234)
    gen = _gen(h, vsd)
    for B, U in [(1, 1), (2, 2), (1, 3), (3, 5)]:
        b = synth.synth_voc_batch(B, U, h, seed=U)
        with torch.no_grad():
            ref = O.code_generator_forward(vsd, h, b["code"], b["spkr"])
        y = gen(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV)).cpu()
        assert float((y - ref).abs().max()) <= 5e-5
    b = synth.synth_voc_batch(3, 9, h, seed=3)
    lens = torch.tensor([9, 1, 4])
    y = gen(code=b["code"].to(DEV), spkr=b["spkr"].to(DEV), unit_lens=lens.to(DEV)).cpu()
    for r in range(3):
        n = int(lens[r])
        with torch.no_grad():
            ref = O.code_generator_forward(vsd, h, b["code"][r:r + 1, :n], b["spkr"][r:r + 1])
        assert float((y[r:r + 1, :, :n * 320] - ref).abs().max()) <= 5e-5
