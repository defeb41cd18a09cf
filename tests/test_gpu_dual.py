"""The dual-window anti-phase fused ResBlock kernels (csrc/resblock_dual.h) against the one-window kernels they replace
(csrc/resblock_split.h): same arithmetic in the same order, so the waveforms must agree BIT FOR BIT -- whole-generator runs at
full width (64-, 32- and 16-channel stages, k = 3 / 7 / 11), interior and sequence-edge windows, ragged rows, odd window counts."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

_CHILD = r"""
import sys, torch
sys.path.insert(0, {root!r})
from parrot_tts_amd import synth
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator
h = synth.default_voc_config()
sd = synth.synth_voc_state_dict(h, seed=5, scale=1.0)
g = CodeGenerator(AttrDict(h)); g.load_state_dict(sd); g = g.eval().to("cuda:0")
out = {{}}
for B, U in {shapes!r}:
    batch = synth.synth_voc_batch(B, U, h, seed=B * 1000 + U)
    lens = torch.tensor([max(1, U - 7 * i) for i in range(B)])
    out[(B, U)] = g(code=batch["code"].to("cuda:0"), spkr=batch["spkr"].to("cuda:0"), unit_lens=lens.to("cuda:0")).cpu()
g.check_inputs()
torch.save(out, {path!r})
"""

SHAPES = [(1, 3), (3, 41), (2, 130), (5, 77), (40, 64)]  # (the last: more window pairs than CUs -> persistent workgroups loop)


def _run(dual, path):
    env = dict(os.environ, PARROT_RB_DUAL=str(dual))
    subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, shapes=SHAPES, path=path)], check=True, env=env, timeout=600)
    return torch.load(path)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_dual_window_kernels_are_bit_identical_to_the_one_window_kernels(tmp_path, mode):
    """PARROT_RB_DUAL=1: lean-VALU kernels, one window per workgroup; =2: dual-window anti-phase workgroups; =3: the persistent
    dual-window kernel (resblock_pdual.h); 0: resblock_split.h."""
    a = _run(mode, str(tmp_path / "dual.pt"))
    b = _run(0, str(tmp_path / "single.pt"))
    for key in a:
        assert torch.isfinite(a[key]).all()
        assert torch.equal(a[key], b[key]), f"mode {mode} waveform differs from the round-2 kernels at (B, U) = {key}: max {float((a[key] - b[key]).abs().max())}"
