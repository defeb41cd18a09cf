"""GPU: the library's runtime switches are read once per process, so each alternative code path is
exercised in its own subprocess on a slice of the parity suite: full-size vocoder + TTE goldens and the ragged-row test.
Every switch the library still reads (PARROT_PRECISION, _FUSED, _MRF_STREAMS, _SMALL_TILES, _MFMA16, _FLASH_ATTN, _TTE_MERGE,
_VALU_KERNELS, _TIE_GUARD) appears below with its non-default value (the precision modes also run in-process in the parity tests)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SLICE = "voc_full_u40 or voc_small_resblock2 or tte_full_forced or ragged_batch_rows"

SWITCHES = [
    {"PARROT_MRF_STREAMS": "1"},                       # MRF branches on one stream (the small-batch default is three)
    {"PARROT_MRF_STREAMS": "3", "PARROT_SMALL_TILES": "0"},  # forced branch streams, no small tiles for under-filled launches
    {"PARROT_MFMA16": "0", "PARROT_FLASH_ATTN": "0"},  # 32x32x16 layer kernel everywhere, fp32-MFMA attention cores
    {"PARROT_TTE_MERGE": "0", "PARROT_VALU_KERNELS": "0"},
    {"PARROT_FUSED": "0", "PARROT_TIE_GUARD": "0"},    # every ResBlock layer by layer, tie guard off
    {"PARROT_FUSED": "1", "PARROT_PRECISION": "f32"},  # exact fp32 MFMA incl. the fused 32-channel whole-block kernel
]


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join(f"{k[7:]}={v}" for k, v in e.items()))
def test_parity_slice_under_switch(env):
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k", SLICE,
                          "-p", "no:cacheprovider"], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, **env), timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    assert " passed" in out.stdout and "failed" not in out.stdout
