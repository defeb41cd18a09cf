"""GPU tests of the round-4 changes: chunk workspace sizing across the branch-stream threshold, the range-safe precision
fallback, the extended tie guard, N > 1 rehearsal of bench.py on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import parrot_oracle as O  # noqa: E402
from parrot_tts_amd import synth  # noqa: E402
from parrot_tts_amd.vocoder import AttrDict, CodeGenerator  # noqa: E402

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen(h, sd):
    g = CodeGenerator(AttrDict(h))
    g.load_state_dict(sd)
    return g.eval().to(DEV)


def test_chunked_workspace_covers_a_trailing_chunk_below_the_branch_stream_threshold():
    """the branch-stream count follows B x U (three concurrent MRF branches up to 8192 units, one above),
    and the chunked path sized its inner workspace for the full span only.  B = 40, U = 415 in 256-unit chunks: span 298 units
    (B x span = 11920 > 8192: one stream, 3 + 3 buffers), trailing chunk 180 units (B x n = 7200 <= 8192: three streams wanted
    3 + 9 buffers) -> PARROT_E_NOMEM before the fix.  Every chunk now runs with the stream count of the allocation; chunked ==
    whole to fp32 round-off."""
    h = synth.default_voc_config()
    g = _gen(h, synth.synth_voc_state_dict(h, seed=1234, scale=1.0))
    b = synth.synth_voc_batch(40, 415, h, seed=9)
    code, spkr = b["code"].to(DEV), b["spkr"].to(DEV)
    whole = g(code=code, spkr=spkr)
    chunked = g.forward_chunked(chunk_units=256, code=code, spkr=spkr)
    torch.cuda.synchronize()
    assert chunked.shape == whole.shape
    assert float((chunked - whole).abs().max()) <= 2e-5
    # ragged rows through the same shape class
    lens = torch.randint(200, 416, (40,), generator=torch.Generator().manual_seed(3)).to(DEV)
    a = g(code=code, spkr=spkr, unit_lens=lens)
    c = g.forward_chunked(chunk_units=256, code=code, spkr=spkr, unit_lens=lens)
    hop = g.upsample_factor
    for r in (0, 17, 39):
        n = int(lens[r]) * hop
        assert float((a[r, :, :n] - c[r, :, :n]).abs().max()) <= 2e-5


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_rehearses_eight_ranks_on_one_gpu():
    """bench.py exactly as the driver launches it for N = 8 (torch.distributed.run, one process per rank), all
    eight ranks on the one visible GPU with gloo carrying the collective: rc 0, ONE JSON line from rank 0 whose `dist` says 8
    ranks, whose global batch is 8 x the per-GPU batch, and whose gather took measurable time."""
    env = dict(os.environ, PARROT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--no-cpu-baseline", "--no-alt"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["dist"]["nranks"] == 8 and res["dist"]["backend"] == "gloo"
    assert res["config"]["global_batch"] == 16 and res["config"]["per_gpu_batch"] == 2
    assert abs(res["value"] * res["ms_per_step"] / 1e3 - 16 * 256 * 320) < 1.0
    assert 0.0 < res["gather_ms"] < res["ms_per_step"]
    assert "10-speaker" in res["config"]["note"]
