"""CPU, 2 processes over gloo: the N>1 path of the pipeline -- batch shard by row, no data-path
collective, one waveform gather (RCCL on the GPU box) and the ragged id gather -- reassembles exactly."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from parrot_tts_amd import dist as pdist, synth
    r, w, _ = pdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    full = synth.synth_tte_batch(6, 9, 40, 3, seed=1, ragged=True)      # same global batch on every rank
    mine = pdist.shard_batch(full, r, w)
    sl = pdist.shard_rows(6, r, w)
    assert torch.equal(mine["phones"], full["phones"][sl]) and mine["phones"].shape[1] == 9  # never re-padded
    # stand-in for the per-rank synthesis: a deterministic function of the row contents
    wav = (mine["phones"].float().sum(1, keepdim=True)[:, :, None] + torch.arange(16.0)[None, None, :]).contiguous()
    rows = [[int(v) for v in p[m]] for p, m in zip(mine["phones"], mine["src_mask"])]
    g = pdist.gather_waveforms(wav, dst=0)
    gr = pdist.gather_ragged_rows(rows, dst=0)
    if r == 0:
        want = full["phones"].float().sum(1, keepdim=True)[:, :, None] + torch.arange(16.0)[None, None, :]
        assert torch.equal(g, want)
        assert gr == [[int(v) for v in p[m]] for p, m in zip(full["phones"], full["src_mask"])]
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    else:
        assert g is None and gr is None
    # ragged shards: 5 rows over 2 ranks (3 + 2) and a data-dependent length per shard (equal-shape
    # receive buffers would hang / corrupt here); twice, so the second call reuses the preallocated receive buffer
    for rep in range(2):
        sl5 = pdist.shard_rows(5, r, w)
        n_r = 12 + 4 * r + rep
        rowsv = torch.arange(sl5.start, sl5.stop, dtype=torch.float32)
        wav_r = (rowsv[:, None, None] * 100 + torch.arange(float(n_r))[None, None, :]).contiguous()
        lens_r = torch.arange(sl5.start, sl5.stop) + 3
        got = pdist.gather_waveforms(wav_r, dst=0, n_samples=lens_r)
        if r == 0:
            gw, gl = got
            nmax = 12 + 4 * (w - 1) + rep
            assert gw.shape == (5, 1, nmax) and gl.tolist() == [3, 4, 5, 6, 7]
            for row in range(5):
                rk = 0 if row < 3 else 1
                n_row = 12 + 4 * rk + rep
                assert torch.equal(gw[row, 0, :n_row], row * 100 + torch.arange(float(n_row)))
                assert float(gw[row, 0, n_row:].abs().sum()) == 0.0  # shorter shards are zero-padded
        else:
            assert got is None
    ge = pdist.gather_waveforms(wav, dst=0, equal_shapes=True)  # fixed-length workloads skip the shape exchange
    if r == 0:
        assert torch.equal(ge, want)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


def _worker8(rank, world, port, out_dir, n_rows):
    """BASELINE configs[3] in shape: 512 utterances (or 509: B % world != 0) sharded over 8 ranks, ragged lengths per row and a
    data-dependent padded length per shard, gathered to rank 0 in rank order -- twice (receive-buffer reuse, latched gather mode)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from parrot_tts_amd import dist as pdist
    r, w, _ = pdist.init_from_env("gloo")
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(3, 41, (n_rows,), generator=g)                      # samples per row (same on every rank)
    full = {"phones": torch.arange(n_rows)[:, None].repeat(1, 4), "speaker": torch.arange(n_rows) % 10, "meta": "kept as is"}
    mine = pdist.shard_batch(full, r, w)
    sl = pdist.shard_rows(n_rows, r, w)
    assert mine["meta"] == "kept as is" and torch.equal(mine["phones"], full["phones"][sl])
    assert sl.stop - sl.start in (n_rows // w, n_rows // w + 1)
    for rep in range(2):
        my_lens = lens[sl]
        n_r = int(my_lens.max())                                              # this shard's padded length (L = max over ITS rows)
        wav = torch.zeros((sl.stop - sl.start, 1, n_r))
        for i, row in enumerate(range(sl.start, sl.stop)):
            wav[i, 0, : int(lens[row])] = row * 1000.0 + rep + torch.arange(float(lens[row]))
        got = pdist.gather_waveforms(wav, dst=0, n_samples=my_lens)
        if r == 0:
            gw, gl = got
            assert gw.shape == (n_rows, 1, max(int(lens[pdist.shard_rows(n_rows, q, w)].max()) for q in range(w)))
            assert torch.equal(gl, lens)
            for row in range(n_rows):
                n = int(lens[row])
                assert torch.equal(gw[row, 0, :n], row * 1000.0 + rep + torch.arange(float(n))), row
                assert float(gw[row, 0, n:].abs().sum()) == 0.0
        else:
            assert got is None
    assert pdist.dist_info()["nranks"] == 8 and pdist.dist_info()["gather"] in ("gather", "allgather")  # latched after the first call
    if r == 0:
        open(os.path.join(out_dir, f"ok{n_rows}"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_shard_and_gather_of_512_and_509_rows(tmp_path):
    """the N = 8 layout of BASELINE configs[3] (512 -> 8 x 64 rows) rehearsed without hardware, plus a row
    count that does not divide (509 -> 64 x 5 + 63 x 3)."""
    for n_rows in (512, 509):
        mp.spawn(_worker8, args=(8, _free_port(), str(tmp_path), n_rows), nprocs=8, join=True)
        assert (tmp_path / f"ok{n_rows}").read_text() == "ok"


def test_single_process_is_a_no_op():
    sys.path.insert(0, ROOT)
    from parrot_tts_amd import dist as pdist
    w = torch.randn(3, 1, 8)
    assert pdist.gather_waveforms(w) is w
    assert pdist.gather_ragged_rows([[1], [2, 3]]) == [[1], [2, 3]]


def _worker_fail_after_latch(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["TORCHELASTIC_RUN_ID"] = "test"  # as under torchrun: a launcher that ends the peers when one rank exits
    from parrot_tts_amd import dist as pdist
    pdist.init_from_env("gloo")
    wav = torch.full((2, 1, 8), float(rank))
    g = pdist.gather_waveforms(wav, dst=0, equal_shapes=True)  # first call: all ranks agree, the mode is latched
    assert pdist.dist_info()["gather"] == "gather"
    if rank == 0:
        assert g.shape == (4, 1, 8)
        open(os.path.join(out_dir, "latched"), "w").write("ok")
    if rank == 1:  # a later failure on ONE rank (new shape, OOM, backend hiccup ...)
        def boom(*a, **k):
            raise RuntimeError("injected gather failure")
        dist.gather = boom
    pdist.gather_waveforms(wav, dst=0, equal_shapes=True)  # rank 0 enters the collective, rank 1 fails before it
    open(os.path.join(out_dir, f"survived{rank}"), "w").write("no")  # (never reached on rank 1; rank 0 is ended by the launcher)


def test_gather_failure_after_the_latch_ends_the_group_fast(tmp_path):
    """once the gather mode is latched there is no per-step agreement, so an error on one rank used to leave its peers
    blocked in the collective.  The failing rank now reports and exits (status 70); the launcher (mp.spawn here, torchrun in
    production) ends the peers at once instead of letting them wait for the backend's watchdog."""
    import time
    from parrot_tts_amd import dist as pdist
    t0 = time.time()
    try:
        mp.spawn(_worker_fail_after_latch, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
        raised = None
    except Exception as e:  # ProcessExitedException: rank 1 ended with GATHER_FATAL_EXIT_CODE
        raised = e
    assert raised is not None and getattr(raised, "exit_code", None) == pdist.GATHER_FATAL_EXIT_CODE, raised
    assert (tmp_path / "latched").read_text() == "ok"
    assert not (tmp_path / "survived1").exists()
    assert time.time() - t0 < 120
