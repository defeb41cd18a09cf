"""RCCL sanity with ONE rank (the test box has one GPU): the primitives of parrot_tts_amd.dist's N > 1 path on this torch / RCCL stack."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from parrot_tts_amd import dist as pdist
w = torch.randn(4, 1, 1000, device="cuda:0")
# the code path of gather_waveforms for world > 1 needs world > 1; exercise the same primitives on RCCL with one rank
out = [torch.empty(4, 1000, device="cuda:0")]
dist.gather(w.reshape(4, 1000).contiguous(), out, dst=0)
assert torch.equal(out[0], w.reshape(4, 1000))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
g = dist.new_group(backend="gloo"); a = [torch.empty(2, dtype=torch.int64)]; dist.all_gather(a, torch.tensor([3, 4]), group=g)
print("rccl single-rank primitives ok", float(t), a[0].tolist())
dist.destroy_process_group()
