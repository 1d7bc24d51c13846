"""One-rank RCCL sanity check of the exchange used by host/sharding.py (run on a GPU box):
python tools/rccl_sanity.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from webgpu_msm_bls12_377_amd.host.sharding import ShardedMsm
sh = ShardedMsm(0, 1, device=torch.device("cuda", 0))
# force the collective path with a fake 2-rank layout on one rank is impossible; exercise the tensors instead
send = torch.arange(3328 * 8, dtype=torch.uint8, device="cuda") % 251
recv = torch.empty_like(send)
dist.all_gather_into_tensor(recv, send)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("rccl ok", bool((recv == send).all()), float(t.item()))
dist.destroy_process_group()
