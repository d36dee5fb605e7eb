"""GPU-box helper (under torchrun, one rank, backend nccl = RCCL): every collective getPOA and bench.py issue on the
N > 1 path, executed once on a world of one -- the 8-GPU node is not ours to use, the calls at least are."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.arange(9, dtype=torch.int64, device=dev)
dist.broadcast(t, src=0)                                     # alignment._shard: the shard bounds
tot = torch.tensor([3, 4], dtype=torch.int64, device=dev)
dist.all_reduce(tot)                                         # small / wrong read totals
out = [None]
dist.gather_object(("headers", [0, 1], True, None), out, dst=0)
pad = torch.ones((5, 26), dtype=torch.int64, device=dev)
got = [torch.zeros((5, 26), dtype=torch.int64, device=dev)]
dist.gather(pad, got, dst=0)                                 # distributed.gather_rows
sizes = [torch.zeros(1, dtype=torch.int64, device=dev)]
dist.all_gather(sizes, torch.tensor([5], dtype=torch.int64, device=dev))
objs = [None]
dist.all_gather_object(objs, {"rank": 0})                    # bench.py: what the process group spans
tm = torch.tensor([1.5, 2.5], dtype=torch.float64, device=dev)
dist.all_reduce(tm, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert t.tolist() == list(range(9)) and tot.tolist() == [3, 4] and out[0][0] == "headers" and bool((got[0] == 1).all()) \
    and int(sizes[0]) == 5 and objs[0]["rank"] == 0 and tm.tolist() == [1.5, 2.5]
print("RCCL collectives of the N > 1 path: all executed (world 1), backend", dist.get_backend())
dist.destroy_process_group()
