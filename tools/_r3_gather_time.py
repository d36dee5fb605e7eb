"""GPU-box helper (under torchrun, one rank): host cost of the pieces of one counters gather over RCCL"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from elector_amd import distributed as ed
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, c = 10001, 25
pipe = ed.GatherPipe(n, c)
local = np.arange(n * c, dtype=np.int64).reshape(n, c)
for _ in range(5):
    pipe.start(local, [n]).wait()
T = {"stage": 0.0, "h2d": 0.0, "gather": 0.0, "wait": 0.0, "d2h": 0.0, "rec": 0.0, "sync": 0.0, "concat": 0.0}
K = 50
for _ in range(K):
    s = pipe.slots[pipe.turn % 4]; pipe.turn += 1
    t = time.perf_counter(); s["stage"].numpy()[:n] = local; T["stage"] += time.perf_counter() - t
    with torch.cuda.stream(pipe.stream):
        t = time.perf_counter(); s["pad"].copy_(s["stage"], non_blocking=True); T["h2d"] += time.perf_counter() - t
        t = time.perf_counter(); w = dist.gather(s["pad"], s["out"], dst=0, async_op=True); T["gather"] += time.perf_counter() - t
        t = time.perf_counter(); w.wait(); T["wait"] += time.perf_counter() - t
        t = time.perf_counter(); s["host"][0].copy_(s["out"][0], non_blocking=True); T["d2h"] += time.perf_counter() - t
        t = time.perf_counter(); s["done"].record(pipe.stream); T["rec"] += time.perf_counter() - t
    t = time.perf_counter(); s["done"].synchronize(); T["sync"] += time.perf_counter() - t
    t = time.perf_counter(); x = np.concatenate([s["host"].numpy()[0, :n]], axis=0); T["concat"] += time.perf_counter() - t
print({k: round(v / K * 1e3, 3) for k, v in T.items()}, "ms per gather")
assert np.array_equal(x, local)
dist.destroy_process_group()
