"""PCIe rates of the box: D2H / H2D of pinned memory through the runtime's copy path (torch copy_), alone and with a
compute kernel running beside it.  Usage: python tools/_r4_pcie.py"""
import os, sys, time
import torch
dev = torch.device("cuda", 0)
n = 360 << 20
d = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
h = torch.empty(n, dtype=torch.uint8).pin_memory()
def rate(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n * reps / (time.perf_counter() - t) / 1e9
print("SDMA env", os.environ.get("HSA_ENABLE_SDMA"))
print("D2H copy_ %.1f GB/s" % rate(lambda: h.copy_(d, non_blocking=True)))
print("H2D copy_ %.1f GB/s" % rate(lambda: d.copy_(h, non_blocking=True)))
s2 = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev)
def busy_d2h():
    with torch.cuda.stream(s2):
        for _ in range(3):
            torch.mm(a, a)
    h.copy_(d, non_blocking=True)
print("D2H beside matmuls %.1f GB/s" % rate(busy_d2h))
# two copies in flight on two streams
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
def two():
    with torch.cuda.stream(s2):
        h2.copy_(d, non_blocking=True)
    h.copy_(d, non_blocking=True)
print("2 x D2H on two streams %.1f GB/s (sum)" % (2 * rate(two)))
