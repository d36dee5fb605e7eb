"""Multi-GPU layer: one process per GPU (torchrun), reads sharded by contiguous
ranges balanced by DP cells (read_cell_estimate), NO data-path collective; a single gather of the
per-piece integer counters to rank 0 at the end.

`backend="nccl"` is RCCL over xGMI on the GPU node; the same code runs on the
`gloo` backend for the CPU tests.  All floating point happens after the gather,
on rank 0, in original read order (elector_amd.computeStats.aggregate), so the
report does not depend on the number of GPUs.
"""
import numpy as np


def shard_bounds(weights, world):
    """Contiguous split of len(weights) items into `world` ranges with near-equal
    weight sums.  -> int64[world + 1] boundaries."""
    n = len(weights)
    b = np.zeros(world + 1, dtype=np.int64)
    if n == 0:
        return b
    cum = np.cumsum(np.asarray(weights, dtype=np.float64))
    total = cum[-1]
    for r in range(1, world):
        t = total * r / world
        k = int(np.searchsorted(cum, t, side="left"))          # first prefix whose sum reaches the target
        # cut after item k when that lands closer to the target than cutting in front of it
        before = cum[k - 1] if k > 0 else 0.0
        b[r] = k + 1 if k < n and (cum[k] - t) <= (t - before) else k
    b[world] = n
    return np.maximum.accumulate(b)


def read_cell_estimate(lr, lc, lu, window=60):
    """DP cells a read triple will cost, known before it is split (shards are balanced
    by sum of cells, Lr*Lc + |PO|*Lu per window).  Windows are about `window` bases long whatever the read
    length, so alignment #1 costs ~window cells per corrected base and alignment #2 ~window cells per
    uncorrected base where the corrected read covers the reference, 1 + 1 cells per base where it does
    not (`N` filler windows: Lr x 1 and (Lr + 1) x Lu / windows)."""
    lr = np.asarray(lr, dtype=np.float64)
    lc = np.minimum(np.asarray(lc, dtype=np.float64), lr)
    lu = np.asarray(lu, dtype=np.float64)
    return window * (lc + lu * 1.05) + (lr - lc)


def gather_sizes(n_local, group=None):
    """Row counts of every rank's block, for gather_rows(sizes=...) when the same counts serve many gathers"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [int(n_local)]
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([int(n_local)], dtype=torch.int64, device=dev), group=group)
    return [int(s.item()) for s in sizes]


def gather_rows(local, group=None, dst=0, sizes=None, device=None):
    """Gather variable-length int64 [k_i, C] blocks from every rank to `dst`, in rank
    order.  Returns the concatenated array on dst and None elsewhere.  sizes: the ranks' row counts when the
    caller already has them (gather_sizes) -- one collective instead of two."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = (torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
           if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    local = np.ascontiguousarray(local, dtype=np.int64)
    ncol = local.shape[1] if local.ndim == 2 else 1
    if sizes is None:
        sizes = gather_sizes(local.shape[0], group)
    elif len(sizes) != world or int(sizes[rank]) != local.shape[0]:
        raise ValueError("gather_rows: sizes do not describe this rank's block")
    cap = max(int(s) for s in sizes)
    pad = torch.zeros((cap, ncol), dtype=torch.int64, device=dev)
    if local.shape[0]:
        pad[: local.shape[0]] = torch.from_numpy(local.reshape(local.shape[0], ncol)).to(dev)
    out = [torch.zeros((cap, ncol), dtype=torch.int64, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst, group=group)
    if rank != dst:
        return None
    return np.concatenate([o[: int(s)].cpu().numpy() for o, s in zip(out, sizes)], axis=0)


class _Gather:
    """a gather_rows in flight: wait() -> the concatenated rows on dst, None elsewhere"""

    def __init__(self, pipe, slot, sizes, is_dst):
        self._pipe, self._slot, self._sizes, self._is_dst = pipe, slot, sizes, is_dst

    def wait(self):
        s = self._slot
        s["done"].synchronize()                    # the collective and (on dst) the copy to pinned memory have run
        s["busy"] = False
        if not self._is_dst:
            return None
        host = s["host"].numpy()
        return np.concatenate([host[r, : int(n)] for r, n in enumerate(self._sizes)], axis=0)


class GatherPipe:
    """Repeated gathers of int64 [k_i, C] blocks to one rank without stalling the caller: a few buffer sets (pinned
    staging, device block, the receive blocks and a pinned landing area on dst) are reused, every transfer and the
    collective itself are queued on a stream of the pipe's own, and nothing waits before _Gather.wait().  With the
    nccl backend (RCCL over xGMI) the blocks travel device to device; with gloo everything stays on the host.
    cap: rows a rank's block may have (the buffers are allocated once, for that)."""

    def __init__(self, cap, ncol, group=None, dst=0, depth=4, device=None):
        import contextlib
        import torch
        import torch.distributed as dist
        self.group, self.dst, self.ncol = group, dst, int(ncol)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.nccl = dist.get_backend(group) == "nccl"
        self.cap = max(int(cap), 1)
        # The device is an argument, not "the current one": HIP's current device is per THREAD and a new thread starts
        # on device 0, so a pipe built or started on a helper thread (bench.py does both) would put its buffers and its
        # stream on GPU 0 in every rank while the process group is bound to GPU local_rank.
        self.device = None
        if self.nccl:
            self.device = int(torch.cuda.current_device() if device is None else device)
        dev = torch.device("cuda", self.device) if self.nccl else torch.device("cpu")
        self._on_device = (lambda: torch.cuda.device(self.device)) if self.nccl else contextlib.nullcontext
        self.stream = torch.cuda.Stream(device=dev) if self.nccl else None
        self.slots = []
        with self._on_device():
            self._alloc(depth, dev, dst)
        if self.nccl:
            # the zero fills above ran on the device's default stream: the pipe's stream starts behind them
            self.stream.wait_stream(torch.cuda.default_stream(dev))
        self.turn = 0

    def _alloc(self, depth, dev, dst):
        import torch
        for _ in range(depth):
            s = {"busy": False}
            s["stage"] = torch.zeros((self.cap, self.ncol), dtype=torch.int64)
            if self.nccl:
                s["stage"] = s["stage"].pin_memory()
                s["pad"] = torch.zeros((self.cap, self.ncol), dtype=torch.int64, device=dev)
            else:
                s["pad"] = s["stage"]
            if self.rank == dst:
                s["out"] = [torch.zeros((self.cap, self.ncol), dtype=torch.int64, device=dev) for _ in range(self.world)]
                s["host"] = torch.zeros((self.world, self.cap, self.ncol), dtype=torch.int64)
                if self.nccl:
                    s["host"] = s["host"].pin_memory()
            s["done"] = torch.cuda.Event() if self.nccl else _HostDone()
            self.slots.append(s)

    def start(self, local, sizes):
        import torch
        import torch.distributed as dist
        local = np.ascontiguousarray(local, dtype=np.int64).reshape(-1, self.ncol)
        sizes = [int(x) for x in sizes]
        if len(sizes) != self.world or local.shape[0] != sizes[self.rank] or max(sizes) > self.cap:
            raise ValueError("GatherPipe: sizes %r do not describe this rank's block of %d rows (room for %d)"
                             % (sizes, local.shape[0], self.cap))
        s = self.slots[self.turn % len(self.slots)]
        self.turn += 1
        if s["busy"]:
            raise RuntimeError("GatherPipe: more gathers in flight than buffer sets; wait() for the oldest first")
        s["busy"] = True
        s["stage"].numpy()[: local.shape[0]] = local
        is_dst = self.rank == self.dst
        if self.nccl:
            with self._on_device(), torch.cuda.stream(self.stream):
                s["pad"].copy_(s["stage"], non_blocking=True)
                dist.gather(s["pad"], s["out"] if is_dst else None, dst=self.dst, group=self.group, async_op=True).wait()
                if is_dst:
                    for r in range(self.world):
                        s["host"][r].copy_(s["out"][r], non_blocking=True)
                s["done"].record(self.stream)
        else:
            work = dist.gather(s["pad"], s["out"] if is_dst else None, dst=self.dst, group=self.group, async_op=True)
            s["done"].set(work, s if is_dst else None)
        return _Gather(self, s, sizes, is_dst)


class _HostDone:
    """the gloo side of GatherPipe's completion event"""

    def set(self, work, slot):
        self._work, self._slot = work, slot

    def synchronize(self):
        self._work.wait()
        if self._slot is not None:
            for r, o in enumerate(self._slot["out"]):
                self._slot["host"][r].copy_(o)


def gather_rows_async(local, sizes, group=None, dst=0, cap=None, device=None, _pipes={}):
    """gather_rows without waiting for it: the collective runs while the caller goes on (bench.py: the next
    steps' kernels are already queued); sizes as from gather_sizes.  -> object with wait().  The calls of a process
    share one GatherPipe per row width (at most four gathers in flight); cap: rows to make room for when the pipe is
    created or has to grow (growing allocates pinned memory: callers that know their largest block say so up front).
    device: the GPU of this rank (nccl backend) -- REQUIRED knowledge when the call comes from a helper thread, whose
    current device is 0 whatever the main thread set; None = the calling thread's current device."""
    local = np.ascontiguousarray(local, dtype=np.int64)
    ncol = local.shape[1] if local.ndim == 2 else 1
    need = max(max(int(x) for x in sizes), int(cap or 0), 1)
    key = (ncol, group, dst, device)              # the group object itself: an id() can be reused after collection
    pipe = _pipes.get(key)
    if pipe is None or pipe.cap < need:
        pipe = _pipes[key] = GatherPipe(need, ncol, group, dst, device=device)
    return pipe.start(local, sizes)


def sharded_counters(pieces, counter_fn, group=None):
    """Every rank computes the counters of its own contiguous range of READS with
    counter_fn(sub_pieces) -> int64[n_pieces_local, C]; rank 0 receives all rows in
    piece order.  `pieces` is the full elector_amd.computeStats.Pieces on every rank."""
    import torch.distributed as dist
    from .computeStats import Pieces, ES_NCOUNTERS
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_reads = len(pieces.read_first) - 1
    cols_per_read = np.add.reduceat(pieces.cols, pieces.read_first[:-1]) if n_reads else np.zeros(0)
    b = shard_bounds(cols_per_read, world)
    r0, r1 = int(b[rank]), int(b[rank + 1])
    p0, p1 = int(pieces.read_first[r0]), int(pieces.read_first[r1])
    sub = Pieces()
    sub.headers, sub.header_nos = pieces.headers[p0:p1], pieces.header_nos[p0:p1]
    sub.cols = np.ascontiguousarray(pieces.cols[p0:p1])
    sub.row_off = np.ascontiguousarray(pieces.row_off[p0:p1 + 1] - pieces.row_off[p0])
    sub.rows = np.ascontiguousarray(pieces.rows[pieces.row_off[p0]:pieces.row_off[p1]])
    sub.read_first = np.ascontiguousarray(pieces.read_first[r0:r1 + 1] - p0)
    local = counter_fn(sub) if p1 > p0 else np.zeros((0, ES_NCOUNTERS), dtype=np.int64)
    return gather_rows(local, group)
