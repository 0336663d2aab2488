"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).

The path shards by independent units: rays never interact and the scene is read-only.  Every
rank holds a full replica of the linearised BVH in its own HBM (1M triangles = 104 MB, 10M =
1.04 GB of 288 GB), traces a contiguous shard of the ray array, and the ONLY exchange is one
gather of the 16-byte hit records to the root rank (north star: "a single RCCL gather over xGMI
for hit records").  No all-reduce, no all-to-all.  The helpers below work on plain byte
tensors, so the same code runs under gloo on CPU in the tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

HIT_BYTES = 16


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of n rays for `rank`: THE shard rule of the C ABI (vt_shard_bounds) -- capacity =
    ceil(n / world) rounded up to 64 rays, shard g = [g * capacity, min(n, (g + 1) * capacity)).  The torch path and the
    native path (vt_trace_closest_gather_dev, vt_gather_hits_dev) therefore lay out the gathered array identically:
    rank r's records start at record r * capacity, which is ray order."""
    from .api import shard_bounds as abi_bounds
    return abi_bounds(n, world, rank)


def shard_capacity(n: int, world: int) -> int:
    from .api import shard_capacity as abi_capacity
    return abi_capacity(n, world)


def gather_records(local: torch.Tensor, count: int, n_total: int, record_bytes: int = HIT_BYTES, dst: int = 0,
                   out: Optional[List[torch.Tensor]] = None) -> Optional[torch.Tensor]:
    """Gather `count` records (record_bytes each, uint8 tensor) from every rank to `dst`.

    dist.gather needs equal sizes, so ragged shards are padded to shard_capacity records.
    Returns the concatenated (n_total * record_bytes) tensor on dst, None elsewhere.
    `out` may hold pre-allocated per-rank receive buffers (reused across batches)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = shard_capacity(n_total, world) * record_bytes
    send = local
    if local.numel() != cap:
        send = torch.zeros(cap, dtype=torch.uint8, device=local.device)
        send[: count * record_bytes] = local[: count * record_bytes]
    if rank == dst:
        bufs = out if out is not None else [torch.empty(cap, dtype=torch.uint8, device=local.device) for _ in range(world)]
        dist.gather(send, bufs, dst=dst)
        # shard r starts at record r * capacity: the concatenation IS ray order, padding only behind the last ray
        return torch.cat(bufs)[: n_total * record_bytes]
    dist.gather(send, None, dst=dst)
    return None


def trace_sharded(trace_fn: Callable[[torch.Tensor, int], torch.Tensor], rays: torch.Tensor, n_total: int,
                  ray_bytes: int = 32, record_bytes: int = HIT_BYTES, dst: int = 0) -> Optional[torch.Tensor]:
    """Strong-scaling form: every rank sees the whole ray array (uint8 tensor), traces its
    contiguous shard with trace_fn(shard_rays, count) -> hit bytes, root gets all hits in order."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n_total, world, rank)
    local = trace_fn(rays[lo * ray_bytes: hi * ray_bytes], hi - lo)
    return gather_records(local, hi - lo, n_total, record_bytes, dst)


def chunk_bounds(n: int, nchunks: int, align: int = 64) -> List[Tuple[int, int]]:
    """Split [0, n) into <= nchunks contiguous pieces whose starts are multiples of `align`."""
    nchunks = max(1, min(nchunks, max(1, n // align)))
    per = ((n + nchunks - 1) // nchunks + align - 1) // align * align
    out = []
    lo = 0
    while lo < n:
        out.append((lo, min(n, lo + per)))
        lo += per
    return out


def pipelined_trace_gather(trace_chunk: Callable[[int, int], None], n: int, hits_local: torch.Tensor,
                           recv_bufs: Optional[List[torch.Tensor]], nchunks: int = 4, record_bytes: int = HIT_BYTES,
                           dst: int = 0, via_host: bool = False) -> None:
    """Weak-scaling step: every rank traces its own n rays in chunks and the hit records of chunk c
    travel to `dst` (async gather) while chunk c+1 is being traced.

    trace_chunk(lo, hi) enqueues the trace of rays [lo, hi) into hits_local (bytes [lo*rb, hi*rb)).
    recv_bufs: on dst, one n*record_bytes uint8 tensor per rank (reused across steps); None elsewhere.
    via_host=True stages every chunk through host memory (for backends without device-tensor gather,
    e.g. gloo when two ranks share one GPU in a test); the RCCL path gathers device tensors directly.
    Returns after every gather has completed with respect to the current stream."""
    rank = dist.get_rank()
    works = []
    for lo, hi in chunk_bounds(n, nchunks):
        trace_chunk(lo, hi)
        send = hits_local[lo * record_bytes: hi * record_bytes]
        if via_host:
            host_recv = [torch.empty(send.numel(), dtype=torch.uint8) for _ in recv_bufs] if rank == dst else None
            dist.gather(send.cpu(), host_recv, dst=dst)
            if rank == dst:
                for b, h in zip(recv_bufs, host_recv):
                    b[lo * record_bytes: hi * record_bytes].copy_(h)
            continue
        recv = [b[lo * record_bytes: hi * record_bytes] for b in recv_bufs] if rank == dst else None
        works.append(dist.gather(send, recv, dst=dst, async_op=True))
    for w in works:
        w.wait()


class HitGatherPipeline:
    """Double-buffered form of pipelined_trace_gather for a stream of batches (bench.py's N > 1 loop).

    Batch b traces into hits[b % 2] and gathers into recv[b % 2]; the gathers of batch b are only waited
    for when buffer b % 2 is needed again (batch b + 2) or at drain(), so the xGMI transfer of one batch
    overlaps the tracing of the next.  Results on the root: recv[b % 2][rank] after drain()/next reuse."""

    def __init__(self, n: int, device, nchunks: int = 2, record_bytes: int = HIT_BYTES, dst: int = 0,
                 via_host: bool = False):
        self.n, self.nchunks, self.rb, self.dst, self.via_host = n, nchunks, record_bytes, dst, via_host
        world, rank = dist.get_world_size(), dist.get_rank()
        self.hits = [torch.empty(n * record_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = [[torch.empty(n * record_bytes, dtype=torch.uint8, device=device) for _ in range(world)]
                     if rank == dst else None for _ in range(2)]
        self.pending: List[List] = [[], []]
        self.batch = 0

    def submit(self, trace_chunk: Callable[[torch.Tensor, int, int], None]) -> int:
        """trace_chunk(hits_buffer, lo, hi) enqueues the trace of rays [lo, hi) of the current batch.
        Returns the buffer index the batch uses."""
        b = self.batch % 2
        for w in self.pending[b]:
            w.wait()
        self.pending[b] = []
        rank = dist.get_rank()
        for lo, hi in chunk_bounds(self.n, self.nchunks):
            trace_chunk(self.hits[b], lo, hi)
            send = self.hits[b][lo * self.rb: hi * self.rb]
            if self.via_host:
                host_recv = [torch.empty(send.numel(), dtype=torch.uint8) for _ in self.recv[b]] if rank == self.dst else None
                dist.gather(send.cpu(), host_recv, dst=self.dst)
                if rank == self.dst:
                    for buf, h in zip(self.recv[b], host_recv):
                        buf[lo * self.rb: hi * self.rb].copy_(h)
                continue
            recv = [buf[lo * self.rb: hi * self.rb] for buf in self.recv[b]] if rank == self.dst else None
            self.pending[b].append(dist.gather(send, recv, dst=self.dst, async_op=True))
        self.batch += 1
        return b

    def drain(self) -> None:
        for p in self.pending:
            for w in p:
                w.wait()
        self.pending = [[], []]
