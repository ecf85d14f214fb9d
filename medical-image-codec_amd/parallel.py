"""Static sharding of independent units over the GPUs of one node.

Strips, MIC2 frames and MIC3 tiles share no data, so the only cross-rank traffic is the assembly of
a container: an all-gather of per-unit compressed sizes (8 B per unit) and a gather of the blobs to
rank 0.  No all-reduce, no data-path collective inside the codec.  The codec is injected, so the
CPU tests can drive the plumbing with gloo and a stand-in codec.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np


def shard_range(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous static partition: unit i belongs to rank i*world//n_units (each rank's output is one
    contiguous blob range)."""
    lo = (n_units * rank) // world
    hi = (n_units * (rank + 1)) // world
    return lo, hi


def write_mic2(width: int, height: int, blobs: Sequence[bytes]) -> bytes:
    """WriteMIC2 (multiframe.go:49-91), independent mode."""
    n = len(blobs)
    hdr = bytearray(20 + 8 * n)
    hdr[0:4] = b"MIC2"
    hdr[4:8] = int(width).to_bytes(4, "little"); hdr[8:12] = int(height).to_bytes(4, "little")
    hdr[12:16] = n.to_bytes(4, "little"); hdr[16] = 0x01
    off = 0
    for i, b in enumerate(blobs):
        hdr[20 + 8 * i: 24 + 8 * i] = off.to_bytes(4, "little")
        hdr[24 + 8 * i: 28 + 8 * i] = len(b).to_bytes(4, "little")
        off += len(b)
    return bytes(hdr) + b"".join(blobs)


def dist_compress_multi_frame(frames_local: Sequence[np.ndarray], width: int, height: int, max_value: int,
                              n_frames_total: int, codec: Callable[[np.ndarray, int, int, int], bytes],
                              group=None) -> Optional[bytes]:
    """Each rank compresses its contiguous shard of an n_frames_total stack (frames_local = the frames of
    shard_range(n_frames_total, world, rank)); rank 0 returns the MIC2 file, the others None."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_range(n_frames_total, world, rank)
    assert len(frames_local) == hi - lo
    blobs = [codec(f, width, height, max_value) for f in frames_local]
    # 1. all-gather of the per-frame sizes (fixed-length vector, zero padded)
    per = (n_frames_total + world - 1) // world + 1
    mine = torch.zeros(per, dtype=torch.int64)
    mine[: len(blobs)] = torch.tensor([len(b) for b in blobs], dtype=torch.int64)
    sizes = [torch.zeros(per, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, mine, group=group)
    # 2. gather of the payloads to rank 0 (byte tensors, padded to the largest shard)
    payload = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    totals = [int(s.sum()) for s in sizes]
    cap = max(max(totals), 1)
    buf = torch.zeros(cap, dtype=torch.uint8)
    buf[: payload.size] = torch.from_numpy(payload.copy())
    gathered = [torch.zeros(cap, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out: List[bytes] = []
    for r in range(world):
        rlo, rhi = shard_range(n_frames_total, world, r)
        data = gathered[r].numpy().tobytes()
        off = 0
        for k in range(rhi - rlo):
            ln = int(sizes[r][k])
            out.append(data[off: off + ln]); off += ln
    return write_mic2(width, height, out)
