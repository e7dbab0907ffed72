"""Multi-GPU partitioning of the hot path.  Every block / frame /
closed GOP is independent for these kernels, so ranks own disjoint contiguous
ranges and never exchange data on the data path; the only collectives are the
timing barrier / max-reduce and an optional all-reduce of fixed-size digests
(sum of eob, sum of SAD, block count, checksum-of-checksums) used to verify a run.
One process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm)."""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous block range [lo, hi) of rank r: [r*n/G, (r+1)*n/G)."""
    assert 0 <= rank < world
    return (n * rank) // world, (n * (rank + 1)) // world


def gop_owner(gop_index: int, world: int) -> int:
    """C5: GOP g (30 frames) is encoded by GPU g mod G."""
    return gop_index % world


def gops_of_rank(n_gops: int, rank: int, world: int):
    return [g for g in range(n_gops) if gop_owner(g, world) == rank]


# Checksum lanes are residues modulo a prime SMALL enough that thousands of them add up without leaving int64 (with 2^61 - 1,
# which round 1 used, four residues already overflow: the sum then depended on how the work was grouped into calls).
DIGEST_MOD = (1 << 31) - 1


def digest(eob: np.ndarray, sad: np.ndarray, qcoeff_checksum: int) -> np.ndarray:
    """Fixed-size int64 digest of a shard's outputs: [blocks, sum eob, sum sad, checksum]."""
    return np.array([int(eob.size), int(eob.astype(np.int64).sum()), int(sad.astype(np.int64).sum()),
                     int(qcoeff_checksum) & 0x7fffffffffffffff], dtype=np.int64)


def checksum_i32(a: np.ndarray) -> int:
    """Order-independent-by-block, position-sensitive-within-block checksum: sum over
    elements of value * (1 + index mod 8191), mod 2^31 - 1.  Shards add."""
    a = np.ascontiguousarray(a).reshape(-1).astype(np.int64)
    w = (np.arange(a.size, dtype=np.int64) % 8191) + 1
    return int((a * w % DIGEST_MOD).sum() % DIGEST_MOD)


def allreduce_digest(d: np.ndarray, device=None, checksum_lanes=(3,)) -> np.ndarray:
    """Sum digests over ranks; the checksum lane(s) modulo 2^31 - 1 (lane 3 of `digest`, lane 2 of frames.FramePass.digest)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return d
    t = torch.from_numpy(d.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    out = t.cpu().numpy()
    for k in checksum_lanes:
        out[k] %= DIGEST_MOD
    return out
