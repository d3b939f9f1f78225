"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

Extraction (unit = frame) and pair matching (unit = (KF_i, KF_j) job) shard with NO data-path collective: every rank
takes a contiguous block of units.  The one exchange step of the path is the all-keyframes loop-closure match
(config #4): the K x N x 32-byte descriptor table lives on one rank and is replicated with a single broadcast, after
which every rank matches its share of the jobs against its replica; the per-job results (a few ints) are gathered.
"""
import numpy as np


def shard_range(n_units, rank, world):
    """contiguous block partition: unit u belongs to the rank whose [lo, hi) holds it; sizes differ by at most 1"""
    base, rem = divmod(int(n_units), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(n_units, world):
    return [shard_range(n_units, r, world)[1] - shard_range(n_units, r, world)[0] for r in range(world)]


def broadcast_descriptor_table(table, counts, src=0):
    """replicate the keyframe descriptor table [K, cap, 32] (uint8) and its per-keyframe counts [K] (int32) from
    rank `src` to every rank — ONE collective per tensor (RCCL broadcast over xGMI on the GPU box).  `table` / `counts`
    must be pre-allocated with the same shape on every rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(table, src=src)
        dist.broadcast(counts, src=src)
    return table, counts


def gather_job_results(local_nmatches, n_jobs):
    """all ranks receive nmatches[n_jobs] in job order (all_gather of equal-size padded shards)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_nmatches
    world = dist.get_world_size()
    sizes = shard_sizes(n_jobs, world)
    pad = max(sizes)
    buf = torch.full((pad,), -1, dtype=local_nmatches.dtype, device=local_nmatches.device)
    buf[:local_nmatches.numel()] = local_nmatches
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def match_jobs_sharded(table, counts, pair_a, pair_b, match_fn, src=0):
    """config #4: broadcast the table, match this rank's block of the (a, b) jobs with `match_fn(table, counts, a, b)
    -> nmatches tensor`, gather the counts.  `match_fn` is the device matcher on the GPU box
    (FeatureMatcher.match_pairs_device) and is injected so the sharding logic can be exercised on CPU with gloo."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    broadcast_descriptor_table(table, counts, src)
    lo, hi = shard_range(pair_a.numel(), rank, world)
    local = match_fn(table, counts, pair_a[lo:hi], pair_b[lo:hi])
    return gather_job_results(local, pair_a.numel())


def lcg_pairs(seed, n_jobs, n_keyframes):
    """10 000 (i, j) pair jobs drawn by LCG"""
    from .synth import lcg_states
    st = lcg_states(seed, 2 * n_jobs)
    a = (st[:n_jobs] % n_keyframes).astype(np.int32)
    b = (st[n_jobs:] % n_keyframes).astype(np.int32)
    b = np.where(a == b, (b + 1) % n_keyframes, b).astype(np.int32)
    return a, b
