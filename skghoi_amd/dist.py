"""Data-parallel plumbing of the interaction head (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

Images are independent units: inference shards the image list over ranks with no data-path collective.
The only exchanges the reference's head performs are the three 1-element `n_p` all-reduces, each behind a barrier
(heads/adamixer_transH_spatial_r50_head.py:167-172, 194-199, 223-228); they are fused here into ONE 3-element
all-reduce without barriers.  Gradient all-reduce belongs to the trainer (DDP / RCCL) and is not part of this file.
"""
import torch
import torch.distributed as dist


def host_cpu_share():
    """CPU cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (a container on a
    256-core host with a 16-core quota reports 256 everywhere else, and torch then starts 128 intra-op threads that
    fight over 16 cores: the host-bound training step ran 3-4x slower that way)."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                    # cgroup v2: "<quota|max> <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                                         # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def shard_range(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` (the first n_items % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(seconds, device=None, group=None):
    """Wall time of the slowest rank (bench contract: barrier-bracketed region, MAX over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def fused_normalisers(n_hoi, n_inter, n_transh, device=None, group=None):
    """(n_p / world) for the three loss terms in one collective: returns three Python floats.

    Reference semantics per term: n_p = all_reduce_sum(n_p) / world_size (HEAD:167-172)."""
    vals = torch.tensor([float(n_hoi), float(n_inter), float(n_transh)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=group)
        vals = vals / dist.get_world_size(group)
    a, b, c = vals.tolist()
    return a, b, c


def gather_counts(value, device=None, group=None):
    """All ranks' integer `value` as a list (used to assemble whole-job throughput / result offsets)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [int(value)]
    w = dist.get_world_size(group)
    out = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(w)]
    dist.all_gather(out, torch.tensor([int(value)], dtype=torch.int64, device=device), group=group)
    return [int(o.item()) for o in out]
