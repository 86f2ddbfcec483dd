"""Data-parallel plumbing of the interaction head (one process per GPU, torch.distributed; "nccl" = RCCL on ROCm).

Images are independent units: inference shards the image list over ranks with no data-path collective.
The only exchanges the reference's head performs are the three 1-element `n_p` all-reduces, each behind a barrier
(heads/adamixer_transH_spatial_r50_head.py:167-172, 194-199, 223-228); they are fused here into ONE 3-element
all-reduce without barriers, issued asynchronously right after label association (`start_normalisers`) and consumed as a
device tensor when the loss scalars are formed -- no host synchronisation in between.  Gradient all-reduce belongs to the
trainer (DDP / RCCL).  `gather_image_results` reassembles the per-rank result dicts of sharded inference / validation
for a single-rank evaluator (the reference's `_synchronise_and_log_results`, utils.py:263-282).
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


class PreparedNormalisers:
    """The fused `n_p` all-reduce of a PREPARED training batch (skghoi_amd.train_fused.prepare_steps): float32 device
    counts [3] (integer-valued, exact), summed in place over the ranks from the stream the preparation runs on, as soon as
    the label association is enqueued.  get() -> the same tensor divided by the world size (HEAD:167-172: all_reduce,
    then n_p / world_size in float32), ordered behind the collective on the stream get() is called on -- the preparation's
    side stream when the batch was prefetched (`finish()`), the step's stream otherwise."""

    def __init__(self, counts, group=None, force=False, native=None, enabled=True):
        self.vals, self.work, self.world, self._done, self.native = counts, None, 1, False, None
        if not enabled:
            return
        if native is not None:
            # the trainer's gradient exchange runs on the HIP library's own RCCL communicator (trainer.NativeComm): this
            # collective joins it there -- ONE communicator, one issue order on every rank [chunks of step i, this]
            from . import _capi
            self.world, self.native = native.world, native
            _capi.check(_capi.lib().skg_comm_all_reduce_begin_f32(native.handle, counts.data_ptr(), counts.numel(),
                                                                  torch.cuda.current_stream(counts.device).cuda_stream),
                        "skg_comm_all_reduce_begin_f32")
        elif dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
            self.world = dist.get_world_size(group)
            self.work = dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def get(self):
        if not self._done:
            if self.native is not None:
                from . import _capi
                _capi.check(_capi.lib().skg_comm_all_reduce_end(self.native.handle,
                                                                torch.cuda.current_stream(self.vals.device).cuda_stream),
                            "skg_comm_all_reduce_end")
            if self.work is not None:
                self.work.wait()
                self.work = None
            if self.world > 1:
                self.vals.div_(self.world)
            self._done = True
        return self.vals


def start_normalisers(counts, distributed=True, group=None, force=False, native=None):
    """counts: device (or CPU, for gloo) tensor of the three per-rank normaliser counts {#positive scored cells,
    #positive pairs, #positive pairs}.  Starts ONE 3-element all-reduce (async) when a process group with more than
    one rank is up and `distributed`; returns a handle whose get() is all_reduce_sum(counts) / world_size as float32 [3].
    The SAME collective as the fused step's preparation issues (PreparedNormalisers: three fp32 words in place, on the
    trainer's communicator): a rank whose batch takes this route -- the autograd path -- meets ranks on the fused route."""
    vals = counts.to(torch.float32).reshape(3).clone()
    if not distributed:
        return PreparedNormalisers(vals, enabled=False)
    return PreparedNormalisers(vals, group=group, force=force, native=native)


def _all_gather_ragged(t, group=None):
    """t: [n, ...] with a rank-dependent n -> list (one entry per rank) of the ranks' tensors."""
    w = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    ns = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(ns, n, group=group)
    ns = [int(v.item()) for v in ns]
    m = max(ns + [1])
    pad = t.new_zeros((m,) + tuple(t.shape[1:]))
    pad[:t.shape[0]] = t
    outs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(outs, pad.contiguous(), group=group)
    return [o[:k] for o, k in zip(outs, ns)]


# the one result key whose ragged dimension is not the first (HEAD:317-322: prior is [2, L])
_RAGGED_DIM = {"prior": 1}


def gather_image_results(results, group=None):
    """Sharded inference / validation: every rank holds the result dicts (HEAD:317-322) of ITS images; returns, on
    every rank, the result dicts of ALL images in rank order (= global image order under `shard_range`).  Tensors stay
    on the device they were on (device tensors travel over RCCL, CPU tensors over gloo).  Per key one padded
    all_gather; the ragged per-image sizes travel in one more.  Single process: returns `results` unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return list(results)
    w = dist.get_world_size(group)
    keys = sorted(results[0].keys()) if results else []
    # every rank must agree on the key set even when it holds no image: take it from the first rank that has one
    keysets = [None] * w
    dist.all_gather_object(keysets, keys, group=group)
    keys = next((k for k in keysets if k), [])
    if not keys:
        return []
    dev = results[0][keys[0]].device if results else None
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    lens = torch.zeros(len(results), len(keys), dtype=torch.int64, device=dev)
    for i, r in enumerate(results):
        for j, k in enumerate(keys):
            lens[i, j] = r[k].shape[_RAGGED_DIM.get(k, 0)]
    all_lens = _all_gather_ragged(lens, group)
    protos = [None] * w                        # dtype / trailing shape of every key (a rank without images has none)
    mine = {k: (str(results[0][k].dtype).replace("torch.", ""), tuple(results[0][k].movedim(_RAGGED_DIM.get(k, 0), 0).shape[1:]))
            for k in keys} if results else None
    dist.all_gather_object(protos, mine, group=group)
    proto = next(p for p in protos if p)
    gathered = {}
    for k in keys:
        d = _RAGGED_DIM.get(k, 0)
        if results:
            flat = torch.cat([r[k].movedim(d, 0) for r in results])
        else:
            flat = torch.zeros((0,) + proto[k][1], dtype=getattr(torch, proto[k][0]), device=dev)
        gathered[k] = _all_gather_ragged(flat.contiguous(), group)
    out = []
    for r in range(w):
        ln = all_lens[r].tolist()                    # [[length per key] per image of rank r]
        base = len(out)
        out.extend({} for _ in ln)
        for j, k in enumerate(keys):
            d = _RAGGED_DIM.get(k, 0)
            parts = gathered[k][r].split([row[j] for row in ln]) if ln else []
            for i, part in enumerate(parts):
                out[base + i][k] = part.movedim(0, d) if d else part
    return out
