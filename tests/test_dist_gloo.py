"""world_size-2 `gloo` tests (CPU) of the N>1 path: image sharding, max-over-ranks timing, the fused 3-element loss
normaliser all-reduce (replaces the reference's three barrier + all-reduce pairs, HEAD:167-172/194-199/223-228)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skghoi_amd import dist as skd


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = skd.shard_range(n_items, rank, world)
    t = skd.max_over_ranks(1.0 + rank)
    # per-rank positives: rank r contributes (10+r, 3+r, 3+r)
    a, b, c = skd.fused_normalisers(10 + rank, 3 + rank, 3 + rank)
    # the reference's way, term by term
    ref = []
    for v in (10 + rank, 3 + rank, 3 + rank):
        x = torch.as_tensor([v]); dist.barrier(); dist.all_reduce(x); ref.append((x / world).item())
    counts = skd.gather_counts(hi - lo)
    q.put((rank, lo, hi, t, (a, b, c), tuple(ref), counts))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_rank_gloo(n_items):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    covered = []
    for rank, lo, hi, t, fused, ref, counts in res:
        covered += list(range(lo, hi))
        assert t == 2.0                                    # max over ranks of (1.0, 2.0)
        assert fused == pytest.approx(ref) and fused == pytest.approx((10.5, 3.5, 3.5))
        assert sum(counts) == n_items
    assert covered == list(range(n_items))                 # disjoint, complete, ordered


def test_single_process_fallbacks():
    assert skd.shard_range(10, 0, 1) == (0, 10)
    assert skd.max_over_ranks(0.25) == 0.25
    assert skd.fused_normalisers(4, 2, 2) == (4.0, 2.0, 2.0)
    assert [skd.shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
