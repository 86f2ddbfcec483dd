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


def _fake_result(i):
    """A result dict shaped like HEAD:317-322 with image-dependent ragged sizes (image 3 is empty)."""
    P, L = (0, 0) if i == 3 else (2 + i % 3, 5 + 2 * i)
    g = torch.Generator().manual_seed(100 + i)
    return dict(boxes_h=torch.rand(P, 4, generator=g), boxes_o=torch.rand(P, 4, generator=g),
                index=torch.randint(0, max(P, 1), (L,), generator=g), prediction=torch.randint(0, 117, (L,), generator=g),
                scores=torch.rand(L, generator=g), object=torch.randint(0, 80, (P,), generator=g),
                prior=torch.rand(2, L, generator=g), weights=torch.rand(P, generator=g))


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
    # the product path: async 3-element all-reduce, consumed as a tensor (no .item())
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t_, *a_, **k_: (calls.append(t_.numel()), orig(t_, *a_, **k_))[1]
    try:
        h = skd.start_normalisers(torch.tensor([10 + rank, 3 + rank, 3 + rank]))
        fused_async = tuple(h.get().tolist())
    finally:
        dist.all_reduce = orig
    assert calls == [3] and h.get().dtype == torch.float32
    # sharded inference: rank r holds the result dicts of items [lo, hi); every rank gets all of them back, in order
    mine = [_fake_result(i) for i in range(lo, hi)]
    allres = skd.gather_image_results(mine)
    ok = len(allres) == n_items
    for i, r in enumerate(allres):
        w = _fake_result(i)
        ok = ok and set(r) == set(w) and all(torch.equal(r[k], w[k]) and r[k].dtype == w[k].dtype for k in w)
    q.put((rank, lo, hi, t, (a, b, c), tuple(ref), counts, fused_async, ok))
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
    for rank, lo, hi, t, fused, ref, counts, fused_async, gathered_ok in res:
        assert fused_async == pytest.approx(ref) and gathered_ok
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
    h = skd.start_normalisers(torch.tensor([4, 2, 2]))
    assert h.work is None and h.get().tolist() == [4.0, 2.0, 2.0]
    res = [_fake_result(0), _fake_result(1)]
    assert skd.gather_image_results(res) == res


def _run_bench(args, env_extra, timeout=180):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_n_launches_its_ranks_itself():
    """`python bench.py --gpus 2` (no launcher, the form the driver uses) starts two rank processes before touching the
    GPU, and rank 0's ONE line says what the process group saw.  Reference: mp.spawn(main, nprocs=world_size)
    (configures/hicodet/adamixer_transH_spatial_r50_main.py:175-179).  CPU rehearsal: gloo, empty steps."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "8"],
                   {"SKG_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist"] == dict(world_size=2, backend="gloo", launcher="bench.py --gpus N")
    assert out["config"]["images_counted"] == 2 * 3 * 8 and out["scaling"] == "weak"


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_launcher_propagates_a_rank_failure():
    """A rank that dies must fail the whole run (non-zero exit), not leave the others waiting in a collective."""
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "2"], {"SKG_BENCH_BACKEND": "gloo", "SKG_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
