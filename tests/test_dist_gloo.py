"""world_size-2 `gloo` tests (CPU) of the N>1 path: image sharding, max-over-ranks timing, the fused 3-element loss
normaliser all-reduce (replaces the reference's three barrier + all-reduce pairs, HEAD:167-172/194-199/223-228)."""
import os

import numpy as np
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


def test_bench_training_leg_guard_exception_is_a_record_and_a_hang_is_an_exit_code():
    """The guard around bench.py's N > 1 training leg (LegGuard), rehearsed with two gloo ranks and the dry run's stand-in leg:
    * the leg comes back: one line, rc 0, the leg's record in it;
    * rank 1 RAISES inside the leg while rank 0 waits in the leg's collective: rank 0 learns of it through the store within
      a second, prints ONE line whose train.bf16.error names rank 1, rc 0 (the run is complete, the failure is on record);
    * rank 1 never comes back (a stuck collective): after the deadline rank 0 prints ONE line with the error and its
      evidence, and the run's exit code is bench.EXIT_LEG_STUCK -- a hang is not a clean run."""
    import json
    env = {"SKG_BENCH_BACKEND": "gloo", "SKG_BENCH_TRAIN_DEADLINE": "4"}
    args = ["--gpus", "2", "--dry-run", "--steps", "2"]

    def line(r):
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, (r.stdout, r.stderr[-1500:])
        return json.loads(lines[0])

    r = _run_bench(args, dict(env, SKG_BENCH_DRY_TRAIN_LEG="ok"))
    assert r.returncode == 0, r.stderr[-1500:]
    assert line(r)["train"]["bf16"]["rehearsal"] is True
    r = _run_bench(args, dict(env, SKG_BENCH_DRY_TRAIN_LEG="raise:1", SKG_BENCH_TRAIN_DEADLINE="60"))
    assert r.returncode == 0, r.stderr[-1500:]
    rec = line(r)["train"]["bf16"]
    assert "rank 1" in rec["error"] and "rehearsed failure" in rec["error"]
    r = _run_bench(args, dict(env, SKG_BENCH_DRY_TRAIN_LEG="raise:0", SKG_BENCH_TRAIN_DEADLINE="60"))
    assert r.returncode == 0, r.stderr[-1500:]
    assert "rank 0" in line(r)["train"]["bf16"]["error"]
    r = _run_bench(args, dict(env, SKG_BENCH_DRY_TRAIN_LEG="hang:1"))
    assert r.returncode == 4, (r.returncode, r.stderr[-1500:])
    rec = line(r)["train"]["bf16"]
    assert "did not finish within" in rec["error"] and rec["evidence"]["route"] == "dry run" and rec["rank"] == 0
    assert "bench.py rank 1:" in r.stderr                      # the stuck rank's own watchdog left its evidence too


def _exchange_worker(rank, world, port, q):
    """One rank of the data-parallel gradient-exchange test (CPU, gloo): oracle gradients of THIS rank's image under the
    reference's data-parallel loss (sum / (all_reduce_sum(n_p) / world), HEAD:167-172), laid out in the gradient arena,
    exchanged chunk by chunk as the fused backward would (stage prefixes), then compared with the single-process oracle
    gradients of the whole two-image batch."""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import cases
    import helpers
    from oracle import skg_oracle as O
    from skghoi_amd import GraphHead, InteractionHead, synth, train_fused, trainer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    case = cases.build_case("train_tiny")
    cfg = case["cfg"]
    tables = helpers.golden_tables(helpers.load_golden("train_tiny"))

    def oracle(lo, hi):
        sd = synth.make_state_dict(cfg["K"], case["C"], case["p"], seed=case["weight_seed"])
        sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        torch.manual_seed(case["rng_seed"])
        off = [0]

        def pool(coords):                       # the cached pooled features are indexed by global box row
            n = sum(len(c) for c in coords)
            return cases.pooled_for(case, 64)[off[0]:off[0] + n]
        results, extras = O.interaction_head_forward(
            sd, case["feat3"][lo:hi], case["detections"][lo:hi], case["shapes"][lo:hi], pool, cfg["K"], cfg["human_idx"],
            case["o2v"], targets=case["targets"][lo:hi], training=True, max_human=case["max_human"],
            max_object=case["max_object"], num_iter=case["num_iter"], tables=tables[lo:hi])
        return sd, results, extras

    # rows of the first image in the pooled cache (the second rank's boxes start after them)
    sd_all, res_all, ex_all = oracle(0, 2)
    n0 = len(ex_all["preprocessed"][0]["boxes"])
    lo, hi = rank, rank + 1
    sd = synth.make_state_dict(cfg["K"], case["C"], case["p"], seed=case["weight_seed"])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.manual_seed(case["rng_seed"])
    results, extras = O.interaction_head_forward(
        sd, case["feat3"][lo:hi], case["detections"][lo:hi], case["shapes"][lo:hi],
        lambda coords: cases.pooled_for(case, 64)[n0 * rank:n0 * rank + sum(len(c) for c in coords)], cfg["K"],
        cfg["human_idx"], case["o2v"], targets=case["targets"][lo:hi], training=True, max_human=case["max_human"],
        max_object=case["max_object"], num_iter=case["num_iter"], tables=tables[lo:hi])
    n_cells = float(sum(int(torch.count_nonzero(r["labels"])) for r in results))
    n_pairs = float(sum(int(torch.count_nonzero(r["unary_labels"])) for r in results))
    # ---- the step's ONE fused normaliser all-reduce, then the arena chunks: census of the python-level collectives
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t_, *a_, **k_: (calls.append(t_.numel()), orig(t_, *a_, **k_))[1]
    try:
        norm = skd.start_normalisers(torch.tensor([n_cells, n_pairs, n_pairs])).get()      # all_reduce_sum / world
        params = list(sd.values())
        g_hoi = torch.autograd.grad(extras["losses"]["hoi_loss"], params, retain_graph=True, allow_unused=True)
        g_int = torch.autograd.grad(extras["losses"]["interactiveness_loss"], params, allow_unused=True)
        # local loss = sum / n_local  ->  data-parallel loss = sum / n_p  (n_p = norm): rescale the local gradients
        a, b = n_cells / float(norm[0]), n_pairs / float(norm[1])
        mine = {k: (0 if gh is None else a * gh) + (0 if gi is None else b * gi) for k, gh, gi in zip(sd, g_hoi, g_int)}
        gh_ = GraphHead(case["C"], case["p"], 1024, 1024, cfg["K"], cfg["human_idx"], case["o2v"])
        head = InteractionHead(torch.nn.Identity(), gh_, torch.nn.Linear(2048, 1), torch.nn.Linear(2048, cfg["K"]),
                               human_idx=cfg["human_idx"], num_classes=cfg["K"])
        st = train_fused.Stacked(head, torch.device("cpu"))
        ga = torch.zeros(st.total)
        views = st.grad_views(ga)
        names = {id(p): n for n, p in head.named_parameters()}
        for p, v in zip(st.src, views):
            g = mine[names[id(p)]]
            if torch.is_tensor(g):
                v.copy_(g)
        ex = trainer.ArenaExchange(head, min_chunk=1 << 18)
        ex.begin(ga)
        for s_, end in enumerate(st.milestone_end):
            ex.on_stage(s_, ga, end, last=(s_ == len(st.milestone_end) - 1))
        ex.finish()
    finally:
        dist.all_reduce = orig
    # ---- against the single-process gradients of the two-image batch
    total = ex_all["losses"]["hoi_loss"] + ex_all["losses"]["interactiveness_loss"]
    want = torch.autograd.grad(total, list(sd_all.values()), allow_unused=True)
    worst = 0.0
    byname = {names[id(p)]: v for p, v in zip(st.src, views)}
    for k, w in zip(sd_all, want):
        if w is None:
            continue
        scale = max(float(w.abs().max()), 1e-6)
        worst = max(worst, float((byname[k] - w).abs().max()) / scale)
    q.put((rank, calls, ex.collectives, st.total, worst))
    dist.barrier()
    dist.destroy_process_group()


def test_arena_gradient_exchange_equals_single_process_batch_gradients():
    """Two gloo ranks, one image each: the arena exchange (chunked all-reduce of the stage prefixes, averaged) of the
    per-rank oracle gradients under the reference's data-parallel normaliser equals the oracle gradients of the two-image
    batch in one process (HEAD:167-172 + DDP's gradient mean); exactly {1 normaliser, k arena} collectives per step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, calls, k, total, worst in res:
        assert calls[0] == 3 and len(calls) == 1 + k and 2 <= k <= 12, calls
        assert sum(calls[1:]) == total
        assert worst <= 1e-4, worst               # fp32 summation order, like the other gradient tests


def _meter_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    from skghoi_amd import evaluate as ev
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rs = np.random.RandomState(5)
    n = 600
    sc = torch.tensor(np.round(rs.uniform(0, 1, n), 3), dtype=torch.float32)
    cl = torch.tensor(rs.randint(0, 9, n)); lb = torch.tensor((rs.uniform(0, 1, n) < 0.25).astype(np.float32))
    whole = ev.DetectionAPMeter(9); whole.append(sc, cl, lb)
    m = ev.DeviceAPMeter(9, device="cpu")
    lo, hi = (0, 250) if rank == 0 else (250, n)                 # ragged shards; rank order = global order
    for a in range(lo, hi, 100):
        m.append(sc[a:min(a + 100, hi)], cl[a:min(a + 100, hi)], lb[a:min(a + 100, hi)])
    got = m.eval()                                               # ONE padded all_gather, every rank gets the APs
    empty = ev.DeviceAPMeter(9, device="cpu") if rank == 1 else m
    _ = empty.eval()                                             # a rank with an empty log still meets its peer
    q.put((rank, torch.equal(got, whole.eval()), float(got.mean())))
    dist.barrier()
    dist.destroy_process_group()


def test_training_meter_gathers_the_ranks_logs_once():
    """utils.py:263-282 all-gathers every iteration's results; the asynchronous meter keeps each rank's log on its device and
    gathers once in eval(): two gloo ranks with ragged shards reproduce the single meter over all detections."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_meter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and res[0][2] == res[1][2] > 0
