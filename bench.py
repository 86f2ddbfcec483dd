#!/usr/bin/env python3
"""bench.py -- images/s through the interaction head on synthetic 20-human x 20-object graphs (BASELINE.json metric).

One "step" = one eval forward of InteractionHead over a batch of B cached images per GPU (preprocess/NMS -> roi-pool
stand-in returning the HBM-resident cached box features -> graph head -> classifier -> scoring -> result dicts).
Inputs are resident in HBM before the timed region.  N GPUs = N independent shards of images (no data-path
collective: images are independent); value = N * B * K / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     : dominant kernel (the fp32-MFMA GEMM) algorithmic FLOP/s from HIP-event timings of every launch
  cpu_baseline : the CPU oracle (faithful restatement of the reference head, oracle/skg_oracle.py) timed on this
                 box's host cores on a bounded sample (rank 0, N=1 only)
and, at N=1, extra legs of the same run: two_stream_chunks (the engine's default stream setting; the timed region itself
runs the chunks on one stream so that the per-launch HIP-event durations of the roofline record are those of un-shared
launches), fp16x2 (the opt-in split-operand GEMM path, own roofline), b1_latency_ms / b4_latency_ms (the reference
evaluates one image per forward), train (the batch-4 training step: forward + backward + AdamW, fp32 and bf16 operands).
The headline value / dtype / roofline are the exact fp32 path (precision="fp32", the head's default).
"""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
# Before the HIP runtime starts: the hardware-queue setting of this process (skghoi_amd/runtime.py).  The same call
# runs in every form this script is started in -- alone, as a rank under torch.distributed.run, as a rank of `--gpus N`
# (those are children of a parent that never touches the GPU) -- so every rank of every launch form gets the same setting.
from skghoi_amd import runtime as _runtime
_runtime.configure()

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: BF16/F16 dense MFMA peak (~2.5 PF)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E ~8 TB/s
COMPULSORY_BYTES_PER_IMAGE = 2.47e6
F16X2_PASSES = 3                      # fp16x2 path: h.h + h.m + m.h = three fp16 MFMA flops per useful flop
N_H = N_O = 20
C_FEAT, POOL = 256, 7


class ResidentPool(torch.nn.Module):
    """box_roi_pool stand-in: the cached AdaMixer-R50 box features, already in HBM, in the head's box order."""

    def __init__(self, pooled):
        super().__init__()
        self.pooled = pooled

    def forward(self, features, boxes, image_shapes):
        return self.pooled


def make_inputs(B, rank, device):
    from skghoi_amd import synth
    dets, pooled, feats, shapes = [], [], [], []
    for i in range(B):
        im = synth.make_image(1000 + rank * B + i, n_h=N_H, n_o=N_O, out_channels=C_FEAT, pool=POOL)
        dets.append(dict(boxes=im["boxes"].to(device), labels=im["labels"].to(device), scores=im["scores"].to(device)))
        pooled.append(im["pooled"]); feats.append(im["feat3"]); shapes.append(im["hw"])
    pooled = torch.cat(pooled).to(device)
    feat3 = torch.cat(feats).to(device)
    return dets, pooled, OrderedDict((k, feat3) for k in "0123"), shapes


def build_head(device, max_human=N_H, max_object=N_O):
    from skghoi_amd import GraphHead, InteractionHead, synth
    o2v = synth.hico_object_to_verb()
    gh = GraphHead(C_FEAT, POOL, 1024, 1024, 117, 49, o2v, num_iter=2)
    head = InteractionHead(torch.nn.Identity(), gh, torch.nn.Linear(2048, 1), torch.nn.Linear(2048, 117),
                           human_idx=49, num_classes=117, max_human=max_human, max_object=max_object)
    head.load_state_dict(synth.make_state_dict(117, C_FEAT, POOL, seed=0))
    return head.to(device).eval()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(budget_s=40.0):
    """The oracle (kind 'port': loop-for-loop restatement of the reference head, incl. its Python row loop) on the
    host cores, same synthetic workload, batch 1 per forward like the reference's inference (utils.py:166-167).
    Protocol of SURVEY 8(d): all cores of the process's share AND one thread; 2 warm-up images, then 5 timed images per
    thread setting (fewer only if the time budget runs out); min and median reported; CPU model printed."""
    from oracle import skg_oracle as O
    from skghoi_amd import synth
    sd = synth.make_state_dict(117, C_FEAT, POOL, seed=0)
    o2v = synth.hico_object_to_verb()
    cores = torch.get_num_threads()               # set in main() to the process's CPU share
    t_start = time.perf_counter()

    def leg(threads, deadline):
        torch.set_num_threads(threads)
        times = []
        with torch.no_grad():
            for i in range(2 + 5):
                im = synth.make_image(1000 + i, n_h=N_H, n_o=N_O, out_channels=C_FEAT, pool=POOL)
                det = [dict(boxes=im["boxes"], labels=im["labels"], scores=im["scores"])]
                torch.manual_seed(i)
                t0 = time.perf_counter()
                O.interaction_head_forward(sd, im["feat3"], det, [im["hw"]], lambda c: im["pooled"], 117, 49, o2v,
                                           max_human=N_H, max_object=N_O, row_loop=True)
                dt = time.perf_counter() - t0
                if i >= 2:                      # two warm-up images
                    times.append(dt)
                if len(times) >= 2 and time.perf_counter() > deadline:
                    break
        return dict(threads=threads, images=len(times), min_s=round(min(times), 4),
                    median_s=round(float(np.median(times)), 4), value=round(1.0 / float(np.median(times)), 4))
    try:
        full = leg(cores, t_start + 0.4 * budget_s)
        one = leg(1, t_start + budget_s)
    finally:
        torch.set_num_threads(cores)
    return dict(value=full["value"], unit="images/s", cores=cores, kind="port", cpu_model=cpu_model(),
                all_cores=full, one_thread=one,
                sample="20x20 synthetic images, batch 1 per forward, fp32 torch CPU; per thread setting 2 warm-up images "
                       "then %d / %d timed images (%d threads / 1 thread); value = 1 / median seconds per image on %d "
                       "threads" % (full["images"], one["images"], cores, cores))


def run_train(B, precision, steps, warmup, device, rank, world, dist_on, prefetch=True, force_exchange=False,
              measure=False, probe=None):
    """The data-parallel training step (forward + backward on the HIP GEMMs, AdamW; with a process group: the gradient
    arena exchanged chunk by chunk behind the backward + the fused normaliser all-reduce, over RCCL) on B synthetic 20x20
    images per GPU with ground truth appended.  force_exchange: take the data-parallel route in a process group of ONE rank
    too.  measure: a few more, untimed steps with HIP events at the phase boundaries of the step's stream and around every
    dense product (skg_train_timer).  Returns (elapsed seconds of `steps` steps on this rank, last loss dict, info)."""
    from skghoi_amd import synth, trainer
    head = build_head(device).train()
    if probe is not None:
        probe["head"] = head                                 # (a watchdog's failure record reads the step's progress off it)
    head.distributed = dist_on
    head.force_collectives = force_exchange
    head.precision = precision
    dets, pooled, feats, shapes = make_inputs(B, rank, device)
    o2v = synth.hico_object_to_verb()
    cpu_dets = [dict(boxes=d["boxes"].cpu(), labels=d["labels"].cpu(), scores=d["scores"].cpu()) for d in dets]
    targets = [{k: v.to(device) for k, v in synth.make_targets(d, 49, o2v, 500 + i, n_gt=4).items()}
               for i, d in enumerate(cpu_dets)]

    class Pool(torch.nn.Module):
        """box_roi_pool stand-in: the cached box features, resident in HBM, for however many boxes the head selected (GT boxes
        are appended in training: the row count follows the head)."""
        cache = {}

        def forward(self, features, boxes, image_shapes):
            n = sum(len(b) for b in boxes)
            t = self.cache.get(n)
            if t is None:
                reps = (n + pooled.shape[0] - 1) // pooled.shape[0]
                t = self.cache[n] = pooled.repeat(reps, 1, 1, 1)[:n].contiguous()
            return t

    head.box_roi_pool = Pool()
    net = trainer.wrap_ddp(head, device, force_exchange=force_exchange)
    opt = trainer.build_optimizer(net, lr=1e-4)
    torch.manual_seed(1234 + rank)
    # lazy=True: the losses stay on the device (no per-step .item() / isnan round trip); all the work of the K steps is
    # still inside the timed region -- it ends on a device synchronisation -- and the losses are read and NaN-checked after
    # prefetch: like a trainer over a loader of cached detections (skghoi_amd.trainer.Trainer), every step hands the NEXT
    # batch to the head, which prepares it (selection, pairs, labels, host RNG) on a side stream while the GPU works on
    # the step just enqueued.  All work of every timed step -- its own preparation included, done during the step before
    # -- falls inside the timed region except the first step's, which the last warm-up step prepared (and the last timed
    # step prepares one batch nobody runs: the counts balance).
    nxt = (feats, dets, shapes, targets) if prefetch else None
    # ... and the batch after that (trainer.Trainer's default: a two-batch look-ahead); SKG_BENCH_LOOKAHEAD=1: one batch
    nxt2 = nxt if os.environ.get("SKG_BENCH_LOOKAHEAD", "2") != "1" else None
    for _ in range(warmup):
        losses, _ = trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt, prefetch2=nxt2)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    from skghoi_amd import gemmx as _gemmx
    _gemmx.path_counts(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        losses, _ = trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt, prefetch2=nxt2)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    info = {}
    pc = _gemmx.path_counts()
    info["product_launches_per_step"] = dict(zip(("exact_fp32", "bf16_register_staged", "bf16_direct_to_lds"),
                                                 (round(v / max(steps, 1), 2) for v in pc)))
    import ctypes
    from skghoi_amd import _capi
    pl = getattr(head, "_last_train_plan", None)
    if pl is not None:
        info["gflop_per_step"] = round(float(_capi.lib().skg_train_flops(ctypes.byref(pl), 2)) / 1e9, 3)
    exs = trainer.exchanges(net)
    if exs:
        # a few more steps, untimed, with HIP events around the point where the step's stream waits for the gradient
        # exchange: the part of the all-reduce that the backward did not cover
        exs[0].timing = True
        waits = []
        for _ in range(5):
            trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt, prefetch2=nxt2)
            waits.append(exs[0].read_timing())
        native = exs[0].native is not None
        info["grad_exchange"] = dict(collectives_per_step=exs[0].collectives + 1,
                                     exposed_wait_ms=round(float(np.median(waits)), 4),
                                     arena_mb=round(exs[0].ga.numel() * 4 / 2 ** 20, 1),
                                     route=("libskghoi_hip's own RCCL communicator: ncclAllReduce per chunk issued by the "
                                            "backward's worker thread (include/skghoi.h skg_comm, "
                                            "skg_ctx_train_backward_exchange_f32)") if native else
                                           "torch.distributed all_reduce per chunk from the Python thread (ArenaExchange.drive)",
                                     note="1 fused normaliser all-reduce (issued while the batch is prepared) + the gradient "
                                          "arena in chunks, each ordered behind the event of the backward stage that completes "
                                          "it (the backward itself is ONE call issued by the library's worker thread); "
                                          "exposed_wait = device time between the backward's last kernel and the end of the "
                                          "last chunk's collective, median of 5 untimed steps")
        exs[0].timing = False
    if measure:
        lib = _capi.lib()
        n_meas = 10
        tm = lib.skg_train_timer_create(256 * n_meas)
        head._train_spans = []
        head._train_timer = tm
        try:
            for _ in range(n_meas):
                trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt, prefetch2=nxt2)
            torch.cuda.synchronize()
            out3 = (ctypes.c_double * 3)()
            _capi.check(lib.skg_train_timer_read(tm, out3), "skg_train_timer_read")
        finally:
            spans, head._train_spans, head._train_timer = head._train_spans, None, None
            head.__dict__.pop("_train_spans", None); head.__dict__.pop("_train_timer", None)
            torch.cuda.synchronize()
            lib.skg_train_timer_destroy(tm)
        ok = [sp for sp in spans if all(k in sp for k in ("f0", "b0", "b1", "o1"))]
        med = lambda xs: round(float(np.median(xs)), 4) if xs else None
        info["phases_ms"] = dict(forward=med([sp["f0"].elapsed_time(sp["b0"]) for sp in ok]),
                                 backward=med([sp["b0"].elapsed_time(sp["b1"]) for sp in ok]),
                                 optimizer=med([sp["b1"].elapsed_time(sp["o1"]) for sp in ok]), steps=len(ok),
                                 note="HIP-event spans on the step's stream in %d untimed steps of this same process: f0 "
                                      "forward begins -> b0 losses done -> b1 behind the backward's last launch (recorded by "
                                      "the worker thread) -> o1 behind AdamW; gaps where the stream waits for the host are "
                                      "inside the spans" % n_meas)
        info["dense_products"] = dict(ms_per_step=round(out3[0] / n_meas, 4), launches_per_step=round(out3[1] / n_meas, 1),
                                      gflop_per_step=round(out3[2] / n_meas / 1e9, 3))
    return elapsed, trainer.read_losses(losses), info


def train_mode(args, device, rank, world, dist_on):
    """Secondary metric: images/s of the training step.  Reference settings: batch 4 per GPU (main:158)."""
    import torch.distributed as dist
    from skghoi_amd import dist as skd
    B = args.batch if args.batch != 256 else 4
    args.precision = args.precision or "fp32"
    elapsed, losses, info = run_train(B, args.precision, args.steps, args.warmup, device, rank, world,
                                      dist_on or args.dp_world1, force_exchange=args.dp_world1, measure=args.measure)
    elapsed = skd.max_over_ranks(elapsed, device=device)
    if rank == 0:
        print(json.dumps(dict(metric="images/sec through the interaction-head TRAINING step (20x20 pairs)",
                              value=round(B * world * args.steps / elapsed, 2), unit="images/s", n_gpus=world,
                              steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
                              higher_is_better=True, scaling="weak", vs_baseline=None,
                              dtype="bf16" if args.precision == "bf16" else "f32", data="synthetic",
                              config=dict(workload="train step: fwd + bwd + AdamW, NegativeSampling + MarginLoss + "
                                                   "two focal terms, 20x20 synthetic images with GT appended",
                                          batch_per_gpu=B, parallelism="dp%d" % world),
                              losses=losses, dist=dist_info(world, args.dp_world1), runtime=runtime_info(), **info)))
    if dist_on or args.dp_world1:
        dist.destroy_process_group()


def runtime_info():
    from skghoi_amd import runtime
    rec = runtime.info()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # the three-queue default was chosen from single-GPU A/Bs (profiles/r05_hw_queues_full_bench.txt); RCCL's own
        # streams share those queues on a multi-rank run and no multi-GPU box has been available to compare against
        rec = dict(rec, note="hardware-queue count measured at world size 1 only; unmeasured for world > 1 "
                             "(SKG_HW_QUEUES=0 keeps the HIP runtime's default)")
    return rec


def dp_world1_child(precision, steps, warmup):
    """The data-parallel route of the training step timed on ONE GPU: a child process of this script that opens an RCCL
    process group of world size 1 (RCCL's streams exist, its kernels run) and installs the gradient exchange -- the code
    path a rank of BASELINE config 4 runs, minus the wire.  Returns the child's JSON record (or an error note)."""
    import socket
    import subprocess
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "train", "--precision", precision, "--dp-world1",
           "--steps", str(steps), "--warmup", str(warmup)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    except subprocess.TimeoutExpired:
        return dict(error="timeout")
    for line in r.stdout.decode(errors="replace").splitlines():
        if line.startswith("{"):
            return json.loads(line)
    return dict(error="rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:]))


GEMM_NAMES = {0: "skg_gemm_kernel<BIAS>", 1: "skg_gemm_kernel<BIAS_RELU>", 2: "skg_gemm_kernel<MUL_RELU>",
              3: "skg_gemm_kernel<RELU_DOT>", 4: "skg_gemm_kernel<BIAS_RES_RELU>", 5: "skg_gemm_group_kernel"}


def timed_infer(step, barrier, steps, time_gemms):
    """Times exactly `steps` forwards between barriers; HIP events on the launch stream around every launch of the
    dominant kernel (MUL_RELU epilogue: the MBF fc_2 GEMMs) when time_gemms.  -> (seconds, [(e0, e1, M, N, K, epi)])."""
    from skghoi_amd import engine
    # an event pair costs a few microseconds of GPU time: ~150 pairs per step (every GEMM) would be 4 % of the step,
    # the ~8 launches of the dominant kernel are not
    engine.GEMM_TIMER = [] if time_gemms else None
    engine.GEMM_TIMER_EPI = {2}
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    timer, engine.GEMM_TIMER = (engine.GEMM_TIMER or []), None
    engine.GEMM_TIMER_EPI = None
    return elapsed, timer


def gemm_roofline(timer, precision, traffic_table):
    """Roofline record of the dominant kernel from per-launch HIP-event durations.  achieved = useful 2*M*N*K flops /
    measured time; peak = fp32 MFMA dense peak for the exact path, fp16 dense MFMA peak / 3 passes for fp16x2."""
    groups = {}
    for e0, e1, M, N, K, epi in timer:
        g = groups.setdefault(epi, [0.0, 0.0, 0])
        g[0] += e0.elapsed_time(e1) * 1e-3; g[1] += 2.0 * M * N * K; g[2] += 1
    if not groups:
        return None, 0.0, 0
    dom = max(groups, key=lambda k: groups[k][0])
    t_dom, f_dom, n_dom = groups[dom]
    split = precision == "fp16x2"
    name = GEMM_NAMES[dom].replace(">", ", fp16x2>") if split and dom != 5 else GEMM_NAMES[dom]
    peak = PEAK_F16_MFMA_TFLOPS / F16X2_PASSES if split else PEAK_F32_MFMA_TFLOPS
    achieved = f_dom / t_dom / 1e12
    traffic = traffic_table.get(name)
    rec = dict(bound="mfma", kernel=name, achieved=round(achieved, 2), peak=round(peak, 1), unit="TFLOP/s",
               frac=round(achieved / peak, 4), traffic=traffic,
               traffic_source=(traffic_table.get("_source") if traffic else None),
               peak_note=("useful (2MNK) flops; peak = fp16 dense MFMA %.0f TF / %d MFMA passes per useful flop"
                          % (PEAK_F16_MFMA_TFLOPS, F16X2_PASSES)) if split else "fp32 MFMA dense peak (v_mfma_f32_32x32x2_f32)",
               launches=n_dom, avg_launch_ms=round(t_dom / max(n_dom, 1) * 1e3, 4))
    return rec, t_dom, n_dom


def small_batch_latency(head, dets, pooled, feats, shapes, B, iters=200, warmup=30):
    """Mean wall time of one eval forward on B images (B = 1 is the reference's own evaluation mode, utils.py:166-167:
    `assert len(output) == 1`), back to back, results left on the device."""
    feat3 = feats["3"][:B]
    f = OrderedDict((k, feat3) for k in "0123")
    old_pool = head.box_roi_pool
    head.box_roi_pool = ResidentPool(pooled[:B * (N_H + N_O)])
    d, sh = dets[:B], shapes[:B]
    try:
        with torch.no_grad():
            for _ in range(warmup):
                head(f, d, sh)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                head(f, d, sh)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3
    finally:
        head.box_roi_pool = old_pool


def launch_ranks(n, argv):
    """Starts `n` fresh rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    rendezvous on 127.0.0.1), relays rank 0's stdout (the ONE JSON line) and returns the worst exit code.  The parent
    never initialises the GPU and never replaces itself: the ranks are ordinary children.  A rank that dies takes the
    others down with it (by PID) instead of leaving them waiting in a collective."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    out0 = b""
    try:
        while alive:
            for p in list(alive):
                try:
                    if p is procs[0]:
                        out0 += p.communicate(timeout=0.5)[0] or b""
                    else:
                        p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                alive.remove(p)
                if p.returncode != 0:
                    rc = rc or p.returncode
                    for q in alive:                      # a dead rank leaves the others hanging in their next collective
                        q.terminate()
    finally:
        for q in alive:
            q.kill()
    for line in out0.decode(errors="replace").splitlines():      # stdout carries the JSON line only; library chatter
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")      # (gloo / RCCL banners) goes to stderr
    sys.stdout.flush()
    return rc if rc >= 0 else 128 - rc


EXIT_LEG_STUCK = 4        # a rank did not come back from the N > 1 training leg within its deadline


class LegGuard:
    """Guards the N > 1 training leg -- the first code of a run to move gradients between real GPUs -- so that whatever goes
    wrong in it cannot cost the run its headline line, and cannot pass for success either:

    * an EXCEPTION on any rank becomes `train.bf16.error` on rank 0's line; the rank that raised says so through the process
      group's store, its peers (who would sit in their next collective until the deadline) see it within a second and leave;
      exit code 0 -- the run is complete, the leg's failure is in the record;
    * a leg that is NOT BACK within the deadline (a rank stuck in a collective: a GPU hang) makes rank 0 print the line with
      the error and every rank's evidence -- rank, backward stages issued, collectives issued, route -- and every rank exit
      with EXIT_LEG_STUCK: the driver's rc says the run was not clean.

    `out` (the line's dict) is only touched under the guard's lock, by whoever gets there first: the leg's own thread
    (finish / failed) or the watchdog (bail) -- one line, never two, never a dict printed while it is being built."""
    KEY = "skg_bench_train_leg_failed"

    def __init__(self, rank, world, out, deadline, evidence=None, emit=None):
        import threading
        self.rank, self.world, self.out, self.deadline = rank, world, out, deadline
        self.evidence = evidence or (lambda: {})
        self.emit = emit or (lambda o: print(json.dumps(o), flush=True))
        self.lock = threading.Lock()
        self.done = threading.Event()
        self.state = "running"
        self.store = None
        try:
            import torch.distributed as dist
            if world > 1 and dist.is_initialized():
                self.store = dist.distributed_c10d._get_default_store()
        except Exception:                                    # noqa: BLE001
            self.store = None
        threading.Thread(target=self._watch, daemon=True).start()

    def _peer_failure(self):
        if self.store is None:
            return None
        try:
            if self.store.check([self.KEY]):
                return self.store.get(self.KEY).decode(errors="replace")
        except Exception:                                    # noqa: BLE001
            return None
        return None

    def _watch(self):
        t_end = time.monotonic() + self.deadline
        while not self.done.wait(0.5):
            msg = self._peer_failure()
            if msg is not None:
                self._bail("a peer failed in the data-parallel training leg -- " + msg, 0)
            if time.monotonic() > t_end:
                self._bail("the data-parallel training leg did not finish within %.0f s" % self.deadline, EXIT_LEG_STUCK)

    def _record(self, error):
        rec = dict(error=error[:600], rank=self.rank)
        try:
            rec["evidence"] = self.evidence()
        except Exception as e:                               # noqa: BLE001
            rec["evidence"] = "unavailable: %s" % e
        return rec

    def _bail(self, error, code):
        with self.lock:
            if self.state != "running":
                return
            self.state = "bailed"
            rec = self._record(error)
            if self.rank == 0:
                self.out["train"] = dict(bf16=rec)
                self.out["runtime"] = runtime_info()
                self.emit(self.out)
                if code == EXIT_LEG_STUCK and self.world > 1:
                    # the peers' watchdogs run on the same deadline with a 0.5 s poll: give them the time to leave their own
                    # evidence on stderr before this rank's exit makes the launcher end them
                    time.sleep(1.5)
            else:
                sys.stderr.write("bench.py rank %d: %s\n" % (self.rank, json.dumps(rec)))
                sys.stderr.flush()
            os._exit(code)                                   # (with the lock held: the leg's own thread never prints after this)

    def finish(self, train_record):
        """The leg came back on this rank.  False if the watchdog got there first (the process is on its way out)."""
        with self.lock:
            if self.state != "running":
                return False
            self.state = "finished"
            self.out["train"] = train_record
        self.done.set()
        return True

    def failed(self, exc):
        """The leg raised on this rank: tell the peers, record the error."""
        msg = "rank %d: %s: %s" % (self.rank, type(exc).__name__, exc)
        peer = self._peer_failure()
        if peer is not None:         # this rank's error is the echo of a peer's (its collective lost the peer that left)
            return self.finish(dict(bf16=self._record("a peer failed in the data-parallel training leg -- %s (here, after that: %s)"
                                                      % (peer, msg))))
        if self.store is not None:
            try:
                self.store.set(self.KEY, msg[:600])
            except Exception:                                # noqa: BLE001
                pass
        return self.finish(dict(bf16=self._record(msg)))


def dry_run(args, rank, world):
    """Launcher rehearsal (no GPU, no head): the same rendezvous, barriers, max-over-ranks timing and count gather as the
    real run around an empty step."""
    import torch.distributed as dist
    from skghoi_amd import dist as skd
    if world > 1:
        dist.init_process_group(os.environ.get("SKG_BENCH_BACKEND", "gloo"))

    def barrier():
        if world > 1:
            dist.barrier()
    if os.environ.get("SKG_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies after the rendezvous
        os._exit(3)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    barrier()
    elapsed = skd.max_over_ranks(time.perf_counter() - t0)
    counts = skd.gather_counts(args.batch * args.steps)
    out = dict(metric="images/sec through interaction head (20x20 pairs)", value=None, unit="images/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f32", data="synthetic", dry_run=True,
               config=dict(workload="launcher rehearsal: no head, empty steps",
                           images_counted=sum(counts), batch_per_gpu=args.batch),
               dist=dist_info(world))
    leg = os.environ.get("SKG_BENCH_DRY_TRAIN_LEG")           # test hook: rehearse the N > 1 training leg's guard --
    clean = True                                              # "ok" | "raise:<rank>" | "hang:<rank>"
    if leg and world > 1:
        guard = LegGuard(rank, world, out, float(os.environ.get("SKG_BENCH_TRAIN_DEADLINE", "300")),
                         evidence=lambda: dict(route="dry run", stages_issued=0, collectives_issued=0))
        kind, _, who = leg.partition(":")
        try:
            if kind == "raise" and who == str(rank):
                raise RuntimeError("rehearsed failure of the training leg")
            if kind == "hang" and who == str(rank):
                time.sleep(3600)
            dist.barrier()                                    # (the leg's collective: the peers of a failed rank wait here)
            guard.finish(dict(bf16=dict(ms_per_step=0.0, rehearsal=True)))
        except Exception as e:                                # noqa: BLE001
            clean = False
            guard.failed(e)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if not clean:
        if rank == 0:
            time.sleep(2.0)          # (rank 0 hosts the store: the peers must get to read this rank's word)
        os._exit(0)                  # (the peers have left or are leaving: no collective teardown with them)
    if world > 1:
        dist.destroy_process_group()


def dist_info(world, forced=False):
    """What the process group itself reports (not what the flags asked for)."""
    import torch.distributed as dist
    if (world > 1 or forced) and dist.is_initialized():
        return dict(world_size=dist.get_world_size(), backend=dist.get_backend(),
                    launcher=os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or
                    ("bench.py --dp-world1" if forced else "bench.py --gpus N"))
    return dict(world_size=1, backend=None, launcher=None)


def b1_stream(device, n_forwards=2048, n_images=256, seed=2024):
    """The reference's evaluation loop (utils.py:160-167: one image per forward over the 9 658 test images) on a stream of
    single images whose (humans, nodes) follow a HICO-DET-like spread at the reference's default caps (max_human =
    max_object = 15): n_h in 1..15 with P ~ 1 / k^1.2, objects in 1..15 with P ~ 1 / k^0.8 (many small graphs, a tail of
    large ones), ~150 distinct shapes.  Two passes over the same stream: the first pays the plan captures (one per shape
    BUCKET, skghoi_amd/small.py), the second is the steady state.  Per-forward latency = call to results complete on
    the device (synchronised per forward, like a loop that consumes each result).  The loop is trainer.test's: it knows its
    next image and hands that image's detections to the head once the current forward is enqueued (look-ahead); a third
    pass without it is reported as steady_state_plain."""
    from skghoi_amd import synth
    head = build_head(device, max_human=15, max_object=15)
    rs = np.random.RandomState(seed)
    kh = np.arange(1, 16); ph = kh ** -1.2; ph /= ph.sum()
    ko = np.arange(1, 16); po = ko ** -0.8; po /= po.sum()
    imgs, shapes_seen = [], set()
    for i in range(n_images):
        nh, no = int(rs.choice(kh, p=ph)), int(rs.choice(ko, p=po))
        im = synth.make_image(50000 + i, n_h=nh, n_o=no, out_channels=C_FEAT, pool=POOL)
        det = [dict(boxes=im["boxes"].to(device), labels=im["labels"].to(device), scores=im["scores"].to(device))]
        f3 = im["feat3"].to(device)
        imgs.append((det, im["pooled"].to(device), OrderedDict((k, f3) for k in "0123"), [im["hw"]]))
        shapes_seen.add((nh, nh + no))
    pool = ResidentPool(None)
    head.box_roi_pool = pool
    order = rs.randint(0, n_images, n_forwards)
    runner = None
    out = {}
    resident = torch.cuda.Event(); resident.record()          # the images are on the device: nothing for a look-ahead to wait for
    with torch.no_grad():
        # first_pass / steady_state: the loop of trainer.test -- once forward i is enqueued, image i + 1's detections go to the
        # head (InteractionHead.prefetch_eval: selection, count read-back and table draw beside the forward in flight);
        # steady_state_plain: the same stream with every forward preparing for itself, as a caller without look-ahead sees it
        for name in ("first_pass", "steady_state", "steady_state_plain"):
            ahead = name != "steady_state_plain"
            lat = np.zeros(n_forwards)
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            for k, i in enumerate(order):
                det, pooled, feats, shp = imgs[int(i)]
                pool.pooled = pooled
                t0 = time.perf_counter()
                head(feats, det, shp)
                if ahead and k + 1 < n_forwards:
                    head.prefetch_eval(imgs[int(order[k + 1])][0], after=resident)
                torch.cuda.synchronize()
                lat[k] = time.perf_counter() - t0
            wall = time.perf_counter() - t_all
            st = head.engine()._small.stats()
            prev = runner or dict(hits=0, misses=0, captures=0, evictions=0)
            calls = st["hits"] + st["misses"] - prev["hits"] - prev["misses"]
            out[name] = dict(forwards=n_forwards, mean_ms=round(float(lat.mean()) * 1e3, 4),
                             p50_ms=round(float(np.percentile(lat, 50)) * 1e3, 4),
                             p95_ms=round(float(np.percentile(lat, 95)) * 1e3, 4),
                             max_ms=round(float(lat.max()) * 1e3, 3), images_per_s=round(n_forwards / wall, 1),
                             plan_hit_rate=round((st["hits"] - prev["hits"]) / max(calls, 1), 4),
                             captures=st["captures"] - prev["captures"], evictions=st["evictions"] - prev["evictions"],
                             look_ahead_hits=st["look_ahead_hits"] - prev.get("look_ahead_hits", 0))
            runner = st
    out.update(distinct_shapes=len(shapes_seen), plans=runner["plans"], max_human=15, max_object=15, precision=head.precision,
               note="single-image eval forwards over %d synthetic images with a HICO-like spread of graph sizes; one "
                    "captured launch plan per bucket of (humans, nodes), true sizes in the device-side meta record" % n_images)
    release_plans(head)
    return out


def b4_validate(device, n_images=256, batch=4, seed=2025):
    """The reference's VALIDATION loop (utils.py:283-299: eval mode, batches of 4 WITH their targets -- main:55-63 -- results
    into a 117-class meter) on a stream of batches whose shapes never repeat: 256 synthetic images with the HICO-like spread
    of b1_stream, ground truth from their own boxes, at the default 15 / 15 caps.  Each batch goes through the head's
    eval-with-targets pass -- the native preparation (selection, pairs, label association, the reference's host RNG) and
    the native launch plan's forward in exact fp32 -- with the next batch prepared on the side stream meanwhile, as
    trainer.Trainer.validate does.  Per-batch latency = call to results complete on the device.  Nothing is captured: a
    launch plan here is a struct on the stack, so there is no per-shape state to miss (no hit rate to report)."""
    from skghoi_amd import synth, trainer
    head = build_head(device, max_human=15, max_object=15).eval()
    rs = np.random.RandomState(seed)
    kh = np.arange(1, 16); ph = kh ** -1.2; ph /= ph.sum()
    ko = np.arange(1, 16); po = ko ** -0.8; po /= po.sum()
    o2v = synth.hico_object_to_verb()
    imgs = []
    for i in range(n_images):
        nh, no = int(rs.choice(kh, p=ph)), int(rs.choice(ko, p=po))
        im = synth.make_image(70000 + i, n_h=nh, n_o=no, out_channels=C_FEAT, pool=POOL)
        det = dict(boxes=im["boxes"], labels=im["labels"], scores=im["scores"])
        tg = synth.make_targets(det, 49, o2v, 800 + i, n_gt=3)
        imgs.append((det, tg, im["pooled"], im["feat3"], im["hw"]))
    batches = []
    for b0 in range(0, n_images, batch):
        chunk = imgs[b0:b0 + batch]
        dets = [{k: v.to(device) for k, v in c[0].items()} for c in chunk]
        tgs = [{k: v.to(device) for k, v in c[1].items()} for c in chunk]
        f3 = torch.cat([c[3] for c in chunk]).to(device)
        batches.append((OrderedDict((k, f3) for k in "0123"), dets, [c[4] for c in chunk], tgs,
                        torch.cat([c[2] for c in chunk]).to(device)))
    pool = ResidentPool(None)
    head.box_roi_pool = pool
    out = {}
    with torch.no_grad():
        for name in ("first_pass", "steady_state"):
            lat = np.zeros(len(batches))
            torch.cuda.synchronize()
            t_all = time.perf_counter()
            for k, (feats, dets, shp, tgs, pooled) in enumerate(batches):
                pool.pooled = pooled
                t0 = time.perf_counter()
                head(feats, dets, shp, tgs)
                if k + 1 < len(batches):                     # the look-ahead of Trainer.validate: the next batch's whole
                    nb = batches[k + 1]                      # preparation while the GPU runs this batch's forward
                    h = trainer.prefetch_batch(head, nb[0], nb[1], nb[2], nb[3])
                    if h is not None and os.environ.get("SKG_BENCH_VALIDATE_FINISH", "1") != "0":
                        h.finish()
                torch.cuda.synchronize()
                lat[k] = time.perf_counter() - t0
            wall = time.perf_counter() - t_all
            out[name] = dict(batches=len(batches), mean_ms=round(float(lat.mean()) * 1e3, 4),
                             p50_ms=round(float(np.percentile(lat, 50)) * 1e3, 4),
                             p95_ms=round(float(np.percentile(lat, 95)) * 1e3, 4),
                             images_per_s=round(n_images / wall, 1))
    out.update(batch=batch, images=n_images, max_human=15, max_object=15, precision="fp32",
               note="eval-mode forwards WITH targets (validation batches of 4: labels associated, sampling RNG consumed) over "
                    "a stream of batch shapes that never repeat; native preparation + native launch plan, next batch prepared "
                    "on the side stream; no captured graphs on this route")
    return out


def train_roofline(prec, ms, inf):
    """Roofline record of one training leg, everything measured in this run: whole-step fraction (plan arithmetic / step
    time) and the dense products' own fraction (HIP events around every skg_gemmx launch of a few untimed steps)."""
    gf = inf.get("gflop_per_step")
    if not gf:
        return None
    peak = PEAK_F16_MFMA_TFLOPS if prec == "bf16" else PEAK_F32_MFMA_TFLOPS
    ach = gf / ms                                                # GFLOP / ms = TFLOP/s
    rec = dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
               gflop_per_step=gf,
               note="2MNK of every dense product of the step (forward + backward, from the launch plan: skg_train_flops) / "
                    "step time / dense MFMA peak of the operand type; a batch-4 step is ~100 dependent launches of 2-120 us")
    dp = inf.get("dense_products")
    if dp and dp["ms_per_step"]:
        a2 = dp["gflop_per_step"] / dp["ms_per_step"]
        rec["dominant_kernel"] = dict(kernel="skg_gemmx_bf16_kernel" if prec == "bf16" else "skg_gemmx_kernel",
                                      ms_per_step=dp["ms_per_step"], launches_per_step=dp["launches_per_step"],
                                      gflop_per_step=dp["gflop_per_step"], achieved=round(a2, 2),
                                      dominant_kernel_frac=round(a2 / peak, 4),
                                      note="HIP events around every skg_gemmx launch of the plan (main kernel + its split-K "
                                           "reduce) in 10 untimed steps of this process; fp32: the four exact fc_2 products "
                                           "of the forward run on skg_gemm_kernel and are not in this figure")
    return rec


def release_plans(head):
    """Destroys the head's captured launch plans (idle device).  Every live hipGraphExec holds streams of its own, i.e.
    hardware queues; with the ~45 plans of the small-batch legs alive the training legs that follow IN THIS PROCESS ran the
    same kernels 0.5 ms per step slower (1.97 instead of 1.40 ms; tools/_trainleg_probe.py, engine.shared_side_stream).
    A training job never holds evaluation graphs: the legs release theirs when they are done."""
    eng = getattr(head, "_engine", None)
    small = getattr(eng, "_small", None)
    if small is not None:
        small.close()
        eng._small = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="override HeadEngine.chunk_images")
    ap.add_argument("--streams", type=int, default=0, help="override HeadEngine.n_streams")
    ap.add_argument("--no-gemm-timer", action="store_true", help="skip the per-launch HIP-event GEMM timing")
    ap.add_argument("--gemm-table", action="store_true", help="per-shape GEMM timing table on stderr")
    ap.add_argument("--precision", choices=["fp16x2", "fp32", "bf16"], default=None,
                    help="dense-layer path of the HEADLINE leg.  infer: fp32 (default; exact fp32 MFMA, the reference's "
                         "arithmetic) or fp16x2 (opt-in split operands on the fp16 MFMA).  train: fp32 (default) or bf16")
    ap.add_argument("--no-legs", action="store_true",
                    help="headline only: skip the extra legs (fp16x2, small-batch latency, training step)")
    ap.add_argument("--b1-stream", action="store_true", help="only the single-image shape-stream leg (prints its record)")
    ap.add_argument("--b4-validate", action="store_true", help="only the batch-4 validation-stream leg (prints its record)")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="train: NegativeSampling+MarginLoss training step (fwd+bwd+AdamW) as the headline instead")
    ap.add_argument("--dp-world1", action="store_true",
                    help="with --mode train on one GPU: open an RCCL process group of world size 1 and take the data-parallel "
                         "route (staged backward on the worker thread, arena chunks + normaliser through RCCL)")
    ap.add_argument("--measure", action="store_true",
                    help="with --mode train: add the phase spans and the dense-product timing of a few untimed steps")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher rehearsal on a box without a GPU: the ranks rendezvous (SKG_BENCH_BACKEND=gloo), run the "
                         "barrier / max-over-ranks / gather plumbing around an empty step and print the line; no head, no "
                         "throughput claim (value is null)")
    args = ap.parse_args()

    # ---- `bench.py --gpus N` launches its N ranks itself (the reference does: mp.spawn(main, nprocs=world_size),
    # configures/hicodet/adamixer_transH_spatial_r50_main.py:175-179).  Nothing in this process has touched the GPU yet.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    dist_on = world > 1
    if args.dry_run:
        return dry_run(args, rank, world)
    local = local % max(torch.cuda.device_count(), 1)          # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # torch's intra-op pool: the cores this process can really use (cgroup quota, not the host's core count), shared
    # between the ranks of the node; the host-bound training step wants few threads (its CPU ops are tiny)
    from skghoi_amd import trainer as _trainer
    from skghoi_amd.dist import host_cpu_share
    if args.mode == "train":
        _trainer.limit_host_threads(world)
    else:
        torch.set_num_threads(max(1, host_cpu_share() // world))
    if dist_on:
        import torch.distributed as dist
        backend = os.environ.get("SKG_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    elif args.dp_world1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
        dist.init_process_group(os.environ.get("SKG_BENCH_BACKEND", "nccl"), rank=0, world_size=1, device_id=device)

    if args.mode == "train":
        return train_mode(args, device, rank, world, dist_on)
    if args.b1_stream:
        print(json.dumps(b1_stream(device)))
        return
    if args.b4_validate:
        print(json.dumps(b4_validate(device)))
        return

    from skghoi_amd import engine
    head = build_head(device)
    dets, pooled, feats, shapes = make_inputs(args.batch, rank, device)
    head.box_roi_pool = ResidentPool(pooled)
    head.precision = args.precision or "fp32"
    if head.precision == "bf16":
        raise SystemExit("--precision bf16 is a training configuration (use --mode train)")
    if args.chunk:
        head.engine().chunk_images = args.chunk
    # The timed region runs the chunks on ONE stream: with the engine's default of two, GEMMs of neighbouring chunks
    # time-share the CUs (+1-2 % images/s) and the HIP events around a launch then time both of them -- the per-launch
    # durations of the roofline record would read twice what the kernel takes (and what rocprofv3, which serialises
    # kernels, reports).  The two-stream figure is reported next to it ("two_stream_chunks").
    default_streams = head.engine().n_streams
    head.engine().n_streams = args.streams or 1

    def step():
        with torch.no_grad():
            return head(feats, dets, shapes)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    traffic_table = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")     # HBM bytes per launch from the rocprofv3 --pmc passes
    if os.path.isfile(tpath):
        try:
            traffic_table = json.load(open(tpath))
        except Exception:
            traffic_table = {}

    # ---- headline leg: W warm-up steps, then exactly K timed steps between barriers
    torch.manual_seed(1234 + rank)
    for _ in range(args.warmup):
        res = step()
    assert len(res) == args.batch and res[0]["boxes_h"].shape == (N_H * (N_H + N_O - 1), 4)
    elapsed, timer = timed_infer(step, barrier, args.steps, not args.no_gemm_timer)
    timer_all = []
    if not args.no_gemm_timer:          # every GEMM launch of ONE extra, untimed step: the all-GEMM figures and the table
        engine.GEMM_TIMER = timer_all
        step()
        torch.cuda.synchronize()
        engine.GEMM_TIMER = None

    from skghoi_amd import dist as skd
    elapsed = skd.max_over_ranks(elapsed, device=device)
    total_images = sum(skd.gather_counts(args.batch * args.steps, device=device))
    value = total_images / elapsed

    t_all = sum(e0.elapsed_time(e1) * 1e-3 for e0, e1, *_ in timer_all) or 1e-9
    f_all = sum(2.0 * M * N * K for _, _, M, N, K, _ in timer_all)
    if args.gemm_table and rank == 0:
        tab = {}
        for e0, e1, M, N, K, epi in timer_all:
            g = tab.setdefault((M, N, K, epi), [0.0, 0])
            g[0] += e0.elapsed_time(e1); g[1] += 1
        for (M, N, K, epi), (ms, n) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
            print("M=%7d N=%5d K=%6d epi=%d  n=%3d  total %8.3f ms  avg %7.3f ms  %6.1f TFLOP/s" % (
                M, N, K, epi, n, ms, ms / n, 2.0 * M * N * K * n / ms / 1e9), file=sys.stderr)
    roofline, t_dom, n_dom = gemm_roofline(timer, head.precision, traffic_table)
    if roofline is None:
        roofline = dict(bound="mfma", kernel=GEMM_NAMES[2], achieved=None, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=None, traffic=None, note="per-launch timing disabled (--no-gemm-timer)")
    else:
        roofline.update(all_gemm_tflops=round(f_all / t_all / 1e12, 2),
                        gemm_share_of_step=round(t_all / (elapsed / args.steps), 4),
                        gflop_per_image=round(f_all / args.batch / 1e9, 3))
    # the north star also asks for the HBM-roofline view (it cannot bind -- ~7000 flop per compulsory byte)
    per_gpu = value / world
    traffic = roofline.get("traffic")
    roofline["hbm"] = dict(peak_gbs=HBM_PEAK_GBS,
                           compulsory_frac=round(per_gpu * COMPULSORY_BYTES_PER_IMAGE / (HBM_PEAK_GBS * 1e9), 5),
                           dominant_kernel_frac=(round(traffic / (t_dom / max(n_dom, 1)) / (HBM_PEAK_GBS * 1e9), 4)
                                                 if traffic and n_dom else None),
                           note="compulsory = 2.47 MB per image (pooled features + TransH tables + outputs, SURVEY 8d); "
                                "dominant kernel = HBM bytes per launch (traffic_source) / measured launch time")

    split = head.precision == "fp16x2"
    out = OrderedDict(metric="images/sec through interaction head (20x20 pairs)", value=round(value, 2),
                      unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                      ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
                      vs_baseline=None,
                      dtype="f32 (fp16x2 split operands, fp32 accumulate)" if split else "f32", data="synthetic",
                      config=dict(precision=head.precision, workload="HICO-DET-shaped synthetic cached detections: 20 humans x 20 objects per "
                                           "image (G=800 grid rows, P=780 pairs, K=117 verbs), eval forward of "
                                           "InteractionHead from cached AdaMixer-R50 box features [40,256,7,7]/image",
                                  batch_per_gpu=args.batch, images_per_step=args.batch * world,
                                  parallelism="dp%d (independent image shards, no data-path collective)" % world),
                      roofline=roofline, dist=dist_info(world))

    # ---- extra legs, same run, one GPU only (the N-GPU runs of the scaling curve stay headline-only)
    if world == 1 and not args.no_legs:
        if not args.streams and default_streams > 1:
            head.engine().n_streams = default_streams
            for _ in range(2):
                step()
            ks = max(2, min(args.steps, 5))
            els, _ = timed_infer(step, barrier, ks, False)
            out["two_stream_chunks"] = dict(value=round(args.batch * ks / els, 2), unit="images/s", steps=ks,
                                            ms_per_step=round(els / ks * 1e3, 3), n_streams=default_streams,
                                            note="HeadEngine's default: chunks alternate over two HIP streams")
            head.engine().n_streams = 1
        if head.precision == "fp32":
            # (a) opt-in fp16x2 path on the same inputs, with its own roofline
            head.precision = "fp16x2"
            k2 = max(2, min(args.steps, 5))
            for _ in range(2):
                step()
            el2, timer2 = timed_infer(step, barrier, k2, not args.no_gemm_timer)
            head.precision = "fp32"
            leg = dict(value=round(args.batch * k2 / el2, 2), unit="images/s", steps=k2,
                       ms_per_step=round(el2 / k2 * 1e3, 3), dtype="f32 (fp16x2 split operands, fp32 accumulate)",
                       note="opt-in precision='fp16x2': every GEMM operand carried as two fp16 numbers (22 significant "
                            "bits), three fp16-MFMA passes, fp32 accumulation; narrower than the reference's fp32")
            r2 = gemm_roofline(timer2, "fp16x2", traffic_table)[0]
            if r2 is not None:
                leg["roofline"] = r2
            out["fp16x2"] = leg
            step()                                  # back on the exact path (re-selects its kernels) before the next leg
        # (b) small batches: the reference's own evaluation runs one image per forward (utils.py:166-167)
        out["b1_latency_ms"] = round(small_batch_latency(head, dets, pooled, feats, shapes, 1), 4)
        out["b4_latency_ms"] = round(small_batch_latency(head, dets, pooled, feats, shapes, min(4, args.batch)), 4)
        out["b1_stream"] = b1_stream(device)
        if "b4_validate" not in os.environ.get("SKG_BENCH_SKIP", ""):        # (developer aid: legs to leave out)
            out["b4_validate"] = b4_validate(device)
        out["small_batch"] = dict(precision=head.precision, b1_images_per_s=round(1e3 / out["b1_latency_ms"], 1),
                                  b4_images_per_s=round(4e3 / out["b4_latency_ms"], 1),
                                  note="mean wall time per eval forward, 200 back-to-back forwards after 30 warm-ups")
        # (c) the training step at the reference's batch 4 per GPU (main:158): fwd + bwd + AdamW
        release_plans(head)
        _trainer.limit_host_threads(world)
        train = {}
        for prec in ("fp32", "bf16"):
            ks = 60                                                          # (steady state: the first ~10 steps still grow pools)
            el_t, losses, inf = run_train(4, prec, ks, 12, device, rank, world, False, measure=True)
            ms = el_t / ks * 1e3
            rec = dict(ms_per_step=round(ms, 3), images_per_s=round(4 * ks / el_t, 2), batch=4, steps=ks,
                       losses={k: round(v, 6) for k, v in losses.items()})
            el_i, _, _ = run_train(4, prec, 10, 4, device, rank, world, False, prefetch=False)
            rec["inline_ms_per_step"] = round(el_i / 10 * 1e3, 3)           # the same step without the look-ahead
            rec["roofline"] = train_roofline(prec, ms, inf)
            if inf.get("phases_ms"):
                rec["phases_ms"] = inf["phases_ms"]
            if inf.get("product_launches_per_step"):
                rec["product_launches_per_step"] = inf["product_launches_per_step"]     # which GEMM loop the step's products ran on
            train[prec] = rec
        # (d) the DATA-PARALLEL route of the bf16 step on this one GPU: RCCL process group of world size 1, arena exchange
        # installed, in a child process (RCCL's queues stay out of this one)
        dp = dp_world1_child("bf16", 60, 12)
        if "error" in dp:
            train["bf16"]["dp_world1"] = dp
        else:
            train["bf16"]["dp_world1"] = dict(ms_per_step=dp["ms_per_step"], images_per_s=dp["value"],
                                              vs_single_process=round(dp["ms_per_step"] / train["bf16"]["ms_per_step"], 4),
                                              grad_exchange=dp.get("grad_exchange"), dist=dp.get("dist"),
                                              runtime=dp.get("runtime"),
                                              note="same step through the data-parallel code path (staged backward on the "
                                                   "worker thread, 1 normaliser + k arena all-reduces through RCCL) in a "
                                                   "process group of ONE rank, child process of this run")
        train["note"] = ("NegativeSampling + MarginLoss + two focal terms, forward + backward + AdamW, 4 synthetic 20x20 "
                         "images with ground truth appended; bf16 = bf16 GEMM operands, fp32 accumulation / master weights; "
                         "every step hands the next batch to the head for preparation (prefetch), as a trainer over a "
                         "loader of cached detections does")
        out["train"] = train
        torch.set_num_threads(max(1, host_cpu_share() // world))
    elif world > 1 and not args.no_legs:
        # ---- N > 1: the data-parallel TRAINING step too (BASELINE config 4: batch 4 per GPU, bf16, gradient all-reduce over
        # RCCL / xGMI) -- every rank runs it; value = whole-job images/s over the max-over-ranks time
        _trainer.limit_host_threads(world)
        ks = 40
        # This leg is the first code to move gradients between real GPUs (a one-GPU box cannot): whatever goes wrong in it
        # must not cost the run its headline line, and a hang must show in the exit code (LegGuard): an exception becomes an
        # error record (rc 0); a leg that is not back within the deadline makes rank 0 print the line with the error and every
        # rank's evidence, and every rank exit with EXIT_LEG_STUCK.
        probe = {}

        def evidence():
            """What this rank can say about where its step stands (no GPU call, no lock a stuck step could hold)."""
            ev = dict(rank=rank)
            head_ = probe.get("head")
            if head_ is None:
                return ev
            from skghoi_amd import _capi
            ctx = head_.__dict__.get("_train_ctx")
            if ctx is not None and getattr(ctx, "_h", None):
                prog = int(_capi.lib().skg_ctx_train_backward_progress(ctx._h))
                ev.update(backward_stages_issued=prog & 0xff, backward_pending=bool(prog & 0x100))
            ex = getattr(head_, "grad_exchange", None)
            if ex is not None:
                native = getattr(ex, "native", None)
                ev["route"] = "libskghoi_hip's own RCCL communicator" if native is not None else "torch.distributed"
                if native is not None:
                    ev["collectives_issued"] = int(_capi.lib().skg_comm_collectives(native.handle))
                ev["python_side_collectives"] = getattr(ex, "collectives", None)
            return ev

        guard = LegGuard(rank, world, out, float(os.environ.get("SKG_BENCH_TRAIN_DEADLINE", "300")), evidence=evidence)
        leg_ok = True
        try:
            el_t, losses, inf = run_train(4, "bf16", ks, 10, device, rank, world, True, measure=True, probe=probe)   # (every rank: the steps hold collectives)
            el_t = skd.max_over_ranks(el_t, device=device)
            ms = el_t / ks * 1e3
            rec = dict(ms_per_step=round(ms, 3), images_per_s=round(4 * world * ks / el_t, 2), batch_per_gpu=4, steps=ks,
                       n_gpus=world, scaling="weak", losses={k: round(v, 6) for k, v in losses.items()},
                       grad_exchange=inf.get("grad_exchange"), dist=dist_info(world))
            rec["roofline"] = train_roofline("bf16", ms, inf)
            if inf.get("phases_ms"):
                rec["phases_ms"] = inf["phases_ms"]
            guard.finish(dict(bf16=rec, note="data-parallel training step on every rank: fwd + bwd + AdamW at batch 4 per GPU, "
                                             "gradient arena exchanged in chunks behind the backward + one fused normaliser "
                                             "all-reduce (RCCL); ms_per_step = max over ranks; roofline of rank 0"))
        except Exception as e:                                   # noqa: BLE001
            leg_ok = False
            guard.failed(e)
        if not leg_ok:
            # the peers are leaving through their watchdogs (they saw this rank's word in the store): print and go, without a
            # collective teardown that would wait for them
            out["runtime"] = runtime_info()
            if rank == 0:
                print(json.dumps(out), flush=True)
                time.sleep(2.0)                  # (rank 0 hosts the store: the peers must get to read this rank's word)
            os._exit(0)
        torch.set_num_threads(max(1, host_cpu_share() // world))
    out["runtime"] = runtime_info()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
