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


def build_head(device):
    from skghoi_amd import GraphHead, InteractionHead, synth
    o2v = synth.hico_object_to_verb()
    gh = GraphHead(C_FEAT, POOL, 1024, 1024, 117, 49, o2v, num_iter=2)
    head = InteractionHead(torch.nn.Identity(), gh, torch.nn.Linear(2048, 1), torch.nn.Linear(2048, 117),
                           human_idx=49, num_classes=117, max_human=N_H, max_object=N_O)
    head.load_state_dict(synth.make_state_dict(117, C_FEAT, POOL, seed=0))
    return head.to(device).eval()


def cpu_baseline(n_images):
    """The oracle (kind 'port': loop-for-loop restatement of the reference head, incl. its Python row loop) on the
    host cores, same synthetic workload, batch 1 per forward like the reference's inference (utils.py:166-167)."""
    from oracle import skg_oracle as O
    from skghoi_amd import synth
    sd = synth.make_state_dict(117, C_FEAT, POOL, seed=0)
    o2v = synth.hico_object_to_verb()
    cores = torch.get_num_threads()               # set in main() to the process's CPU share

    times = []
    with torch.no_grad():
        for i in range(n_images + 1):
            im = synth.make_image(1000 + i, n_h=N_H, n_o=N_O, out_channels=C_FEAT, pool=POOL)
            det = [dict(boxes=im["boxes"], labels=im["labels"], scores=im["scores"])]
            torch.manual_seed(i)
            t0 = time.perf_counter()
            O.interaction_head_forward(sd, im["feat3"], det, [im["hw"]], lambda c: im["pooled"], 117, 49, o2v,
                                       max_human=N_H, max_object=N_O, row_loop=True)
            dt = time.perf_counter() - t0
            if i > 0:                       # first image = warm-up
                times.append(dt)
            if sum(times) > 25.0:
                break
    return dict(value=round(len(times) / sum(times), 4), unit="images/s", cores=cores, kind="port",
                sample="%d images (20x20), batch 1 per forward, fp32 torch CPU, %d threads, after 1 warm-up; "
                       "median %.3f s/image" % (len(times), cores, float(np.median(times))))


def train_mode(args, device, rank, world, dist_on):
    """Secondary metric: images/s of the data-parallel training step (fp32; forward + backward on the HIP GEMMs,
    AdamW, DDP gradient all-reduce over RCCL when world > 1).  Reference settings: batch 4 per GPU (main:158)."""
    import torch.distributed as dist
    from skghoi_amd import synth, trainer
    B = args.batch if args.batch != 256 else 4
    head = build_head(device).train()
    head.distributed = dist_on
    head.precision = args.precision or "fp32"
    args.precision = head.precision
    dets, pooled, feats, shapes = make_inputs(B, rank, device)
    o2v = synth.hico_object_to_verb()
    cpu_dets = [dict(boxes=d["boxes"].cpu(), labels=d["labels"].cpu(), scores=d["scores"].cpu()) for d in dets]
    targets = [{k: v.to(device) for k, v in synth.make_targets(d, 49, o2v, 500 + i, n_gt=4).items()}
               for i, d in enumerate(cpu_dets)]

    class Pool(torch.nn.Module):
        def forward(self, features, boxes, image_shapes):
            n = sum(len(b) for b in boxes)           # GT boxes are appended in training: size follows the head
            reps = (n + pooled.shape[0] - 1) // pooled.shape[0]
            return pooled.repeat(reps, 1, 1, 1)[:n]

    head.box_roi_pool = Pool()
    net = trainer.wrap_ddp(head, device)
    opt = trainer.build_optimizer(net, lr=1e-4)
    torch.manual_seed(1234 + rank)
    for _ in range(args.warmup):
        losses, _ = trainer.train_step(net, opt, feats, dets, shapes, targets=targets)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses, _ = trainer.train_step(net, opt, feats, dets, shapes, targets=targets)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    from skghoi_amd import dist as skd
    elapsed = skd.max_over_ranks(time.perf_counter() - t0, device=device)
    if rank == 0:
        print(json.dumps(dict(metric="images/sec through the interaction-head TRAINING step (20x20 pairs)",
                              value=round(B * world * args.steps / elapsed, 2), unit="images/s", n_gpus=world,
                              steps=args.steps, warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3),
                              higher_is_better=True, scaling="weak", vs_baseline=None,
                              dtype="bf16" if args.precision == "bf16" else "f32", data="synthetic",
                              config=dict(workload="train step: fwd + bwd + AdamW, NegativeSampling + MarginLoss + "
                                                   "two focal terms, 20x20 synthetic images with GT appended",
                                          batch_per_gpu=B, parallelism="dp%d" % world),
                              losses=losses)))
    if dist_on:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=0, help="override HeadEngine.chunk_images")
    ap.add_argument("--streams", type=int, default=0, help="override HeadEngine.n_streams")
    ap.add_argument("--no-gemm-timer", action="store_true", help="skip the per-launch HIP-event GEMM timing")
    ap.add_argument("--gemm-table", action="store_true", help="per-shape GEMM timing table on stderr")
    ap.add_argument("--precision", choices=["fp16x2", "fp32", "bf16"], default=None,
                    help="dense-layer path.  infer: fp16x2 (default; fp32-grade split operands on the fp16 MFMA) or "
                         "fp32 (exact fp32 MFMA).  train: fp32 (default) or bf16")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the extra timed steps on the exact fp32 path")
    ap.add_argument("--mode", choices=["infer", "train"], default="infer",
                    help="train: NegativeSampling+MarginLoss training step (fwd+bwd+AdamW), secondary metric")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
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

    if args.mode == "train":
        return train_mode(args, device, rank, world, dist_on)

    from skghoi_amd import engine
    head = build_head(device)
    dets, pooled, feats, shapes = make_inputs(args.batch, rank, device)
    head.box_roi_pool = ResidentPool(pooled)
    head.precision = args.precision or "fp16x2"
    if head.precision == "bf16":
        raise SystemExit("--precision bf16 is a training configuration (use --mode train)")
    if args.chunk:
        head.engine().chunk_images = args.chunk
    if args.streams:
        head.engine().n_streams = args.streams

    def step():
        with torch.no_grad():
            return head(feats, dets, shapes)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(1234 + rank)
    for _ in range(args.warmup):
        res = step()
    assert len(res) == args.batch and res[0]["boxes_h"].shape == (N_H * (N_H + N_O - 1), 4)

    # HIP events around the launches of the dominant kernel only (MUL_RELU epilogue: the three MBF fc_2 GEMMs per
    # chunk): an event pair costs a few microseconds of GPU time, ~150 pairs per step would be 4 % of the step
    engine.GEMM_TIMER = None if args.no_gemm_timer else []
    engine.GEMM_TIMER_EPI = {2}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    elapsed = time.perf_counter() - t0
    timer, engine.GEMM_TIMER = (engine.GEMM_TIMER or []), None
    engine.GEMM_TIMER_EPI = None
    timer_all = []
    if not args.no_gemm_timer:          # every GEMM launch of ONE extra, untimed step: the all-GEMM figures and the table
        engine.GEMM_TIMER = timer_all
        step()
        torch.cuda.synchronize()
        engine.GEMM_TIMER = None

    # ---- secondary leg, same run: the exact fp32-MFMA path (precision="fp32") on the same inputs
    exact = None
    if head.precision != "fp32" and not args.no_exact_leg:
        main_precision, head.precision = head.precision, "fp32"
        k2 = max(2, min(args.steps, 5))
        step(); step()
        engine.GEMM_TIMER = None if args.no_gemm_timer else []
        engine.GEMM_TIMER_EPI = {2}
        barrier()
        t1 = time.perf_counter()
        for _ in range(k2):
            step()
        barrier()
        el2 = time.perf_counter() - t1
        timer2, engine.GEMM_TIMER = (engine.GEMM_TIMER or []), None
        engine.GEMM_TIMER_EPI = None
        head.precision = main_precision
        t_d = sum(e0.elapsed_time(e1) * 1e-3 for e0, e1, *_ in timer2)
        f_d = sum(2.0 * M * N * K for _, _, M, N, K, _ in timer2)
        exact = (el2, k2, (f_d / t_d / 1e12) if t_d > 0 else None, (t_d / len(timer2) * 1e3) if timer2 else None)

    from skghoi_amd import dist as skd
    elapsed = skd.max_over_ranks(elapsed, device=device)
    total_images = sum(skd.gather_counts(args.batch * args.steps, device=device))
    value = total_images / elapsed

    # ---- roofline of the dominant kernel: per-launch HIP-event durations, grouped by kernel instance (epilogue)
    groups = {}
    for e0, e1, M, N, K, epi in timer:
        g = groups.setdefault(epi, [0.0, 0.0, 0])
        g[0] += e0.elapsed_time(e1) * 1e-3; g[1] += 2.0 * M * N * K; g[2] += 1
    t_all = sum(e0.elapsed_time(e1) * 1e-3 for e0, e1, *_ in timer_all)
    f_all = sum(2.0 * M * N * K for _, _, M, N, K, _ in timer_all)
    if args.gemm_table and rank == 0:
        tab = {}
        for e0, e1, M, N, K, epi in timer_all:
            g = tab.setdefault((M, N, K, epi), [0.0, 0])
            g[0] += e0.elapsed_time(e1); g[1] += 1
        for (M, N, K, epi), (ms, n) in sorted(tab.items(), key=lambda kv: -kv[1][0]):
            print("M=%7d N=%5d K=%6d epi=%d  n=%3d  total %8.3f ms  avg %7.3f ms  %6.1f TFLOP/s" % (
                M, N, K, epi, n, ms, ms / n, 2.0 * M * N * K * n / ms / 1e9), file=sys.stderr)
    names = {0: "skg_gemm_kernel<BIAS>", 1: "skg_gemm_kernel<BIAS_RELU>", 2: "skg_gemm_kernel<MUL_RELU>",
             3: "skg_gemm_kernel<RELU_DOT>", 4: "skg_gemm_kernel<BIAS_RES_RELU>", 5: "skg_gemm_group_kernel"}
    if not groups:
        groups = {2: [1e-9, 0.0, 0]}
    dom = max(groups, key=lambda k: groups[k][0])
    t_dom, f_dom, n_dom = groups[dom]
    t_all = t_all or 1e-9
    achieved = f_dom / t_dom / 1e12
    split = head.precision == "fp16x2"
    peak = PEAK_F16_MFMA_TFLOPS / F16X2_PASSES if split else PEAK_F32_MFMA_TFLOPS
    if split:
        names = {k: v.replace(">", ", fp16x2>") if k != 5 else v for k, v in names.items()}
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")     # HBM bytes per launch from the rocprofv3 --pmc passes
    if os.path.isfile(tpath):
        try:
            traffic = json.load(open(tpath)).get(names[dom])
        except Exception:
            traffic = None
    roofline = dict(bound="mfma", kernel=names[dom], achieved=round(achieved, 2), peak=round(peak, 1),
                    unit="TFLOP/s", frac=round(achieved / peak, 4), traffic=traffic,
                    peak_note=("useful (2MNK) flops; peak = fp16 dense MFMA %.0f TF / %d MFMA passes per useful flop"
                               % (PEAK_F16_MFMA_TFLOPS, F16X2_PASSES)) if split else "fp32 MFMA dense peak",
                    launches=n_dom, avg_launch_ms=round(t_dom / max(n_dom, 1) * 1e3, 4),
                    all_gemm_tflops=round(f_all / t_all / 1e12, 2),
                    gemm_share_of_step=round(t_all / (elapsed / args.steps), 4),
                    gflop_per_image=round(f_all / args.batch / 1e9, 3))
    # the north star also asks for the HBM-roofline view (it cannot bind -- ~7000 flop per compulsory byte)
    per_gpu = value / world
    roofline["hbm"] = dict(peak_gbs=HBM_PEAK_GBS,
                           compulsory_frac=round(per_gpu * COMPULSORY_BYTES_PER_IMAGE / (HBM_PEAK_GBS * 1e9), 5),
                           dominant_kernel_frac=(round(traffic / (t_dom / max(n_dom, 1)) / (HBM_PEAK_GBS * 1e9), 4)
                                                 if traffic else None),
                           note="compulsory = 2.47 MB per image (pooled features + TransH tables + outputs, SURVEY 8d); "
                                "dominant kernel = measured HBM bytes per launch / launch time")

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
                      roofline=roofline)
    if exact is not None:
        el2 = skd.max_over_ranks(exact[0], device=device)
        out["exact_fp32"] = dict(value=round(args.batch * world * exact[1] / el2, 2), unit="images/s", steps=exact[1],
                                 ms_per_step=round(el2 / exact[1] * 1e3, 3),
                                 note="same run, same inputs, precision='fp32' (exact fp32 MFMA everywhere)")
        if exact[2] is not None:
            out["exact_fp32"]["roofline"] = dict(bound="mfma", kernel="skg_gemm_kernel<MUL_RELU>", achieved=round(exact[2], 2),
                                                 peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                                                 frac=round(exact[2] / PEAK_F32_MFMA_TFLOPS, 4),
                                                 avg_launch_ms=round(exact[3], 4))
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(12)
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
