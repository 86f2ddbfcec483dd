"""Developer aid (GPU box): is the batch-4 training loop bound by the host thread?  Wall time per step against the CPU time
of the Python thread (time.thread_time) and of the whole process (HIP runtime threads included), with look-ahead.
usage: train_cpu_probe.py [precision=bf16] [batch=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import synth, trainer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
trainer.limit_host_threads()
device = torch.device("cuda:0")
head = bench.build_head(device).train()
head.precision = prec
dets, pooled, feats, shapes = bench.make_inputs(B, 0, device)
o2v = synth.hico_object_to_verb()
cpu_dets = [dict(boxes=d["boxes"].cpu(), labels=d["labels"].cpu(), scores=d["scores"].cpu()) for d in dets]
targets = [{k: v.to(device) for k, v in synth.make_targets(d, 49, o2v, 500 + i, n_gt=4).items()} for i, d in enumerate(cpu_dets)]


class Pool(torch.nn.Module):
    def forward(self, features, boxes, image_shapes):
        n = sum(len(b) for b in boxes)
        reps = (n + pooled.shape[0] - 1) // pooled.shape[0]
        return pooled.repeat(reps, 1, 1, 1)[:n]


head.box_roi_pool = Pool()
net = trainer.wrap_ddp(head, device)
opt = trainer.build_optimizer(net, lr=1e-4)
nxt = (feats, dets, shapes, targets)
for _ in range(10):
    trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt)
torch.cuda.synchronize()
for rep in range(3):
    N = 100
    w0, c0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
    for _ in range(N):
        trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt)
    c1, p1 = time.thread_time(), time.process_time()
    torch.cuda.synchronize()
    w1 = time.perf_counter()
    print("%s batch %d: wall %.3f ms/step, python thread CPU %.3f, process CPU %.3f" %
          (prec, B, (w1 - w0) / N * 1e3, (c1 - c0) / N * 1e3, (p1 - p0) / N * 1e3), flush=True)
