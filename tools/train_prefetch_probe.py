"""Developer aid (GPU box): host-side phase times of the batch-4 training step with and without prefetch.
usage: train_prefetch_probe.py [precision=bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import synth, trainer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
trainer.limit_host_threads()
device = torch.device("cuda:0")
head = bench.build_head(device).train()
head.precision = prec
dets, pooled, feats, shapes = bench.make_inputs(4, 0, device)
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
pc = time.perf_counter
for mode in ("inline", "prefetch", "inline", "prefetch"):
    nxt = (feats, dets, shapes, targets) if mode != "inline" else None
    for _ in range(6):
        trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt)
    torch.cuda.synchronize()
    N = 40
    acc = dict(zero=0.0, fwd=0.0, bwd=0.0, opt=0.0, pre=0.0)
    t0 = pc()
    for _ in range(N):
        a = pc(); opt.zero_grad(set_to_none=True)
        b = pc(); out = net(feats, dets, shapes, targets); ld = out.pop(); total = sum(ld.values())
        c0 = pc(); ahead = trainer.prefetch_batch(net, *nxt) if nxt is not None else None
        c = pc(); total.backward()
        d0 = pc()
        if ahead is not None:
            ahead.advance()
        d = pc(); opt.step()
        e = pc()
        if ahead is not None:
            ahead.finish()
        f = pc()
        acc["zero"] += b - a; acc["fwd"] += c0 - b; acc["bwd"] += d0 - c; acc["opt"] += e - d
        acc["pre"] += (c - c0) + (d - d0) + (f - e)
    t1 = pc()
    torch.cuda.synchronize()
    t2 = pc()
    print("%-8s host %.3f ms/step (with final sync %.3f)  " % (mode, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3) +
          "  ".join("%s %.3f" % (k, v / N * 1e3) for k, v in acc.items()), flush=True)
    if mode != "inline":
        head._take_prefetched(None, None, None)
