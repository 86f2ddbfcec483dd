"""Developer aid (GPU box): the batch-4 training loop at the bench shapes, N steps with the backward issued by the library's
worker thread against the same N steps with it issued inline: final weights must be bit-identical.
usage: defer_stress.py [precision=bf16] [steps=300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import synth, trainer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
trainer.limit_host_threads()
device = torch.device("cuda:0")
dets, pooled, feats, shapes = bench.make_inputs(4, 0, device)
o2v = synth.hico_object_to_verb()
cpu_dets = [dict(boxes=d["boxes"].cpu(), labels=d["labels"].cpu(), scores=d["scores"].cpu()) for d in dets]
targets = [{k: v.to(device) for k, v in synth.make_targets(d, 49, o2v, 500 + i, n_gt=4).items()} for i, d in enumerate(cpu_dets)]


class Pool(torch.nn.Module):
    def forward(self, features, boxes, image_shapes):
        n = sum(len(b) for b in boxes)
        reps = (n + pooled.shape[0] - 1) // pooled.shape[0]
        return pooled.repeat(reps, 1, 1, 1)[:n]


def run(defer):
    torch.manual_seed(5)
    head = bench.build_head(device).train()
    head.precision = prec
    head.box_roi_pool = Pool()
    if not defer:
        inner = head.fused_step
        head.fused_step = lambda *a, defer_backward=False, **k: inner(*a, defer_backward=False, **k)
    net = trainer.wrap_ddp(head, device)
    opt = trainer.build_optimizer(net, lr=1e-4)
    torch.manual_seed(1234)
    nxt = (feats, dets, shapes, targets)
    for _ in range(N):
        losses, _ = trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt)
    torch.cuda.synchronize()
    return head, trainer.read_losses(losses)


h0, l0 = run(False)
h1, l1 = run(True)
bad = [k for (k, a), (_, b) in zip(h0.state_dict().items(), h1.state_dict().items()) if not torch.equal(a, b)]
print("%s, %d steps: losses inline %s / worker %s; %d of %d tensors differ" % (prec, N, l0, l1, len(bad), len(h0.state_dict())))
sys.exit(1 if bad or l0 != l1 else 0)
