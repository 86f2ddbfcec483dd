"""Train -> validate -> train in ONE process (what the reference's engine does at every epoch boundary, utils.py:232-249):
60 timed batch-4 bf16 training steps, a validation pass (eval mode WITH targets at batch 4, plus single-image eval forwards
that replay a captured launch plan), 60 more training steps.  Prints one JSON line with both blocks' step times.

    python tools/alternation_probe.py [hw_queues]      # 0 / absent = the runtime's default queue count"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from skghoi_amd import runtime  # noqa: E402
runtime.configure(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
import torch  # noqa: E402
import bench  # noqa: E402
from skghoi_amd import synth, trainer  # noqa: E402

dev = torch.device("cuda", 0)
trainer.limit_host_threads(1)
head = bench.build_head(dev).train()
head.precision = "bf16"
dets, pooled, feats, shapes = bench.make_inputs(4, 0, dev)
o2v = synth.hico_object_to_verb()
targets = [{k: v.to(dev) for k, v in synth.make_targets(dict(boxes=d["boxes"].cpu(), labels=d["labels"].cpu(),
                                                             scores=d["scores"].cpu()), 49, o2v, 500 + i, n_gt=4).items()}
           for i, d in enumerate(dets)]


class Pool(torch.nn.Module):
    def forward(self, features, boxes, image_shapes):
        n = sum(len(b) for b in boxes)
        reps = (n + pooled.shape[0] - 1) // pooled.shape[0]
        return pooled.repeat(reps, 1, 1, 1)[:n]


head.box_roi_pool = Pool()
net = trainer.wrap_ddp(head, dev)
opt = trainer.build_optimizer(net, lr=1e-4)
torch.manual_seed(3)
nxt = (feats, dets, shapes, targets)


def block(n=60, warm=12):
    head.train()
    for _ in range(warm):
        trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=True, prefetch=nxt)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = dict(runtime=runtime.info())
out["train_block_1_ms"] = round(block(), 4)
tr = trainer.Trainer(net, opt, None, None, val_loader=[(feats, dets, shapes, targets)] * 3, num_classes=117, device=dev)
ap = tr.validate()                                            # eval mode with targets, batch 4 (utils.py:283-299)
out["validation_map"] = round(float(ap.mean()), 6)
f1 = type(feats)((k, v[:1]) for k, v in feats.items())
with torch.no_grad():
    for _ in range(20):                                       # the reference's test loop: one image per forward
        res = head(f1, dets[:1], shapes[:1])
torch.cuda.synchronize()
out["graph_plans"] = len(head.engine()._small.plans) if head.engine()._small is not None else 0
out["train_block_2_ms"] = round(block(), 4)
out["ratio"] = round(out["train_block_2_ms"] / out["train_block_1_ms"], 4)
print(json.dumps(out))
