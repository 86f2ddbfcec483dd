"""Developer aid: cProfile + kernel summary of the training step (GPU box)."""
import cProfile, pstats, sys, time
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import synth, trainer
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
head = bench.build_head(dev).train()
dets, pooled, feats, shapes = bench.make_inputs(B, 0, dev)
o2v = synth.hico_object_to_verb()
cpu_dets = [dict(boxes=d["boxes"].cpu(), labels=d["labels"].cpu(), scores=d["scores"].cpu()) for d in dets]
targets = [{k: v.to(dev) for k, v in synth.make_targets(d, 49, o2v, 500 + i, n_gt=4).items()} for i, d in enumerate(cpu_dets)]
class Pool(torch.nn.Module):
    def forward(self, features, boxes, image_shapes):
        n = sum(len(b) for b in boxes); reps = (n + pooled.shape[0] - 1) // pooled.shape[0]
        return pooled.repeat(reps, 1, 1, 1)[:n]
head.box_roi_pool = Pool()
opt = trainer.build_optimizer(head)
for _ in range(2):
    trainer.train_step(head, opt, feats, dets, shapes, targets=targets)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(3):
    trainer.train_step(head, opt, feats, dets, shapes, targets=targets)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 3 * 1e3)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
