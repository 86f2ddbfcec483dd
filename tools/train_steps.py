import sys, time
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch, bench
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import synth, trainer
dev = torch.device("cuda", 0); B = 4
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
opt = trainer.build_optimizer(head, lr=1e-4)
torch.manual_seed(1234)
"""Developer aid: wall time of the phases of consecutive training steps (forward / backward / optimizer), each synced."""
import gc
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = head(feats, dets, shapes, targets); loss_dict = out.pop()
    total = sum(loss_dict.values())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    total.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("step %2d fwd %.1f  bwd %.1f  opt %.1f ms   gc %s  alloc %.0f MB reserved %.0f MB" % (
        i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, gc.get_count(),
        torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20), flush=True)
