"""Developer aid (GPU box): cProfile of the host side of the training step at batch 4.
usage: train_host_profile.py [precision=bf16] [batch=4]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from skghoi_amd import synth, trainer

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
LAZY = not (len(sys.argv) > 3 and sys.argv[3] == "eager")
PF = len(sys.argv) > 4 and sys.argv[4] == "prefetch"
trainer.limit_host_threads()
device = torch.device("cuda:0")
if os.environ.get("SKG_PROFILE_AFTER_GC"):        # a first training run, then gc.collect() + empty_cache(): the slow mode
    import gc
    bench.run_train(B, prec, 20, 5, device, 0, 1, False)
    gc.collect(); torch.cuda.empty_cache()
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
NXT = (feats, dets, shapes, targets) if PF else None
for _ in range(5):
    trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=LAZY, prefetch=NXT)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=LAZY, prefetch=NXT)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host %.3f ms/step, with final sync %.3f ms/step" % ((t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    trainer.train_step(net, opt, feats, dets, shapes, targets=targets, lazy=LAZY, prefetch=NXT)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / N * 1e6, ct / N * 1e6, nc / N, "%s:%d(%s)" % (fn.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", ""), line, name)))
print("total under the profiler: %.0f us/step" % sum(r[0] for r in rows))
print("--- by own time (us/step own, us/step cumulative, calls/step)")
for r in sorted(rows, key=lambda r: -r[0])[:70]:
    print("%8.1f %8.1f %7.1f  %s" % r)
print("--- by cumulative time")
for r in sorted(rows, key=lambda r: -r[1])[:60]:
    print("%8.1f %8.1f %7.1f  %s" % r)
