"""Developer aid: wall time of consecutive forward calls at a given batch (no profiler)."""
import sys, time
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
head = bench.build_head(dev)
dets, pooled, feats, shapes = bench.make_inputs(B, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
eng = head.engine()
import skghoi_amd.engine as E
orig_pre = eng.preprocess; orig_graph = eng.graph
T = {}
def tw(name, f):
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
    return g
eng.preprocess = tw("pre", orig_pre); eng.graph = tw("graph", orig_graph)
eng.classify = tw("cls", eng.classify); eng.score = tw("score", eng.score)
head._results = tw("results", head._results)
with torch.no_grad():
    for i in range(12):
        T.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = head(feats, dets, shapes)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("step %2d host %.2f ms  +sync %.2f ms   " % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3) +
              " ".join("%s=%.2f" % (k, v * 1e3) for k, v in T.items()), flush=True)
