"""Developer aid (GPU box): host cProfile + wall time of the single-image eval forward (bucket plans, steady state)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
from collections import OrderedDict
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
dets, pooled, feats, shapes = bench.make_inputs(1, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
with torch.no_grad():
    for _ in range(30):
        head(feats, dets, shapes)
    torch.cuda.synchronize()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N):
        head(feats, dets, shapes)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("B=1: host %.1f us/forward, with final sync %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(N):
        head(feats, dets, shapes)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
