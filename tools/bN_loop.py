"""Developer aid (GPU box): N eval forwards at a small batch, for rocprofv3 --kernel-trace (tools/replay_timeline.py).
usage: bN_loop.py [batch=4] [forwards=60]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
dets, pooled, feats, shapes = bench.make_inputs(B, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
with torch.no_grad():
    for _ in range(20):
        head(feats, dets, shapes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        head(feats, dets, shapes)
    torch.cuda.synchronize()
    print("B=%d: %.3f ms per forward" % (B, (time.perf_counter() - t0) / N * 1e3))
