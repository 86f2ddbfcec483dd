"""Developer aid: cProfile of the host side of one bench step (run on the GPU box)."""
import cProfile, pstats, sys, os
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dets, pooled, feats, shapes = bench.make_inputs(B, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
with torch.no_grad():
    head(feats, dets, shapes); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3):
        head(feats, dets, shapes)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "cumulative").print_stats(45)
