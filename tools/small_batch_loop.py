"""Developer aid (GPU box): B-image eval forwards in a loop, for rocprofv3 --kernel-trace --stats and host profiling.
usage: small_batch_loop.py [B=1] [iters=300] [precision=fp32] [small=1] [cprofile=0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
from collections import OrderedDict
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
small = int(sys.argv[4]) if len(sys.argv) > 4 else 1
prof = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda", 0)
if os.environ.get("SKG_SMALL_MODE"):
    from skghoi_amd import _capi
    _capi.set_tuning(small_mode=int(os.environ["SKG_SMALL_MODE"]))
head = bench.build_head(dev)
head.precision = prec
dets, pooled, feats, shapes = bench.make_inputs(B, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
head.engine().small_batch_max = 8 if small else 0
torch.manual_seed(0)


def loop(n):
    with torch.no_grad():
        for _ in range(n):
            head(feats, dets, shapes)
    torch.cuda.synchronize()


loop(30)
t0 = time.perf_counter()
if prof:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); loop(iters); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
else:
    loop(iters)
dt = (time.perf_counter() - t0) / iters * 1e3
print("B=%d precision=%s small=%d: %.4f ms per forward (%.1f images/s)" % (B, prec, small, dt, B * 1e3 / dt))
