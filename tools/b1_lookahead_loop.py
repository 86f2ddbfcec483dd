"""Developer aid (GPU box): the single-image eval loop with and without the one-image look-ahead (InteractionHead.
prefetch_eval), synchronised per forward like bench.py's b1_stream; prints the per-forward latency and how it splits into
the host's call time and the wait for the device.  usage: b1_lookahead_loop.py [ahead=1] [iters=400] [n_h=20] [n_o=20]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import numpy as np
import torch
from skghoi_amd import runtime as _rt; _rt.configure()
import bench

ahead = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
dets, pooled, feats, shapes = bench.make_inputs(1, 0, dev)
head.box_roi_pool = bench.ResidentPool(pooled)
torch.manual_seed(0)
resident = torch.cuda.Event(); resident.record()


def loop(n, ahead):
    call, pf, wait = np.zeros(n), np.zeros(n), np.zeros(n)
    with torch.no_grad():
        for k in range(n):
            t0 = time.perf_counter()
            head(feats, dets, shapes)
            t1 = time.perf_counter()
            if ahead:
                head.prefetch_eval(dets, after=resident)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            call[k], pf[k], wait[k] = t1 - t0, t2 - t1, t3 - t2
    head.engine()._small.drop_look_ahead()
    return call * 1e3, pf * 1e3, wait * 1e3


loop(40, ahead)
if os.environ.get("SKG_CPROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); loop(iters, ahead); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)
    sys.exit(0)
for mode in ((ahead, 1 - ahead, ahead) if len(sys.argv) <= 3 or True else (ahead,)):
    c, p, w = loop(iters, mode)
    tot = c + p + w
    print("look-ahead=%d: %.4f ms per forward (p50 %.4f) = forward call %.4f + prefetch call %.4f + wait %.4f (medians %.4f / %.4f / %.4f)"
          % (mode, tot.mean(), np.median(tot), c.mean(), p.mean(), w.mean(), np.median(c), np.median(p), np.median(w)))
