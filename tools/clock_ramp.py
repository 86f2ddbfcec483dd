"""Developer aid: launch time of the split GEMM over a long back-to-back run (DVFS ramp)."""
import sys, time
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd.engine import gemm, SplitWeights
M, N, K = 51200, 1024, 1024
g = torch.Generator().manual_seed(0)
A = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
b = torch.rand(N, generator=g).cuda(); C = torch.empty(M, N, device="cuda")
with SplitWeights():
    gemm(A, W, b, C, M, N, K, 1); torch.cuda.synchronize()
    time.sleep(1.0)
    t0 = time.perf_counter()
    for blk in range(30):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            gemm(A, W, b, C, M, N, K, 1)
        e1.record(); torch.cuda.synchronize()
        print("t=%.3f s  %.4f ms/launch" % (time.perf_counter() - t0, e0.elapsed_time(e1) / 100), flush=True)
