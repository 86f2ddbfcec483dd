"""Developer aid (GPU box): time of ONE small exact-fp32 GEMM launch (64 x 64 tiles, latency loop) at M rows, N = 1024, over K
-> us per launch and us per 64-k step.  usage: gemm_k3_time.py [M=400]   (SKG_LIB / SKG_SMALL_MODE select build and loop)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()
from skghoi_amd import _capi
from skghoi_amd.engine import gemm
M = int(sys.argv[1]) if len(sys.argv) > 1 else 400
N = 1024
if os.environ.get("SKG_SMALL_MODE"):
    _capi.set_tuning(small_mode=int(os.environ["SKG_SMALL_MODE"]))
g = torch.Generator().manual_seed(0)
res = {}
for K in (1024, 4096):
    A = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
    b = torch.rand(N, generator=g).cuda(); C = torch.empty(M, N, device="cuda")
    for _ in range(20):
        gemm(A, W, b, C, M, N, K, 1)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        gemm(A, W, b, C, M, N, K, 1)
    e1.record(); torch.cuda.synchronize()
    res[K] = e0.elapsed_time(e1) / 200 * 1e3
print("M=%d N=%d: K=1024 %.1f us, K=4096 %.1f us per launch -> %.3f us per 64-k step" % (M, N, res[1024], res[4096], (res[4096] - res[1024]) / 48))
