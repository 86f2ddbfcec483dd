"""Developer aid (GPU): one image's grid GEMM (M = 800, N = K = 1024, exact fp32) with 1..4 K slices, product + reduce launch,
timed back to back on one stream.  usage: gemm_m800_split.py [M] [epilogue]"""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()
from skghoi_amd import _capi
from skghoi_amd.engine import gemm
M = int(sys.argv[1]) if len(sys.argv) > 1 else 800
epi = int(sys.argv[2]) if len(sys.argv) > 2 else _capi.EPI_BIAS_RELU
N = K = 1024
g = torch.Generator().manual_seed(0)
A = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
b = torch.rand(N, generator=g).cuda(); C = torch.empty(M, N, device="cuda")
ref = None
for S in (1, 2, 3, 4):
    kw = dict(split_k=S, split_ws=torch.empty(S, M, N, device="cuda")) if S > 1 else {}
    for _ in range(200):
        gemm(A, W, b, C, M, N, K, epi, **kw)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(500):
        gemm(A, W, b, C, M, N, K, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 500 * 1e3
    if ref is None:
        ref = C.clone()
    print("M=%d S=%d  %.1f us  %.1f TFLOP/s  max diff vs S=1 %.2e" % (M, S, us, 2.0 * M * N * K / us / 1e6, float((C - ref).abs().max())), flush=True)
