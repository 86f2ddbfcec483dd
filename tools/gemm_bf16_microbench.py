"""Developer aid (GPU box): skg_gemm_bf16 throughput on the hot shapes."""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd.autograd import gemm_bf16
g = torch.Generator().manual_seed(0)
for M, N, K in [(51200, 1024, 1024), (102400, 1024, 1024), (10240, 1024, 12544), (4096, 4096, 4096), (8192, 8192, 8192)]:
    A = (torch.rand(M, K, generator=g) * 2 - 1).cuda().bfloat16(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda().bfloat16()
    b = torch.rand(N, generator=g).cuda()
    for _ in range(100):                 # the shader clock needs ~50 ms of load to settle
        gemm_bf16(A, W, b, M, N, K, True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    R = 100
    for _ in range(R):
        gemm_bf16(A, W, b, M, N, K, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / R
    print("bf16 M=%6d N=%5d K=%6d  %8.4f ms  %7.1f TFLOP/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
