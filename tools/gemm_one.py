"""Developer aid: run ONE GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K epi reps"""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd.engine import gemm
M, N, K, epi, reps = [int(x) for x in sys.argv[1:6]]
g = torch.Generator().manual_seed(0)
A = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
b = torch.rand(N, generator=g).cuda(); C = torch.empty(M, N, device="cuda")
for _ in range(reps):
    gemm(A, W, b, C, M, N, K, epi)
torch.cuda.synchronize()
