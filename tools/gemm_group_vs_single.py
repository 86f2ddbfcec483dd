"""Developer aid: four equal big GEMMs as ONE grouped launch vs four single launches (fp16x2 loop), steady state."""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import _capi
from skghoi_amd.engine import gemm, gemm_group, SplitWeights
M, N, K = (int(sys.argv[1]) if len(sys.argv) > 1 else 51200), 1024, 1024
g = torch.Generator().manual_seed(0)
A = (torch.rand(M, K, generator=g) * 2 - 1).cuda()
Ws = [((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda() for _ in range(4)]
b = torch.rand(N, generator=g).cuda(); Cs = [torch.empty(M, N, device="cuda") for _ in range(4)]
def single():
    for W, C in zip(Ws, Cs):
        gemm(A, W, b, C, M, N, K, _capi.EPI_BIAS_RELU)
def grouped():
    gemm_group([((A, W, b, C, M, N, K, _capi.EPI_BIAS_RELU), {}) for W, C in zip(Ws, Cs)])
with SplitWeights():
    for name, fn in (("4 single launches", single), ("1 grouped launch ", grouped), ("4 single launches", single)):
        for _ in range(80):
            fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(80):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 80
        print("%s  %.4f ms  %.1f TFLOP/s" % (name, ms, 4 * 2.0 * M * N * K / ms / 1e9), flush=True)
