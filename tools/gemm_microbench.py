"""Developer aid (GPU box): times skg_gemm_f32 on the hot shapes of the 20x20 workload.
usage: python tools/gemm_microbench.py [reps] [split]   (SKG_LIB=<path to .so> selects a kernel build; "split" times
the fp16x2 split-operand loop and prints its max deviation from the exact fp32 loop)"""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import _capi
from skghoi_amd.engine import gemm, SplitWeights, _NullCtx

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
split = len(sys.argv) > 2 and sys.argv[2] == "split"
ctx = SplitWeights() if split else _NullCtx()
shapes = [(51200, 1024, 1024, 1), (51200, 1024, 1024, 2), (51200, 1024, 1024, 3), (10240, 1024, 12544, 1),
          (2560, 1024, 1024, 0), (1280, 1024, 1024, 0), (51200, 1024, 256, 1), (199680, 118, 2048, 0)]
g = torch.Generator().manual_seed(0)
for M, N, K, epi in shapes:
    A = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
    b = torch.rand(N, generator=g).cuda(); C = torch.empty(M, N, device="cuda")
    kw = {}
    if epi == 2:
        P = torch.rand(1280, N, generator=g).cuda(); Q = torch.rand(2560, N, generator=g).cuda()
        pi = (torch.arange(M) // 40 % 1280).int().cuda(); qi = (torch.arange(M) % 2560).int().cuda()
        kw = dict(P=P, p_idx=pi, ldp=N, Q=Q, q_idx=qi, ldq=N, mbias=b, C_raw=torch.empty(M, N, device="cuda"), ldc_raw=N)
    if epi == 3:
        kw = dict(dot_w=b, dot_partial=torch.empty(64, M, device="cuda"))
    dev = ""
    if split and epi != 3:
        gemm(A, W, b, C, M, N, K, epi, **kw)
        C0 = C.clone()
    ctx.__enter__()
    for _ in range(3):
        gemm(A, W, b, None if epi == 3 else C, M, N, K, epi, **kw)
    if split and epi != 3:
        dev = "  max|split - fp32| %.2e (max|C| %.2f)" % ((C - C0).abs().max().item(), C0.abs().max().item())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        gemm(A, W, b, None if epi == 3 else C, M, N, K, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    ctx.__exit__()
    ms = e0.elapsed_time(e1) / reps
    print("M=%7d N=%5d K=%6d epi=%d  %8.4f ms  %6.1f TFLOP/s" % (M, N, K, epi, ms, 2.0 * M * N * K / ms / 1e9) + dev, flush=True)
