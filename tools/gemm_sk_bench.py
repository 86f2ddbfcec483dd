"""Developer aid (GPU box): the stream-K launch against the ordinary launch of skg_gemm_f32 at mid-size shapes.
usage: gemm_sk_bench.py [reps=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import _capi
from skghoi_amd import engine as E

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator().manual_seed(0)
shapes = [(800, 1024, 1024), (1600, 1024, 1024), (3200, 1024, 1024), (3120, 1024, 1024), (6400, 1024, 1024),
          (12800, 1024, 1024), (25600, 1024, 1024), (3200, 1024, 256), (3120, 118, 2048), (160, 1024, 12544)]
for M, N, K in shapes:
    A = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
    b = torch.rand(N, generator=g).cuda()
    out = {}
    line = "M=%6d N=%5d K=%6d tiles %4d:" % (M, N, K, ((M + 127) // 128) * ((N + 127) // 128))
    for name, sk in (("std", 0), ("sk256", 256), ("sk512", 512), ("sk768", 768)):
        C = torch.empty(M, N, device="cuda")
        for _ in range(20):
            E.gemm(A, W, b, C, M, N, K, _capi.EPI_BIAS_RELU, stream_k=sk)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            E.gemm(A, W, b, C, M, N, K, _capi.EPI_BIAS_RELU, stream_k=sk)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        out[name] = C
        line += "  %s %7.1f us %6.1f TF" % (name, us, 2.0 * M * N * K / us / 1e6)
    ref = torch.relu(A.double() @ W.double().t() + b.double()).float()
    line += "   max|sk512 - std| %.2e  max|sk512 - f64| %.2e" % ((out["sk512"] - out["std"]).abs().max().item(),
                                                               (out["sk512"] - ref).abs().max().item())
    print(line, flush=True)
