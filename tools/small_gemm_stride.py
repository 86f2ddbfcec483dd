"""Developer aid (GPU box): does the row stride of W (a power of two: 4096 B at K = 1024) hurt the weight-streaming
small-M GEMMs?  Times the grouped 64x64-tile launch of four M=40 GEMMs with ldw = K and ldw = K + pad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import _capi, engine

M, N, K = 40, 1024, 1024
for pad in (0, 16, 32, 64, 272):
    Ws = [torch.randn(N, K + pad).cuda() for _ in range(4)]
    A = torch.randn(M, K).cuda()
    outs = [torch.empty(M, N).cuda() for _ in range(4)]
    specs = [((A, W, None, o, M, N, K, _capi.EPI_BIAS), dict(ldw=K + pad)) for W, o in zip(Ws, outs)]
    for _ in range(10):
        engine.gemm_group(specs)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            engine.gemm_group(specs)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    ref = A @ Ws[0][:, :K].t()
    err = (outs[0] - ref).abs().max().item()
    print("ldw = K + %3d: %7.2f us per grouped launch (4 GEMMs %dx%dx%d)  err %.1e" % (pad, e0.elapsed_time(e1) * 1e3 / 20, M, N, K, err))
# single big-K case: box_head layer 1 at one image
M, N, K = 40, 1024, 12544
for pad in (0, 16, 64):
    W = torch.randn(N, K + pad).cuda(); A = torch.randn(M, K).cuda(); out = torch.empty(M, N).cuda()
    sk = engine.pick_split_k(M, N, K)
    ws = torch.empty(sk, M, N).cuda()
    f = lambda: engine.gemm(A, W, None, out, M, N, K, _capi.EPI_BIAS, ldw=K + pad, split_k=sk, split_ws=ws)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print("bh1 ldw = K + %3d (split %d): %7.2f us" % (pad, sk, e0.elapsed_time(e1) * 1e3 / 20))
