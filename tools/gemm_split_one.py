"""Developer aid: time ONE shape of the split-operand GEMM loop.  usage: gemm_split_one.py [M N K epi reps]"""
import sys
sys.path.insert(0, "."); sys.dont_write_bytecode = True
import os
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd.engine import gemm, SplitWeights
a = [int(x) for x in sys.argv[1:6]]
M, N, K, epi, reps = a + [51200, 1024, 1024, 1, 300][len(a):]
g = torch.Generator().manual_seed(0)
pad = int(os.environ.get("XPAD", "0"))            # extra floats per A row (leading dimension K + pad): L2 channel experiment
A = torch.empty(M, K + pad, device="cuda"); A[:, :K] = (torch.rand(M, K, generator=g) * 2 - 1).cuda(); A = A[:, :K]; W = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).cuda()
b = torch.rand(N, generator=g).cuda(); C = torch.empty(M, N, device="cuda")
kw = {}
if os.environ.get("XA"):        # every A row gathers row 0: A always cache-hot
    kw["a_rows"] = torch.zeros(M, dtype=torch.int32, device="cuda")
import contextlib
ctx = contextlib.nullcontext if os.environ.get("EXACT") else SplitWeights      # EXACT=1: the exact fp32-MFMA loop
with ctx():
    for _ in range(max(3, reps)):            # the shader clock needs ~50 ms of load to settle
        gemm(A, W, b, C, M, N, K, epi, **kw)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        gemm(A, W, b, C, M, N, K, epi, **kw)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("M=%d N=%d K=%d epi=%d  %.4f ms  %.1f TFLOP/s" % (M, N, K, epi, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
if os.environ.get("XT"):            # SKG_XTIME build: per-phase cycle counts (wave w: frag reads, MFMA+loads, split+store, barrier)
    dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
    with SplitWeights():
        gemm(A, W, b, C, M, N, K, epi, split_ws=dbg, **kw)
    torch.cuda.synchronize()
    t = dbg.cpu().view(8, 8)
    for w in range(4):
        n = max(int(t[w, 4]), 1)
        print("wave %d: tiles %d  cycles/tile: reads %.0f  mfma+loads %.0f  split+store %.0f  barrier %.0f  total %.0f" % (
            w, n, t[w, 0] / n, t[w, 1] / n, t[w, 2] / n, t[w, 3] / n, float(t[w, :4].sum()) / n))
