"""Developer aid (GPU box): is skg_gemmx_bf16 bound by operand traffic?  The same forward product with the rows of B (and of
A) all aliased to row 0 (leading dimension 0: every load hits the same line): if the time drops, the loop is waiting for bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import gemmx


def bench(ops, bf16, n=50):
    for _ in range(10):
        gemmx.launch(ops, bf16=bf16)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gemmx.launch(ops, bf16=bf16)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in [(3200, 1024, 1024), (3200, 4096, 1024), (25600, 1024, 1024), (102400, 1024, 1024)]:
    x = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda() * 0.03; b = torch.randn(N).cuda(); y = torch.empty(M, N).cuda()
    f = 2.0 * M * N * K
    for bf16 in (True, False):
        line = "%s fwd M=%6d N=%4d K=%4d:" % ("bf16" if bf16 else "fp32", M, N, K)
        for name, za, zb in (("normal", False, False), ("B rows aliased", False, True), ("A and B aliased", True, True)):
            op = gemmx.forward(x, W, y, bias=b, relu=True)
            if za:
                op.a_sm = 0
            if zb:
                op.b_sn = 0
            us = bench([op], bf16)
            line += "  %s %7.1f us %6.1f TF" % (name, us, f / us / 1e6)
        print(line, flush=True)
