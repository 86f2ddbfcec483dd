"""Developer aid (GPU box): skg_gemmx_f32 throughput on the training step's shapes (batch 4: 3200 grid rows).
usage: gemmx_microbench.py [rows=3200] [bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import gemmx

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"


def bench(name, mk, flops, n=30):
    ops = mk()
    for _ in range(5):
        gemmx.launch(ops, bf16=BF16)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gemmx.launch(ops, bf16=BF16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%-46s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flops / ms / 1e9))


for M, N, K in [(rows, 1024, 1024), (rows, 1024, 256), (160, 1024, 1024), (176, 1024, 12544), (rows, 118, 2048)]:
    x = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda() * 0.03; b = torch.randn(N).cuda()
    y = torch.empty(M, N).cuda(); dz = torch.randn(M, N).cuda(); dx = torch.empty(M, K).cuda()
    dW = torch.empty(N, K).cuda(); db = torch.empty(N).cuda()
    f = 2.0 * M * N * K
    bench("fwd   M=%d N=%d K=%d" % (M, N, K), lambda: [gemmx.forward(x, W, y, bias=b, relu=True)], f)
    bench("dX    M=%d N=%d K=%d" % (M, N, K), lambda: [gemmx.input_grad(dz, W, dx, mask=x)], f)
    bench("dW+db M=%d N=%d K=%d" % (M, N, K), lambda: [gemmx.weight_grad(dz, x, dW, db=db)], f)
    bench("dX|dW one launch", lambda: [gemmx.input_grad(dz, W, dx, mask=x), gemmx.weight_grad(dz, x, dW, db=db)], 2 * f)
