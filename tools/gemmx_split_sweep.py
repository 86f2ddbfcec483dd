"""Developer aid (GPU box): split-K sweep of skg_gemmx on the training step's shapes.  usage: gemmx_split_sweep.py [bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import gemmx

BF16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"


def t(ops, n=30):
    for _ in range(5):
        gemmx.launch(ops, bf16=BF16)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gemmx.launch(ops, bf16=BF16)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K in [(3200, 1024, 1024), (3200, 1024, 4096), (3200, 1024, 2048), (3200, 1024, 256), (160, 1024, 1024), (3120, 118, 2048)]:
    x = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda() * 0.03; b = torch.randn(N).cuda()
    y = torch.empty(M, N).cuda(); dz = torch.randn(M, N).cuda(); dx = torch.empty(M, K).cuda()
    dW = torch.empty(N, K).cuda(); db = torch.empty(N).cuda()
    for name, mk in (("fwd", lambda: gemmx.forward(x, W, y, bias=b, relu=True)),
                     ("dX ", lambda: gemmx.input_grad(dz, W, dx, mask=x)),
                     ("dW ", lambda: gemmx.weight_grad(dz, x, dW, db=db))):
        row = []
        for S in (0, 1, 2, 3, 4, 5, 6, 8, 12, 16):
            op = mk(); op.split_k = S
            auto = gemmx.pick_split(op, bk=32 if BF16 else 16) if S == 0 else S
            row.append("%s%d:%5.1f" % ("auto=" if S == 0 else "S", auto, t([op])))
        print("%s M=%5d N=%5d K=%5d  us  %s" % (name, op.M, op.N, op.K, "  ".join(row)), flush=True)
