"""Developer aid (GPU box): one skg_gemmx product in a loop, for rocprofv3 --pmc.
usage: gemmx_one.py M N K kind(fwd|dx|dw) split iters [bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import gemmx

M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind, S, iters = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
BF16 = len(sys.argv) > 7 and sys.argv[7] == "bf16"
x = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda() * 0.03; b = torch.randn(N).cuda()
y = torch.empty(M, N).cuda(); dz = torch.randn(M, N).cuda(); dx = torch.empty(M, K).cuda()
dW = torch.empty(N, K).cuda(); db = torch.empty(N).cuda()
op = {"fwd": lambda: gemmx.forward(x, W, y, bias=b, relu=True), "dx": lambda: gemmx.input_grad(dz, W, dx, mask=x),
      "dw": lambda: gemmx.weight_grad(dz, x, dW, db=db)}[kind]()
op.split_k = S
for _ in range(iters):
    gemmx.launch([op], bf16=BF16)
torch.cuda.synchronize()
