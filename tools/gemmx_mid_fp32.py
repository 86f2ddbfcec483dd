"""Developer aid (GPU box): skg_gemmx_f32 on the mid-size eval shapes (B = 1 .. 4 images: M = 400 .. 1600 grid rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from gemmx_gpu_time import gpu_us, make
for M in (400, 800, 1600, 3200):
    for N, K in ((1024, 1024),):
        line = "fp32 fwd M=%5d N=%d K=%d:" % (M, N, K)
        for S in (1, 2, 4, 8, 0):
            op, f = make("fwd", M, N, K)
            op.split_k = S
            us = gpu_us([op], False)
            line += "  S=%d %6.1f us %5.1f TF" % (S, us, f / us / 1e6)
        print(line, flush=True)
