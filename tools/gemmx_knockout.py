"""Developer aid (GPU box): the bf16 forward product at M = 3200, N = 1024 over K with the loaded library (use SKG_LIB with
the timing builds of tools/build_gemmx_variants.sh: epilogue removed, plus one piece of the k step removed) -> time per step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from gemmx_gpu_time import gpu_us, make

out = []
for M, N in ((3200, 1024), (102400, 1024)):
    t = {}
    for K in (256, 2048):
        op, _ = make("fwd", M, N, K)
        op.split_k = 1
        t[K] = gpu_us([op], True)
    out.append("M=%d: K=256 %.1f us, K=2048 %.1f us, %.3f us per 32-k step" % (M, t[256], t[2048], (t[2048] - t[256]) / 56))
print(" | ".join(out))
