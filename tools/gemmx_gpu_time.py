"""Developer aid (GPU box): GPU-side duration of one skg_gemmx product, free of the host's launch rate -- `reps` launches are
captured into a hipGraph and the replay is timed (a Python loop of ctypes launches tops out near 25-40 us per launch, which
is longer than the batch-4 products themselves).
usage: gemmx_gpu_time.py [bf16|fp32] [kind ...]      prints a K sweep and the aliased-operand variants per kind"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import torch
from skghoi_amd import runtime as _rt; _rt.configure()      # hardware-queue setting, before the first GPU use
from skghoi_amd import gemmx


def gpu_us(ops, bf16, reps=20, rounds=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            gemmx.launch(ops, bf16=bf16)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                gemmx.launch(ops, bf16=bf16)
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(rounds):
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def make(kind, M, N, K, twins=False):
    x = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda() * 0.03; b = torch.randn(N).cuda()
    y = torch.empty(M, N).cuda(); dz = torch.randn(M, N).cuda(); dx = torch.empty(M, K).cuda()
    dW = torch.empty(N, K).cuda(); db = torch.empty(N).cuda()
    if kind == "fwd":
        op, a, bb, c = gemmx.forward(x, W, y, bias=b, relu=True), x, W, y
    elif kind == "dx":
        op, a, bb, c = gemmx.input_grad(dz, W, dx, mask=x), dz, W, dx
    else:
        op, a, bb, c = gemmx.weight_grad(dz, x, dW, db=db), dz, x, dW
    if twins:                                              # bf16 twins of both operands, and the twin of the output
        op.A16, op.B16 = a.to(torch.bfloat16), bb.to(torch.bfloat16)
        op.C16 = torch.empty(c.shape, dtype=torch.bfloat16, device=c.device)
        op.keep += [op.A16, op.B16, op.C16]
    return op, 2.0 * M * N * K


if __name__ == "__main__":
    bf16 = not (len(sys.argv) > 1 and sys.argv[1] == "fp32")
    kinds = [a for a in sys.argv[2:] if not a.startswith("--")] or ["fwd", "dx", "dw"]
    tag = "bf16" if bf16 else "fp32"
    for kind in kinds:
        for M, N, K, S in [(3200, 1024, 128, 1), (3200, 1024, 256, 1), (3200, 1024, 512, 1), (3200, 1024, 1024, 1),
                           (3200, 1024, 2048, 1), (3200, 1024, 1024, 2), (3200, 1024, 1024, 0), (3200, 4096, 1024, 1),
                           (3200, 256, 1024, 0), (25600, 1024, 1024, 1), (102400, 1024, 1024, 1)]:
            op, f = make(kind, M, N, K)
            op.split_k = S
            line = "%s %-3s M=%6d N=%4d K=%4d S=%d:" % (tag, kind, M, N, K, S)
            us = gpu_us([op], bf16)
            line += "  %7.1f us %6.1f TF" % (us, f / us / 1e6)
            if bf16:
                op3, _ = make(kind, M, N, K, twins=True)
                op3.split_k = S
                us = gpu_us([op3], bf16)
                line += "  | bf16 twins %7.1f us %6.1f TF" % (us, f / us / 1e6)
            if kind == "fwd" and "--aliased" in sys.argv:
                for name, za, zb in (("B aliased", False, True), ("A+B aliased", True, True)):
                    op2, _ = make(kind, M, N, K)
                    op2.split_k = S
                    if za:
                        op2.a_sm = 0
                    if zb:
                        op2.b_sn = 0
                    us = gpu_us([op2], bf16)
                    line += "  | %s %7.1f us %6.1f TF" % (name, us, f / us / 1e6)
            print(line, flush=True)
