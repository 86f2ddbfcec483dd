"""Developer aid (GPU box): where a workgroup of skg_gemmx_t16_kernel spends its time -- needs the -DSKG_XPROBE_STAMPS build
(tools/build_gemmx_variants.sh stamps="-DSKG_XPROBE_STAMPS"; SKG_LIB=build/variants/lib_stamps.so).  Wall-clock stamps (100 MHz) at
entry, after the k loop, after the epilogue's stores have been acknowledged."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.dont_write_bytecode = True
import numpy as np
import torch
from skghoi_amd import runtime as _rt; _rt.configure()
from skghoi_amd import gemmx

M, N, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (3200, 1024, 1024)
x = torch.randn(M, K).cuda(); W = torch.randn(N, K).cuda() * 0.03; b = torch.randn(N).cuda(); y = torch.empty(M, N).cuda()
op = gemmx.forward(x, W, y, bias=b, relu=True)
op.A16, op.B16 = x.to(torch.bfloat16), W.to(torch.bfloat16)
op.C16 = torch.empty(M, N, dtype=torch.bfloat16).cuda()
op.split_k = 1
nb = ((M + 127) // 128) * ((N + 127) // 128)
stamps = torch.zeros(nb, 4, dtype=torch.int64).cuda()
from skghoi_amd import _capi
import ctypes as C
lib = _capi.lib()


def launch():
    arr = (_capi.GemmXDesc * 1)()
    d = arr[0]
    d.A, d.a_sm, d.a_sk = op.A, op.a_sm, op.a_sk
    d.B, d.b_sn, d.b_sk = op.B, op.b_sn, op.b_sk
    d.C, d.ldc, d.M, d.N, d.K, d.relu = op.C, op.ldc, M, N, K, 1
    d.bias = b.data_ptr()
    d.A16, d.B16, d.C16 = op.A16.data_ptr(), op.B16.data_ptr(), op.C16.data_ptr()
    d.split_k = 0
    d.split_ws = stamps.data_ptr()
    _capi.check(lib.skg_gemmx_bf16(arr, 1, torch.cuda.current_stream().cuda_stream), "gemmx")


for _ in range(5):
    launch()
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
t0 = s[:, 0].min()
ent, lp, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - s[:, 0]) / 100.0, (s[:, 2] - s[:, 1]) / 100.0
print("M=%d N=%d K=%d: %d workgroups" % (M, N, K, nb))
print("entry after the first workgroup (us): median %.2f  max %.2f" % (np.median(ent), ent.max()))
print("prologue + k loop (us):               median %.2f  min %.2f  max %.2f" % (np.median(lp), lp.min(), lp.max()))
print("epilogue until stores acked (us):     median %.2f  min %.2f  max %.2f" % (np.median(end), end.min(), end.max()))
print("last workgroup done after the first entry (us): %.2f" % ((s[:, 2].max() - t0) / 100.0))
cus = len(set(s[:, 3].astype(np.int64).tolist()))
print("distinct hardware ids (XCC/SE/CU words): %d" % cus)
