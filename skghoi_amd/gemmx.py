"""Python face of skg_gemmx_f32 / skg_gemmx_bf16 (include/skghoi.h): the free-layout MFMA GEMM of the training step.

    C(m, n) (+)= mask(relu(sum_k A(m, k) B(k, n) + bias[n]))

Operands are described by `Op` records (pointer + strides + optional power-of-two blocking), built from torch tensors by
the helpers below; `launch([...])` enqueues up to GEMMX_GROUP_MAX independent products in one launch (dX and dW of a
layer, the node-row GEMMs of a graph) and picks split-K factors for long contractions with small outputs.
Replaces the autograd of every nn.Linear / MultiBranchFusion on the path (heads/adamixer_transH_spatial_r50_head.py:
469-474, 509-527, 635-701) without the operand transposes a k-contiguous-only kernel needs.
"""
import ctypes as C

import torch

from . import _capi
from .engine import _stream



class Op:
    """One product.  Set fields directly or through the constructors below."""
    __slots__ = ("A", "a_sm", "a_sk", "B", "b_sn", "b_sk", "b_kshift", "b_kstride", "b_nshift", "b_nstride", "C", "ldc",
                 "c_nshift", "c_nstride", "M", "N", "K", "bias", "relu", "mask", "ldmask", "accumulate", "a_rowsum",
                 "split_k", "keep", "A16", "B16", "C16", "a16_ld", "b16_ld")

    def __init__(self, **kw):
        self.b_kshift = self.b_nshift = self.c_nshift = 0
        self.b_kstride = self.b_nstride = self.c_nstride = 0
        self.bias = self.mask = self.a_rowsum = None
        self.ldmask = 0
        self.relu = self.accumulate = False
        self.split_k = 0                       # 0: chosen by launch()
        self.A16 = self.B16 = self.C16 = None  # bf16 twins (torch.bfloat16 tensors laid out like A / B / C), optional
        self.a16_ld = self.b16_ld = 0          # pitch of a PADDED twin (0: the fp32 array's)
        self.keep = []
        for k, v in kw.items():
            setattr(self, k, v)


def _p(t):
    return 0 if t is None else t.data_ptr()


def forward(x, W, out, bias=None, relu=False, M=None, K=None, N=None, w_blocks=None):
    """out[M, N] = act(x[M, K] W[N, K]^T + bias).  w_blocks = (log2 block, block stride): W stored branch-major,
    [K / block][N][block] (the 16 fc_3 weights of an MBF, block 64)."""
    M = x.shape[0] if M is None else M
    K = x.shape[1] if K is None else K
    N = out.shape[1] if N is None else N
    op = Op(A=x.data_ptr(), a_sm=x.stride(0), a_sk=1, B=W.data_ptr(), b_sk=1, C=out.data_ptr(), ldc=out.stride(0), M=M, N=N,
            K=K, bias=bias, relu=relu)
    if w_blocks is None:
        op.b_sn = W.stride(0)
    else:
        op.b_kshift, op.b_kstride = w_blocks
        op.b_sn = 1 << op.b_kshift
    op.keep = [x, W, out, bias]
    return op


def input_grad(dz, W, dx, mask=None, accumulate=False, M=None, N_in=None, K_out=None, w_blocks=None):
    """dx[M, N_in] (+)= dz[M, K_out] W[K_out, N_in], zeroed where mask <= 0 (after the accumulation).  w_blocks as in
    forward()."""
    M = dz.shape[0] if M is None else M
    n_out = dz.shape[1] if K_out is None else K_out
    k_in = dx.shape[1] if N_in is None else N_in
    op = Op(A=dz.data_ptr(), a_sm=dz.stride(0), a_sk=1, B=W.data_ptr(), b_sn=1, C=dx.data_ptr(), ldc=dx.stride(0), M=M,
            N=k_in, K=n_out, mask=mask, ldmask=(mask.stride(0) if mask is not None else 0), accumulate=accumulate)
    if w_blocks is None:
        op.b_sk = W.stride(0)                  # B(k = out index, n = in index) = W[k][n]
    else:
        op.b_nshift, op.b_nstride = w_blocks   # W[(n >> s)][k][n & mask]: blocked along the in index
        op.b_sk = 1 << op.b_nshift
    op.keep = [dz, W, dx, mask]
    return op


def weight_grad(dz, x, dW, db=None, accumulate=False, rows=None, n_out=None, k_in=None, w_blocks=None):
    """dW[N_out, K_in] (+)= dz[rows, N_out]^T x[rows, K_in];  db[N_out] (+)= column sums of dz.  w_blocks: dW stored
    branch-major like the weight."""
    rows = dz.shape[0] if rows is None else rows
    n_out = dz.shape[1] if n_out is None else n_out
    k_in = x.shape[1] if k_in is None else k_in
    op = Op(A=dz.data_ptr(), a_sm=1, a_sk=dz.stride(0), B=x.data_ptr(), b_sn=1, b_sk=x.stride(0), C=dW.data_ptr(), M=n_out,
            N=k_in, K=rows, a_rowsum=db, accumulate=accumulate)
    if w_blocks is None:
        op.ldc = dW.stride(0)
    else:
        op.c_nshift, op.c_nstride = w_blocks
        op.ldc = 1 << op.c_nshift
    op.keep = [dz, x, dW, db]
    return op


def pick_split(op, blocks_so_far=0, target=None, bk=16):
    """Split-K factor of one product.  Measured on MI355X (tools/gemmx_split_sweep.py, training shapes at 3200 grid
    rows, re-swept on the whole training step after the staged epilogue: skg_train_plan.hip split_target): the exact fp32
    loop is best at ~500 workgroups, the bf16 loop at ~256; slices keep >= 128 k; products with more than ~1.5x fewer
    tiles than the target are not split."""
    tiles = ((op.M + 127) // 128) * ((op.N + 127) // 128)
    kt = (op.K + bk - 1) // bk
    if target is None:
        target = 500 if bk == 16 else 256
    cap = min(kt * bk // 128, 64)
    if tiles == 0 or cap < 2:
        return 1
    return int(max(1, min(int(target / tiles + 0.5), cap)))


# INLAUNCH = True: split products are reduced INSIDE their launch (skg_gemmx_desc.split_ctr: tile counters, zero when made,
# left zero by every launch) instead of by a second launch that adds the slices.  Same additions in the same order either way
# (tests/test_gemmx_gpu.py runs every case both ways); measured slower on MI355X at the training shapes (train_fused.py,
# INLAUNCH_SPLIT_REDUCE), hence off by default.
INLAUNCH = False
_COUNTERS = {}
N_COUNTERS = 1 << 14


def counters(device):
    """The tile counters of `device`'s launches through this module (one array: the launches of a step share one stream)."""
    t = _COUNTERS.get(device)
    if t is None:
        t = _COUNTERS[device] = torch.zeros(N_COUNTERS, device=device, dtype=torch.int32)
    return t


def path_counts(reset=False):
    """Launches since the last reset as (exact fp32, bf16 register-staged, bf16 direct-to-LDS) -- statistics for tests."""
    out = (C.c_int64 * 3)()
    _capi.lib().skg_gemmx_path_counts(out, int(reset))
    return tuple(int(v) for v in out)


def launch(ops, bf16=False, inlaunch=None):
    """Enqueues the products (one launch; split products are reduced by their tiles' last workgroups, or -- inlaunch=False,
    or more tiles than counters -- by one more launch).  bf16: operands rounded to bf16 on the way to the matrix core
    (skg_gemmx_bf16), everything in memory stays fp32."""
    lib = _capi.lib()
    inlaunch = INLAUNCH if inlaunch is None else inlaunch
    fn, name, bk = (lib.skg_gemmx_bf16, "skg_gemmx_bf16", 32) if bf16 else (lib.skg_gemmx_f32, "skg_gemmx_f32", 16)
    ops = [o for o in ops if o.M > 0 and o.N > 0]
    for i0 in range(0, len(ops), _capi.GEMMX_GROUP_MAX):
        chunk = ops[i0:i0 + _capi.GEMMX_GROUP_MAX]
        arr = (_capi.GemmXDesc * len(chunk))()
        keep = []
        used = 0
        for d, o in zip(arr, chunk):
            d.A, d.a_sm, d.a_sk = o.A, o.a_sm, o.a_sk
            d.B, d.b_sn, d.b_sk = o.B, o.b_sn, o.b_sk
            d.b_kshift, d.b_nshift, d.b_kstride, d.b_nstride = o.b_kshift, o.b_nshift, o.b_kstride, o.b_nstride
            d.C, d.ldc, d.c_nshift, d.c_nstride = o.C, o.ldc, o.c_nshift, o.c_nstride
            d.accumulate = int(bool(o.accumulate))
            d.M, d.N, d.K, d.relu = o.M, o.N, o.K, int(bool(o.relu))
            d.bias = _p(o.bias); d.mask = _p(o.mask); d.ldmask = o.ldmask; d.a_rowsum = _p(o.a_rowsum)
            d.A16, d.B16, d.C16 = _p(o.A16), _p(o.B16), _p(o.C16)
            d.a16_ld, d.b16_ld = o.a16_ld, o.b16_ld
            sk = o.split_k or pick_split(o, bk=bk)
            d.split_k = sk if sk > 1 else 0
            if sk > 1:
                dev = o.keep[0].device if o.keep else torch.device("cuda")
                ws = torch.empty(sk * (o.M * o.N + o.M), device=dev, dtype=torch.float32)
                keep.append(ws)
                d.split_ws = ws.data_ptr()
                tiles = ((o.M + 127) // 128) * ((o.N + 127) // 128)
                if inlaunch and used + tiles <= N_COUNTERS:
                    d.split_ctr = counters(dev).data_ptr() + 4 * used
                    used += tiles
        _capi.check(fn(arr, len(chunk), _stream()), "%s[%d]" % (name, len(chunk)))
