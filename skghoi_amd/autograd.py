"""Differentiable dense layer on the HIP GEMM (training path).

forward : y = act(x W^T + b)                      -> skg_gemm_f32 (EPI_BIAS / EPI_BIAS_RELU)
backward: dz = dy * 1[y > 0]
          dx = dz W      = gemm(A = dz [M,N],   "W" = W^T [K,N])            -> [M,K]
          dW = dz^T x    = gemm(A = dz^T [N,M], "W" = x^T [K,M])            -> [N,K]
          db = sum_rows dz
All three products run on the same fp32-MFMA kernel (both operands k-contiguous), fed by skg_transpose_f32.
Replaces the autograd of every nn.Linear on the interaction-head path (reference: eager torch.nn.functional.linear).
"""
import torch

from . import _capi
from .engine import gemm, pick_split_k, _stream


def _pad4(n):
    return (n + 3) // 4 * 4


def _gemm_sk(A, W, bias, C, M, N, K, epi):
    """skg_gemm_f32 with split-K when the M x N tile grid is small and K long (dW = dZ^T X has K = batch rows)."""
    sk = pick_split_k(M, N, K)
    ws = torch.empty(sk, M, N, device=A.device, dtype=torch.float32) if sk > 1 else None
    gemm(A, W, bias, C, M, N, K, epi, split_k=sk, split_ws=ws)


def transpose(x, rows, cols, ld_out=None):
    """x: [rows, >=cols] row-major (stride(0) = ld) -> [cols, ld_out] with the first `rows` columns filled."""
    ld_out = _pad4(rows) if ld_out is None else ld_out
    out = torch.zeros(cols, ld_out, device=x.device, dtype=torch.float32) if ld_out != rows \
        else torch.empty(cols, ld_out, device=x.device, dtype=torch.float32)
    _capi.check(_capi.lib().skg_transpose_f32(x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), ld_out,
                                              _stream()), "skg_transpose_f32")
    return out


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        if x.device.type != "cuda":
            raise _capi.SkgError("skghoi_amd layers run on a HIP device only")
        x = x.float()
        M, K = x.shape
        N = weight.shape[0]
        Kp = _pad4(K)
        if Kp != K or not x.is_contiguous():
            xp = torch.zeros(M, Kp, device=x.device); xp[:, :K] = x
        else:
            xp = x
        w = weight.float()
        if Kp != K or not w.is_contiguous():
            wp = torch.zeros(N, Kp, device=x.device); wp[:, :K] = w
        else:
            wp = w
        y = torch.empty(M, N, device=x.device)
        if M:
            _gemm_sk(xp, wp, bias.float().contiguous() if bias is not None else None, y, M, N, Kp,
                     _capi.EPI_BIAS_RELU if relu else _capi.EPI_BIAS)
        ctx.relu = relu
        ctx.K = K
        ctx.has_bias = bias is not None
        ctx.save_for_backward(xp, wp, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, wp, y = ctx.saved_tensors
        M, Kp = xp.shape
        N = wp.shape[0]
        K = ctx.K
        dz = dy.float()
        if ctx.relu:
            dz = dz * (y > 0)
        dz = dz.contiguous()
        dx = dw = db = None
        Np = _pad4(N)
        if Np != N:
            dzp = torch.zeros(M, Np, device=dz.device); dzp[:, :N] = dz
        else:
            dzp = dz
        if ctx.needs_input_grad[0]:
            wT = transpose(wp, N, Kp, ld_out=Np)                     # [Kp, Np]
            dxp = torch.empty(M, Kp, device=dz.device)
            if M:
                _gemm_sk(dzp, wT, None, dxp, M, Kp, Np, _capi.EPI_BIAS)
            dx = dxp[:, :K]
        if ctx.needs_input_grad[1]:
            Mp = _pad4(M)
            dzT = transpose(dz, M, N, ld_out=Mp)                     # [N, Mp]
            xT = transpose(xp, M, Kp, ld_out=Mp)                     # [Kp, Mp]
            dwp = torch.empty(N, Kp, device=dz.device)
            if M:
                _gemm_sk(dzT, xT, None, dwp, N, Kp, Mp, _capi.EPI_BIAS)
            else:
                dwp.zero_()
            dw = dwp[:, :K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dz.sum(dim=0)
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    return LinearFn.apply(x, weight, bias, relu)


# ------------------------------------------------------------------------------------------------ bf16 configuration
def _pad_to(n, m):
    return (n + m - 1) // m * m


def gemm_bf16(A, W, bias, M, N, K, relu=False, out_dtype=torch.bfloat16):
    """C = act(A W^T + bias) on skg_gemm_bf16 (A [M,>=K] bf16, W [N,>=K] bf16, K % 64 == 0)."""
    import ctypes as C
    out = torch.empty(M, N, device=A.device, dtype=out_dtype)
    if M == 0:
        return out
    d = _capi.GemmBf16Desc()
    d.A = A.data_ptr(); d.lda = A.stride(0); d.W = W.data_ptr(); d.ldw = W.stride(0)
    d.bias = bias.data_ptr() if bias is not None else 0
    d.C = out.data_ptr(); d.ldc = out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.relu = int(relu); d.out_bf16 = int(out_dtype == torch.bfloat16)
    blocks = ((M + 127) // 128) * ((N + 127) // 128)
    sk = 1
    if blocks < 512 and K >= 1024:
        sk = int(max(1, min(-(-512 // blocks), K // 256, 64)))
    ws = torch.empty(sk, M, N, device=A.device, dtype=torch.float32) if sk > 1 else None
    d.split_k = sk; d.split_ws = ws.data_ptr() if ws is not None else 0
    _capi.check(_capi.lib().skg_gemm_bf16(C.byref(d), _stream()), "skg_gemm_bf16[%dx%dx%d]" % (M, N, K))
    return out


def _bf16_padded(x, cols_to):
    """[rows, cols] any float dtype -> contiguous bf16 [rows, cols_to] (zero padded)."""
    r, c = x.shape
    if c == cols_to and x.dtype == torch.bfloat16 and x.is_contiguous():
        return x
    out = torch.zeros(r, cols_to, device=x.device, dtype=torch.bfloat16) if c != cols_to else \
        torch.empty(r, cols_to, device=x.device, dtype=torch.bfloat16)
    out[:, :c] = x
    return out


def transpose_bf16(x, rows, cols, ld_out):
    out = torch.zeros(cols, ld_out, device=x.device, dtype=torch.bfloat16)
    _capi.check(_capi.lib().skg_transpose_bf16(x.data_ptr(), x.stride(0), rows, cols, out.data_ptr(), ld_out,
                                               _stream()), "skg_transpose_bf16")
    return out


class LinearBf16Fn(torch.autograd.Function):
    """Autocast-style layer: operands rounded to bf16, fp32 accumulation, bf16 activations out; parameters and their
    gradients stay fp32 (dW is accumulated in fp32 by the kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, out_dtype=torch.float32):
        M, K = x.shape
        N = weight.shape[0]
        Kp = _pad_to(K, 64)
        xb = _bf16_padded(x, Kp); wb = _bf16_padded(weight.detach(), Kp)
        y = gemm_bf16(xb, wb, bias.detach().float().contiguous() if bias is not None else None, M, N, Kp, relu,
                      out_dtype=out_dtype)
        ctx.relu, ctx.K, ctx.has_bias = relu, K, bias is not None
        ctx.save_for_backward(xb, wb, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wb, y = ctx.saved_tensors
        M, Kp = xb.shape
        N = wb.shape[0]
        K = ctx.K
        dz = dy
        if ctx.relu:
            dz = dz * (y > 0)
        dx = dw = db = None
        Np = _pad_to(N, 64)
        dzb = _bf16_padded(dz, Np)
        if ctx.needs_input_grad[0]:
            wT = transpose_bf16(wb, N, Kp, Np)                           # [Kp, Np]
            dx = gemm_bf16(dzb, wT, None, M, Kp, Np, out_dtype=torch.float32)[:, :K]
        if ctx.needs_input_grad[1]:
            Mp = _pad_to(M, 64)
            dzT = transpose_bf16(dzb, M, N, Mp)                          # [N, Mp]
            xT = transpose_bf16(xb, M, Kp, Mp)                           # [Kp, Mp]
            dw = gemm_bf16(dzT, xT, None, N, Kp, Mp, out_dtype=torch.float32)[:, :K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dz.float().sum(dim=0)
        return dx, dw, db, None, None


def linear_bf16(x, weight, bias=None, relu=False, out_dtype=torch.float32):
    """bf16 operands on the matrix cores, fp32 accumulation; activations are handed on in `out_dtype` (fp32 by default:
    the element-wise glue of the head stays in fp32)."""
    return LinearBf16Fn.apply(x, weight, bias, relu, out_dtype)
