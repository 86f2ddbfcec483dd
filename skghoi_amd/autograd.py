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
