// skg_train.hip -- the non-GEMM kernels of the fused TRAINING step of the interaction head (forward pieces that must keep
// what the backward needs, and the hand-written backward of every element-wise / graph stage).
//
// Reference (all heads/adamixer_transH_spatial_r50_head.py unless noted): message passing and normalisation HEAD:892-925,
// read-out HEAD:966-973, fc_head / fc_tail inputs HEAD:884-885, focal losses HEAD:153-205 with ops.py:159-211.  The
// reference leaves the backward to autograd: ~1000 small element-wise kernels per step.  Here each stage is one kernel
// forward and one or two backward; reductions over graph neighbourhoods are done per destination row (no atomics, fixed
// order, deterministic), exactly like the forward aggregation in skg_graph.hip.  All row widths are the head's 1024.
#include "skg_common.h"

#define TR_COLS 1024

// ------------------------------------------------------------------------------------------------ row dot
// out[r] = sum_c X[r, c] * w[c]   (the adjacency Linear(1024 -> 1) without its bias, HEAD:897).  One wavefront per row.
__global__ __launch_bounds__(256) void skg_rowdot_kernel(const float* __restrict__ X, int64_t ld,
                                                         const float* __restrict__ w, int rows, int cols,
                                                         float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* x = X + (int64_t)r * ld;
    float s = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + c);
        const float4 b = *reinterpret_cast<const float4*>(w + c);
        s += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    s = skg_wave_sum(s);
    if (lane == 0) out[r] = s;
}

extern "C" int skg_rowdot_f32(const float* X, int64_t ld, const float* w, int rows, int cols, float* out, void* stream) {
    if (rows < 0 || cols <= 0 || (cols & 3)) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!X || !w || !out) return SKG_E_ARG;
    if ((ld & 3) || !skg_aligned16(X) || !skg_aligned16(w)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_rowdot_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, X, ld, w, rows, cols, out);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ add + LayerNorm
// x = a + b (node + message, HEAD:912-914 / 923-925), y = LayerNorm(x); keeps x and (mean, rstd) for the backward.
__device__ __forceinline__ void skg_add_layernorm_row(const float* __restrict__ a, int64_t lda,
                                                      const float* __restrict__ b, int64_t ldb,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, float* __restrict__ xsum, float* __restrict__ y,
                                                      float* __restrict__ stats, int r, uint16_t* __restrict__ y16) {
    __shared__ float sred[4];
    const int c = threadIdx.x * 4;
    const float4 va = *reinterpret_cast<const float4*>(a + (int64_t)r * lda + c);
    const float4 vb = *reinterpret_cast<const float4*>(b + (int64_t)r * ldb + c);
    const float4 v = make_float4(va.x + vb.x, va.y + vb.y, va.z + vb.z, va.w + vb.w);
    const float mean = skg_block_sum256((v.x + v.y) + (v.z + v.w), sred) / (float)TR_COLS;
    const float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    const float var = skg_block_sum256((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w), sred) / (float)TR_COLS;
    const float rstd = 1.f / sqrtf(var + eps);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 bb = *reinterpret_cast<const float4*>(beta + c);
    *reinterpret_cast<float4*>(xsum + (int64_t)r * TR_COLS + c) = v;
    const float4 yo = make_float4(d.x * rstd * g.x + bb.x, d.y * rstd * g.y + bb.y, d.z * rstd * g.z + bb.z, d.w * rstd * g.w + bb.w);
    *reinterpret_cast<float4*>(y + (int64_t)r * TR_COLS + c) = yo;
    if (y16) skg_store_twin4(y16 + (int64_t)r * TR_COLS + c, yo);
    if (threadIdx.x == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
}

__global__ __launch_bounds__(256) void skg_add_layernorm_kernel(const float* __restrict__ a, int64_t lda,
                                                                const float* __restrict__ b, int64_t ldb,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps,
                                                                float* __restrict__ xsum, float* __restrict__ y,
                                                                float* __restrict__ stats, uint16_t* __restrict__ y16) {
    skg_add_layernorm_row(a, lda, b, ldb, gamma, beta, eps, xsum, y, stats, (int)blockIdx.x, y16);
}

struct skg_add_layernorm_pack { skg_add_layernorm_args a[SKG_MULTI_MAX]; };
__global__ __launch_bounds__(256) void skg_add_layernorm_multi_kernel(const skg_add_layernorm_pack pk, float eps) {
    const skg_add_layernorm_args& a = pk.a[blockIdx.y];
    if ((int)blockIdx.x >= a.rows) return;                 // (uniform per workgroup)
    skg_add_layernorm_row(a.a, a.lda, a.b, a.ldb, a.gamma, a.beta, eps, a.xsum, a.y, a.stats, (int)blockIdx.x, a.y16);
}

int skg_add_layernorm_multi(const skg_add_layernorm_args* calls, int n, float eps, void* stream) {
    if (!calls || n < 1 || n > SKG_MULTI_MAX) return SKG_E_ARG;
    skg_add_layernorm_pack pk;
    int m = 0, rows = 0;
    for (int i = 0; i < n; ++i) {
        const skg_add_layernorm_args& a = calls[i];
        if (a.rows < 0) return SKG_E_ARG;
        if (a.rows == 0) continue;
        if (!a.a || !a.b || !a.gamma || !a.beta || !a.xsum || !a.y || !a.stats) return SKG_E_ARG;
        if ((a.lda & 3) || (a.ldb & 3) || !skg_aligned16(a.a) || !skg_aligned16(a.b) || !skg_aligned16(a.gamma) ||
            !skg_aligned16(a.beta) || !skg_aligned16(a.xsum) || !skg_aligned16(a.y))
            return SKG_E_ALIGN;
        pk.a[m] = a;
        pk.a[m++].y16 = skg_twin(a.y);
        rows = a.rows > rows ? a.rows : rows;
    }
    if (m == 0) return 0;
    hipLaunchKernelGGL(skg_add_layernorm_multi_kernel, dim3(rows, m), dim3(256), 0, (hipStream_t)stream, pk, eps);
    return skg_launch_status();
}

extern "C" int skg_add_layernorm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma,
                                     const float* beta, int rows, float eps, float* xsum, float* y, float* stats,
                                     void* stream) {
    if (rows < 0) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!a || !b || !gamma || !beta || !xsum || !y || !stats) return SKG_E_ARG;
    if ((lda & 3) || (ldb & 3) || !skg_aligned16(a) || !skg_aligned16(b) || !skg_aligned16(gamma) ||
        !skg_aligned16(beta) || !skg_aligned16(xsum) || !skg_aligned16(y))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_add_layernorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, gamma,
                       beta, eps, xsum, y, stats, skg_twin(y));
    return skg_launch_status();
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd.
__device__ __forceinline__ void skg_layernorm_bwd_row(const float* __restrict__ dy, int64_t lddy,
                                                      const float* __restrict__ x, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, float* __restrict__ dx,
                                                      const float* __restrict__ relu_src, float* __restrict__ dx_masked,
                                                      int r, uint16_t* __restrict__ dx16, uint16_t* __restrict__ dxm16) {
    __shared__ float sred[4];
    const int c = threadIdx.x * 4;
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    const float4 vx = *reinterpret_cast<const float4*>(x + (int64_t)r * TR_COLS + c);
    const float4 vd = *reinterpret_cast<const float4*>(dy + (int64_t)r * lddy + c);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + c);
    const float4 xh = make_float4((vx.x - mean) * rstd, (vx.y - mean) * rstd, (vx.z - mean) * rstd, (vx.w - mean) * rstd);
    const float4 g = make_float4(vd.x * gm.x, vd.y * gm.y, vd.z * gm.z, vd.w * gm.w);
    const float c1 = skg_block_sum256((g.x + g.y) + (g.z + g.w), sred) / (float)TR_COLS;
    const float c2 = skg_block_sum256((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w), sred) / (float)TR_COLS;
    const float4 o = make_float4(rstd * (g.x - c1 - xh.x * c2), rstd * (g.y - c1 - xh.y * c2),
                                 rstd * (g.z - c1 - xh.z * c2), rstd * (g.w - c1 - xh.w * c2));
    *reinterpret_cast<float4*>(dx + (int64_t)r * TR_COLS + c) = o;
    if (dx16) skg_store_twin4(dx16 + (int64_t)r * TR_COLS + c, o);
    if (dx_masked) {
        const float4 m = *reinterpret_cast<const float4*>(relu_src + (int64_t)r * TR_COLS + c);
        const float4 om = make_float4(m.x > 0.f ? o.x : 0.f, m.y > 0.f ? o.y : 0.f, m.z > 0.f ? o.z : 0.f, m.w > 0.f ? o.w : 0.f);
        *reinterpret_cast<float4*>(dx_masked + (int64_t)r * TR_COLS + c) = om;
        if (dxm16) skg_store_twin4(dxm16 + (int64_t)r * TR_COLS + c, om);
    }
}

__global__ __launch_bounds__(256) void skg_layernorm_bwd_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ gamma,
                                                                float* __restrict__ dx,
                                                                const float* __restrict__ relu_src,
                                                                float* __restrict__ dx_masked,
                                                                uint16_t* __restrict__ dx16, uint16_t* __restrict__ dxm16) {
    skg_layernorm_bwd_row(dy, lddy, x, stats, gamma, dx, relu_src, dx_masked, (int)blockIdx.x, dx16, dxm16);
}

// dgamma[c] = sum_r dy[r, c] * xhat[r, c],  dbeta[c] = sum_r dy[r, c]: one thread per column and row group, rows in order.
// 64 columns x 16 row groups per workgroup (16 workgroups of 1024 threads): rows r = g, g + 16, ... per group, groups added
// in a fixed tree.  (Four row groups walked ~80 node rows in 20 dependent round trips: 14 us for 0.7 MB; sixteen: 5.)
#define TR_RG 16
__device__ __forceinline__ void skg_layernorm_param_grad_cols(const float* __restrict__ dy, int64_t lddy,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ stats, int rows,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              int blk) {
    __shared__ float sg_s[TR_RG][64], sb_s[TR_RG][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blk * 64 + cl;
    float sg = 0.f, sb = 0.f;
    for (int r = g; r < rows; r += TR_RG) {
        const float d = dy[(int64_t)r * lddy + c];
        sg += d * ((x[(int64_t)r * TR_COLS + c] - stats[2 * r]) * stats[2 * r + 1]);
        sb += d;
    }
    sg_s[g][cl] = sg; sb_s[g][cl] = sb;
    __syncthreads();
    if (g == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < TR_RG; q += 4) {
            a += (sg_s[q][cl] + sg_s[q + 1][cl]) + (sg_s[q + 2][cl] + sg_s[q + 3][cl]);
            b += (sb_s[q][cl] + sb_s[q + 1][cl]) + (sb_s[q + 2][cl] + sb_s[q + 3][cl]);
        }
        dgamma[c] = a;
        dbeta[c] = b;
    }
}

__global__ __launch_bounds__(64 * TR_RG) void skg_layernorm_param_grad_kernel(const float* __restrict__ dy, int64_t lddy,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ stats, int rows,
                                                                       float* __restrict__ dgamma,
                                                                       float* __restrict__ dbeta) {
    skg_layernorm_param_grad_cols(dy, lddy, x, stats, rows, dgamma, dbeta, (int)blockIdx.x);
}

// Both LayerNorms of the step in two launches instead of four: (a) every row's input gradient, (b) the parameter gradients
struct skg_layernorm_bwd_pack { skg_layernorm_bwd_args a[SKG_MULTI_MAX]; };
__global__ __launch_bounds__(256) void skg_layernorm_bwd_multi_kernel(const skg_layernorm_bwd_pack pk) {
    const skg_layernorm_bwd_args& a = pk.a[blockIdx.y];
    if ((int)blockIdx.x >= a.rows) return;                 // (uniform per workgroup)
    skg_layernorm_bwd_row(a.dy, a.lddy, a.x, a.stats, a.gamma, a.dx, a.relu_src, a.dx_masked, (int)blockIdx.x, a.dx16,
                          a.dx_masked16);
}
__global__ __launch_bounds__(64 * TR_RG) void skg_layernorm_param_grad_multi_kernel(const skg_layernorm_bwd_pack pk) {
    const skg_layernorm_bwd_args& a = pk.a[blockIdx.y];
    skg_layernorm_param_grad_cols(a.dy, a.lddy, a.x, a.stats, a.rows, a.dgamma, a.dbeta, (int)blockIdx.x);
}

int skg_layernorm_bwd_multi(const skg_layernorm_bwd_args* calls, int n, void* stream) {
    if (!calls || n < 1 || n > SKG_MULTI_MAX) return SKG_E_ARG;
    skg_layernorm_bwd_pack pk;
    int rows = 0;
    for (int i = 0; i < n; ++i) {
        const skg_layernorm_bwd_args& a = calls[i];
        if (a.rows < 0) return SKG_E_ARG;
        if (!a.dy || !a.x || !a.stats || !a.gamma || !a.dx || !a.dgamma || !a.dbeta || (a.dx_masked && !a.relu_src)) return SKG_E_ARG;
        if ((a.lddy & 3) || !skg_aligned16(a.dy) || !skg_aligned16(a.x) || !skg_aligned16(a.gamma) || !skg_aligned16(a.dx) ||
            !skg_aligned16(a.relu_src) || !skg_aligned16(a.dx_masked))
            return SKG_E_ALIGN;
        pk.a[i] = a;
        pk.a[i].dx16 = skg_twin(a.dx);
        pk.a[i].dx_masked16 = a.dx_masked ? skg_twin(a.dx_masked) : nullptr;
        rows = a.rows > rows ? a.rows : rows;
    }
    if (rows)
        hipLaunchKernelGGL(skg_layernorm_bwd_multi_kernel, dim3(rows, n), dim3(256), 0, (hipStream_t)stream, pk);
    hipLaunchKernelGGL(skg_layernorm_param_grad_multi_kernel, dim3(TR_COLS / 64, n), dim3(64 * TR_RG), 0, (hipStream_t)stream, pk);
    return skg_launch_status();
}

extern "C" int skg_layernorm_bwd_f32(const float* dy, int64_t lddy, const float* x, const float* stats,
                                     const float* gamma, int rows, float* dx, const float* relu_src, float* dx_masked,
                                     float* dgamma, float* dbeta, void* stream) {
    if (rows < 0) return SKG_E_ARG;
    if (!dy || !x || !stats || !gamma || !dx || !dgamma || !dbeta || (dx_masked && !relu_src)) return SKG_E_ARG;
    if ((lddy & 3) || !skg_aligned16(dy) || !skg_aligned16(x) || !skg_aligned16(gamma) || !skg_aligned16(dx) ||
        !skg_aligned16(relu_src) || !skg_aligned16(dx_masked))
        return SKG_E_ALIGN;
    if (rows)
        hipLaunchKernelGGL(skg_layernorm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, stats,
                           gamma, dx, relu_src, dx_masked, skg_twin(dx), dx_masked ? skg_twin(dx_masked) : (uint16_t*)nullptr);
    hipLaunchKernelGGL(skg_layernorm_param_grad_kernel, dim3(TR_COLS / 64), dim3(64 * TR_RG), 0, (hipStream_t)stream, dy, lddy,
                       x, stats, rows, dgamma, dbeta);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ fc_1 * fc_2 backward
// Forward (MBF / MessageMBF, HEAD:469-474, 509-527):  t = relu(m * f),  m = P[pi] + Q[qi] + mbias,  f = F[fi].
// Given g = dt (already zeroed where t <= 0):   dF[fi] (+)= g * m,   dm = g * f  (written over g, in place).
__global__ __launch_bounds__(256) void skg_mul_bwd_kernel(float* __restrict__ g, int64_t ldg,
                                                          const float* __restrict__ F, const int32_t* __restrict__ f_idx,
                                                          int64_t ldf, const float* __restrict__ P,
                                                          const int32_t* __restrict__ p_idx, int64_t ldp,
                                                          const float* __restrict__ Q, const int32_t* __restrict__ q_idx,
                                                          int64_t ldq, const float* __restrict__ mbias,
                                                          float* __restrict__ dF, int64_t lddf, int accumulate,
                                                          uint16_t* __restrict__ dF16) {
    const int r = blockIdx.x;
    const int c = threadIdx.x * 4;
    const int fi = f_idx ? f_idx[r] : r;
    float4 m = *reinterpret_cast<const float4*>(P + (int64_t)(p_idx ? p_idx[r] : r) * ldp + c);
    if (Q) {
        const float4 t = *reinterpret_cast<const float4*>(Q + (int64_t)(q_idx ? q_idx[r] : r) * ldq + c);
        m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
    }
    if (mbias) {
        const float4 t = *reinterpret_cast<const float4*>(mbias + c);
        m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
    }
    float4* gp = reinterpret_cast<float4*>(g + (int64_t)r * ldg + c);
    const float4 gv = *gp;
    const float4 fv = *reinterpret_cast<const float4*>(F + (int64_t)fi * ldf + c);
    float4* dp = reinterpret_cast<float4*>(dF + (int64_t)fi * lddf + c);
    float4 o = make_float4(gv.x * m.x, gv.y * m.y, gv.z * m.z, gv.w * m.w);
    if (accumulate) { const float4 t = *dp; o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
    *dp = o;
    if (dF16) skg_store_twin4(dF16 + (int64_t)fi * lddf + c, o);
    *gp = make_float4(gv.x * fv.x, gv.y * fv.y, gv.z * fv.z, gv.w * fv.w);
}

struct skg_mul_bwd_pack { skg_mul_bwd_args a[SKG_MULTI_MAX]; };
__global__ __launch_bounds__(256) void skg_mul_bwd_multi_kernel(const skg_mul_bwd_pack pk) {
    const skg_mul_bwd_args& a = pk.a[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= a.rows) return;
    const int c = threadIdx.x * 4;
    const int fi = a.f_idx ? a.f_idx[r] : r;
    float4 m = *reinterpret_cast<const float4*>(a.P + (int64_t)(a.p_idx ? a.p_idx[r] : r) * a.ldp + c);
    if (a.Q) {
        const float4 t = *reinterpret_cast<const float4*>(a.Q + (int64_t)(a.q_idx ? a.q_idx[r] : r) * a.ldq + c);
        m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
    }
    if (a.mbias) {
        const float4 t = *reinterpret_cast<const float4*>(a.mbias + c);
        m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
    }
    float4* gp = reinterpret_cast<float4*>(a.g + (int64_t)r * a.ldg + c);
    const float4 gv = *gp;
    const float4 fv = *reinterpret_cast<const float4*>(a.F + (int64_t)fi * a.ldf + c);
    float4* dp = reinterpret_cast<float4*>(a.dF + (int64_t)fi * a.lddf + c);
    float4 o = make_float4(gv.x * m.x, gv.y * m.y, gv.z * m.z, gv.w * m.w);
    if (a.accumulate) { const float4 t = *dp; o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
    *dp = o;
    if (a.dF16) skg_store_twin4(a.dF16 + (int64_t)fi * a.lddf + c, o);
    *gp = make_float4(gv.x * fv.x, gv.y * fv.y, gv.z * fv.z, gv.w * fv.w);
}

int skg_mul_bwd_multi(const skg_mul_bwd_args* calls, int n, void* stream) {
    if (!calls || n < 1 || n > SKG_MULTI_MAX) return SKG_E_ARG;
    skg_mul_bwd_pack pk;
    int m = 0, rows = 0;
    for (int i = 0; i < n; ++i) {
        const skg_mul_bwd_args& a = calls[i];
        if (a.rows < 0) return SKG_E_ARG;
        if (a.rows == 0) continue;
        if (!a.g || !a.F || !a.P || !a.dF) return SKG_E_ARG;
        if ((a.ldg & 3) || (a.ldf & 3) || (a.ldp & 3) || (a.lddf & 3) || (a.Q && (a.ldq & 3))) return SKG_E_ALIGN;
        if (!skg_aligned16(a.g) || !skg_aligned16(a.F) || !skg_aligned16(a.P) || !skg_aligned16(a.dF) || !skg_aligned16(a.Q) ||
            !skg_aligned16(a.mbias))
            return SKG_E_ALIGN;
        pk.a[m] = a;
        pk.a[m++].dF16 = skg_twin(a.dF);
        rows = a.rows > rows ? a.rows : rows;
    }
    if (m == 0) return 0;
    hipLaunchKernelGGL(skg_mul_bwd_multi_kernel, dim3(rows, m), dim3(256), 0, (hipStream_t)stream, pk);
    return skg_launch_status();
}

extern "C" int skg_mul_bwd_f32(float* g, int64_t ldg, const float* F, const int32_t* f_idx, int64_t ldf, const float* P,
                               const int32_t* p_idx, int64_t ldp, const float* Q, const int32_t* q_idx, int64_t ldq,
                               const float* mbias, int rows, float* dF, int64_t lddf, int accumulate, void* stream) {
    if (rows < 0) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!g || !F || !P || !dF) return SKG_E_ARG;
    if ((ldg & 3) || (ldf & 3) || (ldp & 3) || (lddf & 3) || (Q && (ldq & 3))) return SKG_E_ALIGN;
    if (!skg_aligned16(g) || !skg_aligned16(F) || !skg_aligned16(P) || !skg_aligned16(dF) || (Q && !skg_aligned16(Q)) ||
        (mbias && !skg_aligned16(mbias)))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_mul_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, ldg, F, f_idx, ldf, P, p_idx,
                       ldp, Q, q_idx, ldq, mbias, dF, lddf, accumulate, skg_twin(dF));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ neighbourhood sums
// Sums of the rows of `src` that belong to one human / one node / one image: the gradients of the gathered fc_1 tables.
//   mode 0  src = grid rows (r = grid_off + i * n + j):   outH[(a, i)] = sum_j src[r],   outN[(a, j)] = sum_i src[r]
//   mode 1  src = kept pairs (p = pair_off + i * (n - 1) + jj):  outH[(a, i)] = sum_jj src[p];
//                                                          outN[(a, j)] = sum_{i != j} src[pair of (i, j)]
//   mode 2  src = kept pairs:  outH[meta[a].image] = sum of all pairs of active image a   (outN unused)
// One workgroup per destination row, rows added in index order.
__device__ __forceinline__ void skg_segment_sum_body(const float* __restrict__ src, int64_t ld,
                                                     const skg_image_meta* __restrict__ meta,
                                                     const int32_t* __restrict__ hum_img,
                                                     const int32_t* __restrict__ node_img, int n_dst_h, int mode,
                                                     float* __restrict__ outH, float* __restrict__ outN, int accumulate,
                                                     int blk, uint16_t* __restrict__ outH16, uint16_t* __restrict__ outN16) {
    const bool to_h = blk < n_dst_h;
    const int dst = to_h ? blk : blk - n_dst_h;
    float* out = to_h ? outH : outN;
    uint16_t* out16 = to_h ? outH16 : outN16;
    if (!out) return;
    const int a = mode == 2 ? dst : (to_h ? hum_img[dst] : node_img[dst]);
    const skg_image_meta mt = meta[a];
    const int c = threadIdx.x * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add = [&](int64_t row) {
        const float4 v = *reinterpret_cast<const float4*>(src + row * ld + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    };
    if (mode == 0) {
        if (to_h) { const int i = dst - mt.hum_off; for (int j = 0; j < mt.n; ++j) add((int64_t)mt.grid_off + (int64_t)i * mt.n + j); }
        else { const int j = dst - mt.node_off; for (int i = 0; i < mt.n_h; ++i) add((int64_t)mt.grid_off + (int64_t)i * mt.n + j); }
    } else if (mode == 1) {
        if (to_h) { const int i = dst - mt.hum_off; for (int jj = 0; jj < mt.n - 1; ++jj) add((int64_t)mt.pair_off + (int64_t)i * (mt.n - 1) + jj); }
        else {
            const int j = dst - mt.node_off;
            for (int i = 0; i < mt.n_h; ++i) if (i != j) add((int64_t)mt.pair_off + (int64_t)i * (mt.n - 1) + (j < i ? j : j - 1));
        }
    } else {
        // all kept pairs of the image: hundreds of rows through one workgroup -- four independent chains keep four loads in
        // flight per lane (a single chain walks 780 rows of a 20 x 20 image at one L2 round trip each: 190 us, measured)
        const int P = mt.n_h * (mt.n - 1);
        float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1, a3 = a1;
        const float* base = src + (int64_t)mt.pair_off * ld + c;
        int p = 0;
        for (; p + 3 < P; p += 4) {
            const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)p * ld);
            const float4 v1 = *reinterpret_cast<const float4*>(base + (int64_t)(p + 1) * ld);
            const float4 v2 = *reinterpret_cast<const float4*>(base + (int64_t)(p + 2) * ld);
            const float4 v3 = *reinterpret_cast<const float4*>(base + (int64_t)(p + 3) * ld);
            acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
            a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
            a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
            a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; p < P; ++p) add((int64_t)mt.pair_off + p);
        acc.x = (acc.x + a1.x) + (a2.x + a3.x); acc.y = (acc.y + a1.y) + (a2.y + a3.y);
        acc.z = (acc.z + a1.z) + (a2.z + a3.z); acc.w = (acc.w + a1.w) + (a2.w + a3.w);
    }
    float4* o = reinterpret_cast<float4*>(out + (int64_t)(mode == 2 ? mt.image : dst) * TR_COLS + c);
    if (accumulate) { const float4 t = *o; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
    *o = acc;
    if (out16) skg_store_twin4(out16 + (int64_t)(mode == 2 ? mt.image : dst) * TR_COLS + c, acc);
}

__global__ __launch_bounds__(256) void skg_segment_sum_kernel(const float* __restrict__ src, int64_t ld,
                                                              const skg_image_meta* __restrict__ meta,
                                                              const int32_t* __restrict__ hum_img,
                                                              const int32_t* __restrict__ node_img, int n_dst_h, int mode,
                                                              float* __restrict__ outH, float* __restrict__ outN,
                                                              int accumulate, uint16_t* __restrict__ outH16,
                                                              uint16_t* __restrict__ outN16) {
    skg_segment_sum_body(src, ld, meta, hum_img, node_img, n_dst_h, mode, outH, outN, accumulate, (int)blockIdx.x, outH16, outN16);
}

struct skg_segment_sum_pack { skg_segment_sum_args a[SKG_MULTI_MAX]; };
__global__ __launch_bounds__(256) void skg_segment_sum_multi_kernel(const skg_segment_sum_pack pk,
                                                                    const skg_image_meta* __restrict__ meta,
                                                                    const int32_t* __restrict__ hum_img,
                                                                    const int32_t* __restrict__ node_img, int n_dst_h) {
    const skg_segment_sum_args& a = pk.a[blockIdx.y];
    skg_segment_sum_body(a.src, a.ld, meta, hum_img, node_img, n_dst_h, a.mode, a.outH, a.outN, a.accumulate, (int)blockIdx.x,
                         a.outH16, a.outN16);
}

int skg_segment_sum_multi(const skg_segment_sum_args* calls, int n, const skg_image_meta* meta, const int32_t* hum_img,
                          const int32_t* node_img, int sum_h, int sum_n, void* stream) {
    if (!calls || n < 1 || n > SKG_MULTI_MAX || sum_h < 0 || sum_n < 0) return SKG_E_ARG;
    if (sum_h + sum_n == 0) return 0;
    if (!meta || !hum_img || !node_img) return SKG_E_ARG;
    skg_segment_sum_pack pk;
    for (int i = 0; i < n; ++i) {
        const skg_segment_sum_args& a = calls[i];
        if (a.mode < 0 || a.mode > 1 || !a.src || (!a.outH && !a.outN)) return SKG_E_ARG;
        if ((a.ld & 3) || !skg_aligned16(a.src) || !skg_aligned16(a.outH) || !skg_aligned16(a.outN)) return SKG_E_ALIGN;
        pk.a[i] = a;
        pk.a[i].outH16 = a.outH ? skg_twin(a.outH) : nullptr;
        pk.a[i].outN16 = a.outN ? skg_twin(a.outN) : nullptr;
    }
    hipLaunchKernelGGL(skg_segment_sum_multi_kernel, dim3(sum_h + sum_n, n), dim3(256), 0, (hipStream_t)stream, pk, meta,
                       hum_img, node_img, sum_h);
    return skg_launch_status();
}

// mode 2 on its own grid: (active image, block of 64 columns).  One workgroup per image pulls hundreds of 4-KB rows through
// ONE CU's L2 port (66 us for a 20 x 20 image, measured: ~70 GB/s per CU); here 16 workgroups share an image, each thread
// owns one float4 column quad of its 64-column block and every 16th row, and the 16 row groups are added in fixed order.
__global__ __launch_bounds__(256) void skg_segment_sum_image_kernel(const float* __restrict__ src, int64_t ld,
                                                                    const skg_image_meta* __restrict__ meta,
                                                                    float* __restrict__ out, int accumulate,
                                                                    uint16_t* __restrict__ out16) {
    __shared__ float4 red[16][16];
    const skg_image_meta mt = meta[blockIdx.x];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.y * 64 + 4 * q;
    const int P = mt.n_h * (mt.n - 1);
    const float* base = src + (int64_t)mt.pair_off * ld + c;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int p = g;
    for (; p + 16 < P; p += 32) {
        const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)p * ld);
        const float4 v1 = *reinterpret_cast<const float4*>(base + (int64_t)(p + 16) * ld);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    if (p < P) {
        const float4 v0 = *reinterpret_cast<const float4*>(base + (int64_t)p * ld);
        a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    red[g][q] = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    __syncthreads();
    if (g == 0) {
        float4 s = red[0][q];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 t = red[k][q]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        float4* o = reinterpret_cast<float4*>(out + (int64_t)mt.image * TR_COLS + c);
        if (accumulate) { const float4 t = *o; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
        *o = s;
        if (out16) skg_store_twin4(out16 + (int64_t)mt.image * TR_COLS + c, s);
    }
}

extern "C" int skg_segment_sum_f32(const float* src, int64_t ld, const skg_image_meta* meta, int n_active,
                                   const int32_t* hum_img, const int32_t* node_img, int sum_h, int sum_n, int mode,
                                   float* outH, float* outN, int accumulate, void* stream) {
    if (n_active < 0 || sum_h < 0 || sum_n < 0 || mode < 0 || mode > 2) return SKG_E_ARG;
    const int nh = mode == 2 ? n_active : sum_h, nn = mode == 2 ? 0 : sum_n;
    if (nh + nn == 0) return 0;
    if (!src || !meta || (mode != 2 && (!hum_img || !node_img)) || (!outH && !outN)) return SKG_E_ARG;
    if ((ld & 3) || !skg_aligned16(src) || !skg_aligned16(outH) || !skg_aligned16(outN)) return SKG_E_ALIGN;
    if (mode == 2) {
        if (!outH) return SKG_E_ARG;
        hipLaunchKernelGGL(skg_segment_sum_image_kernel, dim3(n_active, TR_COLS / 64), dim3(256), 0, (hipStream_t)stream, src,
                           ld, meta, outH, accumulate, skg_twin(outH));
        return skg_launch_status();
    }
    hipLaunchKernelGGL(skg_segment_sum_kernel, dim3(nh + nn), dim3(256), 0, (hipStream_t)stream, src, ld, meta, hum_img,
                       node_img, nh, mode, outH, outN, accumulate, outH ? skg_twin(outH) : (uint16_t*)nullptr,
                       outN ? skg_twin(outN) : (uint16_t*)nullptr);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ aggregation backward
// Forward (skg_graph_aggregate): U[h] = sum_j alpha[(i,j)] Tos[(i,j)],  V[o] = sum_i beta[(i,j)] Tso[(i,j)].
// Stage 1, one wavefront per grid row r = (i, j):
//   dTos[r] = alpha[r] * dU[h] where Tos[r] > 0 else 0       (Tos = relu(...): the mask of the fc_1 * fc_2 stage)
//   dTso[r] = beta[r]  * dV[o] where Tso[r] > 0 else 0
//   da[r]   = dU[h] . Tos[r],   db[r] = dV[o] . Tso[r]       (gradients of the softmax weights)
__global__ __launch_bounds__(256) void skg_aggregate_bwd_rows_kernel(
    const float* __restrict__ dU, const float* __restrict__ dV, const float* __restrict__ Tos,
    const float* __restrict__ Tso, const float* __restrict__ alpha, const float* __restrict__ beta,
    const int32_t* __restrict__ grid_h, const int32_t* __restrict__ grid_o, int rows, float* __restrict__ dTos,
    float* __restrict__ dTso, float* __restrict__ da, float* __restrict__ db) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float al = alpha[r], be = beta[r];
    const float* u = dU + (int64_t)grid_h[r] * TR_COLS;
    const float* v = dV + (int64_t)grid_o[r] * TR_COLS;
    const float* ts = Tos + (int64_t)r * TR_COLS;
    const float* tv = Tso + (int64_t)r * TR_COLS;
    float sa = 0.f, sb = 0.f;
    for (int c = lane * 4; c < TR_COLS; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(u + c), t = *reinterpret_cast<const float4*>(ts + c);
        const float4 b = *reinterpret_cast<const float4*>(v + c), s = *reinterpret_cast<const float4*>(tv + c);
        sa += (a.x * t.x + a.y * t.y) + (a.z * t.z + a.w * t.w);
        sb += (b.x * s.x + b.y * s.y) + (b.z * s.z + b.w * s.w);
        *reinterpret_cast<float4*>(dTos + (int64_t)r * TR_COLS + c) =
            make_float4(t.x > 0.f ? al * a.x : 0.f, t.y > 0.f ? al * a.y : 0.f, t.z > 0.f ? al * a.z : 0.f, t.w > 0.f ? al * a.w : 0.f);
        *reinterpret_cast<float4*>(dTso + (int64_t)r * TR_COLS + c) =
            make_float4(s.x > 0.f ? be * b.x : 0.f, s.y > 0.f ? be * b.y : 0.f, s.z > 0.f ? be * b.z : 0.f, s.w > 0.f ? be * b.w : 0.f);
    }
    sa = skg_wave_sum(sa); sb = skg_wave_sum(sb);
    if (lane == 0) { da[r] = sa; db[r] = sb; }
}

// Stage 2, softmax backward per destination (one workgroup each, like the forward): for human (a, i) over its n senders
//   dadj_h[r] = alpha[r] * (da[r] - sum_j alpha[(i,j)] da[(i,j)]),  for node (a, j) over its n_h senders likewise with
// beta / db into dadj_n.  The adjacency logit feeds both softmaxes: its gradient is dadj_h + dadj_n.
__global__ __launch_bounds__(64) void skg_aggregate_bwd_softmax_kernel(
    const skg_image_meta* __restrict__ meta, const int32_t* __restrict__ hum_img, const int32_t* __restrict__ node_img,
    int sum_h, const float* __restrict__ alpha, const float* __restrict__ beta, const float* __restrict__ da,
    const float* __restrict__ db, float* __restrict__ dadj_h, float* __restrict__ dadj_n) {
    const bool to_h = (int)blockIdx.x < sum_h;
    const int dst = to_h ? blockIdx.x : blockIdx.x - sum_h;
    const skg_image_meta mt = meta[to_h ? hum_img[dst] : node_img[dst]];
    const int local = to_h ? dst - mt.hum_off : dst - mt.node_off;
    const int cnt = to_h ? mt.n : mt.n_h;
    const int64_t row0 = to_h ? (int64_t)mt.grid_off + (int64_t)local * mt.n : (int64_t)mt.grid_off + local;
    const int64_t step = to_h ? 1 : mt.n;
    const float* w = to_h ? alpha : beta;
    const float* d = to_h ? da : db;
    float* out = to_h ? dadj_h : dadj_n;
    float s = 0.f;
    for (int t = threadIdx.x; t < cnt; t += 64) s += w[row0 + t * step] * d[row0 + t * step];
    s = skg_wave_sum(s);
    for (int t = threadIdx.x; t < cnt; t += 64) {
        const int64_t r = row0 + t * step;
        out[r] = w[r] * (d[r] - s);
    }
}

extern "C" int skg_aggregate_bwd_f32(const float* dU, const float* dV, const float* Tos, const float* Tso,
                                     const float* alpha, const float* beta, const int32_t* grid_h, const int32_t* grid_o,
                                     int sum_g, const skg_image_meta* meta, const int32_t* hum_img,
                                     const int32_t* node_img, int sum_h, int sum_n, float* dTos, float* dTso,
                                     float* da, float* db, float* dadj_h, float* dadj_n, void* stream) {
    if (sum_g < 0 || sum_h < 0 || sum_n < 0) return SKG_E_ARG;
    if (sum_g == 0) return 0;
    if (!dU || !dV || !Tos || !Tso || !alpha || !beta || !grid_h || !grid_o || !meta || !hum_img || !node_img || !dTos ||
        !dTso || !da || !db || !dadj_h || !dadj_n)
        return SKG_E_ARG;
    if (!skg_aligned16(dU) || !skg_aligned16(dV) || !skg_aligned16(Tos) || !skg_aligned16(Tso) || !skg_aligned16(dTos) ||
        !skg_aligned16(dTso))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_aggregate_bwd_rows_kernel, dim3((sum_g + 3) / 4), dim3(256), 0, (hipStream_t)stream, dU, dV, Tos,
                       Tso, alpha, beta, grid_h, grid_o, sum_g, dTos, dTso, da, db);
    hipLaunchKernelGGL(skg_aggregate_bwd_softmax_kernel, dim3(sum_h + sum_n), dim3(64), 0, (hipStream_t)stream, meta, hum_img,
                       node_img, sum_h, alpha, beta, da, db, dadj_h, dadj_n);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ adjacency backward
// adj[r] = Wt[r] . w + b with Wt = relu(...) (HEAD:896-897):  dadj[r] = dadj_h[r] + dadj_n[r],
// dWt[r, c] = dadj[r] * w[c] where Wt[r, c] > 0.  (dw and db come from the GEMM  dadj^T Wt  on the matrix core.)
__global__ __launch_bounds__(256) void skg_adjacency_bwd_kernel(const float* __restrict__ dadj_h,
                                                                const float* __restrict__ dadj_n,
                                                                const float* __restrict__ w,
                                                                const float* __restrict__ Wt, float* __restrict__ dadj,
                                                                float* __restrict__ dWt, uint16_t* __restrict__ dWt16) {
    const int r = blockIdx.x;
    const int c = threadIdx.x * 4;
    const float d = dadj_h[r] + dadj_n[r];
    if (threadIdx.x == 0) dadj[r] = d;
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    const float4 t = *reinterpret_cast<const float4*>(Wt + (int64_t)r * TR_COLS + c);
    const float4 o = make_float4(t.x > 0.f ? d * wv.x : 0.f, t.y > 0.f ? d * wv.y : 0.f, t.z > 0.f ? d * wv.z : 0.f,
                                 t.w > 0.f ? d * wv.w : 0.f);
    *reinterpret_cast<float4*>(dWt + (int64_t)r * TR_COLS + c) = o;
    if (dWt16) skg_store_twin4(dWt16 + (int64_t)r * TR_COLS + c, o);
}

extern "C" int skg_adjacency_bwd_f32(const float* dadj_h, const float* dadj_n, const float* w, const float* Wt, int rows,
                                     float* dadj, float* dWt, void* stream) {
    if (rows < 0) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!dadj_h || !dadj_n || !w || !Wt || !dadj || !dWt) return SKG_E_ARG;
    if (!skg_aligned16(w) || !skg_aligned16(Wt) || !skg_aligned16(dWt)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_adjacency_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, dadj_h, dadj_n, w, Wt, dadj,
                       dWt, skg_twin(dWt));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ fc_head / fc_tail inputs
// Backward of skg_concat_entity: d_enc[e] = (dX[hum_of[e]] + dX[sum_h + node_of[e]])[:1024], zeroed where enc[e] <= 0 (the
// ReLU of box_head's second layer, HEAD:639).  hum_of / node_of: the human / node row that reads encoding row e, or -1.
__global__ __launch_bounds__(256) void skg_entity_rows_bwd_kernel(const float* __restrict__ dX, int64_t ldx,
                                                                  const int32_t* __restrict__ hum_of,
                                                                  const int32_t* __restrict__ node_of, int sum_h,
                                                                  const float* __restrict__ enc,
                                                                  float* __restrict__ d_enc, uint16_t* __restrict__ d_enc16) {
    const int e = blockIdx.x;
    const int c = threadIdx.x * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int h = hum_of[e], o = node_of[e];
    if (h >= 0) { const float4 v = *reinterpret_cast<const float4*>(dX + (int64_t)h * ldx + c); acc = v; }
    if (o >= 0) {
        const float4 v = *reinterpret_cast<const float4*>(dX + (int64_t)(sum_h + o) * ldx + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 m = *reinterpret_cast<const float4*>(enc + (int64_t)e * TR_COLS + c);
    const float4 r4 = make_float4(m.x > 0.f ? acc.x : 0.f, m.y > 0.f ? acc.y : 0.f, m.z > 0.f ? acc.z : 0.f, m.w > 0.f ? acc.w : 0.f);
    *reinterpret_cast<float4*>(d_enc + (int64_t)e * TR_COLS + c) = r4;
    if (d_enc16) skg_store_twin4(d_enc16 + (int64_t)e * TR_COLS + c, r4);
}

extern "C" int skg_entity_rows_bwd_f32(const float* dX, int64_t ldx, const int32_t* hum_of, const int32_t* node_of,
                                       int sum_h, int n_enc, const float* enc, float* d_enc, void* stream) {
    if (n_enc < 0 || sum_h < 0) return SKG_E_ARG;
    if (n_enc == 0) return 0;
    if (!dX || !hum_of || !node_of || !enc || !d_enc) return SKG_E_ARG;
    if ((ldx & 3) || !skg_aligned16(dX) || !skg_aligned16(enc) || !skg_aligned16(d_enc)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_entity_rows_bwd_kernel, dim3(n_enc), dim3(256), 0, (hipStream_t)stream, dX, ldx, hum_of, node_of,
                       sum_h, enc, d_enc, skg_twin(d_enc));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ focal losses
// compute_interaction_classification_loss + compute_interactiveness_loss (HEAD:153-205) with binary_focal_loss
// (ops.py:159-211: |1 - y - alpha| * (|y - x| + eps)^gamma * BCE(x, y), alpha 0.5, eps 1e-6; gamma 0.2 on the scored
// cells, 2.0 on the pair weights), forward AND the gradient w.r.t. the logits in one pass over what postprocess emitted:
//   cell c of image a: pair p = pair_off + index[c], verb v = pred[c]; x = scores[c] = sigmoid(lp) * ph * po * w.detach()
//   pair p: w = sigmoid(ls), y = unary[p] = min(sum_v labels[p, v], 1)
// BCE and its derivative as torch defines them (log clamped at -100; (x - y) / max(x (1 - x), 1e-12)).
// SKG_LOSS_CHUNKS workgroups per image (strided over its cells and pairs); partial[a][chunk] = {sum of cell losses, sum
// of pair losses, #positive cells, #positive pairs}: the caller adds them up; the two counts are the loss normalisers
// n_p (HEAD:162-165, 190-192).
// dlogits [sumP, ldl] must be zero-filled; columns < K receive d(sum cell loss)/dlp, column K d(sum pair loss)/dls.
__device__ __forceinline__ float skg_focal(float x, float y, float gamma, float& dldx) {
    const float alpha = 0.5f, eps = 1e-6f;
    const float c = fabsf(1.f - y - alpha);
    const float u = fabsf(y - x) + eps;
    const float lx = fmaxf(logf(x), -100.f), l1x = fmaxf(logf(1.f - x), -100.f);
    const float bce = -(y * lx + (1.f - y) * l1x);
    const float pw = powf(u, gamma);
    const float sgn = (x > y) ? 1.f : ((x < y) ? -1.f : 0.f);
    const float dbce = (x - y) / fmaxf((1.f - x) * x, 1e-12f);
    dldx = c * (gamma * powf(u, gamma - 1.f) * sgn * bce + pw * dbce);
    return c * pw * bce;
}

__global__ __launch_bounds__(256) void skg_hoi_loss_kernel(
    const float* __restrict__ logits, int64_t ldl, int K, const skg_image_meta* __restrict__ meta, int n_active,
    int64_t cells_total, const int64_t* __restrict__ index, const int64_t* __restrict__ pred,
    const float* __restrict__ scores, const float* __restrict__ labels, float* __restrict__ cell_labels,
    float* __restrict__ unary, float* __restrict__ partial, float* __restrict__ dlogits) {
    __shared__ float sred[4];
    const int a = blockIdx.x;
    const skg_image_meta mt = meta[a];
    const int64_t c0 = mt.out_off;
    const int64_t c1 = (a + 1 < n_active) ? (int64_t)meta[a + 1].out_off : cells_total;
    const int P = mt.n_h * (mt.n - 1);
    const int stride = 256 * gridDim.y, first = blockIdx.y * 256 + threadIdx.x;
    float s1 = 0.f, s2 = 0.f, n1 = 0.f, n2 = 0.f;
    for (int64_t c = c0 + first; c < c1; c += stride) {
        const int64_t p = (int64_t)mt.pair_off + index[c];
        const int v = (int)pred[c];
        const float y = labels[p * K + v];
        const float x = scores[c];
        cell_labels[c] = y;
        if (y != 0.f) n1 += 1.f;
        float dldx;
        s1 += skg_focal(x, y, 0.2f, dldx);
        const float sg = 1.f / (1.f + expf(-logits[p * ldl + v]));
        dlogits[p * ldl + v] = dldx * x * (1.f - sg);
    }
    // a wave per pair: its lanes read the K labels of the row side by side (one thread walking a row alone was 117 loads, each
    // from another cache line than its neighbours' -- 25 of the kernel's 35 us); the labels are 0 / 1, their sum is exact in any order
    const int lane = threadIdx.x & 63;
    for (int pl = blockIdx.y * 4 + (threadIdx.x >> 6); pl < P; pl += 4 * gridDim.y) {
        const int64_t p = (int64_t)mt.pair_off + pl;
        float ys = 0.f;
        for (int v = lane; v < K; v += 64) ys += labels[p * K + v];
        ys = skg_wave_sum(ys);
        if (lane != 0) continue;
        const float y = fminf(ys, 1.f);
        unary[p] = y;
        if (y != 0.f) n2 += 1.f;
        const float w = 1.f / (1.f + expf(-logits[p * ldl + K]));
        float dldw;
        s2 += skg_focal(w, y, 2.0f, dldw);
        dlogits[p * ldl + K] = dldw * w * (1.f - w);
    }
    s1 = skg_block_sum256(s1, sred);
    s2 = skg_block_sum256(s2, sred);
    n1 = skg_block_sum256(n1, sred);
    n2 = skg_block_sum256(n2, sred);
    if (threadIdx.x == 0) {
        float* o = partial + 4 * ((int64_t)a * gridDim.y + blockIdx.y);
        o[0] = s1; o[1] = s2; o[2] = n1; o[3] = n2;
    }
}

extern "C" int skg_hoi_loss_f32(const float* logits, int64_t ldl, int K, const skg_image_meta* meta, int n_active,
                                int64_t cells_total, const int64_t* index, const int64_t* pred, const float* scores,
                                const float* labels, float* cell_labels, float* unary, float* partial, float* dlogits,
                                void* stream) {
    if (n_active < 0 || K <= 0 || ldl <= K || cells_total < 0) return SKG_E_ARG;
    if (n_active == 0) return 0;
    if (!logits || !meta || !index || !pred || !scores || !labels || !cell_labels || !unary || !partial || !dlogits)
        return SKG_E_ARG;
    hipLaunchKernelGGL(skg_hoi_loss_kernel, dim3(n_active, SKG_LOSS_CHUNKS), dim3(256), 0, (hipStream_t)stream, logits, ldl, K, meta, n_active,
                       cells_total, index, pred, scores, labels, cell_labels, unary, partial, dlogits);
    return skg_launch_status();
}

// The three per-rank normaliser counts of the loss terms from what the PREPARATION of a batch knows (HEAD:162-172, 190-199,
// 219-228: n_p = #non-zero labels among the scored cells / #pairs with any label, twice): they depend on the detections, the
// verb table and the associated labels, not on the logits, so a data-parallel trainer forms them -- and starts their
// all-reduce -- while the batch is being prepared instead of between the two halves of the loss.  A wave per kept pair;
// the counts are small integers, exact in fp32 whatever the order of the atomic adds.
__global__ __launch_bounds__(256) void skg_count_positives_kernel(
    const float* __restrict__ labels, int K, const float* __restrict__ det_scores, const int64_t* __restrict__ det_labels,
    const skg_image_meta* __restrict__ meta, const int64_t* __restrict__ x_keep, const int64_t* __restrict__ y_keep,
    const int32_t* __restrict__ verb_off, const int32_t* __restrict__ verb_list, int num_obj_classes, float prior_pow,
    float* __restrict__ counts) {
    __shared__ float sred[4];
    const skg_image_meta mt = meta[blockIdx.x];
    const int P = mt.n_h * (mt.n - 1);
    const int lane = threadIdx.x & 63;
    float n1 = 0.f, n2 = 0.f;
    for (int pl = blockIdx.y * 4 + (threadIdx.x >> 6); pl < P; pl += 4 * gridDim.y) {
        const int64_t p = (int64_t)mt.pair_off + pl;
        const int bh = mt.box_off + (int)x_keep[p];
        const int64_t lab = det_labels[mt.box_off + (int)y_keep[p]];
        const int cls = (int)lab;
        const int nv = (lab >= 0 && lab < num_obj_classes && powf(det_scores[bh], prior_pow) != 0.f)
                           ? verb_off[cls + 1] - verb_off[cls] : 0;                 // the pair's scored cells (skg_pair_cells)
        float ys = 0.f, c1 = 0.f;
        for (int v = lane; v < K; v += 64) ys += labels[p * K + v];
        for (int t = lane; t < nv; t += 64) c1 += labels[p * K + verb_list[verb_off[cls] + t]] != 0.f ? 1.f : 0.f;
        ys = skg_wave_sum(ys);
        c1 = skg_wave_sum(c1);
        if (lane == 0) { n1 += c1; if (ys != 0.f) n2 += 1.f; }
    }
    n1 = skg_block_sum256(n1, sred);
    n2 = skg_block_sum256(n2, sred);
    if (threadIdx.x == 0 && (n1 != 0.f || n2 != 0.f)) {
        atomicAdd(counts, n1); atomicAdd(counts + 1, n2); atomicAdd(counts + 2, n2);
    }
}

extern "C" int skg_count_positives_f32(const float* labels, int K, const float* det_scores, const int64_t* det_labels,
                                       const skg_image_meta* meta, int n_active, const int64_t* x_keep,
                                       const int64_t* y_keep, const int32_t* verb_off, const int32_t* verb_list,
                                       int num_obj_classes, float prior_pow, float* counts, void* stream) {
    if (n_active < 0 || K <= 0 || num_obj_classes <= 0 || !counts) return SKG_E_ARG;
    hipError_t e = hipMemsetAsync(counts, 0, 3 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (n_active == 0) return 0;
    if (!labels || !det_scores || !det_labels || !meta || !x_keep || !y_keep || !verb_off || !verb_list) return SKG_E_ARG;
    hipLaunchKernelGGL(skg_count_positives_kernel, dim3(n_active, 16), dim3(256), 0, (hipStream_t)stream, labels, K,
                       det_scores, det_labels, meta, x_keep, y_keep, verb_off, verb_list, num_obj_classes, prior_pow, counts);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ TransH pos / neg sampling
// HEAD:936-963 + the intended MarginLoss (HEAD:207-235; heads/NegativeSampling.py:52-56, heads/MarginLoss.py:28-36):
// per image the TransH scores of the positive cells (labels != 0, row-major over (pair, verb)) and of as many sampled
// negatives -- the cells of rank perm[i] among the image's ZERO cells (nonzero(labels == 0)[randperm(#zeros)[:m]]).  The
// reference materialises both index lists with nonzero() over ~10^5 cells per image; positives are a few dozen, so:
//   kernel 1 (image x SKG_SAMPLE_CHUNKS workgroups): every workgroup compacts the positive cells of its slice of the
//            image, in order (wave ballots), into its own short list
//   kernel 2 (one workgroup per image): concatenates the lists (= all positives in row-major order), emits their
//            scores, and finds the zero cell of rank r as r + #(positives at or before it) by walking the positive list
// Outputs pos_scores / neg_scores [sum m] (image a at pos_off[a]) and partial[a] = sum_i max(pos_i - neg_i, -margin).
#define SKG_SAMPLE_CHUNKS 16

__global__ __launch_bounds__(256) void skg_transh_compact_kernel(const float* __restrict__ labels, int K,
                                                                 const skg_image_meta* __restrict__ meta, int cap,
                                                                 int32_t* __restrict__ chunk_cells,
                                                                 int32_t* __restrict__ chunk_count) {
    __shared__ int wcnt[4];
    __shared__ int base;
    const int a = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const skg_image_meta mt = meta[a];
    const int cells = mt.n_h * (mt.n - 1) * K;
    const int per = ((cells + SKG_SAMPLE_CHUNKS - 1) / SKG_SAMPLE_CHUNKS + 255) / 256 * 256;
    const int c0 = min(ch * per, cells), c1 = min(c0 + per, cells);
    const float* lab = labels + (int64_t)mt.pair_off * K;
    int32_t* out = chunk_cells + ((int64_t)a * SKG_SAMPLE_CHUNKS + ch) * cap;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int t0 = c0; t0 < c1; t0 += 256) {
        const int c = t0 + tid;
        const bool p = c < c1 && lab[c] != 0.f;
        const unsigned long long bal = __ballot(p);
        if (lane == 0) wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wcnt[w];
        const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (p && k < cap) out[k] = c;
        __syncthreads();
        if (tid == 0) base += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        __syncthreads();
    }
    if (tid == 0) chunk_count[a * SKG_SAMPLE_CHUNKS + ch] = base;
}

__global__ __launch_bounds__(256) void skg_transh_sample_kernel(const float* __restrict__ scores, int K,
                                                                const skg_image_meta* __restrict__ meta, int cap,
                                                                const int32_t* __restrict__ chunk_cells,
                                                                const int32_t* __restrict__ chunk_count,
                                                                const int32_t* __restrict__ pos_off,
                                                                const int64_t* __restrict__ perm, float margin,
                                                                int32_t* __restrict__ pos_cells,
                                                                float* __restrict__ pos_scores,
                                                                float* __restrict__ neg_scores,
                                                                float* __restrict__ partial) {
    __shared__ int cbase[SKG_SAMPLE_CHUNKS + 1];
    __shared__ float sred[4];
    const int a = blockIdx.x, tid = threadIdx.x;
    const skg_image_meta mt = meta[a];
    const float* sc = scores + (int64_t)mt.pair_off * K;
    const int o0 = pos_off[a], m = pos_off[a + 1] - o0;
    if (tid == 0) {
        cbase[0] = 0;
        for (int c = 0; c < SKG_SAMPLE_CHUNKS; ++c) cbase[c + 1] = cbase[c] + chunk_count[a * SKG_SAMPLE_CHUNKS + c];
    }
    __syncthreads();
    // positives in row-major order = the chunk lists back to back
    for (int c = 0; c < SKG_SAMPLE_CHUNKS; ++c) {
        const int n = cbase[c + 1] - cbase[c];
        const int32_t* src = chunk_cells + ((int64_t)a * SKG_SAMPLE_CHUNKS + c) * cap;
        for (int i = tid; i < n; i += 256) {
            const int k = cbase[c] + i;
            if (k < m) { const int cell = src[i]; pos_cells[o0 + k] = cell; pos_scores[o0 + k] = sc[cell]; }
        }
    }
    __threadfence_block();
    __syncthreads();
    // zero cell of rank r: c = r + (number of positive cells <= c); the positive list is sorted
    float s = 0.f;
    for (int i = tid; i < m; i += 256) {
        int64_t c = perm[o0 + i];
        for (int j = 0; j < m; ++j) {
            if ((int64_t)pos_cells[o0 + j] <= c) ++c; else break;
        }
        const float nv = sc[c];
        neg_scores[o0 + i] = nv;
        s += fmaxf(pos_scores[o0 + i] - nv, -margin);
    }
    s = skg_block_sum256(s, sred);
    if (tid == 0) partial[a] = s;
}

extern "C" int64_t skg_transh_sample_ws_ints(int n_active, int max_pos_per_image) {
    if (n_active < 0 || max_pos_per_image < 0) return SKG_E_ARG;
    const int64_t cap = max_pos_per_image > 0 ? max_pos_per_image : 1;
    return (int64_t)n_active * SKG_SAMPLE_CHUNKS * (cap + 1);
}

extern "C" int skg_transh_sample_f32(const float* labels, const float* scores, int K, const skg_image_meta* meta,
                                     int n_active, const int32_t* pos_off, int max_pos_per_image, const int64_t* perm,
                                     float margin, int32_t* ws, int32_t* pos_cells, float* pos_scores, float* neg_scores,
                                     float* partial, void* stream) {
    if (n_active < 0 || K <= 0 || max_pos_per_image < 0) return SKG_E_ARG;
    if (n_active == 0) return 0;
    if (!labels || !scores || !meta || !pos_off || !perm || !ws || !pos_cells || !pos_scores || !neg_scores || !partial)
        return SKG_E_ARG;
    const int cap = max_pos_per_image > 0 ? max_pos_per_image : 1;
    int32_t* counts = ws + (int64_t)n_active * SKG_SAMPLE_CHUNKS * cap;
    hipLaunchKernelGGL(skg_transh_compact_kernel, dim3(n_active, SKG_SAMPLE_CHUNKS), dim3(256), 0, (hipStream_t)stream,
                       labels, K, meta, cap, ws, counts);
    hipLaunchKernelGGL(skg_transh_sample_kernel, dim3(n_active), dim3(256), 0, (hipStream_t)stream, scores, K, meta, cap,
                       ws, counts, pos_off, perm, margin, pos_cells, pos_scores, neg_scores, partial);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ loss tail
// The scalars of the three loss terms from what the loss / sampling kernels left behind (HEAD:162-177, 190-205, 228-234):
//   sums = sum over rows of partial [rows, 4] = {cell loss, pair loss, #positive cells, #positive pairs}
//   n_p  = norm_in (the data-parallel normalisers all_reduce_sum(counts) / world, HEAD:167-172) or {sums[2], sums[3], sums[3]}
//   losses = {sums[0] / n_p[0], sums[1] / n_p[1], (sum(mpart) / max(M_pos, 1) + margin) / n_p[2]}
//   scale  = {1 / n_p[0], 1 / n_p[1]}: what d(loss)/d(logit) of the loss kernel has to be multiplied with
//   counts_out (optional) = {sums[2], sums[3], sums[3]}: the per-rank counts a data-parallel caller all-reduces first
// One wavefront; fixed summation order (deterministic).
__global__ __launch_bounds__(64) void skg_loss_finish_kernel(const float* __restrict__ partial, int rows,
                                                             const float* __restrict__ mpart, int n_img, float m_pos,
                                                             float margin, float grad_share,
                                                             const float* __restrict__ norm_in,
                                                             float* __restrict__ losses, float* __restrict__ scale,
                                                             float* __restrict__ counts_out) {
    const int lane = threadIdx.x;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, sm = 0.f;
    for (int r = lane; r < rows; r += 64) {
        s0 += partial[4 * r]; s1 += partial[4 * r + 1]; s2 += partial[4 * r + 2]; s3 += partial[4 * r + 3];
    }
    for (int a = lane; a < n_img; a += 64) sm += mpart[a];
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 += __shfl_xor(s3, o);
        sm += __shfl_xor(sm, o);
    }
    if (lane == 0) {
        if (counts_out) { counts_out[0] = s2; counts_out[1] = s3; counts_out[2] = s3; }
        if (losses) {
            const float n0 = norm_in ? norm_in[0] : s2, n1 = norm_in ? norm_in[1] : s3, n2 = norm_in ? norm_in[2] : s3;
            losses[0] = s0 / n0; losses[1] = s1 / n1;
            losses[2] = (sm / (m_pos > 1.f ? m_pos : 1.f) + margin) / n2;
            scale[0] = (1.f / n0) * grad_share; scale[1] = (1.f / n1) * grad_share;
        }
    }
}

extern "C" int skg_loss_finish_f32(const float* partial, int rows, const float* mpart, int n_img, int64_t m_pos,
                                   float margin, float grad_share, const float* norm_in, float* losses, float* scale,
                                   float* counts_out, void* stream) {
    if (rows < 0 || n_img < 0 || !partial || !mpart || (!losses && !counts_out) || (losses && !scale)) return SKG_E_ARG;
    hipLaunchKernelGGL(skg_loss_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, rows, mpart, n_img,
                       (float)m_pos, margin, grad_share, norm_in, losses, scale, counts_out);
    return skg_launch_status();
}

// dl[r, c] *= (c < K ? scale[0] * g[0] : scale[1] * g[1])  -- d(total)/d(logits) from the loss kernel's d(sum)/d(logits):
// scale = 1 / n_p of the two focal terms (skg_loss_finish_f32), g = the upstream gradients of the two loss scalars.
__global__ __launch_bounds__(256) void skg_scale_dlogits_kernel(float* __restrict__ dl, int64_t ld, int rows, int K,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ g0,
                                                                const float* __restrict__ g1,
                                                                float* __restrict__ out) {
    const float a = scale[0] * g0[0], b = scale[1] * g1[0];
    const int64_t n = (int64_t)rows * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % ld);
        out[i] = dl[i] * (c < K ? a : b);
    }
}

extern "C" int skg_scale_dlogits_f32(const float* dl, int64_t ld, int rows, int K, const float* scale, const float* g0,
                                     const float* g1, float* out, void* stream) {
    if (rows < 0 || K <= 0 || ld <= K || !dl || !scale || !g0 || !g1 || !out) return SKG_E_ARG;
    if (rows == 0) return 0;
    int64_t n = (int64_t)rows * ld;
    int blocks = (int)((n + 256 * 4 - 1) / (256 * 4)); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(skg_scale_dlogits_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, const_cast<float*>(dl), ld,
                       rows, K, scale, g0, g1, out);
    return skg_launch_status();
}
