// skg_graph.hip -- bipartite message aggregation and LayerNorm for the interaction head's graph.
//
// Reference: heads/adamixer_transH_spatial_r50_head.py:897-925.
//   adjacency = Linear(1024 -> 1)(attention_head(...)).reshape(n_h, n)                                  (HEAD:897)
//   messages_to_h = relu( sum_j softmax_j(adj)[i, j]   * obj_to_sub(o_j, s_ij) )                        (HEAD:907-910)
//   messages_to_o = relu( sum_i softmax_i(adj^T)[j, i] * sub_to_obj(h_i, s_ij) )                        (HEAD:916-922)
// MessageMBF ends in fc_3, a linear map with no activation (HEAD:509-527), and each softmax row sums to one, so the
// weighted sum is taken over the PRE-fc_3 rows T (= relu(fc_1 * fc_2), [G, 1024]) and fc_3 then runs on n_h (resp. n)
// rows instead of n_h * n rows: sum_j a_ij (W3 T_ij + b3) = W3 (sum_j a_ij T_ij) + b3.
//
// One workgroup per destination node; its adjacency row (or column) and softmax live in LDS; the T rows are streamed
// with coalesced 16-byte loads (the kernel is HBM-bound: it reads each T row exactly once).
#include "skg_common.h"

__global__ __launch_bounds__(256) void skg_graph_aggregate_kernel(
    const float* __restrict__ dot_partial, int n_partial, int64_t partial_ld, float adj_bias,
    const skg_image_meta* __restrict__ meta, const int32_t* __restrict__ hum_img, const int32_t* __restrict__ node_img,
    int sum_h, const float* __restrict__ T_os, const float* __restrict__ T_so, int64_t ldt, int cols,
    float* __restrict__ U, float* __restrict__ V, int64_t ldu, float* __restrict__ adj_out,
    float* __restrict__ alpha_out, float* __restrict__ beta_out, uint16_t* __restrict__ U16, uint16_t* __restrict__ V16) {
    __shared__ float sw[SKG_MAX_NODES];
    __shared__ float sred[4];
    const int tid = threadIdx.x;
    const bool to_human = (int)blockIdx.x < sum_h;
    const int dst = to_human ? blockIdx.x : blockIdx.x - sum_h;
    const int a = to_human ? hum_img[dst] : node_img[dst];
    const skg_image_meta mt = meta[a];
    const int local = to_human ? dst - mt.hum_off : dst - mt.node_off;
    // capacity padding (plans sized for a bucket of shapes): rows past the image's humans / nodes are nobody's destination
    if (local >= (to_human ? mt.n_h : mt.n)) return;
    const int cnt = to_human ? mt.n : mt.n_h;                 // number of senders
    const int64_t row0 = to_human ? (int64_t)mt.grid_off + (int64_t)local * mt.n : (int64_t)mt.grid_off + local;
    const int64_t rstep = to_human ? 1 : mt.n;

    // adjacency logits of the senders (sum of the per-64-column partial dots, fixed order -> deterministic)
    for (int t = tid; t < cnt; t += 256) {
        const int64_t gr = row0 + t * rstep;
        float s = 0.f;
        for (int k = 0; k < n_partial; ++k) s += dot_partial[(int64_t)k * partial_ld + gr];
        s += adj_bias;
        sw[t] = s;
        if (to_human && adj_out) adj_out[gr] = s;
    }
    __syncthreads();
    // softmax over the senders
    float mx = -INFINITY;
    for (int t = tid; t < cnt; t += 256) mx = fmaxf(mx, sw[t]);
    mx = skg_wave_max(mx);
    if ((tid & 63) == 0) sred[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    float se = 0.f;
    for (int t = tid; t < cnt; t += 256) {
        const float e = expf(sw[t] - mx);
        sw[t] = e;
        se += e;
    }
    se = skg_block_sum256(se, sred);                           // barriers inside also publish sw[]
    const float inv = 1.f / se;
    // training: the softmax weights themselves (alpha over a human's senders, beta over a node's), kept for the backward
    float* wout = to_human ? alpha_out : beta_out;
    if (wout)
        for (int t = tid; t < cnt; t += 256) wout[row0 + t * rstep] = sw[t] * inv;

    const float* T = to_human ? T_os : T_so;
    float* out = (to_human ? U : V) + (int64_t)dst * ldu;
    uint16_t* out16 = to_human ? U16 : V16;
    if (out16) out16 += (int64_t)dst * ldu;
    for (int c = tid * 4; c < cols; c += 1024) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = T + row0 * ldt + c;
        for (int t = 0; t < cnt; ++t) {
            const float wt = sw[t] * inv;
            const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)t * rstep * ldt);
            acc.x += wt * v.x; acc.y += wt * v.y; acc.z += wt * v.z; acc.w += wt * v.w;
        }
        *reinterpret_cast<float4*>(out + c) = acc;
        if (out16) skg_store_twin4(out16 + c, acc);
    }
}

extern "C" int skg_graph_aggregate_f32(const float* dot_partial, int n_partial, int64_t partial_ld, float adj_bias,
                                       const skg_image_meta* meta, int n_active, const int32_t* hum_img,
                                       const int32_t* node_img, int sum_h, int sum_n, const float* T_os,
                                       const float* T_so, int64_t ldt, int cols, float* U, float* V, int64_t ldu,
                                       float* adj_out, void* stream) {
    if (n_active < 0 || sum_h < 0 || sum_n < 0 || n_partial <= 0 || cols <= 0 || (cols & 3)) return SKG_E_ARG;
    if (sum_h + sum_n == 0) return 0;
    if (!dot_partial || !meta || !hum_img || !node_img || !T_os || !T_so || !U || !V) return SKG_E_ARG;
    if ((ldt & 3) || (ldu & 3) || !skg_aligned16(T_os) || !skg_aligned16(T_so) || !skg_aligned16(U) ||
        !skg_aligned16(V))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_graph_aggregate_kernel, dim3(sum_h + sum_n), dim3(256), 0, (hipStream_t)stream, dot_partial,
                       n_partial, partial_ld, adj_bias, meta, hum_img, node_img, sum_h, T_os, T_so, ldt, cols, U, V,
                       ldu, adj_out, (float*)nullptr, (float*)nullptr, (uint16_t*)nullptr, (uint16_t*)nullptr);
    return skg_launch_status();
}

// The same aggregation for the training step: also returns the softmax weights alpha[sumG] (over the senders of every
// human) and beta[sumG] (over the senders of every node), which the backward (skg_aggregate_bwd_f32) needs.
extern "C" int skg_graph_aggregate_train_f32(const float* dot_partial, int n_partial, int64_t partial_ld, float adj_bias,
                                             const skg_image_meta* meta, int n_active, const int32_t* hum_img,
                                             const int32_t* node_img, int sum_h, int sum_n, const float* T_os,
                                             const float* T_so, int64_t ldt, int cols, float* U, float* V, int64_t ldu,
                                             float* adj_out, float* alpha_out, float* beta_out, void* stream) {
    if (n_active < 0 || sum_h < 0 || sum_n < 0 || n_partial <= 0 || cols <= 0 || (cols & 3)) return SKG_E_ARG;
    if (sum_h + sum_n == 0) return 0;
    if (!dot_partial || !meta || !hum_img || !node_img || !T_os || !T_so || !U || !V || !alpha_out || !beta_out)
        return SKG_E_ARG;
    if ((ldt & 3) || (ldu & 3) || !skg_aligned16(T_os) || !skg_aligned16(T_so) || !skg_aligned16(U) ||
        !skg_aligned16(V))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_graph_aggregate_kernel, dim3(sum_h + sum_n), dim3(256), 0, (hipStream_t)stream, dot_partial,
                       n_partial, partial_ld, adj_bias, meta, hum_img, node_img, sum_h, T_os, T_so, ldt, cols, U, V,
                       ldu, adj_out, alpha_out, beta_out, skg_twin(U), skg_twin(V));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// nn.LayerNorm(1024) (HEAD:658-659): biased variance, eps inside the sqrt.  One workgroup per row, two passes over
// registers (mean, then centred sum of squares).
__global__ __launch_bounds__(256) void skg_layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int cols, float eps,
                                                            float* __restrict__ out, int64_t ldo) {
    __shared__ float sred[4];
    const int r = blockIdx.x;
    const int c = threadIdx.x * 4;
    const bool in = c < cols;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) v = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c);
    const float mean = skg_block_sum256((v.x + v.y) + (v.z + v.w), sred) / (float)cols;
    float4 dlt = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    float sq = in ? (dlt.x * dlt.x + dlt.y * dlt.y) + (dlt.z * dlt.z + dlt.w * dlt.w) : 0.f;
    const float var = skg_block_sum256(sq, sred) / (float)cols;
    const float rstd = 1.f / sqrtf(var + eps);
    if (in) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c);
        const float4 bb = *reinterpret_cast<const float4*>(beta + c);
        *reinterpret_cast<float4*>(out + (int64_t)r * ldo + c) =
            make_float4(dlt.x * rstd * g.x + bb.x, dlt.y * rstd * g.y + bb.y, dlt.z * rstd * g.z + bb.z,
                        dlt.w * rstd * g.w + bb.w);
    }
}

// Two LayerNorms in one launch (norm_h on the human rows, norm_o on the node rows: HEAD:912-914, 923-925): workgroups
// [0, rows0) take the first, the rest the second; per row the same arithmetic as skg_layernorm_kernel, bit for bit.
struct skg_ln_seg { const float* x; int64_t ldx; const float* gamma; const float* beta; float* out; int64_t ldo; int rows; };

__global__ __launch_bounds__(256) void skg_layernorm2_kernel(const skg_ln_seg a, const skg_ln_seg b, int cols, float eps) {
    __shared__ float sred[4];
    const bool first = (int)blockIdx.x < a.rows;
    const skg_ln_seg& g = first ? a : b;
    const int r = first ? (int)blockIdx.x : (int)blockIdx.x - a.rows;
    const int c = threadIdx.x * 4;
    const bool in = c < cols;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) v = *reinterpret_cast<const float4*>(g.x + (int64_t)r * g.ldx + c);
    const float mean = skg_block_sum256((v.x + v.y) + (v.z + v.w), sred) / (float)cols;
    float4 dlt = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    float sq = in ? (dlt.x * dlt.x + dlt.y * dlt.y) + (dlt.z * dlt.z + dlt.w * dlt.w) : 0.f;
    const float var = skg_block_sum256(sq, sred) / (float)cols;
    const float rstd = 1.f / sqrtf(var + eps);
    if (in) {
        const float4 gm = *reinterpret_cast<const float4*>(g.gamma + c);
        const float4 bb = *reinterpret_cast<const float4*>(g.beta + c);
        *reinterpret_cast<float4*>(g.out + (int64_t)r * g.ldo + c) =
            make_float4(dlt.x * rstd * gm.x + bb.x, dlt.y * rstd * gm.y + bb.y, dlt.z * rstd * gm.z + bb.z,
                        dlt.w * rstd * gm.w + bb.w);
    }
}

extern "C" int skg_layernorm2_f32(const float* x0, int64_t ldx0, const float* gamma0, const float* beta0, int rows0,
                                  float* out0, int64_t ldo0, const float* x1, int64_t ldx1, const float* gamma1,
                                  const float* beta1, int rows1, float* out1, int64_t ldo1, int cols, float eps,
                                  void* stream) {
    if (rows0 < 0 || rows1 < 0 || cols <= 0 || cols > 1024 || (cols & 3)) return SKG_E_ARG;
    if (rows0 + rows1 == 0) return 0;
    if ((rows0 && (!x0 || !gamma0 || !beta0 || !out0)) || (rows1 && (!x1 || !gamma1 || !beta1 || !out1))) return SKG_E_ARG;
    if ((ldx0 & 3) || (ldo0 & 3) || (ldx1 & 3) || (ldo1 & 3)) return SKG_E_ALIGN;
    if (rows0 && (!skg_aligned16(x0) || !skg_aligned16(out0) || !skg_aligned16(gamma0) || !skg_aligned16(beta0))) return SKG_E_ALIGN;
    if (rows1 && (!skg_aligned16(x1) || !skg_aligned16(out1) || !skg_aligned16(gamma1) || !skg_aligned16(beta1))) return SKG_E_ALIGN;
    const skg_ln_seg a{x0, ldx0, gamma0, beta0, out0, ldo0, rows0}, b{x1, ldx1, gamma1, beta1, out1, ldo1, rows1};
    hipLaunchKernelGGL(skg_layernorm2_kernel, dim3(rows0 + rows1), dim3(256), 0, (hipStream_t)stream, a, b, cols, eps);
    return skg_launch_status();
}

extern "C" int skg_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int rows,
                                 int cols, float eps, float* out, int64_t ldo, void* stream) {
    if (rows < 0 || cols <= 0 || cols > 1024 || (cols & 3)) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!x || !gamma || !beta || !out) return SKG_E_ARG;
    if ((ldx & 3) || (ldo & 3) || !skg_aligned16(x) || !skg_aligned16(out) || !skg_aligned16(gamma) ||
        !skg_aligned16(beta))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_layernorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, cols,
                       eps, out, ldo);
    return skg_launch_status();
}
