// skg_pairs.hip -- pair enumeration + 46-d box-pair spatial encoding, plus the small row-wise helpers.
//
// Reference: heads/adamixer_transH_spatial_r50_head.py:847-868 (meshgrid / nonzero(x != y), NaN scrub) and
// ops.py:85-157 (compute_spatial_ratio_encodings; feature order ops.py:134-152, eps 1e-10).
// One workgroup (4 wavefronts) owns one image: its boxes are staged in LDS, every lane walks grid rows r = t, t+256, ...
// (one wavefront per image until round 2: at a single image that is one wave on the whole chip walking 800 rows)
// The reference builds a G x G IoU matrix to take its diagonal (ops.py:119); only the diagonal is computed here.
// Compiled with -ffp-contract=off: each feature is the same sequence of IEEE fp32 operations as the CPU code
// (the log is the only non-correctly-rounded function).
#include "skg_common.h"

#define PS_THREADS 256
__global__ __launch_bounds__(PS_THREADS) void skg_pairs_spatial_kernel(
    const float* __restrict__ boxes, const skg_image_meta* __restrict__ meta, int32_t* __restrict__ grid_h,
    int32_t* __restrict__ grid_o, int32_t* __restrict__ grid_pair, int32_t* __restrict__ grid_img,
    int32_t* __restrict__ pair_grid, int64_t* __restrict__ x_keep, int64_t* __restrict__ y_keep,
    int32_t* __restrict__ pair_h, int32_t* __restrict__ pair_o, float* __restrict__ spatial, int scrub_nan,
    int grid_cap, int pair_cap) {
    __shared__ float4 sbox[SKG_MAX_NODES];
    const int a = blockIdx.x;
    const skg_image_meta mt = meta[a];
    const int lane = threadIdx.x;
    const int n = mt.n, n_h = mt.n_h;
    for (int t = lane; t < n; t += PS_THREADS)
        sbox[t] = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)(mt.box_off + t));
    __syncthreads();

    const float h = mt.img_h, w = mt.img_w;
    const float eps = 1e-10f;
    const int G = n_h * n;
    bool any_nan = false;
    for (int r = lane; r < G; r += PS_THREADS) {
        const int i = r / n, j = r - i * n;
        const int gr = mt.grid_off + r;
        grid_h[gr] = mt.hum_off + i;
        grid_o[gr] = mt.node_off + j;
        grid_img[gr] = mt.image;
        if (i != j) {
            const int pl = i * (n - 1) + (j < i ? j : j - 1);
            const int p = mt.pair_off + pl;
            grid_pair[gr] = p;
            pair_grid[p] = gr;
            x_keep[p] = i;
            y_keep[p] = j;
            pair_h[p] = mt.hum_off + i;
            pair_o[p] = mt.node_off + j;
        } else {
            grid_pair[gr] = -1;
        }
        const float4 b1 = sbox[i], b2 = sbox[j];
        const float c1x = (b1.x + b1.z) / 2.f, c1y = (b1.y + b1.w) / 2.f;
        const float c2x = (b2.x + b2.z) / 2.f, c2y = (b2.y + b2.w) / 2.f;
        const float b1w = b1.z - b1.x, b1h = b1.w - b1.y;
        const float b2w = b2.z - b2.x, b2h = b2.w - b2.y;
        const float dx = fabsf(c2x - c1x) / (b1w + eps);
        const float dy = fabsf(c2y - c1y) / (b1h + eps);
        // IoU (torchvision box_iou): inter / (area1 + area2 - inter), no eps
        const float ltx = fmaxf(b1.x, b2.x), lty = fmaxf(b1.y, b2.y);
        const float rbx = fminf(b1.z, b2.z), rby = fminf(b1.w, b2.w);
        const float iw = fmaxf(rbx - ltx, 0.f), ih = fmaxf(rby - lty, 0.f);
        const float inter = iw * ih;
        const float iou = inter / (b1w * b1h + b2w * b2h - inter);
        const float c1xw = c1x / w, c1yh = c1y / h, c2xw = c2x / w, c2yh = c2y / h;
        const float b1ww = b1w / w, b1hh = b1h / h, b2ww = b2w / w, b2hh = b2h / h;
        const float hw = h * w;
        const float a1 = b1w * b1h / hw, a2 = b2w * b2h / hw;
        float f[23];
        f[0] = c1xw; f[1] = c1yh; f[2] = c2xw; f[3] = c2yh;
        f[4] = c1xw / (c2xw + eps); f[5] = c1yh / (c2yh + eps);
        f[6] = b1ww; f[7] = b1hh; f[8] = b2ww; f[9] = b2hh;
        f[10] = b1ww / (b2ww + eps); f[11] = b1hh / (b2hh + eps);
        f[12] = a1; f[13] = a2; f[14] = a1 / (a2 + eps);
        f[15] = b2w * b2h / (b1w * b1h + eps);
        f[16] = b1w / (b1h + eps); f[17] = b2w / (b2h + eps);
        f[18] = iou;
        f[19] = (c2x > c1x ? 1.f : 0.f) * dx; f[20] = (c2x < c1x ? 1.f : 0.f) * dx;
        f[21] = (c2y > c1y ? 1.f : 0.f) * dy; f[22] = (c2y < c1y ? 1.f : 0.f) * dy;
        float o[SKG_SPATIAL_LD];
#pragma unroll
        for (int k = 0; k < 23; ++k) {
            o[k] = f[k];
            o[23 + k] = logf(f[k] + eps);
            any_nan |= (o[k] != o[k]) | (o[23 + k] != o[23 + k]);
        }
        o[46] = 0.f; o[47] = 0.f;
        float4* dst = reinterpret_cast<float4*>(spatial + (int64_t)gr * SKG_SPATIAL_LD);
#pragma unroll
        for (int k = 0; k < SKG_SPATIAL_LD / 4; ++k) dst[k] = make_float4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
    }
    // Capacity padding (captured small-batch plans sized for a BUCKET of shapes, skghoi_amd/small.py): the image owns
    // grid_cap grid rows and pair_cap pair rows of which it uses G and n_h (n - 1).  The unused tail gets index entries that
    // are safe to gather / scatter through -- a valid human / node / grid row to read, -1 (= not stored) to scatter to -- and
    // zero spatial features, so that the launches that walk all capacity rows compute finite garbage nobody reads.
    if (grid_cap > G) {
        for (int r = G + lane; r < grid_cap; r += PS_THREADS) {
            const int gr = mt.grid_off + r;
            grid_h[gr] = mt.hum_off; grid_o[gr] = mt.node_off; grid_img[gr] = mt.image; grid_pair[gr] = -1;
            float4* dst = reinterpret_cast<float4*>(spatial + (int64_t)gr * SKG_SPATIAL_LD);
#pragma unroll
            for (int k = 0; k < SKG_SPATIAL_LD / 4; ++k) dst[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int P = n_h * (n - 1);
    if (pair_cap > P) {
        for (int q = P + lane; q < pair_cap; q += PS_THREADS) {
            const int p = mt.pair_off + q;
            pair_grid[p] = mt.grid_off; x_keep[p] = 0; y_keep[p] = 0; pair_h[p] = mt.hum_off; pair_o[p] = mt.node_off;
        }
    }
    // torch.nan_to_num over the whole image tensor if it holds any NaN (HEAD:866-868): NaN -> 0, +-inf -> +-FLT_MAX
    if (__syncthreads_or(scrub_nan && any_nan)) {
        for (int r = lane; r < G; r += PS_THREADS) {
            float* row = spatial + (int64_t)(mt.grid_off + r) * SKG_SPATIAL_LD;     // rows this lane wrote itself
            for (int k = 0; k < 46; ++k) {
                const float v = row[k];
                if (v != v) row[k] = 0.f;
                else if (v == INFINITY) row[k] = 3.4028234663852886e38f;
                else if (v == -INFINITY) row[k] = -3.4028234663852886e38f;
            }
        }
    }
}

extern "C" int skg_pairs_spatial_f32(const float* boxes, const skg_image_meta* meta, int n_active, int32_t* grid_h,
                                     int32_t* grid_o, int32_t* grid_pair, int32_t* grid_img, int32_t* pair_grid,
                                     int64_t* x_keep, int64_t* y_keep, int32_t* pair_h, int32_t* pair_o,
                                     float* spatial, int scrub_nan, void* stream) {
    return skg_pairs_spatial_padded_f32(boxes, meta, n_active, grid_h, grid_o, grid_pair, grid_img, pair_grid, x_keep,
                                        y_keep, pair_h, pair_o, spatial, scrub_nan, 0, 0, stream);
}

extern "C" int skg_pairs_spatial_padded_f32(const float* boxes, const skg_image_meta* meta, int n_active, int32_t* grid_h,
                                            int32_t* grid_o, int32_t* grid_pair, int32_t* grid_img, int32_t* pair_grid,
                                            int64_t* x_keep, int64_t* y_keep, int32_t* pair_h, int32_t* pair_o,
                                            float* spatial, int scrub_nan, int grid_cap, int pair_cap, void* stream) {
    if (n_active < 0 || grid_cap < 0 || pair_cap < 0) return SKG_E_ARG;
    if (n_active == 0) return 0;
    if (!boxes || !meta || !grid_h || !grid_o || !grid_pair || !grid_img || !pair_grid || !x_keep || !y_keep ||
        !pair_h || !pair_o || !spatial)
        return SKG_E_ARG;
    if (!skg_aligned16(boxes) || !skg_aligned16(spatial)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_pairs_spatial_kernel, dim3(n_active), dim3(PS_THREADS), 0, (hipStream_t)stream, boxes, meta, grid_h,
                       grid_o, grid_pair, grid_img, pair_grid, x_keep, y_keep, pair_h, pair_o, spatial, scrub_nan, grid_cap, pair_cap);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ global avg pool
// AdaptiveAvgPool2d(1) (HEAD:811): one wavefront per (image, channel) plane, 4 planes per block.
__global__ __launch_bounds__(256) void skg_avgpool_kernel(const float* __restrict__ in, int planes, int HW,
                                                          float* __restrict__ out) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63;
    const float* p = in + (int64_t)plane * HW;
    float s = 0.f;
    for (int t = lane; t < HW; t += 64) s += p[t];
    s = skg_wave_sum(s);
    if (lane == 0) out[plane] = s / (float)HW;
}

extern "C" int skg_global_avgpool_f32(const float* in, int B, int C, int HW, float* out, void* stream) {
    if (B < 0 || C <= 0 || HW <= 0) return SKG_E_ARG;
    if (B == 0) return 0;
    if (!in || !out) return SKG_E_ARG;
    const int planes = B * C;
    hipLaunchKernelGGL(skg_avgpool_kernel, dim3((planes + 3) / 4), dim3(256), 0, (hipStream_t)stream, in, planes, HW,
                       out);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ fc_head/fc_tail input
__global__ __launch_bounds__(256) void skg_concat_entity_kernel(const float* __restrict__ enc, int64_t ld_enc,
                                                                const int32_t* __restrict__ enc_row,
                                                                const float* __restrict__ ent,
                                                                const int32_t* __restrict__ ent_img,
                                                                const int32_t* __restrict__ ent_row,
                                                                float* __restrict__ out, int64_t out_ld,
                                                                uint16_t* __restrict__ out16) {
    const int r = blockIdx.x;
    const float* src = enc + (int64_t)enc_row[r] * ld_enc;
    float* dst = out + (int64_t)r * out_ld;
    const int t = threadIdx.x;
    const float4 v = reinterpret_cast<const float4*>(src)[t];
    reinterpret_cast<float4*>(dst)[t] = v;                                                 // 256 x 4 = 1024 columns
    if (out16) skg_store_twin4(out16 + (int64_t)r * out_ld + 4 * t, v);
    if (t < 64) {
        const float* e = ent + ((int64_t)ent_img[r] * SKG_TRANSH_ENT + ent_row[r]) * SKG_TRANSH_DIM;
        const float x = (t < SKG_TRANSH_DIM) ? e[t] : 0.f;
        dst[1024 + t] = x;                                                                 // 1024..1087
        if (out16) out16[(int64_t)r * out_ld + 1024 + t] = (uint16_t)skg_pack_bf16(x, 0.f);
    }
}

extern "C" int skg_concat_entity_f32(const float* enc, int64_t ld_enc, const int32_t* enc_row, const float* ent,
                                     const int32_t* ent_img, const int32_t* ent_row, int rows, float* out,
                                     int64_t out_ld, void* stream) {
    if (rows < 0) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!enc || !enc_row || !ent || !ent_img || !ent_row || !out || out_ld < 1088) return SKG_E_ARG;
    if (!skg_aligned16(enc) || !skg_aligned16(out) || (ld_enc & 3) || (out_ld & 3)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_concat_entity_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, enc, ld_enc, enc_row,
                       ent, ent_img, ent_row, out, out_ld, skg_twin(out));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ read-out fc_1 * fc_2
__global__ __launch_bounds__(256) void skg_rows_mul_relu_kernel(const float* __restrict__ P,
                                                                const int32_t* __restrict__ p_idx, int64_t ldp,
                                                                const float* __restrict__ Q,
                                                                const int32_t* __restrict__ q_idx, int64_t ldq,
                                                                const float* __restrict__ mbias,
                                                                const float* __restrict__ F,
                                                                const int32_t* __restrict__ f_idx, int64_t ldf,
                                                                int cols, float* __restrict__ out, int64_t ldo,
                                                                uint16_t* __restrict__ out16) {
    const int r = blockIdx.x;
    const float* p = P + (int64_t)(p_idx ? p_idx[r] : r) * ldp;
    const float* q = Q ? Q + (int64_t)(q_idx ? q_idx[r] : r) * ldq : nullptr;
    const float* f = F + (int64_t)(f_idx ? f_idx[r] : r) * ldf;
    float* o = out + (int64_t)r * ldo;
    uint16_t* o16 = out16 ? out16 + (int64_t)r * ldo : nullptr;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        float4 m = *reinterpret_cast<const float4*>(p + c);
        if (q) {
            const float4 t = *reinterpret_cast<const float4*>(q + c);
            m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
        }
        if (mbias) {
            const float4 t = *reinterpret_cast<const float4*>(mbias + c);
            m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
        }
        const float4 v = *reinterpret_cast<const float4*>(f + c);
        const float4 res = make_float4(fmaxf(m.x * v.x, 0.f), fmaxf(m.y * v.y, 0.f), fmaxf(m.z * v.z, 0.f), fmaxf(m.w * v.w, 0.f));
        *reinterpret_cast<float4*>(o + c) = res;
        if (o16) skg_store_twin4(o16 + c, res);
    }
}

struct skg_rows_mul_pack { skg_rows_mul_args a[SKG_MULTI_MAX]; };
__global__ __launch_bounds__(256) void skg_rows_mul_relu_multi_kernel(const skg_rows_mul_pack pk) {
    const skg_rows_mul_args& a = pk.a[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= a.rows) return;
    const float* p = a.P + (int64_t)(a.p_idx ? a.p_idx[r] : r) * a.ldp;
    const float* q = a.Q ? a.Q + (int64_t)(a.q_idx ? a.q_idx[r] : r) * a.ldq : nullptr;
    const float* f = a.F + (int64_t)(a.f_idx ? a.f_idx[r] : r) * a.ldf;
    float* o = a.out + (int64_t)r * a.ldo;
    uint16_t* o16 = a.out16 ? a.out16 + (int64_t)r * a.ldo : nullptr;
    for (int c = threadIdx.x * 4; c < a.cols; c += 1024) {
        float4 m = *reinterpret_cast<const float4*>(p + c);
        if (q) {
            const float4 t = *reinterpret_cast<const float4*>(q + c);
            m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
        }
        if (a.mbias) {
            const float4 t = *reinterpret_cast<const float4*>(a.mbias + c);
            m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
        }
        const float4 v = *reinterpret_cast<const float4*>(f + c);
        const float4 res = make_float4(fmaxf(m.x * v.x, 0.f), fmaxf(m.y * v.y, 0.f), fmaxf(m.z * v.z, 0.f), fmaxf(m.w * v.w, 0.f));
        *reinterpret_cast<float4*>(o + c) = res;
        if (o16) skg_store_twin4(o16 + c, res);
    }
}

int skg_rows_mul_relu_multi(const skg_rows_mul_args* calls, int n, void* stream) {
    if (!calls || n < 1 || n > SKG_MULTI_MAX) return SKG_E_ARG;
    skg_rows_mul_pack pk;
    int m = 0, rows = 0;
    for (int i = 0; i < n; ++i) {
        const skg_rows_mul_args& a = calls[i];
        if (a.rows < 0 || a.cols <= 0 || (a.cols & 3)) return SKG_E_ARG;
        if (a.rows == 0) continue;
        if (!a.P || !a.F || !a.out) return SKG_E_ARG;
        if ((a.ldp & 3) || (a.ldf & 3) || (a.ldo & 3) || (a.Q && (a.ldq & 3))) return SKG_E_ALIGN;
        if (!skg_aligned16(a.P) || !skg_aligned16(a.F) || !skg_aligned16(a.out) || !skg_aligned16(a.Q) || !skg_aligned16(a.mbias))
            return SKG_E_ALIGN;
        pk.a[m] = a;
        pk.a[m++].out16 = skg_twin(a.out);
        rows = a.rows > rows ? a.rows : rows;
    }
    if (m == 0) return 0;
    hipLaunchKernelGGL(skg_rows_mul_relu_multi_kernel, dim3(rows, m), dim3(256), 0, (hipStream_t)stream, pk);
    return skg_launch_status();
}

extern "C" int skg_rows_mul_relu_f32(const float* P, const int32_t* p_idx, int64_t ldp, const float* Q,
                                     const int32_t* q_idx, int64_t ldq, const float* mbias, const float* F,
                                     const int32_t* f_idx, int64_t ldf, int rows, int cols, float* out, int64_t ldo,
                                     void* stream) {
    if (rows < 0 || cols <= 0 || (cols & 3)) return SKG_E_ARG;
    if (rows == 0) return 0;
    if (!P || !F || !out) return SKG_E_ARG;
    if ((ldp & 3) || (ldf & 3) || (ldo & 3) || (Q && (ldq & 3))) return SKG_E_ALIGN;
    if (!skg_aligned16(P) || !skg_aligned16(F) || !skg_aligned16(out) || (Q && !skg_aligned16(Q)) ||
        (mbias && !skg_aligned16(mbias)))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_rows_mul_relu_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, P, p_idx, ldp, Q,
                       q_idx, ldq, mbias, F, f_idx, ldf, cols, out, ldo, skg_twin(out));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ transpose
// out[c, r] = in[r, c] for r < rows, c < cols; out has ld_out >= rows columns, columns rows..ld_out-1 are zeroed by the
// caller.  Used by the backward GEMMs (dA = dZ W needs W^T rows, dW = dZ^T A needs both operands k-contiguous).
__global__ __launch_bounds__(256) void skg_transpose_kernel(const float* __restrict__ in, int64_t ld_in, int rows,
                                                            int cols, float* __restrict__ out, int64_t ld_out) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)r * ld_in + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(int64_t)c * ld_out + r] = tile[tx][i];
    }
}

extern "C" int skg_transpose_f32(const float* in, int64_t ld_in, int rows, int cols, float* out, int64_t ld_out,
                                 void* stream) {
    if (rows < 0 || cols < 0) return SKG_E_ARG;
    if (rows == 0 || cols == 0) return 0;
    if (!in || !out || ld_in < cols || ld_out < rows) return SKG_E_ARG;
    hipLaunchKernelGGL(skg_transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0,
                       (hipStream_t)stream, in, ld_in, rows, cols, out, ld_out);
    return skg_launch_status();
}
