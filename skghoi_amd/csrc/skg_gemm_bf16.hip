// skg_gemm_bf16.hip -- bf16 dense layer on the CDNA4 matrix cores (training config "bf16", BASELINE config 3):
//   C = act(A x W^T + bias),  A [M,K] bf16, W [N,K] bf16 (nn.Linear layout), fp32 accumulation, C fp32 or bf16.
// v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate).  128x128x64 tiles, 4 waves x (2x2) MFMA tiles, double-buffered
// LDS filled by global_load_lds_dwordx4 (8 rows x 128 B per wave instruction); the 16-byte chunk index is XOR-ed with
// (row & 7) on the global SOURCE side and on the fragment reads, which makes the ds_read_b128 conflict-free.
// Used by skghoi_amd.autograd for the forward, dX and dW products of the bf16 training path; split-K (fp32 partials)
// covers dW, whose reduction dimension is the batch-row count.
#include "skg_common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define HB_M 128
#define HB_N 128
#define HB_K 64
#define HB_TILE (HB_M * HB_K)            // bf16 elements per operand tile

__device__ __forceinline__ unsigned short skg_f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);     // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__global__ __launch_bounds__(256, 2) void skg_gemm_bf16_kernel(const skg_gemm_bf16_desc d) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[4 * HB_TILE];            // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, lh = lane >> 5;
    const int nbn = (d.N + HB_N - 1) / HB_N, nbm = (d.M + HB_M - 1) / HB_M;
    int block_id = blockIdx.x, slice = 0;
    const int tiles = nbm * nbn;
    int kt_begin = 0, kt_end = d.K / HB_K;
    if (d.split_k > 1) {
        slice = block_id / tiles; block_id -= slice * tiles;
        const int per = (kt_end + d.split_k - 1) / d.split_k;
        kt_begin = slice * per;
        kt_end = kt_begin + per < kt_end ? kt_begin + per : kt_end;
    }
    const int bn = block_id % nbn, bm = block_id / nbn;
    const int m0 = bm * HB_M, n0 = bn * HB_N;

    // staging map: each wave fills 32 rows of A and of B per tile = 4 + 4 wave instructions of 8 rows x 128 B
    const int rl = lane >> 3, pos = lane & 7;
    const unsigned short* ga[4];
    const unsigned short* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wid * 32 + i * 8 + rl;
        const int ar = m0 + r, wrw = n0 + r;
        const int chunk = pos ^ (r & 7);
        ga[i] = d.A + (int64_t)(ar < d.M ? ar : 0) * d.lda + 8 * chunk;
        gw[i] = d.W + (int64_t)(wrw < d.N ? wrw : 0) * d.ldw + 8 * chunk;
    }
    auto stage = [&](int buf, int kt) {
        unsigned short* a_s = smem + buf * HB_TILE + (wid * 32) * HB_K;
        unsigned short* b_s = smem + 2 * HB_TILE + buf * HB_TILE + (wid * 32) * HB_K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kt * HB_K),
                                             (__attribute__((address_space(3))) void*)(a_s + i * 8 * HB_K), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[i] + kt * HB_K),
                                             (__attribute__((address_space(3))) void*)(b_s + i * 8 * HB_K), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int sw = li & 7;                       // (row & 7) of the rows this lane reads (tile rows are multiples of 32 + li)
    if (kt_begin < kt_end) {
        stage(kt_begin & 1, kt_begin);
        __syncthreads();
    }
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = kt & 1;
        const unsigned short* a_s = smem + cur * HB_TILE + (wr * 64 + li) * HB_K;
        const unsigned short* b_s = smem + 2 * HB_TILE + cur * HB_TILE + (wc * 64 + li) * HB_K;
        bf16x8 a[4][2], b[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = 8 * ((2 * ks + lh) ^ sw);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[ks][i] = *reinterpret_cast<const bf16x8*>(a_s + i * 32 * HB_K + co);
                b[ks][i] = *reinterpret_cast<const bf16x8*>(b_s + i * 32 * HB_K + co);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < kt_end) stage(cur ^ 1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][mi], b[ks][ni], acc[mi][ni], 0, 0, 0);
        __syncthreads();
    }

    // epilogue: transpose through LDS (fp32), then row-wise 4 columns per lane
    constexpr int EST_LD = 68;
    float* est = reinterpret_cast<float*>(smem) + wid * (32 * EST_LD);
    const bool vec_ok = d.split_k <= 1 && (d.ldc & 3) == 0 && skg_aligned16_dev(d.C) && skg_aligned16_dev(d.bias);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                est[((r & 3) + 8 * (r >> 2) + 4 * lh) * EST_LD + ni * 32 + li] = acc[mi][ni][r];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * 64 + lane;
            const int rw = idx >> 4, c4 = (idx & 15) * 4;
            const int row = m0 + wr * 64 + mi * 32 + rw;
            const int col = n0 + wc * 64 + c4;
            const float4 a4 = *reinterpret_cast<const float4*>(est + rw * EST_LD + c4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            if (row >= d.M || col >= d.N) continue;
            if (d.split_k > 1) {
                float* wsp = d.split_ws + ((int64_t)slice * d.M + row) * d.N + col;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (col + c < d.N) wsp[c] = v[c];
                continue;
            }
            if (vec_ok && col + 3 < d.N) {                            // 16-byte (fp32) / 8-byte (bf16) row segments
                if (d.bias) {
                    const float4 b4 = *reinterpret_cast<const float4*>(d.bias + col);
                    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
                }
                if (d.relu) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
                }
                if (d.out_bf16) {
                    const uint32_t lo = (uint32_t)skg_f2bf(v[0]) | ((uint32_t)skg_f2bf(v[1]) << 16);
                    const uint32_t hi = (uint32_t)skg_f2bf(v[2]) | ((uint32_t)skg_f2bf(v[3]) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(d.C) + (int64_t)row * d.ldc + col) = make_uint2(lo, hi);
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.C) + (int64_t)row * d.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                }
                continue;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (col + c >= d.N) continue;
                float x = v[c] + (d.bias ? d.bias[col + c] : 0.f);
                if (d.relu) x = fmaxf(x, 0.f);
                if (d.out_bf16) reinterpret_cast<unsigned short*>(d.C)[(int64_t)row * d.ldc + col + c] = skg_f2bf(x);
                else reinterpret_cast<float*>(d.C)[(int64_t)row * d.ldc + col + c] = x;
            }
        }
    }
}

__global__ __launch_bounds__(256) void skg_bf16_splitk_reduce_kernel(const skg_gemm_bf16_desc d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)d.M * d.N;
    if (i >= total) return;
    const int row = (int)(i / d.N), col = (int)(i % d.N);
    float v = 0.f;
    for (int s = 0; s < d.split_k; ++s) v += d.split_ws[(int64_t)s * total + i];
    if (d.bias) v += d.bias[col];
    if (d.relu) v = fmaxf(v, 0.f);
    if (d.out_bf16) reinterpret_cast<unsigned short*>(d.C)[(int64_t)row * d.ldc + col] = skg_f2bf(v);
    else reinterpret_cast<float*>(d.C)[(int64_t)row * d.ldc + col] = v;
}

extern "C" int skg_gemm_bf16(const skg_gemm_bf16_desc* dh, void* stream) {
    if (!dh) return SKG_E_ARG;
    const skg_gemm_bf16_desc d = *dh;
    if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.W || !d.C) return SKG_E_ARG;
    if (d.M == 0) return 0;
    if ((d.K % HB_K) || (d.lda & 7) || (d.ldw & 7)) return SKG_E_ALIGN;
    if (!skg_aligned16(d.A) || !skg_aligned16(d.W)) return SKG_E_ALIGN;
    if (d.split_k > 1 && (!d.split_ws || d.split_k > 64)) return SKG_E_ARG;
    const int64_t tiles = (int64_t)((d.M + HB_M - 1) / HB_M) * ((d.N + HB_N - 1) / HB_N);
    const int64_t nblk = tiles * (d.split_k > 1 ? d.split_k : 1);
    if (nblk > 0x7fffffffLL) return SKG_E_LIMIT;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(skg_gemm_bf16_kernel, dim3((unsigned)nblk), dim3(256), 0, s, d);
    if (d.split_k > 1) {
        const int64_t total = (int64_t)d.M * d.N;
        hipLaunchKernelGGL(skg_bf16_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
    }
    return skg_launch_status();
}

// out[c, r] = in[r, c] for 16-bit elements (bf16 operands of the backward products)
__global__ __launch_bounds__(256) void skg_transpose16_kernel(const unsigned short* __restrict__ in, int64_t ld_in,
                                                              int rows, int cols, unsigned short* __restrict__ out,
                                                              int64_t ld_out) {
    __shared__ unsigned short tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(int64_t)r * ld_in + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) out[(int64_t)c * ld_out + r] = tile[tx][i];
    }
}

extern "C" int skg_transpose_bf16(const void* in, int64_t ld_in, int rows, int cols, void* out, int64_t ld_out,
                                  void* stream) {
    if (rows < 0 || cols < 0) return SKG_E_ARG;
    if (rows == 0 || cols == 0) return 0;
    if (!in || !out || ld_in < cols || ld_out < rows) return SKG_E_ARG;
    hipLaunchKernelGGL(skg_transpose16_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0,
                       (hipStream_t)stream, (const unsigned short*)in, ld_in, rows, cols, (unsigned short*)out, ld_out);
    return skg_launch_status();
}
