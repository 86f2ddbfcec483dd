// Internal helpers shared by the HIP translation units of libskghoi_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "skghoi.h"

#define SKG_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int skg_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static inline bool skg_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
__device__ __forceinline__ bool skg_aligned16_dev(const void* p) { return (((uintptr_t)p) & 15u) == 0; }   // null counts as aligned

__device__ __forceinline__ float skg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float skg_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Block-wide sum for 256-thread blocks (4 waves); `red` is a 4-float LDS scratch.  All threads get the result.
__device__ __forceinline__ float skg_block_sum256(float v, float* red) {
    v = skg_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- eval-path epilogues on the free-layout GEMM (skg_gemm_x.hip), used when skg_gemm.hip routes a mid-size exact-fp32
// launch there (a few images: 100-400 tiles of 128 x 128, where the register-pipelined 128 x 128 loop beats both eval loops).
// Semantics of include/skghoi.h skg_gemm_desc: kind = SKG_EPI_MUL_RELU (C[out_rows[r]] = relu((acc + bias) * (mbias +
// P[p_idx[r]] + Q[q_idx[r]])), optional raw copy of acc + bias to C_raw[r]) or SKG_EPI_BIAS_RES_RELU (C = relu(acc + bias)
// + res[r]); kind 0 with out_rows = a plain product whose rows are scattered.  Only with the staged epilogue (16-byte
// aligned everything, N % 4 == 0) and without split-K.
struct skg_gemmx_fused {
    int kind;
    const float* P; const int32_t* p_idx; int64_t ldp;
    const float* Q; const int32_t* q_idx; int64_t ldq;
    const float* mbias;
    float* C_raw; int64_t ldc_raw;
    const int32_t* out_rows;
    const float* res; int64_t ldres;
};
// 1 if the product (with these fused parameters, NULL = none) can take skg_gemmx_f32's staged epilogue
int skg_gemmx_can_fuse(const skg_gemmx_desc* d, const skg_gemmx_fused* f);
int skg_gemmx_f32_fused(const skg_gemmx_desc* descs_host, const skg_gemmx_fused* fused_host, int n, void* stream);
