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
