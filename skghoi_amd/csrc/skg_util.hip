// skg_util.hip -- small support kernels of the interaction-head hot path (gfx950).
//
//   skg_param_checksum : order-independent 64-bit checksum over a table of fp32 parameter chunks.  The host engine packs
//                        the head's 408 parameters into kernel-friendly layouts once (skghoi_amd/engine.py,
//                        PackedWeights) and must notice ANY later change of the live parameters -- including writes
//                        that bypass autograd's version counters (`p.data.mul_(2)`).  The reference reads its
//                        nn.Linear weights afresh in every forward (heads/adamixer_transH_spatial_r50_head.py:812-973),
//                        so a stale packed copy would be a silent parity bug.  HBM-bound: 118 MB read once
//                        (the bytes of the parameters), one 16-byte load per lane and step.
#include "skg_common.h"

#define CK_THREADS 256

__device__ __forceinline__ unsigned long long skg_ck_mix(uint32_t bits, uint32_t gidx) {
    // position-dependent so that permuted or shifted contents change the sum; odd multiplier keeps it a bijection per slot
    return (unsigned long long)(bits + 0x9E3779B9u * (gidx + 1u)) * (unsigned long long)(2u * gidx + 1u);
}

__global__ __launch_bounds__(CK_THREADS) void skg_param_checksum_kernel(const skg_param_chunk* __restrict__ chunks,
                                                                       unsigned long long* __restrict__ out) {
    const skg_param_chunk c = chunks[blockIdx.x];
    const uint32_t* __restrict__ p = reinterpret_cast<const uint32_t*>(c.ptr);
    const uint32_t n = c.count, g0 = c.first;
    unsigned long long acc = 0;
    const uint32_t n4 = n >> 2;
    const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(p);
    for (uint32_t i = threadIdx.x; i < n4; i += CK_THREADS) {
        const uint4 v = p4[i];
        const uint32_t g = g0 + 4u * i;
        acc += skg_ck_mix(v.x, g) + skg_ck_mix(v.y, g + 1u) + skg_ck_mix(v.z, g + 2u) + skg_ck_mix(v.w, g + 3u);
    }
    for (uint32_t i = 4u * n4 + threadIdx.x; i < n; i += CK_THREADS) acc += skg_ck_mix(p[i], g0 + i);
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
    // wave reduction of a 64-bit sum through two 32-bit lanes' worth of shuffles
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, off, 64) << 32) |
                                     (uint32_t)__shfl_xor((int)lo, off, 64);
        acc += o;
        lo = (uint32_t)acc; hi = (uint32_t)(acc >> 32);
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

extern "C" int skg_param_checksum(const skg_param_chunk* chunks, int n_chunks, uint64_t* out, void* stream) {
    if (n_chunks < 0 || !out || (n_chunks > 0 && !chunks)) return SKG_E_ARG;
    if ((((uintptr_t)out) & 7u) != 0) return SKG_E_ALIGN;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(skg_param_checksum_kernel, dim3(n_chunks), dim3(CK_THREADS), 0, (hipStream_t)stream, chunks,
                       reinterpret_cast<unsigned long long*>(out));
    return skg_launch_status();
}
