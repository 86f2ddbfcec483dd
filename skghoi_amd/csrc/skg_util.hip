// skg_util.hip -- small support kernels of the interaction-head hot path (gfx950).
//
//   skg_param_checksum : order-independent 64-bit checksum over a table of fp32 parameter chunks.  The host engine packs
//                        the head's 408 parameters into kernel-friendly layouts once (skghoi_amd/engine.py,
//                        PackedWeights) and must notice ANY later change of the live parameters -- including writes
//                        that bypass autograd's version counters (`p.data.mul_(2)`).  The reference reads its
//                        nn.Linear weights afresh in every forward (heads/adamixer_transH_spatial_r50_head.py:812-973),
//                        so a stale packed copy would be a silent parity bug.  HBM-bound: 118 MB read once
//                        (the bytes of the parameters), one 16-byte load per lane and step.
//   skg_adamw          : AdamW over every parameter of the training step in one launch (chunk table).
#include "skg_common.h"

#define CK_THREADS 256

__device__ __forceinline__ unsigned long long skg_ck_mix(uint32_t bits, uint32_t gidx) {
    // position-dependent so that permuted or shifted contents change the sum; the odd 32-bit multiplier is a bijection per
    // slot (one v_mul_lo_u32 + one v_mul_hi_u32)
    return (unsigned long long)(bits + 0x9E3779B9u * (gidx + 1u)) * (unsigned long long)(2u * gidx + 1u);
}

// Grid-stride over the chunk table: a fixed, small grid (SKG_CHECKSUM_PARTIALS blocks) so that the result is a short
// array of per-block partial sums the host adds up -- no atomics (thousands of same-address atomics serialise in L2: the
// first version of this kernel spent 100 us on them), no memset, and the order of the additions is fixed.
__global__ __launch_bounds__(CK_THREADS) void skg_param_checksum_kernel(const skg_param_chunk* __restrict__ chunks,
                                                                       int n_chunks,
                                                                       unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sred[CK_THREADS / 64];
    unsigned long long acc = 0;
    for (int ci = blockIdx.x; ci < n_chunks; ci += gridDim.x) {
        const skg_param_chunk c = chunks[ci];
        const uint32_t* __restrict__ p = reinterpret_cast<const uint32_t*>(c.ptr);
        const uint32_t n = c.count, g0 = c.first;
        const uint32_t n4 = n >> 2;
        const uint4* __restrict__ p4 = reinterpret_cast<const uint4*>(p);
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < n4; i += CK_THREADS) {
            const uint4 v = p4[i];
            const uint32_t g = g0 + 4u * i;
            acc += skg_ck_mix(v.x, g) + skg_ck_mix(v.y, g + 1u) + skg_ck_mix(v.z, g + 2u) + skg_ck_mix(v.w, g + 3u);
        }
        for (uint32_t i = 4u * n4 + threadIdx.x; i < n; i += CK_THREADS) acc += skg_ck_mix(p[i], g0 + i);
    }
    uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)hi, off, 64) << 32) |
                                     (uint32_t)__shfl_xor((int)lo, off, 64);
        acc += o;
        lo = (uint32_t)acc; hi = (uint32_t)(acc >> 32);
    }
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (sred[0] + sred[1]) + (sred[2] + sred[3]);
}

extern "C" int skg_param_checksum(const skg_param_chunk* chunks, int n_chunks, uint64_t* out, void* stream) {
    if (n_chunks < 0 || !out || (n_chunks > 0 && !chunks)) return SKG_E_ARG;
    if ((((uintptr_t)out) & 7u) != 0) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_param_checksum_kernel, dim3(SKG_CHECKSUM_PARTIALS), dim3(CK_THREADS), 0, (hipStream_t)stream,
                       chunks, n_chunks, reinterpret_cast<unsigned long long*>(out));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ AdamW, one launch
// One workgroup per chunk (<= SKG_ADAMW_CHUNK elements of one tensor).  16-byte accesses when the four pointers allow it
// (uniform per workgroup), scalar otherwise (a bias slice that starts at an odd element).
__device__ __forceinline__ void skg_adamw_one(float& p, float g, float& m, float& v, float decay, float c1, float beta2,
                                              float c2, float step_size, float inv_sqrt_b2, float eps) {
    p *= decay;
    m += (g - m) * c1;
    v = beta2 * v + c2 * g * g;
    p -= step_size * (m / (sqrtf(v) * inv_sqrt_b2 + eps));
}

__global__ __launch_bounds__(256) void skg_adamw_kernel(const skg_adamw_chunk* __restrict__ chunks, float decay, float c1,
                                                        float beta2, float c2, float step_size, float inv_sqrt_b2,
                                                        float eps, float* __restrict__ steps, int n_steps) {
    if (blockIdx.x == 0)                         // the optimizer state's per-parameter step counters (torch keeps them as tensors)
        for (int i = threadIdx.x; i < n_steps; i += 256) steps[i] += 1.f;
    const skg_adamw_chunk c = chunks[blockIdx.x];
    const uint32_t n = c.count;
    const bool vec = ((((uintptr_t)c.p) | ((uintptr_t)c.g) | ((uintptr_t)c.m) | ((uintptr_t)c.v)) & 15u) == 0;
    uint32_t done = 0;
    if (vec) {
        const uint32_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(c.p); const float4* g4 = reinterpret_cast<const float4*>(c.g);
        float4* m4 = reinterpret_cast<float4*>(c.m); float4* v4 = reinterpret_cast<float4*>(c.v);
        for (uint32_t i = threadIdx.x; i < n4; i += 256) {
            float4 p = p4[i], m = m4[i], v = v4[i];
            const float4 g = g4[i];
            skg_adamw_one(p.x, g.x, m.x, v.x, decay, c1, beta2, c2, step_size, inv_sqrt_b2, eps);
            skg_adamw_one(p.y, g.y, m.y, v.y, decay, c1, beta2, c2, step_size, inv_sqrt_b2, eps);
            skg_adamw_one(p.z, g.z, m.z, v.z, decay, c1, beta2, c2, step_size, inv_sqrt_b2, eps);
            skg_adamw_one(p.w, g.w, m.w, v.w, decay, c1, beta2, c2, step_size, inv_sqrt_b2, eps);
            p4[i] = p; m4[i] = m; v4[i] = v;
        }
        done = 4u * n4;
    }
    for (uint32_t i = done + threadIdx.x; i < n; i += 256) {
        float p = c.p[i], m = c.m[i], v = c.v[i];
        skg_adamw_one(p, c.g[i], m, v, decay, c1, beta2, c2, step_size, inv_sqrt_b2, eps);
        c.p[i] = p; c.m[i] = m; c.v[i] = v;
    }
}

// The scalar factors are formed in double on the host and rounded once (1 - beta2 in float arithmetic is 1.3e-5 off).
extern "C" int skg_adamw_f32(const skg_adamw_chunk* chunks, int n_chunks, double lr, double beta1, double beta2, double eps,
                             double weight_decay, double bias1, double bias2, float* steps, int n_steps, void* stream) {
    if (n_chunks < 0 || (n_chunks > 0 && !chunks) || n_steps < 0 || (n_steps > 0 && !steps)) return SKG_E_ARG;
    if (!(bias1 > 0.0) || !(bias2 > 0.0) || !(eps >= 0.0)) return SKG_E_ARG;
    if (n_chunks == 0) return 0;
    hipLaunchKernelGGL(skg_adamw_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, chunks,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)(lr / bias1), (float)(1.0 / sqrt(bias2)), (float)eps, steps, n_steps);
    return skg_launch_status();
}

// a slice of the chunk table with the same factors (the optimizer inside the backward, skg_exchange.adamw)
int skg_adamw_slice(const skg_adamw_chunk* chunks, int first, int last, const skg_exchange& x, bool with_steps, hipStream_t stream) {
    if (last <= first) return 0;
    hipLaunchKernelGGL(skg_adamw_kernel, dim3((unsigned)(last - first)), dim3(256), 0, stream, chunks + first,
                       (float)(1.0 - x.lr * x.weight_decay), (float)(1.0 - x.beta1), (float)x.beta2, (float)(1.0 - x.beta2),
                       (float)(x.lr / x.bias1), (float)(1.0 / sqrt(x.bias2)), (float)x.eps,
                       with_steps ? x.adamw_steps : (float*)nullptr, with_steps ? x.adamw_n_steps : 0);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ bf16 twins
thread_local skg_twin_map skg_tls_twin = {{nullptr, nullptr}, {nullptr, nullptr}, {0, 0}};
thread_local const skg_tuning* skg_tls_tuning = nullptr;     // the calling thread's current context's switches (skg_ctx_make_current)

// dst[i] = bf16(src[i]) (round to nearest even): the twin of a whole fp32 buffer in one pass (the parameter arena of the
// bf16 training step: 118 MB read, 59 MB written, ~30 us).
__global__ __launch_bounds__(256) void skg_twin_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        skg_store_twin4(dst + 4 * i, reinterpret_cast<const float4*>(src)[i]);
}

extern "C" int skg_twin_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
    if (n < 0 || (n & 3)) return SKG_E_ARG;
    if (n == 0) return 0;
    if (!src || !dst) return SKG_E_ARG;
    if (!skg_aligned16(src) || (((uintptr_t)dst) & 7u)) return SKG_E_ALIGN;
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(skg_twin_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n4);
    return skg_launch_status();
}
