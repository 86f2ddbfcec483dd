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

// ---- tuning of the calling thread's CURRENT context (include/skghoi.h: skg_tuning, skg_ctx_make_current).  The library
// itself keeps no process-wide tuning state: every switch below lives in a context the caller created and is read through
// this thread-local pointer (NULL: the built-in defaults, constants).
extern thread_local const skg_tuning* skg_tls_tuning;
static inline int skg_tune_small_mode() { const skg_tuning* t = skg_tls_tuning; return t && t->small_mode > 0 ? t->small_mode : 3; }
static inline int skg_tune_small_tiles() { const skg_tuning* t = skg_tls_tuning; return t && t->small_tiles > 0 ? t->small_tiles : 384; }
static inline int skg_tune_route_tiles() { const skg_tuning* t = skg_tls_tuning; return t && t->route_tiles > 0 ? t->route_tiles : 200; }
static inline int skg_tune_khalves_blocks() { const skg_tuning* t = skg_tls_tuning; return t && t->khalves_blocks > 0 ? t->khalves_blocks : 320; }

// ---- bf16 twins of fp32 tensors (skg_gemmx_desc.A16 / B16: what the direct-to-LDS GEMM of the bf16 training step reads).
// The training plan (skg_train_plan.hip) keeps, beside its fp32 workspace, a twin workspace with the SAME element indexing
// and announces both for the duration of one of its calls in this thread-local map; the per-row kernels it launches look
// their outputs up and, where an output lies inside a mapped range, also store its bf16 rounding (round to nearest even) --
// 2 more bytes per element written, instead of the 4 the GEMM would read later.  The map is empty outside a plan call: the
// public entry points behave exactly as before for every other caller.
struct skg_twin_map { const float* base[2]; uint16_t* base16[2]; int64_t n[2]; };
extern thread_local skg_twin_map skg_tls_twin;
static inline uint16_t* skg_twin(const float* p) {
    const skg_twin_map& m = skg_tls_twin;
    for (int i = 0; i < 2; ++i)
        if (m.base16[i] && p >= m.base[i] && p < m.base[i] + m.n[i]) return m.base16[i] + (p - m.base[i]);
    return nullptr;
}
typedef __bf16 skg_bf2 __attribute__((ext_vector_type(2)));
typedef float skg_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t skg_pack_bf16(float a, float b) {          // two floats -> two bf16, round to nearest even
    const skg_f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, skg_bf2));
}
__device__ __forceinline__ void skg_store_twin4(uint16_t* p, const float4 v) {   // p 8-byte aligned
    *reinterpret_cast<uint2*>(p) = make_uint2(skg_pack_bf16(v.x, v.y), skg_pack_bf16(v.z, v.w));
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

// ---- several independent launches of one per-row kernel as ONE launch (blockIdx.y picks the call).  A dependent chain of
// ~100 kernels pays ~4.5 us per kernel boundary whatever the kernel does (a 1-wave kernel takes 4.7 us here): the training
// plan (skg_train_plan.hip) issues its back-to-back calls of these kernels through the *_multi forms.  Same arithmetic,
// same results; the calls of one launch must not depend on each other.  n <= SKG_MULTI_MAX.
#define SKG_MULTI_MAX 4
struct skg_rows_mul_args {
    const float* P; const int32_t* p_idx; int64_t ldp; const float* Q; const int32_t* q_idx; int64_t ldq; const float* mbias;
    const float* F; const int32_t* f_idx; int64_t ldf; int rows, cols; float* out; int64_t ldo;
    uint16_t* out16;             // filled by the launcher from the twin map (callers leave it alone)
};
int skg_rows_mul_relu_multi(const skg_rows_mul_args* calls, int n, void* stream);
struct skg_mul_bwd_args {
    float* g; int64_t ldg; const float* F; const int32_t* f_idx; int64_t ldf; const float* P; const int32_t* p_idx; int64_t ldp;
    const float* Q; const int32_t* q_idx; int64_t ldq; const float* mbias; int rows; float* dF; int64_t lddf; int accumulate;
    uint16_t* dF16;              // filled by the launcher from the twin map
};
int skg_mul_bwd_multi(const skg_mul_bwd_args* calls, int n, void* stream);
struct skg_segment_sum_args {           // modes 0 / 1 of skg_segment_sum_f32 (mode 2 has a grid of its own)
    const float* src; int64_t ld; int mode; float* outH; float* outN; int accumulate;
    uint16_t* outH16; uint16_t* outN16;      // filled by the launcher from the twin map
};
int skg_segment_sum_multi(const skg_segment_sum_args* calls, int n, const skg_image_meta* meta, const int32_t* hum_img,
                          const int32_t* node_img, int sum_h, int sum_n, void* stream);
struct skg_add_layernorm_args {
    const float* a; int64_t lda; const float* b; int64_t ldb; const float* gamma; const float* beta; int rows;
    float* xsum; float* y; float* stats;
    uint16_t* y16;               // filled by the launcher from the twin map
};
int skg_add_layernorm_multi(const skg_add_layernorm_args* calls, int n, float eps, void* stream);
struct skg_layernorm_bwd_args {
    const float* dy; int64_t lddy; const float* x; const float* stats; const float* gamma; int rows; float* dx;
    const float* relu_src; float* dx_masked; float* dgamma; float* dbeta;
    uint16_t* dx16; uint16_t* dx_masked16;   // filled by the launcher from the twin map
};
int skg_layernorm_bwd_multi(const skg_layernorm_bwd_args* calls, int n, void* stream);

// ---- skg_comm.cpp: what the backward's worker thread calls between stages (data parallel).  skg_comm_chunk: the exchange
// stream waits for `after` (a recorded event, or NULL), then all-reduces p[0, n) in place (sum, fp32) there;
// skg_comm_close_step: `stream` ordered behind every collective issued so far.
int skg_comm_chunk(skg_comm* c, hipEvent_t after, float* p, int64_t n);
int skg_comm_close_step(skg_comm* c, hipStream_t stream);
int skg_comm_chunk_in_stream(skg_comm* c, hipStream_t stream, float* p, int64_t n);   // a step's LAST chunk, on the step's stream
hipStream_t skg_comm_stream(skg_comm* c);
// skg_util.hip: entries [first, last) of an AdamW chunk table with the factors carried by an skg_exchange
int skg_adamw_slice(const skg_adamw_chunk* chunks, int first, int last, const skg_exchange& x, bool with_steps, hipStream_t stream);
