// skg_gemm_x.hip -- MFMA GEMM with free operand layouts, for the TRAINING step of the interaction head.
//
//   C(m, n) (+)= epilogue( sum_k A(m, k) * B(k, n) )
//
// The backward pass of every dense layer of the head (reference: autograd of nn.Linear / the 16-branch MultiBranchFusion,
// heads/adamixer_transH_spatial_r50_head.py:469-474, 509-527, 635-701) needs products whose operands are NOT both
// "row = output index, k contiguous" like the forward's:
//     dX = dZ  W        A = dZ [rows, out] (k contiguous),   B(k, n) = W[k][n]      (n contiguous)
//     dW = dZ^T X       A(m, k) = dZ[k][m] (m contiguous),   B(k, n) = X[k][n]      (n contiguous)
// Transposing operands first costs three extra kernels and three extra HBM round trips per layer.  Here either operand
// may be contiguous along k or along its own index, the 16 MBF branch weights may stay in their branch-major storage
// ([16][1024][64] blocks along k or n), and the epilogue fuses what the surrounding autograd would launch separately:
// bias, ReLU, the ReLU mask of the producing layer (dZ_prev = dX * (Y_prev > 0)), gradient accumulation, and the bias
// gradient as row sums of the A operand (db = dZ^T 1).  Up to SKG_GEMMX_GROUP_MAX independent products share one
// launch (dX and dW of a layer; the node-row GEMMs of the graph), with split-K for long contractions and small outputs.
//
// Two kernels, same descriptors and epilogue:
//   skg_gemmx_kernel       exact fp32: v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain), 128x128x16 tiles
//   skg_gemmx_bf16_kernel  operands rounded to bf16 on their way into LDS, fp32 accumulation on
//                          v_mfma_f32_32x32x16_bf16, 128x128x32 tiles (precision="bf16" training)
// Both: 4 waves (2x2), each 64x64 = 2x2 MFMA tiles; double-buffered LDS, register prefetch of the next tile across
// the MFMA loop, one barrier per k-tile.  The operand layouts are COMPILE-TIME cases of the main loop (the workgroup
// picks its case once, uniformly), and the loop over the k-tiles that lie fully inside the slice has no bounds checks,
// no branches and one address addition per load -- all loads of a tile are in flight together (a loop that decides
// per quad serialises them behind s_waitcnt and runs at a fraction of the speed).  A ragged last tile, operands
// without 16-byte alignment and row-contiguous operands whose extent is not a multiple of 4 take the generic loop.
#include "skg_common.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <atomic>

#define XBM 128
#define XBN 128
#define XBK 16
#define XLD 130
#define XTILE (XBK * XLD)
#ifndef SKG_XXCD
#define SKG_XXCD 1
#endif

struct skg_gemmx_group {
    skg_gemmx_desc d[SKG_GEMMX_GROUP_MAX];
    int start[SKG_GEMMX_GROUP_MAX + 1];      // block ranges
    skg_gemmx_fused f[SKG_GEMMX_GROUP_MAX];  // eval-path epilogues (kind 0: none), staged epilogue only
    int vec[SKG_GEMMX_GROUP_MAX];            // bit 0: A fast loop allowed, bit 1: B, bit 2: C 8-byte stores, bit 3: staged
                                             // epilogue, bit 4 / 5: bf16 twin of A / B readable by the fast loop
    int n;
};

__device__ __forceinline__ int64_t xoff(int idx, int shift, int64_t bstride, int64_t estride) {
    // index -> element offset with optional power-of-two blocking: (idx >> shift) * bstride + (idx & mask) * estride
    if (shift <= 0) return (int64_t)idx * estride;
    return (int64_t)(idx >> shift) * bstride + (int64_t)(idx & ((1 << shift) - 1)) * estride;
}

struct XOperand {
    const float* base;
    int64_t s_row, s_k;          // element strides along the operand's own index / along k (one of them is 1)
    int rshift, kshift;          // power-of-two blocking of either index (0 = none)
    int64_t rstride, kstride;
    int rows;                    // extent of the own index (M or N)
    bool vec;                    // 16-byte loads allowed
    __amdgpu_buffer_rsrc_t rsrc; // buffer descriptor over the operand (fast loop)
    __amdgpu_buffer_rsrc_t rsrc16;   // ... over its bf16 twin (skg_gemmx_bf16 only)
    bool vec16;                  // the twin exists and may be read with 16-byte loads
    int64_t s_row16, s_k16;      // the TWIN's strides (skg_gemmx_desc.a16_ld / b16_ld; normally those of the fp32 array)
};

struct XQ2 { float4 a, b; };

// Fast tiles address their loads as  (uniform byte offset of the k-step) + (loop-invariant 32-bit byte offset of the
// lane) through a buffer descriptor of the operand (buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen): the step offset
// advances on the scalar unit, so a k-step spends no vector instruction on addresses.  The first version formed a 64-bit address per load per step and zeroed
// clamped rows with 32 selects per step: ~110 vector instructions beside 8 MFMAs, and a lone wave per SIMD issues them
// one after the other (1200 cycles per step for 256 cycles of MFMA, measured from a K sweep on captured launches).
// Rows outside the operand are clamped to a valid row and NOT zeroed: they only feed output rows / columns >= M / N,
// which the epilogue never stores (nor the bias gradient of rows >= M).
struct XFast { uint32_t o0, o1, o2, o3; };       // (scalars: an array member ended up in scratch memory)
#ifndef SKG_XBUF
#define SKG_XBUF 1
#endif
__device__ __forceinline__ float4 xldo(const XOperand& op, uint32_t soff, uint32_t voff) {
#if SKG_XBUF
    typedef unsigned int xu4 __attribute__((__vector_size__(16)));
    const xu4 r = __builtin_amdgcn_raw_buffer_load_b128(op.rsrc, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
#else
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(op.base) + soff + voff);
#endif
}
// uniform byte offset of the k-step starting at k0 (k0 a multiple of the k-tile; k blocks, when present, are multiples of it)
template <bool KC>
__device__ __forceinline__ uint32_t xstepbase(const XOperand& op, int k0) {
    return (uint32_t)(xoff(k0, op.kshift, op.kstride, KC ? 1 : op.s_k) * 4);
}

__device__ __forceinline__ float4 xzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 xld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Four consecutive elements along the contiguous index c (extent cend); zero outside.
__device__ __forceinline__ float4 xquad(const float* p, int c, int cend, bool vec) {
    float4 t = xzero4();
    if (vec && c + 3 < cend) {
        t = xld4(p);
    } else {
        if (c < cend) t.x = p[0];
        if (c + 1 < cend) t.y = p[1];
        if (c + 2 < cend) t.z = p[2];
        if (c + 3 < cend) t.w = p[3];
    }
    return t;
}

// Generic quad: KC -- 4 consecutive k of `row`; otherwise 4 consecutive rows at `k`.  Every bound checked.
template <bool KC>
__device__ __forceinline__ float4 xgen(const XOperand& op, int row, int k, int kend) {
    if (row >= op.rows || k >= kend) return xzero4();
    if (KC)
        return xquad(op.base + xoff(row, op.rshift, op.rstride, op.s_row) + xoff(k, op.kshift, op.kstride, 1), k, kend,
                     op.vec);
    return xquad(op.base + xoff(k, op.kshift, op.kstride, op.s_k) + xoff(row, op.rshift, op.rstride, 1), row, op.rows,
                 op.vec);
}

// An operand may take the fast loop from row0 when 16-byte loads are allowed and no row-contiguous quad straddles its end.
template <bool KC>
__device__ __forceinline__ bool xfast_ok(const XOperand& op, int row0) {
    return op.vec && (KC || (op.rows & 3) == 0 || row0 + 128 <= op.rows);
}

// ================================================================================================ block -> tile
// Workgroups are dealt round-robin over the 8 XCDs, each with a private 4 MiB L2.  Walking N fastest (the first version)
// gave XCD x the column tiles tn = x mod 8 of EVERY row panel: all eight XCDs fetched the whole A operand.  Here the
// product's blocks are renumbered so that an XCD owns one contiguous range of a tile order made of column GROUPS: g
// column tiles whose B slice (g x 128 x K fp32) fits in ~2 MiB, all row panels of the group, the g tiles and the split
// slices of one panel adjacent.  An XCD then keeps its B slice L2-resident and an A panel is fetched by the XCDs of
// nbn / g groups instead of 8 (M = 102400, N = K = 1024, bf16: 569 -> 502 us).  The renumbering is a bijection for every
// block count (the remainder is spread over the first XCDs).
struct XTileId { int tm, tn, slice; };
__device__ __forceinline__ XTileId xtile_of(int b, int nb, int nbm, int nbn, int S, int K) {
#if SKG_XXCD
    {
        const int q = nb >> 3, r = nb & 7, x = b & 7;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    int g = (int)((2 << 20) / ((int64_t)XBN * 4 * max(K, 1)));
    g = max(1, min(g, nbn));
    const int per_group = nbm * g * S;
    const int grp = min(b / per_group, (nbn - 1) / g);
    const int w = min(g, nbn - grp * g);                  // the last group may be narrower
    const int rr = b - grp * per_group;
    XTileId t;
    t.slice = rr % S;
    t.tn = grp * g + (rr / S) % w;
    t.tm = rr / (S * w);
    return t;
#else
    XTileId t;
    t.slice = b % S; b /= S;
    t.tn = b % nbn; t.tm = b / nbn;                       // consecutive blocks walk N
    return t;
#endif
}

typedef __bf16 xbf2 __attribute__((ext_vector_type(2)));
typedef float xf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t ypack(float a, float b) {          // two floats -> two bf16 (round to nearest even)
    const xf2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xbf2));
}

// ================================================================================================ epilogue
// The accumulators of a wave (64 x 64 outputs) leave through LDS, 32 rows at a time: the MFMA layout gives a lane ONE
// element per row (32 lanes = 128 bytes of a row), and 64 dword stores per thread kept the store unit busy for ~20 us
// on a 3200 x 1024 output -- half of the whole product (measured with the k loop / the epilogue compiled out).  From
// the staged image a lane reads four consecutive columns: 16 dwordx4 stores per thread, 256 contiguous bytes per row,
// and bias / accumulate / mask travel as 16-byte loads as well.  Each wave stages in its own XEP_WAVE floats (no
// barrier: the LDS serves one wave's instructions in order).  Needs N % 4 == 0 and 16-byte aligned C / bias / mask /
// workspace rows (bit 3 of the product's `vec`); anything else takes the element-wise epilogue.
#define XEP_LD 72                            // floats per staged row: 64 + 8 (rows r and r + 4 land 32 banks apart)
#define XEP_WAVE (32 * XEP_LD)
#define XEP_FLOATS (4 * XEP_WAVE)            // 36 KiB per workgroup

// ---- split-K reduced INSIDE the product launch (skg_gemmx_desc.split_ctr).  Every slice's workgroup stores its partial tile
// (and partial row sums) to split_ws as before, then ARRIVES at the tile's counter: atomicInc wraps at S - 1, so the workgroup
// that reads S - 1 is the last one and the counter is zero again for the next launch -- no reset pass, no second kernel.  The
// last arriver adds the slices IN SLICE ORDER (its own from its registers at its position in that order: the same additions
// in the same order as skg_gemmx_reduce_kernel, whichever slice happens to finish last -- deterministic, no float atomics) and
// applies the epilogue.
// Partials cross XCDs, whose L2s are private: they are stored WRITE-THROUGH (sc1, 16 bytes per lane) and read back with sc1
// loads only -- the counter form of the inter-workgroup hand-off (cdna_hip_programming.md, Guideline 16 R1 / in-launch
// split-K): every storing wave drains its stores, the workgroup meets, ONE lane adds to the counter, and the workgroup whose
// add came last loads behind a barrier that lane joins.  The first version stored plainly and released with an agent-scope
// fence (buffer_wbl2: the whole XCD's dirty L2 written back by each of ~20 workgroups per XCD): +40 us per launch, the batch-4
// bf16 step 1.28 -> 1.91 ms.
struct XRed { const float* ws; int64_t MN; int S, slice; };
typedef unsigned int xu4s __attribute__((__vector_size__(16)));
#define XSC1 16                              // aux bits of a buffer access: sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xrsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void xst_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4& v) {
    const xu4s u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)byte_off, 0, XSC1);
}
__device__ __forceinline__ float4 xld_sc1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    const xu4s u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, XSC1);
    return make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
}

// Arrival of one slice's workgroup at its tile's counter; true for the last one.  `flag`: a free word of the kernel's ONE
// LDS array (a second __shared__ object beside a DMA-staged ring makes hipcc drain the ring in front of every k-step's
// first ds_read).
__device__ __forceinline__ bool xsplit_last(uint32_t* ctr, int S, volatile uint32_t* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) *flag = atomicInc(ctr, (uint32_t)(S - 1)) == (uint32_t)(S - 1) ? 1u : 0u;
    __syncthreads();
    return *flag != 0u;
}

// Epilogue of four consecutive columns of one output row (16-byte accesses).
__device__ __forceinline__ void xep_apply(const skg_gemmx_desc& d, const skg_gemmx_fused& f, float4 v, int row, int col,
                                          int64_t coff, bool hb, const float4& bv, const float4& mb) {
    if (hb) { v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
    if (f.kind | (f.out_rows != nullptr)) {            // eval-path epilogues (skg_gemm_desc), uniform per product
        const int orow = f.out_rows ? f.out_rows[row] : row;
        if (f.kind == SKG_EPI_MUL_RELU) {
            if (f.C_raw) *reinterpret_cast<float4*>(f.C_raw + (int64_t)row * f.ldc_raw + col) = v;
            if (orow < 0) return;
            float4 m = mb;
            if (f.P) {
                const float4 t = xld4(f.P + (int64_t)(f.p_idx ? f.p_idx[row] : row) * f.ldp + col);
                m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
            }
            if (f.Q) {
                const float4 t = xld4(f.Q + (int64_t)(f.q_idx ? f.q_idx[row] : row) * f.ldq + col);
                m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
            }
            *reinterpret_cast<float4*>(d.C + coff + (int64_t)orow * d.ldc) =
                make_float4(fmaxf(v.x * m.x, 0.f), fmaxf(v.y * m.y, 0.f), fmaxf(v.z * m.z, 0.f), fmaxf(v.w * m.w, 0.f));
            return;
        }
        if (orow < 0) return;
        if (d.relu || f.kind == SKG_EPI_BIAS_RES_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (f.kind == SKG_EPI_BIAS_RES_RELU) {
            const float4 t = xld4(f.res + (int64_t)row * f.ldres + col);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        *reinterpret_cast<float4*>(d.C + coff + (int64_t)orow * d.ldc) = v;
        return;
    }
    if (d.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float* p = d.C + coff + (int64_t)row * d.ldc;
    if (d.accumulate) { const float4 o = xld4(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    if (d.mask) {                                      // accumulate first, mask last
        const float4 m = xld4(d.mask + (int64_t)row * d.ldmask + col);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4*>(p) = v;
    if (d.C16)                                         // the bf16 twin of what was just stored (next product's operand)
        *reinterpret_cast<uint2*>(d.C16 + coff + (int64_t)row * d.ldc) = make_uint2(ypack(v.x, v.y), ypack(v.z, v.w));
}

// Rows [row0, row0 + 32) x columns [col0, col0 + 64) of the product from the wave's staged block.  ws: this slice's partial
// tile goes there instead (split-K).
template <bool WT = false>                   // WT: the partial tile is stored write-through (reduced in this launch)
__device__ __forceinline__ void xep_rows(const skg_gemmx_desc& d, const skg_gemmx_fused& f, const float* stage, int lane,
                                         int row0, int col0, float* ws) {
    const int c4 = 4 * (lane & 15), rr = lane >> 4, col = col0 + c4;
    if (col >= d.N) return;
    const bool hb = !ws && d.bias;
    float4 bv = xzero4(), mb = xzero4();
    if (hb) bv = xld4(d.bias + col);
    if (f.kind == SKG_EPI_MUL_RELU && f.mbias) mb = xld4(f.mbias + col);
    const int64_t coff = xoff(col, d.c_nshift, d.c_nstride, 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + 4 * i + rr;
        if (row >= d.M) continue;
        const float4 v = *reinterpret_cast<const float4*>(stage + (4 * i + rr) * XEP_LD + c4);
        if (WT) { xst_sc1(xrsrc(ws), ((uint32_t)row * (uint32_t)d.N + (uint32_t)col) * 4u, v); continue; }
        if (ws) { *reinterpret_cast<float4*>(ws + (int64_t)row * d.N + col) = v; continue; }
        xep_apply(d, f, v, row, col, coff, hb, bv, mb);
    }
}

// The same rows by the tile's LAST ARRIVER: every slice added in slice order, then the epilogue.  OWN: this workgroup's slice
// (red.slice) comes from its staged block at its position in that order; otherwise it is read back like the others.  Loads
// are unconditional (rows past M clamped, never stored) and two slices deep: sixteen 16-byte loads in flight per lane.
template <bool OWN, int R>
__device__ __forceinline__ void xep_reduce(const skg_gemmx_desc& d, const skg_gemmx_fused& f, const float* stage, int lane,
                                           int row0, int col0, const XRed& red) {
    static_assert(R == 4 || R == 8, "rows per lane and pass");
    const int c4 = 4 * (lane & 15), rr = lane >> 4, col = col0 + c4;
    if (col >= d.N) return;
    const bool hb = d.bias != nullptr;
    float4 bv = xzero4(), mb = xzero4();
    if (hb) bv = xld4(d.bias + col);
    if (f.kind == SKG_EPI_MUL_RELU && f.mbias) mb = xld4(f.mbias + col);
    const int64_t coff = xoff(col, d.c_nshift, d.c_nstride, 1);
#pragma unroll 1
    for (int h = 0; h < 8 / R; ++h) {                      // (R = 4: two passes of four rows -- half the registers)
        uint32_t ro[R];                                    // byte offsets inside a slice (M * N < 2^29: checked on the host)
        float4 sum[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            ro[i] = ((uint32_t)min(row0 + 4 * (R * h + i) + rr, d.M - 1) * (uint32_t)d.N + (uint32_t)col) * 4u;
            sum[i] = xzero4();
        }
        auto add_slices = [&](int s0, int s1) {
            int s = s0;
            for (; s + 1 < s1; s += 2) {
                const __amdgpu_buffer_rsrc_t w0 = xrsrc(red.ws + (int64_t)s * red.MN);
                const __amdgpu_buffer_rsrc_t w1 = xrsrc(red.ws + (int64_t)(s + 1) * red.MN);
                float4 q0[R], q1[R];
#pragma unroll
                for (int i = 0; i < R; ++i) { q0[i] = xld_sc1(w0, ro[i]); q1[i] = xld_sc1(w1, ro[i]); }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    sum[i].x += q0[i].x; sum[i].y += q0[i].y; sum[i].z += q0[i].z; sum[i].w += q0[i].w;
                    sum[i].x += q1[i].x; sum[i].y += q1[i].y; sum[i].z += q1[i].z; sum[i].w += q1[i].w;
                }
            }
            if (s < s1) {
                const __amdgpu_buffer_rsrc_t w0 = xrsrc(red.ws + (int64_t)s * red.MN);
                float4 q0[R];
#pragma unroll
                for (int i = 0; i < R; ++i) q0[i] = xld_sc1(w0, ro[i]);
#pragma unroll
                for (int i = 0; i < R; ++i) { sum[i].x += q0[i].x; sum[i].y += q0[i].y; sum[i].z += q0[i].z; sum[i].w += q0[i].w; }
            }
        };
        if (OWN) {
            add_slices(0, red.slice);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float4 q = *reinterpret_cast<const float4*>(stage + (4 * (R * h + i) + rr) * XEP_LD + c4);
                sum[i].x += q.x; sum[i].y += q.y; sum[i].z += q.z; sum[i].w += q.w;
            }
            add_slices(red.slice + 1, red.S);
        } else {
            add_slices(0, red.S);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int row = row0 + 4 * (R * h + i) + rr;
            if (row < d.M) xep_apply(d, f, sum[i], row, col, coff, hb, bv, mb);
        }
    }
}

// One output element of a split product from the slices in split_ws (slice order), with the epilogue: the element-wise
// form of the reduction (products whose C does not admit 16-byte accesses), shared by the reduce kernel and the last arriver.
__device__ __forceinline__ void xreduce_elem(const skg_gemmx_desc& d, int64_t MN, int row, int col) {
    const int64_t i = (int64_t)row * d.N + col;
    float v = 0.f;
    for (int s = 0; s < d.split_k; ++s) v += d.split_ws[(int64_t)s * MN + i];
    if (d.bias) v += d.bias[col];
    if (d.relu) v = fmaxf(v, 0.f);
    float* p = d.C + xoff(col, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
    if (d.accumulate) v += *p;
    if (d.mask && !(d.mask[(int64_t)row * d.ldmask + col] > 0.f)) v = 0.f;      // accumulate first, mask last
    *p = v;
    if (d.C16) d.C16[p - d.C] = (uint16_t)ypack(v, 0.f);
}
// bias gradient of a split product: the slices' partial row sums, slice order
__device__ __forceinline__ void xreduce_rowsum(const skg_gemmx_desc& d, int64_t MN, int row) {
    float v = 0.f;
    for (int s = 0; s < d.split_k; ++s) v += d.split_ws[(int64_t)d.split_k * MN + (int64_t)s * d.M + row];
    d.a_rowsum[row] = d.accumulate ? d.a_rowsum[row] + v : v;
}
// the same by a tile's last arriver: the partial sums were stored write-through, and are read past the caches
__device__ __forceinline__ void xreduce_rowsum_sc1(const skg_gemmx_desc& d, int64_t MN, int row) {
    float v = 0.f;
    for (int s = 0; s < d.split_k; ++s)
        v += __hip_atomic_load(d.split_ws + (int64_t)d.split_k * MN + (int64_t)s * d.M + row, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    d.a_rowsum[row] = d.accumulate ? d.a_rowsum[row] + v : v;
}
__device__ __forceinline__ void xstore_rowsum_part(float* p, float v, bool wt) {
    if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// ================================================================================================ exact fp32
// Operand tiles live in LDS K-MAJOR ([16 k][128 rows], row stride 130 dwords): a lane's fragment is ONE ds_read_b64 =
// two adjacent rows at its k (the rows a MFMA tile covers are interleaved -- the tile does not care which rows it is
// given -- so no operand is ever transposed on its way to the matrix core).  Operands contiguous along their own index
// are copied into that image as they are (512-byte coalesced rows); operands contiguous along k are loaded as 16-byte
// k-quads and written transposed (130 = 2 mod 32 spreads the four k-quads of a row over four bank groups).
//   thread -> quads of a 128 x 16 tile:   KC : quad u = k0 + 4 (tid & 3) .. +3 of row (tid >> 2) + 64 u
//                                          RC : quad u = rows 4 (tid & 31) .. +3 at k = k0 + (tid >> 5) + 8 u
template <bool KC>
__device__ __forceinline__ void xprep(const XOperand& op, int row0, int tid, XFast& F) {
    F.o2 = F.o3 = 0;
    if (KC) {
        const int r = row0 + (tid >> 2), last = op.rows - 1, kl = 4 * (tid & 3);
        F.o0 = (uint32_t)((xoff(min(r, last), op.rshift, op.rstride, op.s_row) + kl) * 4);
        F.o1 = (uint32_t)((xoff(min(r + 64, last), op.rshift, op.rstride, op.s_row) + kl) * 4);
    } else {
        const int64_t ro = xoff(max(0, min(row0 + 4 * (tid & 31), op.rows - 4)), op.rshift, op.rstride, 1);
        const int kl = tid >> 5;
        F.o0 = (uint32_t)((ro + (int64_t)kl * op.s_k) * 4);
        F.o1 = (uint32_t)((ro + (int64_t)(kl + 8) * op.s_k) * 4);
    }
}

template <bool KC, bool FAST>
__device__ __forceinline__ XQ2 xtile(const XOperand& op, const XFast& F, int row0, int k0, int kend, int tid) {
    XQ2 v;
    if (FAST) {
        const uint32_t sb = xstepbase<KC>(op, k0);
        v.a = xldo(op, sb, F.o0); v.b = xldo(op, sb, F.o1);
    } else if (KC) {
        v.a = xgen<true>(op, row0 + (tid >> 2), k0 + 4 * (tid & 3), kend);
        v.b = xgen<true>(op, row0 + (tid >> 2) + 64, k0 + 4 * (tid & 3), kend);
    } else {
        v.a = xgen<false>(op, row0 + 4 * (tid & 31), k0 + (tid >> 5), kend);
        v.b = xgen<false>(op, row0 + 4 * (tid & 31), k0 + (tid >> 5) + 8, kend);
    }
    return v;
}

template <bool KC>
__device__ __forceinline__ void xstore_lds(float* tile, int tid, const XQ2& v) {
    if (KC) {
        float* p = tile + (4 * (tid & 3)) * XLD + (tid >> 2);
        p[0] = v.a.x; p[XLD] = v.a.y; p[2 * XLD] = v.a.z; p[3 * XLD] = v.a.w;
        p += 64;
        p[0] = v.b.x; p[XLD] = v.b.y; p[2 * XLD] = v.b.z; p[3 * XLD] = v.b.w;
    } else {
        float2* p = reinterpret_cast<float2*>(tile + (tid >> 5) * XLD + 4 * (tid & 31));
        p[0] = make_float2(v.a.x, v.a.y); p[1] = make_float2(v.a.z, v.a.w);
        p = reinterpret_cast<float2*>(tile + ((tid >> 5) + 8) * XLD + 4 * (tid & 31));
        p[0] = make_float2(v.b.x, v.b.y); p[1] = make_float2(v.b.z, v.b.w);
    }
}

struct XCtx {
    int m0, n0, kend, tid, wm, wn, li, lk;
    bool do_rowsum;
};

// k-tiles [ka, kb) of one workgroup.  `par` = LDS buffer the first tile goes to (flips per tile).
// k-tiles [ka, kb) of one workgroup.  `par` = LDS buffer the first tile goes to (flips per tile).
// XNST register stages: the loads of tile t + XNST are issued while tile t is multiplied, so a tile has XNST steps
// (~1 us each) to arrive -- one stage (the first version) left the HBM / L2 round trip exposed in every step.
#define XNST 3
template <bool AK, bool BK_, bool FAST>
__device__ __forceinline__ void xrun(const XOperand& A, const XOperand& B, const XFast& LA, const XFast& LB, const XCtx& c,
                                     int ka, int kb, int& par, float* smem, f32x16 (&acc)[2][2], float& rsum) {
    if (ka >= kb) return;
    const int tid = c.tid;
    XQ2 sa[XNST], sb[XNST];
    auto load = [&](XQ2& a, XQ2& b, int t) {              // past the end: the last tile again (never stored)
        const int tt = min(t, kb - 1);
        a = xtile<AK, FAST>(A, LA, c.m0, tt * XBK, c.kend, tid);
        b = xtile<BK_, FAST>(B, LB, c.n0, tt * XBK, c.kend, tid);
    };
    auto store = [&](XQ2& a, XQ2& b, int buf) {
        xstore_lds<AK>(smem + buf * 2 * XTILE, tid, a);
        xstore_lds<BK_>(smem + buf * 2 * XTILE + XTILE, tid, b);
    };
#pragma unroll
    for (int j = 0; j < XNST; ++j) load(sa[j], sb[j], ka + j);
    store(sa[0], sb[0], par);
    load(sa[0], sb[0], ka + XNST);
    __syncthreads();
    auto step = [&](auto J, int kt) {                      // tile kt is in LDS[par]; tile kt + 1 in stage (J + 1) % XNST
        constexpr int nx = (decltype(J)::value + 1) % XNST;
        const float* As = smem + par * 2 * XTILE;
        const float* Bs = As + XTILE;
#pragma unroll
        for (int ks = 0; ks < XBK / 2; ++ks) {
            const int kk = 2 * ks + c.lk;
            const float2 a = *reinterpret_cast<const float2*>(As + kk * XLD + c.wm * 64 + 2 * c.li);
            const float2 bq = *reinterpret_cast<const float2*>(Bs + kk * XLD + c.wn * 64 + 2 * c.li);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc[1][1], 0, 0, 0);
        }
        if (c.do_rowsum && tid < XBM) {
#pragma unroll
            for (int kk = 0; kk < XBK; ++kk) rsum += As[kk * XLD + tid];
        }
        if (kt + 1 < kb) store(sa[nx], sb[nx], par ^ 1);
        load(sa[nx], sb[nx], kt + 1 + XNST);
        __syncthreads();
        par ^= 1;
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    static_assert(XNST == 3, "the step sequence below is written for three stages");
    int kt = ka;
    for (; kt + XNST <= kb; kt += XNST) {
        step(I0{}, kt);
        step(I1{}, kt + 1);
        step(I2{}, kt + 2);
    }
    if (kt < kb) step(I0{}, kt);
    if (kt + 1 < kb) step(I1{}, kt + 1);
}

template <bool AK, bool BK_>
__device__ __forceinline__ void xmain(const XOperand& A, const XOperand& B, const XCtx& c, int kt0, int kt1, float* smem,
                                      f32x16 (&acc)[2][2], float& rsum) {
    XFast LA, LB;
    xprep<AK>(A, c.m0, c.tid, LA);
    xprep<BK_>(B, c.n0, c.tid, LB);
    int ktf = kt0;                                         // [kt0, ktf): tiles inside the slice, fast loop
    if (xfast_ok<AK>(A, c.m0) && xfast_ok<BK_>(B, c.n0) && kt1 > kt0) ktf = (c.kend == kt1 * XBK) ? kt1 : kt1 - 1;
    int par = 0;
    xrun<AK, BK_, true>(A, B, LA, LB, c, kt0, ktf, par, smem, acc, rsum);
    xrun<AK, BK_, false>(A, B, LA, LB, c, ktf, kt1, par, smem, acc, rsum);
}

__device__ __forceinline__ void xoperands(const skg_gemmx_desc& d, int vecbits, XOperand& A, XOperand& B) {
    A.base = d.A; A.s_row = d.a_sm; A.s_k = d.a_sk; A.rshift = 0; A.kshift = 0; A.rstride = 0; A.kstride = 0;
    A.rows = d.M; A.vec = vecbits & 1;
    A.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, 0x7fffffff, 0x00020000);
    B.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, 0x7fffffff, 0x00020000);
    A.rsrc16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.A16), 0, 0x7fffffff, 0x00020000);
    B.rsrc16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(d.B16), 0, 0x7fffffff, 0x00020000);
    A.vec16 = (vecbits >> 4) & 1; B.vec16 = (vecbits >> 5) & 1;
    A.s_row16 = (d.a_sk == 1 && d.a16_ld) ? d.a16_ld : d.a_sm; A.s_k16 = (d.a_sk != 1 && d.a16_ld) ? d.a16_ld : d.a_sk;
    B.s_row16 = (d.b_sk == 1 && d.b16_ld) ? d.b16_ld : d.b_sn; B.s_k16 = (d.b_sk != 1 && d.b16_ld) ? d.b16_ld : d.b_sk;
    B.base = d.B; B.s_row = d.b_sn; B.s_k = d.b_sk; B.rshift = d.b_nshift; B.kshift = d.b_kshift;
    B.rstride = d.b_nstride; B.kstride = d.b_kstride; B.rows = d.N; B.vec = (vecbits >> 1) & 1;
}

__global__ __launch_bounds__(256, 2) void skg_gemmx_kernel(const skg_gemmx_group g) {
    static_assert(XEP_FLOATS >= 4 * XTILE, "the staged epilogue reuses the operand tiles' LDS");
    __shared__ __attribute__((aligned(16))) float smem[XEP_FLOATS + 4]; // A0 | B0 | A1 | B1, then the staged epilogue (+ the arrival flag)
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int vecbits = g.vec[gi];
    const int S = d.split_k > 1 ? d.split_k : 1;
    const int nbn = (d.N + XBN - 1) / XBN;
    const XTileId tid3 = xtile_of(blockIdx.x - g.start[gi], g.start[gi + 1] - g.start[gi], (d.M + XBM - 1) / XBM, nbn, S, d.K);
    const int slice = tid3.slice, tn = tid3.tn, tm = tid3.tm;
    // k range of this slice, in whole k-tiles
    const int nkt = (d.K + XBK - 1) / XBK;
    const int per = (nkt + S - 1) / S;
    const int kt0 = slice * per, kt1 = min(nkt, kt0 + per);

    XOperand A, B;
    xoperands(d, vecbits, A, B);
    XCtx c;
    c.m0 = tm * XBM; c.n0 = tn * XBN; c.kend = min(d.K, kt1 * XBK);
    c.tid = threadIdx.x;
    const int lane = c.tid & 63, wave = c.tid >> 6;
    c.wm = wave >> 1; c.wn = wave & 1; c.li = lane & 31; c.lk = lane >> 5;
    c.do_rowsum = d.a_rowsum != nullptr && tn == 0;
    const int m0 = c.m0, n0 = c.n0, tid = c.tid, wm = c.wm, wn = c.wn, li = c.li, lk = c.lk;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float rsum = 0.f;
    if (d.a_sk == 1) {
        if (d.b_sk == 1) xmain<true, true>(A, B, c, kt0, kt1, smem, acc, rsum);
        else xmain<true, false>(A, B, c, kt0, kt1, smem, acc, rsum);
    } else {
        if (d.b_sk == 1) xmain<false, true>(A, B, c, kt0, kt1, smem, acc, rsum);
        else xmain<false, false>(A, B, c, kt0, kt1, smem, acc, rsum);
    }

    // ---- epilogue.  Lane (li, lk) of wave (wm, wn) holds, in acc[mb][nb][4*gq + t], the element
    //      row m0 + wm*64 + 2*(8*gq + 4*lk + t) + mb,  column n0 + wn*64 + 2*li + nb.
    const bool split = S > 1;
    const int64_t MN = (int64_t)d.M * d.N;
    float* ws = split ? d.split_ws + (int64_t)slice * MN : nullptr;           // [S][M * N] then [S][M] row sums
    const bool inl = split && d.split_ctr != nullptr;      // reduced in this launch by the tile's last arriver (staged epilogue only: the host checked)
    if (c.do_rowsum && tid < XBM && m0 + tid < d.M) {
        if (split) xstore_rowsum_part(d.split_ws + (int64_t)S * MN + (int64_t)slice * d.M + m0 + tid, rsum, inl);
        else d.a_rowsum[m0 + tid] = d.accumulate ? d.a_rowsum[m0 + tid] + rsum : rsum;
    }
    if (vecbits & 8) {                                     // staged: the k loop ended on a barrier, LDS is free
        float* stage = smem + wave * XEP_WAVE;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int e = 4 * (2 * ps + g2) + t;
                        *reinterpret_cast<float2*>(stage + (2 * (8 * g2 + 4 * lk + t) + mb) * XEP_LD + 2 * li) =
                            make_float2(acc[mb][0][e], acc[mb][1][e]);
                    }
            if (inl) xep_rows<true>(d, g.f[gi], stage, lane, m0 + wm * 64 + 32 * ps, n0 + wn * 64, ws);
            else xep_rows(d, g.f[gi], stage, lane, m0 + wm * 64 + 32 * ps, n0 + wn * 64, ws);
        }
        if (!inl || !xsplit_last(d.split_ctr + tm * nbn + tn, S, reinterpret_cast<uint32_t*>(smem + XEP_FLOATS))) return;
        // the last slice of this tile to arrive: all slices in slice order + the epilogue
        if (c.do_rowsum && tid < XBM && m0 + tid < d.M) xreduce_rowsum_sc1(d, MN, m0 + tid);
        // (slice -1: this workgroup's own partial is read back like the others -- keeping the 64 accumulators alive across
        //  the arrival would cost this kernel its third wave per SIMD: 164 -> 179 VGPRs)
        const XRed red = {d.split_ws, MN, S, -1};
#pragma unroll 1
        for (int ps = 0; ps < 2; ++ps)
            xep_reduce<false, 4>(d, g.f[gi], stage, lane, m0 + wm * 64 + 32 * ps, n0 + wn * 64, red);
        return;
    }
    const bool vecC = (vecbits >> 2) & 1;
    const int col = n0 + wn * 64 + 2 * li;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int row = m0 + wm * 64 + 2 * (8 * gq + 4 * lk + t) + mb;
                if (row >= d.M || col >= d.N) continue;
                float v0 = acc[mb][0][4 * gq + t], v1 = acc[mb][1][4 * gq + t];
                if (split) {
                    float* p = ws + (int64_t)row * d.N + col;
                    if (col + 1 < d.N && (d.N & 1) == 0) *reinterpret_cast<float2*>(p) = make_float2(v0, v1);
                    else { p[0] = v0; if (col + 1 < d.N) p[1] = v1; }
                    continue;
                }
                const bool two = col + 1 < d.N;
                if (d.bias) { v0 += d.bias[col]; if (two) v1 += d.bias[col + 1]; }
                if (d.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                // accumulate first, mask last: a gradient that reaches a ReLU output from several consumers is summed
                // and THEN cut by the ReLU (the mask belongs to the tensor C describes, not to this one contribution)
                bool k0 = true, k1 = true;
                if (d.mask) {
                    const float* mp = d.mask + (int64_t)row * d.ldmask + col;
                    k0 = mp[0] > 0.f;
                    k1 = two && mp[1] > 0.f;
                }
                float* p = d.C + xoff(col, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
                if (two && vecC) {
                    float2* p2 = reinterpret_cast<float2*>(p);
                    if (d.accumulate) { const float2 o = *p2; v0 += o.x; v1 += o.y; }
                    v0 = k0 ? v0 : 0.f; v1 = k1 ? v1 : 0.f;
                    *p2 = make_float2(v0, v1);
                    if (d.C16) *reinterpret_cast<uint32_t*>(d.C16 + (p - d.C)) = ypack(v0, v1);
                } else {
                    if (d.accumulate) v0 += p[0];
                    v0 = k0 ? v0 : 0.f;
                    p[0] = v0;
                    if (d.C16) d.C16[p - d.C] = (uint16_t)ypack(v0, 0.f);
                    if (two) {
                        float* q = d.C + xoff(col + 1, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
                        if (d.accumulate) v1 += q[0];
                        v1 = k1 ? v1 : 0.f;
                        q[0] = v1;
                        if (d.C16) d.C16[q - d.C] = (uint16_t)ypack(v1, 0.f);
                    }
                }
            }
}

// ================================================================================================ bf16 operands
// Same products, same descriptors, same epilogue; the operands stay fp32 in HBM and are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS, accumulation in fp32 on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
// This is what precision="bf16" training runs (BASELINE config 3: autocast of the reference's nn.Linear layers).
//
// 128x128x32 block tile.  LDS image of an operand tile: four k-planes (8 k each) of [128 rows][8 bf16 = 16 B], plane
// stride 520 dwords, row slot XOR-swizzled (slot = row ^ ((row >> 3) & 3)): a lane's MFMA fragment is ONE
// ds_read_b128, conflict-free, and both source layouts are written conflict-free too --
//   k-contiguous source  : a thread has 4 k of one row        -> one ds_write_b64
//   row-contiguous source: a thread loads 4 rows x 4 k (four float4 along the rows), transposes in registers
//                                                              -> four ds_write_b64 (one per row)
//   thread -> quads of a 128 x 32 tile:   KC : quad u = k0 + 4 (tid & 7) .. +3 of row (tid >> 3) + 32 u
//                                          RC : quad u = rows 4 rq .. +3 at k = k0 + 4 kq4 + u,
//                                               rq = (tid & 7) | ((tid >> 4) & 3) << 3,  kq4 = ((tid >> 3) & 1) | (tid >> 6) << 1
// The bias gradient (row sums of A) is accumulated from the fp32 registers before rounding.
#define YBK 32
#define YPLANE 1040                          // bf16 elements per k-plane: 128 rows * 8 + 16 pad (520 dwords)
#define YTILE (4 * YPLANE)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int yslot(int row) { return row ^ ((row >> 3) & 3); }
__device__ __forceinline__ int yrq(int tid) { return (tid & 7) | (((tid >> 4) & 3) << 3); }
__device__ __forceinline__ int ykq4(int tid) { return ((tid >> 3) & 1) | ((tid >> 6) << 1); }

// ---- operand tiles in flight.  H = false: the fp32 source (four 16-byte loads per thread, rounded on the way into LDS);
// H = true: the operand's bf16 TWIN in memory (skg_gemmx_desc.A16 / B16: two 16-byte loads per thread, no conversion) --
// half the bytes through the L2 -> CU path, which is what bounds this loop (one workgroup per CU draws ~22 B/clk of the
// ~29 B/clk a CU can fetch from L2; measured from K sweeps with the loop / the epilogue compiled out).
typedef uint32_t xu4v __attribute__((ext_vector_type(4)));
template <bool H> struct YStage;
typedef float xf4v __attribute__((ext_vector_type(4)));
// (native vectors as named members: with HIP's float4 -- a struct around a union -- or with array members the stages of
// the mixed fp32 / twin loops were kept in scratch memory)
template <> struct YStage<false> { xf4v a, b, c, d; };
template <> struct YStage<true> { xu4v a, b; };
__device__ __forceinline__ xf4v xnat(float4 t) { xf4v o; o.x = t.x; o.y = t.y; o.z = t.z; o.w = t.w; return o; }

__device__ __forceinline__ xu4v xldo16(const XOperand& op, uint32_t soff, uint32_t voff) {
    typedef unsigned int xu4 __attribute__((__vector_size__(16)));
    const xu4 r = __builtin_amdgcn_raw_buffer_load_b128(op.rsrc16, (int)voff, (int)soff, 0);
    xu4v o; o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = r[3];
    return o;
}

//   thread -> pieces of a 128 x 32 bf16 tile (16 bytes = 8 elements each):
//     KC : piece u = 8 k of plane (tid & 3) of row (tid >> 2) + 64 u                    (4 lanes = the row's 64 bytes)
//     RC : piece u = rows 8 (tid >> 4) .. +7 at k = k0 + 2 (tid & 15) + u              (a k pair per thread)
template <bool KC, bool H>
__device__ __forceinline__ void yprep(const XOperand& op, int row0, int tid, XFast& F) {
    if (!H) {
        if (KC) {
            const int r = row0 + (tid >> 3), last = op.rows - 1, kl = 4 * (tid & 7);
            F.o0 = (uint32_t)((xoff(min(r, last), op.rshift, op.rstride, op.s_row) + kl) * 4);
            F.o1 = (uint32_t)((xoff(min(r + 32, last), op.rshift, op.rstride, op.s_row) + kl) * 4);
            F.o2 = (uint32_t)((xoff(min(r + 64, last), op.rshift, op.rstride, op.s_row) + kl) * 4);
            F.o3 = (uint32_t)((xoff(min(r + 96, last), op.rshift, op.rstride, op.s_row) + kl) * 4);
        } else {
            const int64_t ro = xoff(max(0, min(row0 + 4 * yrq(tid), op.rows - 4)), op.rshift, op.rstride, 1);
            const int kl = 4 * ykq4(tid);
            F.o0 = (uint32_t)((ro + (int64_t)kl * op.s_k) * 4);
            F.o1 = (uint32_t)((ro + (int64_t)(kl + 1) * op.s_k) * 4);
            F.o2 = (uint32_t)((ro + (int64_t)(kl + 2) * op.s_k) * 4);
            F.o3 = (uint32_t)((ro + (int64_t)(kl + 3) * op.s_k) * 4);
        }
    } else {
        F.o2 = F.o3 = 0;
        if (KC) {
            const int r = row0 + (tid >> 2), last = op.rows - 1, kl = 8 * (tid & 3);
            F.o0 = (uint32_t)((xoff(min(r, last), op.rshift, op.rstride, op.s_row) + kl) * 2);
            F.o1 = (uint32_t)((xoff(min(r + 64, last), op.rshift, op.rstride, op.s_row) + kl) * 2);
        } else {
            const int64_t ro = xoff(max(0, min(row0 + 8 * (tid >> 4), op.rows - 8)), op.rshift, op.rstride, 1);
            const int kl = 2 * (tid & 15);
            F.o0 = (uint32_t)((ro + (int64_t)kl * op.s_k) * 2);
            F.o1 = (uint32_t)((ro + (int64_t)(kl + 1) * op.s_k) * 2);
        }
    }
}

template <bool KC, bool FAST>
__device__ __forceinline__ void ytile(const XOperand& op, const XFast& F, int row0, int k0, int kend, int tid,
                                      YStage<false>& s) {
    if (FAST) {
        const uint32_t sb = xstepbase<KC>(op, k0);
        s.a = xnat(xldo(op, sb, F.o0)); s.b = xnat(xldo(op, sb, F.o1));
        s.c = xnat(xldo(op, sb, F.o2)); s.d = xnat(xldo(op, sb, F.o3));
    } else if (KC) {
        const int r = row0 + (tid >> 3), k = k0 + 4 * (tid & 7);
        s.a = xnat(xgen<true>(op, r, k, kend)); s.b = xnat(xgen<true>(op, r + 32, k, kend));
        s.c = xnat(xgen<true>(op, r + 64, k, kend)); s.d = xnat(xgen<true>(op, r + 96, k, kend));
    } else {
        const int row = row0 + 4 * yrq(tid), k = k0 + 4 * ykq4(tid);
        s.a = xnat(xgen<false>(op, row, k, kend)); s.b = xnat(xgen<false>(op, row, k + 1, kend));
        s.c = xnat(xgen<false>(op, row, k + 2, kend)); s.d = xnat(xgen<false>(op, row, k + 3, kend));
    }
}
template <bool KC, bool FAST>
__device__ __forceinline__ void ytile(const XOperand& op, const XFast& F, int row0, int k0, int kend, int tid,
                                      YStage<true>& s) {
    static_assert(FAST, "the bf16 twin is read by the fast loop only");
    const uint32_t sb = xstepbase<KC>(op, k0) >> 1;         // the same element offset, two bytes per element
    s.a = xldo16(op, sb, F.o0); s.b = xldo16(op, sb, F.o1);
}

template <bool KC>
__device__ __forceinline__ void ystore_lds(uint16_t* tile, int tid, const YStage<false>& s) {
    const xf4v va = s.a, vb = s.b, vc = s.c, vd = s.d;
    if (KC) {
        const int kq4 = tid & 7, r = tid >> 3;
        uint16_t* base = tile + (kq4 >> 1) * YPLANE + (kq4 & 1) * 4;
        *reinterpret_cast<uint2*>(base + yslot(r) * 8) = make_uint2(ypack(va.x, va.y), ypack(va.z, va.w));
        *reinterpret_cast<uint2*>(base + yslot(r + 32) * 8) = make_uint2(ypack(vb.x, vb.y), ypack(vb.z, vb.w));
        *reinterpret_cast<uint2*>(base + yslot(r + 64) * 8) = make_uint2(ypack(vc.x, vc.y), ypack(vc.z, vc.w));
        *reinterpret_cast<uint2*>(base + yslot(r + 96) * 8) = make_uint2(ypack(vd.x, vd.y), ypack(vd.z, vd.w));
    } else {
        const int kq4 = ykq4(tid), row = 4 * yrq(tid);
        uint16_t* base = tile + (kq4 >> 1) * YPLANE + (kq4 & 1) * 4;
        *reinterpret_cast<uint2*>(base + yslot(row + 0) * 8) = make_uint2(ypack(va.x, vb.x), ypack(vc.x, vd.x));
        *reinterpret_cast<uint2*>(base + yslot(row + 1) * 8) = make_uint2(ypack(va.y, vb.y), ypack(vc.y, vd.y));
        *reinterpret_cast<uint2*>(base + yslot(row + 2) * 8) = make_uint2(ypack(va.z, vb.z), ypack(vc.z, vd.z));
        *reinterpret_cast<uint2*>(base + yslot(row + 3) * 8) = make_uint2(ypack(va.w, vb.w), ypack(vc.w, vd.w));
    }
}
template <bool KC>
__device__ __forceinline__ void ystore_lds(uint16_t* tile, int tid, const YStage<true>& s) {
    if (KC) {                                              // a piece IS a plane entry: one ds_write_b128 each
        const int pl = tid & 3, r = tid >> 2;
        *reinterpret_cast<xu4v*>(tile + pl * YPLANE + yslot(r) * 8) = s.a;
        *reinterpret_cast<xu4v*>(tile + pl * YPLANE + yslot(r + 64) * 8) = s.b;
    } else {                                               // rows r8 .. r8 + 7 at k and k + 1: one dword {k, k + 1} per row
        const int kp = tid & 15, r8 = 8 * (tid >> 4);
        uint16_t* base = tile + (kp >> 2) * YPLANE + 2 * (kp & 3);
        const xu4v a = s.a, b = s.b;
#define YROWPAIR(j, aw, bw)                                                                                              \
        *reinterpret_cast<uint32_t*>(base + yslot(r8 + 2 * (j)) * 8) = __builtin_amdgcn_perm(bw, aw, 0x05040100u);       \
        *reinterpret_cast<uint32_t*>(base + yslot(r8 + 2 * (j) + 1) * 8) = __builtin_amdgcn_perm(bw, aw, 0x07060302u);
        YROWPAIR(0, a.x, b.x) YROWPAIR(1, a.y, b.y) YROWPAIR(2, a.z, b.z) YROWPAIR(3, a.w, b.w)
#undef YROWPAIR
    }
}

// fp32 row sums of the thread's share of an A tile (before rounding)
template <bool KC>
__device__ __forceinline__ void yrowsum(const YStage<false>& s, float4& rs) {
    const xf4v va = s.a, vb = s.b, vc = s.c, vd = s.d;
    if (KC) {
        rs.x += (va.x + va.y) + (va.z + va.w); rs.y += (vb.x + vb.y) + (vb.z + vb.w);
        rs.z += (vc.x + vc.y) + (vc.z + vc.w); rs.w += (vd.x + vd.y) + (vd.z + vd.w);
    } else {
        rs.x += (va.x + vb.x) + (vc.x + vd.x); rs.y += (va.y + vb.y) + (vc.y + vd.y);
        rs.z += (va.z + vb.z) + (vc.z + vd.z); rs.w += (va.w + vb.w) + (vc.w + vd.w);
    }
}
template <bool KC>
__device__ __forceinline__ void yrowsum(const YStage<true>&, float4&) {}     // (row-sum workgroups read the fp32 operand)

// One register stage (the next tile in flight across the MFMAs): kept for products whose operands are both contiguous
// along their own index and fp32 (dW = dZ^T X) -- measured, a second stage costs them a third of their rate at 102400
// rows (475 -> 295 TFLOP/s: eight more 16-byte loads per thread in flight, two 512-byte segments each), while it gains
// the k-contiguous layouts 15-35 %.
template <bool AK, bool BK_, bool FAST>
__device__ __forceinline__ void yrun1(const XOperand& A, const XOperand& B, const XFast& LA, const XFast& LB, const XCtx& c,
                                      int ka, int kb, int& par, uint16_t* smem, f32x16 (&acc)[2][2], float4& rs) {
    if (ka >= kb) return;
    const int tid = c.tid;
    // fragment addresses (bf16 elements) inside a tile; k-step ks adds 2 ks planes
    const int fa0 = c.lk * YPLANE + yslot(c.wm * 64 + c.li) * 8, fa1 = c.lk * YPLANE + yslot(c.wm * 64 + 32 + c.li) * 8;
    const int fb0 = c.lk * YPLANE + yslot(c.wn * 64 + c.li) * 8, fb1 = c.lk * YPLANE + yslot(c.wn * 64 + 32 + c.li) * 8;
    YStage<false> sa, sb;
    ytile<AK, FAST>(A, LA, c.m0, ka * YBK, c.kend, tid, sa);
    ytile<BK_, FAST>(B, LB, c.n0, ka * YBK, c.kend, tid, sb);
    if (c.do_rowsum) yrowsum<AK>(sa, rs);
    ystore_lds<AK>(smem + par * 2 * YTILE, tid, sa);
    ystore_lds<BK_>(smem + par * 2 * YTILE + YTILE, tid, sb);
    __syncthreads();
    for (int kt = ka; kt < kb; ++kt) {
        const uint16_t* As = smem + par * 2 * YTILE;
        const uint16_t* Bs = As + YTILE;
        const bool more = kt + 1 < kb;
        if (more) {
            ytile<AK, FAST>(A, LA, c.m0, (kt + 1) * YBK, c.kend, tid, sa);
            ytile<BK_, FAST>(B, LB, c.n0, (kt + 1) * YBK, c.kend, tid, sb);
        }
        const bf16x8 a00 = *reinterpret_cast<const bf16x8*>(As + fa0), a01 = *reinterpret_cast<const bf16x8*>(As + fa1);
        const bf16x8 b00 = *reinterpret_cast<const bf16x8*>(Bs + fb0), b01 = *reinterpret_cast<const bf16x8*>(Bs + fb1);
        const bf16x8 a10 = *reinterpret_cast<const bf16x8*>(As + 2 * YPLANE + fa0);
        const bf16x8 a11 = *reinterpret_cast<const bf16x8*>(As + 2 * YPLANE + fa1);
        const bf16x8 b10 = *reinterpret_cast<const bf16x8*>(Bs + 2 * YPLANE + fb0);
        const bf16x8 b11 = *reinterpret_cast<const bf16x8*>(Bs + 2 * YPLANE + fb1);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, b00, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, b01, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b00, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b01, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, b10, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, b11, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, b10, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, b11, acc[1][1], 0, 0, 0);
        if (more) {
            uint16_t* An = smem + (par ^ 1) * 2 * YTILE;
            if (c.do_rowsum) yrowsum<AK>(sa, rs);
            ystore_lds<AK>(An, tid, sa);
            ystore_lds<BK_>(An + YTILE, tid, sb);
        }
        __syncthreads();
        par ^= 1;
    }
}

// YNST register stages of global prefetch (see xrun): here a step's MFMAs take 256 cycles, so the loop is a pure memory
// round trip per step unless several tiles are in flight; two stages (64 VGPRs of fp32 operands) are what the register
// budget of two waves per SIMD leaves next to the 64 accumulators.
#define YNST 2
#ifndef SKG_YORDER
#define SKG_YORDER 1
#endif
template <bool AK, bool BK_, bool FAST, bool AH, bool BH>
__device__ __forceinline__ void yrun(const XOperand& A, const XOperand& B, const XFast& LA, const XFast& LB, const XCtx& c,
                                     int ka, int kb, int& par, uint16_t* smem, f32x16 (&acc)[2][2], float4& rs) {
    if (ka >= kb) return;
    const int tid = c.tid;
    // fragment addresses (bf16 elements) inside a tile; k-step ks adds 2 ks planes
    const int fa0 = c.lk * YPLANE + yslot(c.wm * 64 + c.li) * 8, fa1 = c.lk * YPLANE + yslot(c.wm * 64 + 32 + c.li) * 8;
    const int fb0 = c.lk * YPLANE + yslot(c.wn * 64 + c.li) * 8, fb1 = c.lk * YPLANE + yslot(c.wn * 64 + 32 + c.li) * 8;
    YStage<AH> sa0, sa1;                                   // (separate objects: arrays of stages ended up in scratch memory)
    YStage<BH> sb0, sb1;
    auto load = [&](auto S, int t) {                      // past the end: the last tile again (never stored)
#ifdef SKG_YKNOCK_LOAD                                     // timing builds (tools/build_gemmx_variants.sh): one piece of the step removed
        if (t > ka + 2) return;
#endif
        const int tt = min(t, kb - 1);
        if constexpr (decltype(S)::value == 0) {
            ytile<AK, FAST>(A, LA, c.m0, tt * YBK, c.kend, tid, sa0);
            ytile<BK_, FAST>(B, LB, c.n0, tt * YBK, c.kend, tid, sb0);
        } else {
            ytile<AK, FAST>(A, LA, c.m0, tt * YBK, c.kend, tid, sa1);
            ytile<BK_, FAST>(B, LB, c.n0, tt * YBK, c.kend, tid, sb1);
        }
    };
    auto store = [&](auto S, int buf) {
#ifdef SKG_YKNOCK_STORE
        if (buf >= 0) return;
#endif
        if constexpr (decltype(S)::value == 0) {
            if (c.do_rowsum) yrowsum<AK>(sa0, rs);
            ystore_lds<AK>(smem + buf * 2 * YTILE, tid, sa0);
            ystore_lds<BK_>(smem + buf * 2 * YTILE + YTILE, tid, sb0);
        } else {
            if (c.do_rowsum) yrowsum<AK>(sa1, rs);
            ystore_lds<AK>(smem + buf * 2 * YTILE, tid, sa1);
            ystore_lds<BK_>(smem + buf * 2 * YTILE + YTILE, tid, sb1);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    static_assert(YNST == 2, "the step sequence below is written for two stages");
    load(I0{}, ka);
    load(I1{}, ka + 1);
    store(I0{}, par);
    load(I0{}, ka + 2);
    __syncthreads();
    // MORE: a tile follows (known at compile time inside the main loop: a branch around the store would cut the
    // scheduling region the interleave below needs)
    auto step = [&](auto NX, auto MORE, int kt) {          // tile kt is in LDS[par]; tile kt + 1 in stage NX
        constexpr bool more_ct = decltype(MORE)::value;
        const uint16_t* As = smem + par * 2 * YTILE;
        const uint16_t* Bs = As + YTILE;
        const bf16x8 a00 = *reinterpret_cast<const bf16x8*>(As + fa0), a01 = *reinterpret_cast<const bf16x8*>(As + fa1);
        const bf16x8 b00 = *reinterpret_cast<const bf16x8*>(Bs + fb0), b01 = *reinterpret_cast<const bf16x8*>(Bs + fb1);
        const bf16x8 a10 = *reinterpret_cast<const bf16x8*>(As + 2 * YPLANE + fa0);
        const bf16x8 a11 = *reinterpret_cast<const bf16x8*>(As + 2 * YPLANE + fa1);
        const bf16x8 b10 = *reinterpret_cast<const bf16x8*>(Bs + 2 * YPLANE + fb0);
        const bf16x8 b11 = *reinterpret_cast<const bf16x8*>(Bs + 2 * YPLANE + fb1);
#if SKG_YORDER == 1
        // the next tile goes to the other LDS buffer while this tile's fragments are still on their way to registers
        __builtin_amdgcn_sched_barrier(0);
        if (more_ct || kt + 1 < kb) store(NX, par ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef SKG_YKNOCK_MFMA
        acc[0][0][0] += (float)a00[0] + (float)b00[0] + (float)a01[0] + (float)b01[0] + (float)a10[0] + (float)b10[0] +
                        (float)a11[0] + (float)b11[0];
        if (false)
#endif
        {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, b00, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, b01, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b00, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b01, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, b10, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, b11, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, b10, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, b11, acc[1][1], 0, 0, 0);
        }
#if SKG_YORDER != 1
        if (more_ct || kt + 1 < kb) store(NX, par ^ 1);
#endif
#if SKG_YORDER == 2
        // conversions and LDS writes of the next tile issued in the shadow of the MFMAs (an MFMA holds the matrix pipe for 32
        // cycles; two packed conversions and one ds_write fit behind each)
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#endif
        load(NX, kt + 1 + YNST);
#ifndef SKG_YKNOCK_BARRIER
        __syncthreads();
#endif
        par ^= 1;
    };
    using Yes = std::integral_constant<bool, true>;
    using Maybe = std::integral_constant<bool, false>;
    int kt = ka;
    for (; kt + 2 < kb; kt += 2) {
        step(I1{}, Yes{}, kt);
        step(I0{}, Yes{}, kt + 1);
    }
    if (kt < kb) step(I1{}, Maybe{}, kt);
    if (kt + 1 < kb) step(I0{}, Maybe{}, kt + 1);
}

// 16-bit reads of an operand from row0: the twin exists, its descriptor passed the host's alignment checks, and no
// 8-row piece straddles the operand's end (row-contiguous layout)
template <bool KC>
__device__ __forceinline__ bool yhalf_ok(const XOperand& op, int row0) {
    return op.vec16 && (KC || (op.rows & 7) == 0 || row0 + 128 <= op.rows);
}

template <bool AK, bool BK_, bool AH, bool BH>
__device__ __forceinline__ void yfast(const XOperand& A, const XOperand& B, const XCtx& c, int kt0, int ktf, int& par,
                                      uint16_t* smem, f32x16 (&acc)[2][2], float4& rs) {
    XFast LA, LB;
    yprep<AK, AH>(A, c.m0, c.tid, LA);
    yprep<BK_, BH>(B, c.n0, c.tid, LB);
    if (!AK && !BK_ && !AH && !BH) yrun1<AK, BK_, true>(A, B, LA, LB, c, kt0, ktf, par, smem, acc, rs);
    else yrun<AK, BK_, true, AH, BH>(A, B, LA, LB, c, kt0, ktf, par, smem, acc, rs);
}

template <bool AK, bool BK_>
__device__ __forceinline__ void ymain(const XOperand& A, const XOperand& B, const XCtx& c, int kt0, int kt1, uint16_t* smem,
                                      f32x16 (&acc)[2][2], float4& rs) {
    int ktf = kt0;
    if (xfast_ok<AK>(A, c.m0) && xfast_ok<BK_>(B, c.n0) && kt1 > kt0) ktf = (c.kend == kt1 * YBK) ? kt1 : kt1 - 1;
    int par = 0;
    // the workgroups that form the bias gradient sum the UNROUNDED fp32 operand: they keep the fp32 source of A
    const bool ah = yhalf_ok<AK>(A, c.m0) && !c.do_rowsum, bh = yhalf_ok<BK_>(B, c.n0);
    if (ah) {
        if (bh) yfast<AK, BK_, true, true>(A, B, c, kt0, ktf, par, smem, acc, rs);
        else yfast<AK, BK_, true, false>(A, B, c, kt0, ktf, par, smem, acc, rs);
    } else {
        if (bh) yfast<AK, BK_, false, true>(A, B, c, kt0, ktf, par, smem, acc, rs);
        else yfast<AK, BK_, false, false>(A, B, c, kt0, ktf, par, smem, acc, rs);
    }
    {
        XFast LA, LB;                                      // (unused by the generic loop)
        LA.o0 = LA.o1 = LA.o2 = LA.o3 = 0; LB.o0 = LB.o1 = LB.o2 = LB.o3 = 0;
        if (!AK && !BK_) yrun1<AK, BK_, false>(A, B, LA, LB, c, ktf, kt1, par, smem, acc, rs);
        else yrun<AK, BK_, false, false, false>(A, B, LA, LB, c, ktf, kt1, par, smem, acc, rs);
    }
    if (c.do_rowsum) {                                     // uniform per workgroup; the k loop ended on a barrier
        float* part = reinterpret_cast<float*>(smem);      // [8][128] partial sums
        if (AK) {
            float* q = part + (c.tid & 7) * 128 + (c.tid >> 3);
            q[0] = rs.x; q[32] = rs.y; q[64] = rs.z; q[96] = rs.w;
        } else {
            float* q = part + ykq4(c.tid) * 128 + 4 * yrq(c.tid);
            q[0] = rs.x; q[1] = rs.y; q[2] = rs.z; q[3] = rs.w;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256, 2) void skg_gemmx_bf16_kernel(const skg_gemmx_group g) {
    static_assert(2 * XEP_FLOATS >= 4 * YTILE, "the staged epilogue reuses the operand tiles' LDS");
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * XEP_FLOATS + 8];  // A0 | B0 | A1 | B1, then the staged epilogue (+ flag)
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int vecbits = g.vec[gi];
    const int S = d.split_k > 1 ? d.split_k : 1;
    const int nbn = (d.N + XBN - 1) / XBN;
    const XTileId tid3 = xtile_of(blockIdx.x - g.start[gi], g.start[gi + 1] - g.start[gi], (d.M + XBM - 1) / XBM, nbn, S, d.K);
    const int slice = tid3.slice, tn = tid3.tn, tm = tid3.tm;
    const int nkt = (d.K + YBK - 1) / YBK;
    const int per = (nkt + S - 1) / S;
#ifdef SKG_XPROBE_NOLOOP                                   // timing builds (tools/build_gemmx_variants.sh): what the k loop costs
    const int kt0 = slice * per, kt1 = kt0;
#else
    const int kt0 = slice * per, kt1 = min(nkt, kt0 + per);
#endif

    XOperand A, B;
    xoperands(d, vecbits, A, B);
    XCtx c;
    c.m0 = tm * XBM; c.n0 = tn * XBN; c.kend = min(d.K, kt1 * YBK);
    c.tid = threadIdx.x;
    const int lane = c.tid & 63, wave = c.tid >> 6;
    c.wm = wave >> 1; c.wn = wave & 1; c.li = lane & 31; c.lk = lane >> 5;
    c.do_rowsum = d.a_rowsum != nullptr && tn == 0;
    const int m0 = c.m0, n0 = c.n0, tid = c.tid, wm = c.wm, wn = c.wn, li = c.li, lk = c.lk;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d.a_sk == 1) {
        if (d.b_sk == 1) ymain<true, true>(A, B, c, kt0, kt1, smem, acc, rs);
        else ymain<true, false>(A, B, c, kt0, kt1, smem, acc, rs);
    } else {
        if (d.b_sk == 1) ymain<false, true>(A, B, c, kt0, kt1, smem, acc, rs);
        else ymain<false, false>(A, B, c, kt0, kt1, smem, acc, rs);
    }

#ifdef SKG_XPROBE_NOEPI                                    // timing builds: what the epilogue costs
    if (acc[0][0][0] != 12345.678f) return;
#endif
    // ---- epilogue.  acc[mi][ni][4*gq + t] = row m0 + wm*64 + mi*32 + 8*gq + 4*lk + t, column n0 + wn*64 + ni*32 + li.
    const bool split = S > 1;
    const int64_t MN = (int64_t)d.M * d.N;
    float* ws = split ? d.split_ws + (int64_t)slice * MN : nullptr;           // [S][M * N] then [S][M] row sums
    const bool inl = split && d.split_ctr != nullptr;      // reduced in this launch by the tile's last arriver (staged epilogue only: the host checked)
    if (c.do_rowsum) {                                     // uniform per workgroup; ymain left the partial sums in LDS
        if (tid < XBM && m0 + tid < d.M) {
            const float* part = reinterpret_cast<const float*>(smem);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += part[q * 128 + tid];
            if (split) xstore_rowsum_part(d.split_ws + (int64_t)S * MN + (int64_t)slice * d.M + m0 + tid, s, inl);
            else d.a_rowsum[m0 + tid] = d.accumulate ? d.a_rowsum[m0 + tid] + s : s;
        }
        __syncthreads();                                   // read before the staged epilogue overwrites them
    }
    if (vecbits & 8) {
        float* stage = reinterpret_cast<float*>(smem) + wave * XEP_WAVE;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        stage[(8 * gq + 4 * lk + t) * XEP_LD + ni * 32 + li] = acc[mi][ni][4 * gq + t];
            if (inl) xep_rows<true>(d, g.f[gi], stage, lane, m0 + wm * 64 + mi * 32, n0 + wn * 64, ws);
            else xep_rows(d, g.f[gi], stage, lane, m0 + wm * 64 + mi * 32, n0 + wn * 64, ws);
        }
        if (inl && xsplit_last(d.split_ctr + tm * nbn + tn, S, reinterpret_cast<uint32_t*>(smem) + XEP_FLOATS)) {
            // the last slice of this tile to arrive: all slices in slice order + the epilogue
            if (c.do_rowsum && c.tid < XBM && m0 + c.tid < d.M) xreduce_rowsum_sc1(d, MN, m0 + c.tid);
            const XRed red = {d.split_ws, MN, S, slice};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            stage[(8 * gq + 4 * lk + t) * XEP_LD + ni * 32 + li] = acc[mi][ni][4 * gq + t];
                xep_reduce<true, 8>(d, g.f[gi], stage, lane, m0 + wm * 64 + mi * 32, n0 + wn * 64, red);
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = m0 + wm * 64 + mi * 32 + 8 * gq + 4 * lk + t;
                if (row >= d.M) continue;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = n0 + wn * 64 + ni * 32 + li;
                    if (col >= d.N) continue;
                    float v = acc[mi][ni][4 * gq + t];
                    if (split) { ws[(int64_t)row * d.N + col] = v; continue; }
                    if (d.bias) v += d.bias[col];
                    if (d.relu) v = fmaxf(v, 0.f);
                    float* p = d.C + xoff(col, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
                    if (d.accumulate) v += *p;
                    if (d.mask && !(d.mask[(int64_t)row * d.ldmask + col] > 0.f)) v = 0.f;
                    *p = v;
                    if (d.C16) d.C16[p - d.C] = (uint16_t)ypack(v, 0.f);
                }
            }
}

// ================================================================================================ bf16 twins, direct to LDS
// skg_gemmx_t16_kernel: the same products when BOTH operands come with their bf16 twins (skg_gemmx_desc.A16 / B16) -- what
// the training plan hands over once its producer kernels and the optimizer write twins.  Nothing passes through VGPRs on the
// way in: the operand tiles are staged by global_load_lds_dwordx4 straight into a ring of TNB LDS buffers (TNB - 1 tiles in
// flight across the one barrier of a k-step, counted s_waitcnt vmcnt), there is no conversion and no ds_write in the loop,
// and the k-step is 64 deep -- the loop the round-3 decomposition asked for (global loads, conversion + ds_write and
// fragment reads + MFMAs of the register-staged loop ran one after the other: ~1000 cycles per 32 k for 256 of MFMA).
//
// LDS images (16 KiB per operand tile, lane-linear per wave instruction as the DMA requires; swizzles live on the SOURCE side):
//   k-contiguous operand  : [128 rows][64 k] bf16, 128-byte rows; 16-byte slot s of row r holds k-piece s ^ ((r >> 1) & 7):
//                           the ds_read_b128 of a fragment (32 rows, one k-piece) is conflict-free
//   row-contiguous operand: [64 k][128 rows] bf16, 256-byte rows; slot s of k-row k holds the 8-row chunk
//                           s ^ (((k & 3) << 2) | ((k >> 2) & 3)); fragments come out by ds_read_b64_tr_b16 (the hardware
//                           transpose read: 16 lanes fetch a 4 k x 16 row block, each receives its row's 4 k) -- no operand
//                           is ever transposed in registers or memory
// A ragged last k-tile (K or the split slice not a multiple of 64) is filled through registers, zero beyond the end, into the
// same images.  The bias gradient (row sums of A over k) is one more MFMA per A fragment against a fragment of ones in the
// workgroups of column tile 0: the fp32 sum of the bf16-rounded operand -- what the reference's autocast backward sums
// (grad_output is bf16 there).  Same tiles (128 x 128, 4 waves of 64 x 64), same split-K, same epilogues as above.
#define TBK 64
// TNB = LDS buffers of the ring (TNB - 1 tiles in flight per workgroup).  Measured (tools/t16_time.sh, profiles/r04_gemmx_t16_*):
// 3 and 4 buffers at one workgroup per CU time the same -- the loop then runs at the L2 -> CU rate, ~1250 cycles per 64-k step
// for the 32 KiB a 128 x 128 tile draws per step -- while 2 buffers at TWO workgroups per CU (64 KiB of LDS each) are as fast on
// one tile per CU and 1.4x faster once a CU has several tiles (N = 4096: 76 -> 53 us; M = 102400: 523 -> 364 us): the other
// workgroup's loads, and its epilogue, run under this one's MFMAs.
#ifndef TNB
#define TNB 2
#endif
#define TOPB (128 * TBK * 2)                 // bytes of one operand tile
#define TBUFB (2 * TOPB)                     // A | B

typedef short t16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tswz_rc(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

// Pins a wave-uniform pointer in SGPRs so that `base + per-lane 32-bit offset` selects the saddr + voffset form.
__device__ __forceinline__ const char* t_uniform_ptr(const char* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

struct TLoad { uint32_t o[4]; };             // per-lane byte offsets of the four 16-byte pieces a thread stages per tile

template <bool KC>
__device__ __forceinline__ void tprep(const XOperand& op, int row0, int tid, TLoad& L) {
    const int w = tid >> 6, l = tid & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (KC) {                            // chunk 4 w + i = rows 32 w + 8 i .. + 7; lane: row l >> 3, slot l & 7
            const int r = 32 * w + 8 * i + (l >> 3);
            const int q = (l & 7) ^ ((r >> 1) & 7);
            L.o[i] = (uint32_t)((xoff(min(row0 + r, op.rows - 1), op.rshift, op.rstride, op.s_row16) + 8 * q) * 2);
        } else {                             // chunk 4 w + i = k-rows 16 w + 4 i .. + 3; lane: k-row l >> 4, slot l & 15
            const int k = 16 * w + 4 * i + (l >> 4);
            const int ch = (l & 15) ^ tswz_rc(k);
            const int64_t ro = xoff(max(0, min(row0 + 8 * ch, ((op.rows + 7) & ~7) - 8)), op.rshift, op.rstride, 1);
            L.o[i] = (uint32_t)((ro + (int64_t)k * op.s_k16) * 2);
        }
    }
}

// the four DMA instructions of one operand tile of this wave: 4 x 1 KiB, LDS destination wave-uniform
template <bool KC>
__device__ __forceinline__ void tissue(const XOperand& op, const uint16_t* base16, const TLoad& L, int k0, uint8_t* dst, int wu) {
    const char* gb = t_uniform_ptr(reinterpret_cast<const char*>(base16) +
                                     2 * xoff(k0, op.kshift, op.kstride, KC ? 1 : op.s_k16));
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + L.o[i]),
                                         (__attribute__((address_space(3))) void*)(dst + (4 * wu + i) * 1024), 16, 0, 0);
}

// ragged tile through registers: the thread's four pieces, zero where k >= kend, into the image the DMA would have written
template <bool KC>
__device__ __forceinline__ void tfill(const XOperand& op, const uint16_t* base16, int row0, int k0, int kend, int tid,
                                      uint8_t* dst) {
    const int w = tid >> 6, l = tid & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (KC) {
            const int r = 32 * w + 8 * i + (l >> 3);
            const int q = (l & 7) ^ ((r >> 1) & 7);
            const int k = k0 + 8 * q;
            const uint16_t* p = base16 + xoff(min(row0 + r, op.rows - 1), op.rshift, op.rstride, op.s_row16) +
                                xoff(k, op.kshift, op.kstride, 1);
            if (k + 7 < kend) v = *reinterpret_cast<const uint4*>(p);
            else if (k < kend) {
                uint16_t e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = k + j < kend ? p[j] : (uint16_t)0;
                v = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16),
                               e[6] | ((uint32_t)e[7] << 16));
            }
        } else {
            const int kl = 16 * w + 4 * i + (l >> 4);
            const int ch = (l & 15) ^ tswz_rc(kl);
            if (k0 + kl < kend)
                v = *reinterpret_cast<const uint4*>(base16 + xoff(max(0, min(row0 + 8 * ch, ((op.rows + 7) & ~7) - 8)), op.rshift, op.rstride, 1) +
                                                    xoff(k0 + kl, op.kshift, op.kstride, op.s_k16));
        }
        *reinterpret_cast<uint4*>(dst + (4 * w + i) * 1024 + l * 16) = v;
    }
}

// ---- fragment reads.  Written as inline asm: for a C++ LDS read hipcc orders the access behind every LDS-DMA still in flight
// (s_waitcnt vmcnt(0) in front of the first ds_read of the step -- seen in the ISA), which would drain the ring every step.
// The asm reads are ordered by hand: the step's counted vmcnt + barrier before them, counted lgkmcnt waits (tied to the
// fragment registers, so that the MFMAs cannot move above them) after.
template <int OFF>
__device__ __forceinline__ bf16x8 t_ds_b128(uint32_t addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ t16x4 t_ds_tr(uint32_t addr) {      // (EXEC is all ones here: no divergence around the k loop)
    t16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// Per-lane LDS byte offsets of an operand's fragments inside its tile (loop invariant).
//   k-contiguous : f[ks] = address of k-step ks for the first 32-row block; the second block is + 4096 (32 rows x 128 B)
//   row-contig.  : f[2 t + j] = address of block t (32 rows), k half j (4 k) for k-step 0; k-step ks is + 4096 (16 k-rows x 256 B)
struct TFrag { uint32_t f[4]; };
template <bool KC>
__device__ __forceinline__ void tfrag_prep(int rb, int lane, TFrag& F) {
    if (KC) {
        const int r = rb + (lane & 31), sw = (r >> 1) & 7, h = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) F.f[ks] = (uint32_t)(r * 128 + (((2 * ks + h) ^ sw) << 4));
    } else {
        // group g = lane >> 4 of 16 lanes reads the block k = 8 (g >> 1) + 4 j .. + 3, rows rb + 32 t + 16 (g & 1) .. + 15; lane
        // 4 q + p of the group addresses k-row q, 8-byte piece p of the block's 32 bytes per k-row
        const int g = lane >> 4, i = lane & 15, pq = i >> 2, pp = i & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ch = ((rb + 32 * t + 16 * (g & 1)) >> 3) + (pp >> 1);
                const int k = 8 * (g >> 1) + 4 * j + pq;
                F.f[2 * t + j] = (uint32_t)(k * 256 + ((ch ^ tswz_rc(k)) << 4) + 8 * (pp & 1));
            }
    }
}

// Both fragments (32-row blocks 0 and 1) of k-step KS exactly as the asm reads deliver them: nothing may touch these registers
// (not even the copy that joins two 64-bit halves) before the counted wait that ties them.
template <bool KC> struct TPair;
template <> struct TPair<true> {
    bf16x8 v[2];
    __device__ __forceinline__ bf16x8 get(int t) const { return v[t]; }
};
template <> struct TPair<false> {
    t16x4 lo[2], hi[2];
    __device__ __forceinline__ bf16x8 get(int t) const { return __builtin_shufflevector(lo[t], hi[t], 0, 1, 2, 3, 4, 5, 6, 7); }
};
template <int KS>
__device__ __forceinline__ void tread(uint32_t base, const TFrag& F, TPair<true>& P) {
    P.v[0] = t_ds_b128<0>(base + F.f[KS]);
    P.v[1] = t_ds_b128<4096>(base + F.f[KS]);
}
template <int KS>
__device__ __forceinline__ void tread(uint32_t base, const TFrag& F, TPair<false>& P) {
    P.lo[0] = t_ds_tr<4096 * KS>(base + F.f[0]); P.hi[0] = t_ds_tr<4096 * KS>(base + F.f[1]);
    P.lo[1] = t_ds_tr<4096 * KS>(base + F.f[2]); P.hi[1] = t_ds_tr<4096 * KS>(base + F.f[3]);
}
// LEFT: LDS instructions that may still be outstanding (-1: no wait, only the tie)
template <int LEFT>
__device__ __forceinline__ void ttie(TPair<true>& P) {
    if (LEFT < 0) asm volatile("" : "+v"(P.v[0]), "+v"(P.v[1]));
    else if (LEFT == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.v[0]), "+v"(P.v[1]));
    else if (LEFT == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(P.v[0]), "+v"(P.v[1]));
    else if (LEFT == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(P.v[0]), "+v"(P.v[1]));
    else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(P.v[0]), "+v"(P.v[1]));
}
template <int LEFT>
__device__ __forceinline__ void ttie(TPair<false>& P) {
    if (LEFT < 0) asm volatile("" : "+v"(P.lo[0]), "+v"(P.hi[0]), "+v"(P.lo[1]), "+v"(P.hi[1]));
    else if (LEFT == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(P.lo[0]), "+v"(P.hi[0]), "+v"(P.lo[1]), "+v"(P.hi[1]));
    else if (LEFT == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(P.lo[0]), "+v"(P.hi[0]), "+v"(P.lo[1]), "+v"(P.hi[1]));
    else if (LEFT == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(P.lo[0]), "+v"(P.hi[0]), "+v"(P.lo[1]), "+v"(P.hi[1]));
    else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(P.lo[0]), "+v"(P.hi[0]), "+v"(P.lo[1]), "+v"(P.hi[1]));
}

// One 64-deep k-step from the tile pair at LDS byte address `buf` (A at + 0, B at + TOPB).  Reads of k-step ks + 2 are issued
// before the MFMAs of k-step ks: at most two k-steps of reads are outstanding (the LGKM counter holds 15), and the LDS pipe
// works while the matrix pipe does.
template <bool AK, bool BK_>
__device__ __forceinline__ void tstep(uint32_t buf, const TFrag& FA, const TFrag& FB, f32x16 (&acc)[2][2], f32x16 (&rsa)[2],
                                      bool rowsum) {
    constexpr int NR = (AK ? 2 : 4) + (BK_ ? 2 : 4);          // LDS instructions per k-step
    TPair<AK> a[4];
    TPair<BK_> b[4];
    const uint32_t ab = buf, bb = buf + TOPB;
#define T_READ(KS) { tread<KS>(ab, FA, a[KS]); tread<KS>(bb, FB, b[KS]); }
#define T_WAIT(KS, LEFT) { ttie<(LEFT) ? NR : 0>(a[KS]); ttie<-1>(b[KS]); }
#define T_MFMA(KS)                                                                                                        \
    {                                                                                                                     \
        const bf16x8 fa0 = a[KS].get(0), fa1 = a[KS].get(1), fb0 = b[KS].get(0), fb1 = b[KS].get(1);                      \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb0, acc[0][0], 0, 0, 0);                                \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, fb1, acc[0][1], 0, 0, 0);                                \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb0, acc[1][0], 0, 0, 0);                                \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, fb1, acc[1][1], 0, 0, 0);                                \
        if (rowsum) {                                                                                                     \
            const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};                         \
            rsa[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0, ones, rsa[0], 0, 0, 0);                                 \
            rsa[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1, ones, rsa[1], 0, 0, 0);                                 \
        }                                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
    T_READ(0) T_READ(1)
    T_WAIT(0, 1) T_READ(2) T_MFMA(0)
    T_WAIT(1, 1) T_READ(3) T_MFMA(1)
    T_WAIT(2, 1) T_MFMA(2)
    T_WAIT(3, 0) T_MFMA(3)
#undef T_READ
#undef T_WAIT
#undef T_MFMA
}

template <bool AK, bool BK_>
__device__ __forceinline__ void tmain(const skg_gemmx_desc& d, const XOperand& A, const XOperand& B, const XCtx& c, int kt0,
                                      int kt1, uint8_t* smem, f32x16 (&acc)[2][2], f32x16 (&rsa)[2]) {
    const int tid = c.tid, lane = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: the DMA's LDS base stays scalar
    const bool rowsum = c.do_rowsum && c.wn == 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;   // LDS byte address of the ring
    TLoad LA, LB;
    tprep<AK>(A, c.m0, tid, LA);
    tprep<BK_>(B, c.n0, tid, LB);
    TFrag FA, FB;
    tfrag_prep<AK>(c.wm * 64, lane, FA);
    tfrag_prep<BK_>(c.wn * 64, lane, FB);
    const int ktf = (c.kend == kt1 * TBK) ? kt1 : max(kt0, kt1 - 1);       // [kt0, ktf): whole tiles
    const int nt = ktf - kt0;
    const uint16_t* const a16 = d.A16;                     // (locals: no descriptor reloads inside the k loop)
    const uint16_t* const b16 = d.B16;
    auto issue = [&](int t, int buf) {
        uint8_t* dst = smem + buf * TBUFB;
        tissue<AK>(A, a16, LA, (kt0 + t) * TBK, dst, wu);
        tissue<BK_>(B, b16, LB, (kt0 + t) * TBK, dst + TOPB, wu);
    };
    static_assert(TNB >= 2 && TNB <= 4, "the waits below count TNB - 1 tiles in flight");
#pragma unroll
    for (int t = 0; t < TNB - 1; ++t)
        if (t < nt) issue(t, t);
    int cur = 0;                                           // buffer of tile t; tile t + TNB - 1 goes to the one before it in the ring
    for (int t = 0; t < nt; ++t) {
        // tile t has landed once all but the newer tiles' DMA instructions of THIS wave (eight per tile) are done ...
        const int ahead = nt - 1 - t;                      // tiles issued after tile t
        if (TNB == 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (TNB >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and every wave has passed this barrier; behind it nobody reads buffer (t - 1) % TNB any more, which is where
        // tile t + TNB - 1 goes
        asm volatile("s_barrier" ::: "memory");
        const int prev = cur == 0 ? TNB - 1 : cur - 1;
        if (t + TNB - 1 < nt) issue(t + TNB - 1, prev);
        tstep<AK, BK_>(lds0 + cur * TBUFB, FA, FB, acc, rsa, rowsum);
        cur = cur + 1 == TNB ? 0 : cur + 1;
    }
    if (kt1 > ktf) {                                       // the ragged last tile of the slice
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint8_t* dst = smem + cur * TBUFB;                 // (last read TNB steps ago)
        tfill<AK>(A, d.A16, c.m0, ktf * TBK, c.kend, tid, dst);
        tfill<BK_>(B, d.B16, c.n0, ktf * TBK, c.kend, tid, dst + TOPB);
        __syncthreads();
        tstep<AK, BK_>(lds0 + cur * TBUFB, FA, FB, acc, rsa, rowsum);
    }
    __syncthreads();                                       // the staged epilogue reuses the buffers
}

__global__ __launch_bounds__(256, TNB == 2 ? 2 : 1) void skg_gemmx_t16_kernel(const skg_gemmx_group g) {
    constexpr int TSMEM = TNB * TBUFB > 4 * XEP_FLOATS + 16 ? TNB * TBUFB : 4 * XEP_FLOATS + 16;   // ring, then the staged epilogue + flag
    __shared__ __attribute__((aligned(1024))) uint8_t smem[TSMEM];
#ifdef SKG_XPROBE_STAMPS                                    // timing builds: wall-clock stamps (100 MHz) of the workgroup's phases
    const uint64_t stamp0 = wall_clock64();
#endif
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int vecbits = g.vec[gi];
    const int S = d.split_k > 1 ? d.split_k : 1;
    const int nbn = (d.N + XBN - 1) / XBN;
    const XTileId tid3 = xtile_of(blockIdx.x - g.start[gi], g.start[gi + 1] - g.start[gi], (d.M + XBM - 1) / XBM, nbn, S, d.K);
    const int slice = tid3.slice, tn = tid3.tn, tm = tid3.tm;
    const int nkt = (d.K + TBK - 1) / TBK;
    const int per = (nkt + S - 1) / S;
#ifdef SKG_XPROBE_NOLOOP                                   // timing builds (tools/build_gemmx_variants.sh): what the k loop costs
    const int kt0 = min(nkt, slice * per), kt1 = kt0;
#else
    const int kt0 = min(nkt, slice * per), kt1 = min(nkt, kt0 + per);
#endif

    XOperand A, B;
    xoperands(d, vecbits, A, B);
    XCtx c;
    c.m0 = tm * XBM; c.n0 = tn * XBN; c.kend = min(d.K, kt1 * TBK);
    c.tid = threadIdx.x;
    const int lane = c.tid & 63, wave = c.tid >> 6;
    c.wm = wave >> 1; c.wn = wave & 1; c.li = lane & 31; c.lk = lane >> 5;
    c.do_rowsum = d.a_rowsum != nullptr && tn == 0;
    const int m0 = c.m0, n0 = c.n0, wm = c.wm, wn = c.wn, li = c.li, lk = c.lk;

    f32x16 acc[2][2], rsa[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        acc[0][0][e] = 0.f; acc[0][1][e] = 0.f; acc[1][0][e] = 0.f; acc[1][1][e] = 0.f; rsa[0][e] = 0.f; rsa[1][e] = 0.f;
    }
    if (d.a_sk == 1) {
        if (d.b_sk == 1) tmain<true, true>(d, A, B, c, kt0, kt1, smem, acc, rsa);
        else tmain<true, false>(d, A, B, c, kt0, kt1, smem, acc, rsa);
    } else {
        if (d.b_sk == 1) tmain<false, true>(d, A, B, c, kt0, kt1, smem, acc, rsa);
        else tmain<false, false>(d, A, B, c, kt0, kt1, smem, acc, rsa);
    }

#ifdef SKG_XPROBE_STAMPS
    const uint64_t stamp1 = wall_clock64();
#endif
#ifdef SKG_XPROBE_NOEPI                                    // timing builds: what the epilogue costs
    if (acc[0][0][0] != 12345.678f) return;
#endif
    // ---- epilogue.  acc[mi][ni][4*gq + t] = row m0 + wm*64 + mi*32 + 8*gq + 4*lk + t, column n0 + wn*64 + ni*32 + li.
    const bool split = S > 1;
    const int64_t MN = (int64_t)d.M * d.N;
    float* ws = split ? d.split_ws + (int64_t)slice * MN : nullptr;           // [S][M * N] then [S][M] row sums
    const bool inl = split && d.split_ctr != nullptr;      // reduced in this launch by the tile's last arriver (staged epilogue only: the host checked)
    if (c.do_rowsum && wn == 0 && li == 0) {               // every column of rsa holds the row sums: column 0's lanes write
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + mi * 32 + 8 * (e >> 2) + 4 * lk + (e & 3);
                if (row >= d.M) continue;
                if (split) xstore_rowsum_part(d.split_ws + (int64_t)S * MN + (int64_t)slice * d.M + row, rsa[mi][e], inl);
                else d.a_rowsum[row] = d.accumulate ? d.a_rowsum[row] + rsa[mi][e] : rsa[mi][e];
            }
    }
    if (vecbits & 8) {
        float* stage = reinterpret_cast<float*>(smem) + wave * XEP_WAVE;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        stage[(8 * gq + 4 * lk + t) * XEP_LD + ni * 32 + li] = acc[mi][ni][4 * gq + t];
            if (inl) xep_rows<true>(d, g.f[gi], stage, lane, m0 + wm * 64 + mi * 32, n0 + wn * 64, ws);
            else xep_rows(d, g.f[gi], stage, lane, m0 + wm * 64 + mi * 32, n0 + wn * 64, ws);
        }
        if (inl && xsplit_last(d.split_ctr + tm * nbn + tn, S, reinterpret_cast<uint32_t*>(smem) + XEP_FLOATS)) {
            // the last slice of this tile to arrive: all slices in slice order + the epilogue
            if (c.do_rowsum && c.tid < XBM && m0 + c.tid < d.M) xreduce_rowsum_sc1(d, MN, m0 + c.tid);
            const XRed red = {d.split_ws, MN, S, slice};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            stage[(8 * gq + 4 * lk + t) * XEP_LD + ni * 32 + li] = acc[mi][ni][4 * gq + t];
                xep_reduce<true, 8>(d, g.f[gi], stage, lane, m0 + wm * 64 + mi * 32, n0 + wn * 64, red);
            }
        }
#ifdef SKG_XPROBE_STAMPS                                   // (split_ws of an unsplit product doubles as the stamp buffer: 4 x u64 per workgroup)
        if (!split && d.split_ws && c.tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint64_t* o = reinterpret_cast<uint64_t*>(d.split_ws) + 4 * (uint64_t)blockIdx.x;
            o[0] = stamp0; o[1] = stamp1; o[2] = wall_clock64(); o[3] = __smid();
        }
#endif
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = m0 + wm * 64 + mi * 32 + 8 * gq + 4 * lk + t;
                if (row >= d.M) continue;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int col = n0 + wn * 64 + ni * 32 + li;
                    if (col >= d.N) continue;
                    float v = acc[mi][ni][4 * gq + t];
                    if (split) { ws[(int64_t)row * d.N + col] = v; continue; }
                    if (d.bias) v += d.bias[col];
                    if (d.relu) v = fmaxf(v, 0.f);
                    float* p = d.C + xoff(col, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
                    if (d.accumulate) v += *p;
                    if (d.mask && !(d.mask[(int64_t)row * d.ldmask + col] > 0.f)) v = 0.f;
                    *p = v;
                    if (d.C16) d.C16[p - d.C] = (uint16_t)ypack(v, 0.f);
                }
            }
}

// (An exact-fp32 twin of this kernel -- skg_gemmx_d32_kernel: 32-deep k-steps, ds_read_b128 / ds_read_b32 fragments -- was built,
// tested on every layout and measured in round 4: not faster than skg_gemmx_kernel, dW 1.3-1.6x slower.  The register-staged fp32
// loop is not staging-bound.  Removed again; the record is profiles/r04_fp32_direct_lds_experiment.txt.)

// Adds the split-K slices in slice order and applies the epilogue.  One thread per output element (coalesced along n).
__global__ __launch_bounds__(256) void skg_gemmx_reduce_kernel(const skg_gemmx_group g) {
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int64_t MN = (int64_t)d.M * d.N;
    const int64_t w = (int64_t)(blockIdx.x - g.start[gi]) * 256 + threadIdx.x;
    const bool quads = g.vec[gi] & 8;                     // four columns per thread, 16-byte loads and stores
    const int64_t nmat = quads ? MN / 4 : MN;
    if (w >= nmat) {                                       // bias gradient (row sums of A)
        const int64_t m = w - nmat;
        if (m >= d.M || !d.a_rowsum) return;
        xreduce_rowsum(d, MN, (int)m);
        return;
    }
    if (quads) {
        const int64_t i = 4 * w;
        float4 v = xzero4();
        for (int s = 0; s < d.split_k; ++s) {
            const float4 q = xld4(d.split_ws + (int64_t)s * MN + i);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        const int row = (int)(i / d.N), col = (int)(i % d.N);
        if (d.bias) { const float4 bv = xld4(d.bias + col); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
        if (d.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        float* p = d.C + xoff(col, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
        if (d.accumulate) { const float4 o = xld4(p); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        if (d.mask) {                                      // accumulate first, mask last
            const float4 m = xld4(d.mask + (int64_t)row * d.ldmask + col);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<float4*>(p) = v;
        if (d.C16) *reinterpret_cast<uint2*>(d.C16 + (p - d.C)) = make_uint2(ypack(v.x, v.y), ypack(v.z, v.w));
        return;
    }
    xreduce_elem(d, MN, (int)(w / d.N), (int)(w % d.N));
}

static int skg_gemmx_validate(const skg_gemmx_desc& d) {
    if (d.M < 0 || d.N < 0 || d.K < 0) return SKG_E_ARG;
    if (d.M == 0 || d.N == 0) return 0;
    if (!d.C || d.ldc < 1) return SKG_E_ARG;
    if (d.K > 0 && (!d.A || !d.B)) return SKG_E_ARG;
    if (d.a_sm != 1 && d.a_sk != 1) return SKG_E_ARG;
    if (d.b_sn != 1 && d.b_sk != 1) return SKG_E_ARG;
    if (d.b_kshift < 0 || d.b_kshift > 30 || d.b_nshift < 0 || d.b_nshift > 30 || d.c_nshift < 0 || d.c_nshift > 30)
        return SKG_E_ARG;
    if ((d.b_kshift == 1) || (d.b_nshift == 1)) return SKG_E_ARG;            // blocks of at least 4 (16-byte quads)
    if (d.mask && d.ldmask < d.N) return SKG_E_ARG;
    if (d.split_k > 1 && (!d.split_ws || d.split_k > 256)) return SKG_E_ARG;
    if ((((uintptr_t)d.C16) & 1u) || (((uintptr_t)d.A16) & 1u) || (((uintptr_t)d.B16) & 1u)) return SKG_E_ALIGN;
    if (((uintptr_t)d.split_ctr) & 3u) return SKG_E_ALIGN;
    if (d.a16_ld < 0 || d.b16_ld < 0 || (d.b16_ld && (d.b_kshift || d.b_nshift))) return SKG_E_ARG;
    if (d.a16_ld && d.a16_ld < (d.a_sk == 1 ? d.K : d.M)) return SKG_E_ARG;
    if (d.b16_ld && d.b16_ld < (d.b_sk == 1 ? d.K : d.N)) return SKG_E_ARG;
    return 0;
}

static bool xmul4(int64_t v) { return (v & 3) == 0; }
static bool xmul8(int64_t v) { return (v & 7) == 0; }

// Largest element offset an operand reaches (host copy of xoff): the fast loop keeps 32-bit byte offsets per lane.
static int64_t xspan(int64_t idx, int shift, int64_t bstride, int64_t estride) {
    if (idx <= 0) return 0;
    if (shift <= 0) return idx * estride;
    return (idx >> shift) * bstride + (idx & ((1LL << shift) - 1)) * estride;
}
static bool xfits32(int64_t elems) { return elems >= 0 && elems < (1LL << 29); }     // < 2 GiB in bytes

extern "C" int64_t skg_gemmx_ws_floats(const skg_gemmx_desc* d) {
    if (!d) return SKG_E_ARG;
    return d->split_k > 1 ? (int64_t)d->split_k * ((int64_t)d->M * d->N + d->M) : 0;
}

static bool xfused_ptrs_ok(const skg_gemmx_fused& f) {
    return skg_aligned16(f.P) && skg_aligned16(f.Q) && skg_aligned16(f.mbias) && skg_aligned16(f.C_raw) &&
           skg_aligned16(f.res) && xmul4(f.ldp) && xmul4(f.ldq) && xmul4(f.ldc_raw) && xmul4(f.ldres);
}

// launches per main loop since the last reset (statistics for tests and profiles: which products reach the direct-to-LDS kernel)
static std::atomic<long long> g_path_launches[3];
extern "C" void skg_gemmx_path_counts(int64_t* out3_host, int reset) {
    for (int i = 0; i < 3; ++i) {
        if (out3_host) out3_host[i] = g_path_launches[i].load(std::memory_order_relaxed);
        if (reset) g_path_launches[i].store(0, std::memory_order_relaxed);
    }
}

static bool g_t16_enabled = getenv("SKG_GEMMX_T16") == nullptr || atoi(getenv("SKG_GEMMX_T16")) != 0;   // developer A/B switch

static int skg_gemmx_launch(const skg_gemmx_desc* descs_host, int n, void* stream, bool bf16,
                            const skg_gemmx_fused* fused_host = nullptr) {
    if (!descs_host || n < 1 || n > SKG_GEMMX_GROUP_MAX) return SKG_E_ARG;
    skg_gemmx_group g, r;
    memset(g.f, 0, sizeof(g.f)); memset(r.f, 0, sizeof(r.f));
    g.n = r.n = 0;
    int64_t blocks = 0, rblocks = 0;
    for (int i = 0; i < n; ++i) {
        const skg_gemmx_desc& d = descs_host[i];
        const int rc = skg_gemmx_validate(d);
        if (rc) return rc;
        if (d.M == 0 || d.N == 0) continue;
        const int S = d.split_k > 1 ? d.split_k : 1;
        const int64_t nb = (int64_t)((d.M + XBM - 1) / XBM) * ((d.N + XBN - 1) / XBN) * S;
        if (blocks + nb > 0x7fffffffLL) return SKG_E_LIMIT;
        int vec = 0;
        // bits 0 / 1 admit an operand to the fast loop: 16-byte loads, 32-bit byte offsets, k blocks of whole k-tiles
        if (skg_aligned16(d.A) && (d.a_sk == 1 ? xmul4(d.a_sm) : xmul4(d.a_sk)) &&
            xfits32((int64_t)(d.M - 1) * d.a_sm + (int64_t)(d.K - 1) * d.a_sk + 4))
            vec |= 1;
        if (skg_aligned16(d.B) && (d.b_sk == 1 ? xmul4(d.b_sn) : xmul4(d.b_sk)) &&
            (d.b_kshift == 0 || (xmul4(d.b_kstride) && d.b_kshift >= 5)) && (d.b_nshift == 0 || xmul4(d.b_nstride)) &&
            xfits32(xspan(d.N - 1, d.b_nshift, d.b_nstride, d.b_sn) + xspan(d.K - 1, d.b_kshift, d.b_kstride, d.b_sk) + 4))
            vec |= 2;
        if ((((uintptr_t)d.C) & 7u) == 0 && (d.ldc & 1) == 0 && (d.c_nshift == 0 || (d.c_nstride & 1) == 0)) vec |= 4;
        // bit 3: the staged epilogue / four-column reduce (16-byte accesses to C, bias, mask and the workspace rows)
        if ((d.N & 3) == 0 && skg_aligned16(d.C) && xmul4(d.ldc) &&
            (d.c_nshift == 0 || (d.c_nshift >= 2 && xmul4(d.c_nstride))) && skg_aligned16(d.bias) &&
            skg_aligned16(d.mask) && (!d.mask || xmul4(d.ldmask)) && (S == 1 || skg_aligned16(d.split_ws)) &&
            (((uintptr_t)d.C16) & 7u) == 0)
            vec |= 8;
        // bits 4 / 5: the bf16 twin of A / B may feed the fast loop (16-byte pieces of 8 elements)
        // (a twin with a leading dimension of its own -- a16_ld / b16_ld: a PADDED copy of an operand whose fp32 rows are not
        //  16-byte multiples -- does not need bits 0 / 1: only the direct-to-LDS kernel reads it, and it reads nothing else)
        const int64_t lda16 = d.a16_ld ? d.a16_ld : (d.a_sk == 1 ? d.a_sm : d.a_sk);
        const int64_t ldb16 = d.b16_ld ? d.b16_ld : (d.b_sk == 1 ? d.b_sn : d.b_sk);
        if (bf16 && d.A16 && ((vec & 1) || d.a16_ld) && skg_aligned16(d.A16) && xmul8(lda16) &&
            xfits32((int64_t)(d.a_sk == 1 ? d.M : d.K) * lda16 + 8))
            vec |= 16;
        if (bf16 && d.B16 && ((vec & 2) || d.b16_ld) && skg_aligned16(d.B16) && xmul8(ldb16) &&
            (d.b_kshift == 0 || xmul8(d.b_kstride)) && (d.b_nshift == 0 || (d.b_nshift >= 3 && xmul8(d.b_nstride))) &&
            (!d.b16_ld || xfits32((int64_t)(d.b_sk == 1 ? d.N : d.K) * ldb16 + 8)))
            vec |= 32;
        if (fused_host && (fused_host[i].kind || fused_host[i].out_rows)) {
            // eval-path epilogues exist in the staged epilogue only, and not behind a split-K reduce
            if (!(vec & 8) || (S > 1 && !(d.split_ctr && (int64_t)d.M * d.N < (1LL << 29))) || d.c_nshift || d.accumulate || d.mask || d.C16 ||
                !xfused_ptrs_ok(fused_host[i]))
                return SKG_E_ARG;
            g.f[g.n] = fused_host[i];
        }
        g.d[g.n] = d; g.vec[g.n] = vec; g.start[g.n] = (int)blocks;
        // (16-byte write-through partials, addressed with 32-bit byte offsets inside a slice; anything else: the reduce launch)
        const bool inlaunch = S > 1 && d.split_ctr && (vec & 8) && (int64_t)d.M * d.N < (1LL << 29);
        if (!inlaunch) g.d[g.n].split_ctr = nullptr;
        ++g.n;
        blocks += nb;
        if (S > 1 && !inlaunch) {                          // (with counters the slices are reduced inside the product launch)
            const int64_t MN = (int64_t)d.M * d.N;
            const int64_t total = ((vec & 8) ? MN / 4 : MN) + (d.a_rowsum ? d.M : 0);
            const int64_t nr = (total + 255) / 256;
            if (rblocks + nr > 0x7fffffffLL) return SKG_E_LIMIT;
            r.d[r.n] = d; r.vec[r.n] = vec; r.start[r.n] = (int)rblocks; ++r.n;
            rblocks += nr;
        }
    }
    if (g.n == 0) return 0;
    for (int i = g.n; i <= SKG_GEMMX_GROUP_MAX; ++i) g.start[i] = (int)blocks;
    // every product of the launch with both twins, whole 8-row pieces and k blocks of whole tiles: the direct-to-LDS kernel
    bool t16 = bf16 && g_t16_enabled;
    for (int i = 0; i < g.n && t16; ++i) {
        const skg_gemmx_desc& d = g.d[i];
        // row-contiguous twins are staged in 8-row pieces: whole pieces, or a k-stride that leaves room for the last one
        // (the rows past the end only feed outputs that are never stored)
        const int64_t lda16 = d.a16_ld ? d.a16_ld : d.a_sk, ldb16 = d.b16_ld ? d.b16_ld : d.b_sk;
        t16 = (g.vec[i] & 48) == 48 && d.K > 0 && (d.a_sk == 1 || (d.M & 7) == 0 || lda16 >= ((d.M + 7) & ~7)) &&
              (d.b_sk == 1 || (d.b_nshift == 0 ? ((d.N & 7) == 0 || ldb16 >= ((d.N + 7) & ~7)) : (d.N & 7) == 0)) &&
              (d.b_kshift == 0 || d.b_kshift >= 6);
    }
    if (!t16)                                              // the register-staged loops index a twin like its fp32 array
        for (int i = 0; i < g.n; ++i) {
            if (g.d[i].a16_ld) g.vec[i] &= ~16;
            if (g.d[i].b16_ld) g.vec[i] &= ~32;
        }
    g_path_launches[t16 ? 2 : (bf16 ? 1 : 0)].fetch_add(1, std::memory_order_relaxed);
    if (t16) hipLaunchKernelGGL(skg_gemmx_t16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else if (bf16) hipLaunchKernelGGL(skg_gemmx_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(skg_gemmx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    if (r.n) {
        for (int i = r.n; i <= SKG_GEMMX_GROUP_MAX; ++i) r.start[i] = (int)rblocks;
        hipLaunchKernelGGL(skg_gemmx_reduce_kernel, dim3((unsigned)rblocks), dim3(256), 0, (hipStream_t)stream, r);
    }
    return skg_launch_status();
}

extern "C" int skg_gemmx_f32(const skg_gemmx_desc* descs_host, int n, void* stream) {
    return skg_gemmx_launch(descs_host, n, stream, false);
}

int skg_gemmx_f32_fused(const skg_gemmx_desc* descs_host, const skg_gemmx_fused* fused_host, int n, void* stream) {
    return skg_gemmx_launch(descs_host, n, stream, false, fused_host);
}

int skg_gemmx_can_fuse(const skg_gemmx_desc* dp, const skg_gemmx_fused* f) {
    const skg_gemmx_desc& d = *dp;
    if (skg_gemmx_validate(d)) return 0;
    const bool staged = (d.N & 3) == 0 && skg_aligned16(d.C) && xmul4(d.ldc) && d.c_nshift == 0 && skg_aligned16(d.bias);
    if (!f || !(f->kind || f->out_rows)) return 1;
    return staged && (d.split_k <= 1 || d.split_ctr) && !d.accumulate && !d.mask && xfused_ptrs_ok(*f);
}

extern "C" int skg_gemmx_bf16(const skg_gemmx_desc* descs_host, int n, void* stream) {
    return skg_gemmx_launch(descs_host, n, stream, true);
}
