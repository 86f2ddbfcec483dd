// skg_gemm_x.hip -- fp32 MFMA GEMM with free operand layouts, for the TRAINING step of the interaction head.
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
// Design (gfx950): 128x128x16 block tile, 4 waves (2x2), each 64x64 = 2x2 v_mfma_f32_32x32x2_f32 tiles (exact fp32,
// bit-for-bit an fmaf chain).  Both operand tiles live in LDS K-MAJOR ([16 k][128 rows], row stride 130 dwords): a
// lane's fragment is ONE ds_read_b64 = two adjacent rows at its k (the rows a MFMA tile covers are interleaved -- the
// tile does not care which rows it is given -- so no operand is ever transposed on its way to the matrix core).
// Operands contiguous along their own index are copied into that image as they are (512-byte coalesced rows);
// operands contiguous along k are loaded as 16-byte k-quads and written transposed (bank-conflict free: 130 = 2 mod 32
// spreads the four k-quads of a row over four bank groups).  Double-buffered LDS, register prefetch of the next tile
// across the MFMA loop, one barrier per k-tile.  The fp32 MFMA is slow enough (64 cycles per 32x32x2) that LDS and
// VALU work hide behind it.
#include "skg_common.h"

#define XBM 128
#define XBN 128
#define XBK 16
#define XLD 130
#define XTILE (XBK * XLD)

struct skg_gemmx_group {
    skg_gemmx_desc d[SKG_GEMMX_GROUP_MAX];
    int start[SKG_GEMMX_GROUP_MAX + 1];      // block ranges
    int vec[SKG_GEMMX_GROUP_MAX];            // bit 0: A 16-byte loads allowed, bit 1: B, bit 2: C 8-byte stores allowed
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
    bool kcontig, vec;
};

// One thread's share of a 128 x 16 operand tile: two quads.  kcontig: quad = 4 consecutive k of one row; otherwise
// quad = 4 consecutive rows at one k.  Out-of-range elements read as zero.
__device__ __forceinline__ void xload(const XOperand& op, int row0, int k0, int kend, int tid, float4 (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u;
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (op.kcontig) {
            const int row = row0 + (f >> 2), k = k0 + 4 * (f & 3);
            if (row < op.rows && k < kend) {
                const float* p = op.base + xoff(row, op.rshift, op.rstride, op.s_row) + xoff(k, op.kshift, op.kstride, 1);
                if (op.vec && k + 3 < kend) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (k + c < kend) r[c] = p[c];
                }
            }
        } else {
            const int k = k0 + (f >> 5), row = row0 + 4 * (f & 31);
            if (k < kend && row < op.rows) {
                const float* p = op.base + xoff(k, op.kshift, op.kstride, op.s_k) + xoff(row, op.rshift, op.rstride, 1);
                if (op.vec && row + 3 < op.rows) {
                    const float4 t = *reinterpret_cast<const float4*>(p);
                    r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (row + c < op.rows) r[c] = p[c];
                }
            }
        }
        v[u] = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// Loop-invariant part of one thread's loads of one operand: the addresses along the operand's own index are formed
// once, the main loop adds one k offset per tile and skips every bounds check on tiles that lie inside the slice
// ("fast" tiles: all but possibly the last).  Quads that straddle the end of the operand take the generic path.
struct XLane {
    const float* p0; const float* p1; const float* p2; const float* p3;
    int ok;              // bit u: quad u lies inside the operand along its own index
    int straddle;        // row-contiguous quad crossing the operand's end
    int kloc;
};

__device__ __forceinline__ float4 xld4(const float* p, bool ok) {
    return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void xprep(const XOperand& op, int row0, int tid, XLane& L) {
    L.p1 = L.p2 = L.p3 = nullptr;
    if (op.kcontig) {
        const int r = row0 + (tid >> 2);
        L.kloc = 4 * (tid & 3);
        L.p0 = op.base + xoff(r, op.rshift, op.rstride, op.s_row);
        L.p1 = op.base + xoff(r + 64, op.rshift, op.rstride, op.s_row);
        L.ok = (r < op.rows ? 1 : 0) | (r + 64 < op.rows ? 2 : 0);
        L.straddle = 0;
    } else {
        const int row = row0 + 4 * (tid & 31);
        L.kloc = tid >> 5;
        L.p0 = op.base + xoff(row, op.rshift, op.rstride, 1);
        L.ok = row + 3 < op.rows ? 3 : 0;
        L.straddle = row < op.rows && row + 3 >= op.rows;
    }
}

// one 128 x 16 tile at k0 (fast when it lies inside [.., kend) and 16-byte loads are allowed)
__device__ __forceinline__ void xtile(const XOperand& op, const XLane& L, int row0, int k0, int kend, int tid,
                                      float4 (&v)[2]) {
    if (op.vec && k0 + XBK <= kend && !L.straddle) {
        if (op.kcontig) {
            const int64_t ko = xoff(k0 + L.kloc, op.kshift, op.kstride, 1);
            v[0] = xld4(L.p0 + ko, L.ok & 1);
            v[1] = xld4(L.p1 + ko, L.ok & 2);
        } else {
            v[0] = xld4(L.p0 + xoff(k0 + L.kloc, op.kshift, op.kstride, op.s_k), L.ok & 1);
            v[1] = xld4(L.p0 + xoff(k0 + L.kloc + 8, op.kshift, op.kstride, op.s_k), L.ok & 2);
        }
    } else {
        xload(op, row0, k0, kend, tid, v);
    }
}

__device__ __forceinline__ void xstore_lds(float* tile, bool kcontig, int tid, const float4 (&v)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int f = tid + 256 * u;
        if (kcontig) {
            const int r = f >> 2, kq = f & 3;
            float* p = tile + (4 * kq) * XLD + r;
            p[0] = v[u].x; p[XLD] = v[u].y; p[2 * XLD] = v[u].z; p[3 * XLD] = v[u].w;
        } else {
            const int kk = f >> 5, rq = f & 31;
            float2* p = reinterpret_cast<float2*>(tile + kk * XLD + 4 * rq);
            p[0] = make_float2(v[u].x, v[u].y); p[1] = make_float2(v[u].z, v[u].w);
        }
    }
}

__global__ __launch_bounds__(256, 2) void skg_gemmx_kernel(const skg_gemmx_group g) {
    __shared__ __attribute__((aligned(16))) float smem[4 * XTILE];      // A0 | B0 | A1 | B1
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int vecbits = g.vec[gi];
    const int S = d.split_k > 1 ? d.split_k : 1;
    const int nbn = (d.N + XBN - 1) / XBN;
    const int nbm = (d.M + XBM - 1) / XBM;
    int b = blockIdx.x - g.start[gi];
    const int slice = b % S; b /= S;
    const int tn = b % nbn, tm = b / nbn;                 // consecutive blocks walk N: they share the A panel in L2
    const int m0 = tm * XBM, n0 = tn * XBN;
    // k range of this slice, in whole k-tiles
    const int nkt = (d.K + XBK - 1) / XBK;
    const int per = (nkt + S - 1) / S;
    const int kt0 = slice * per, kt1 = min(nkt, kt0 + per);
    const int kend = min(d.K, kt1 * XBK);

    XOperand A, B;
    A.base = d.A; A.s_row = d.a_sm; A.s_k = d.a_sk; A.rshift = 0; A.kshift = 0; A.rstride = 0; A.kstride = 0;
    A.rows = d.M; A.kcontig = d.a_sk == 1; A.vec = vecbits & 1;
    B.base = d.B; B.s_row = d.b_sn; B.s_k = d.b_sk; B.rshift = d.b_nshift; B.kshift = d.b_kshift;
    B.rstride = d.b_nstride; B.kstride = d.b_kstride; B.rows = d.N; B.kcontig = d.b_sk == 1; B.vec = (vecbits >> 1) & 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool do_rowsum = d.a_rowsum != nullptr && tn == 0 && tid < XBM;
    float rsum = 0.f;

    XLane LA, LB;
    xprep(A, m0, tid, LA);
    xprep(B, n0, tid, LB);
    float4 va[2], vb[2];
    if (kt0 < kt1) {
        xtile(A, LA, m0, kt0 * XBK, kend, tid, va);
        xtile(B, LB, n0, kt0 * XBK, kend, tid, vb);
        xstore_lds(smem, A.kcontig, tid, va);
        xstore_lds(smem + XTILE, B.kcontig, tid, vb);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const float* As = smem + cur * 2 * XTILE;
        const float* Bs = As + XTILE;
        const bool more = kt + 1 < kt1;
        if (more) {
            xtile(A, LA, m0, (kt + 1) * XBK, kend, tid, va);
            xtile(B, LB, n0, (kt + 1) * XBK, kend, tid, vb);
        }
#pragma unroll
        for (int ks = 0; ks < XBK / 2; ++ks) {
            const int kk = 2 * ks + lk;
            const float2 a = *reinterpret_cast<const float2*>(As + kk * XLD + wm * 64 + 2 * li);
            const float2 bq = *reinterpret_cast<const float2*>(Bs + kk * XLD + wn * 64 + 2 * li);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.x, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq.y, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.x, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq.y, acc[1][1], 0, 0, 0);
        }
        if (do_rowsum) {
#pragma unroll
            for (int kk = 0; kk < XBK; ++kk) rsum += As[kk * XLD + tid];
        }
        if (more) {
            float* An = smem + (cur ^ 1) * 2 * XTILE;
            xstore_lds(An, A.kcontig, tid, va);
            xstore_lds(An + XTILE, B.kcontig, tid, vb);
        }
        __syncthreads();
    }

    // ---- epilogue.  Lane (li, lk) of wave (wm, wn) holds, in acc[mb][nb][4*gq + t], the element
    //      row m0 + wm*64 + 2*(8*gq + 4*lk + t) + mb,  column n0 + wn*64 + 2*li + nb.
    const bool split = S > 1;
    const int64_t MN = (int64_t)d.M * d.N;
    float* ws = split ? d.split_ws + (int64_t)slice * (MN + d.M) : nullptr;
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
                    *p2 = make_float2(k0 ? v0 : 0.f, k1 ? v1 : 0.f);
                } else {
                    if (d.accumulate) v0 += p[0];
                    p[0] = k0 ? v0 : 0.f;
                    if (two) {
                        float* q = d.C + xoff(col + 1, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
                        if (d.accumulate) v1 += q[0];
                        q[0] = k1 ? v1 : 0.f;
                    }
                }
            }
    if (do_rowsum && m0 + tid < d.M) {
        if (split) ws[MN + m0 + tid] = rsum;
        else d.a_rowsum[m0 + tid] = d.accumulate ? d.a_rowsum[m0 + tid] + rsum : rsum;
    }
}

// ------------------------------------------------------------------------------------------------ bf16 operands
// Same products, same descriptors, same epilogue; the operands stay fp32 in HBM and are rounded to bf16 (RNE,
// v_cvt_pk_bf16_f32) on their way into LDS, accumulation in fp32 on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
// This is what precision="bf16" training runs (BASELINE config 3: autocast of the reference's nn.Linear layers).
//
// 128x128x32 block tile, 4 waves (2x2) x (2x2) MFMA tiles.  LDS image of an operand tile: four k-planes (8 k each) of
// [128 rows][8 bf16 = 16 B], plane stride 520 dwords, row slot XOR-swizzled (slot = row ^ ((row >> 3) & 3)): a lane's
// MFMA fragment is ONE ds_read_b128, conflict-free, and both source layouts are written conflict-free too --
//   k-contiguous source : a thread has 4 k of one row        -> one ds_write_b64
//   row-contiguous source: a thread loads 4 rows x 4 k (four float4 along the rows), transposes in registers
//                                                             -> four ds_write_b64 (one per row)
// The bias gradient (row sums of A) is accumulated from the fp32 registers before rounding.
#define YBK 32
#define YPLANE 1040                          // bf16 elements per k-plane: 128 rows * 8 + 16 pad (520 dwords)
#define YTILE (4 * YPLANE)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf2 __attribute__((ext_vector_type(2)));
typedef float xf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t ypack(float a, float b) {
    const xf2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xbf2));
}
__device__ __forceinline__ int yslot(int row) { return row ^ ((row >> 3) & 3); }

// Four consecutive elements along the contiguous index `c` (extent cend) at fixed other index; zero outside.
__device__ __forceinline__ float4 yquad(const float* p, int c, int cend, bool vec) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && c + 3 < cend) {
        t = *reinterpret_cast<const float4*>(p);
    } else {
        if (c < cend) t.x = p[0];
        if (c + 1 < cend) t.y = p[1];
        if (c + 2 < cend) t.z = p[2];
        if (c + 3 < cend) t.w = p[3];
    }
    return t;
}

// One thread's share of a 128 x 32 operand tile: four quads (generic path: every bound checked).
//   kcontig : quad u = 4 consecutive k (k0 + 4 (tid & 7)) of row (tid >> 3) + 32 u
//   else    : quad u = rows 4 rq .. 4 rq + 3 at k = k0 + 4 kq4 + u,  rq = (tid & 7) | ((tid >> 4) & 3) << 3,
//             kq4 = ((tid >> 3) & 1) | (tid >> 6) << 1
__device__ __forceinline__ float4 yload1(const XOperand& op, int row0, int k0, int kend, int tid, int u) {
    if (op.kcontig) {
        const int row = row0 + (tid >> 3) + 32 * u, k = k0 + 4 * (tid & 7);
        if (row >= op.rows || k >= kend) return make_float4(0.f, 0.f, 0.f, 0.f);
        return yquad(op.base + xoff(row, op.rshift, op.rstride, op.s_row) + xoff(k, op.kshift, op.kstride, 1), k, kend, op.vec);
    }
    const int rq = (tid & 7) | (((tid >> 4) & 3) << 3), kq4 = ((tid >> 3) & 1) | ((tid >> 6) << 1);
    const int k = k0 + 4 * kq4 + u, row = row0 + 4 * rq;
    if (k >= kend || row >= op.rows) return make_float4(0.f, 0.f, 0.f, 0.f);
    return yquad(op.base + xoff(k, op.kshift, op.kstride, op.s_k) + xoff(row, op.rshift, op.rstride, 1), row, op.rows, op.vec);
}

#define YLOAD(op, row0, k0, v)                                                                            \
    do {                                                                                                  \
        v##0 = yload1(op, row0, k0, kend, tid, 0); v##1 = yload1(op, row0, k0, kend, tid, 1);            \
        v##2 = yload1(op, row0, k0, kend, tid, 2); v##3 = yload1(op, row0, k0, kend, tid, 3);            \
    } while (0)

__device__ __forceinline__ void yprep(const XOperand& op, int row0, int tid, XLane& L) {
    L.p1 = L.p2 = L.p3 = nullptr;
    if (op.kcontig) {
        const int r = row0 + (tid >> 3);
        L.kloc = 4 * (tid & 7);
        L.p0 = op.base + xoff(r, op.rshift, op.rstride, op.s_row);
        L.p1 = op.base + xoff(r + 32, op.rshift, op.rstride, op.s_row);
        L.p2 = op.base + xoff(r + 64, op.rshift, op.rstride, op.s_row);
        L.p3 = op.base + xoff(r + 96, op.rshift, op.rstride, op.s_row);
        L.ok = (r < op.rows ? 1 : 0) | (r + 32 < op.rows ? 2 : 0) | (r + 64 < op.rows ? 4 : 0) | (r + 96 < op.rows ? 8 : 0);
        L.straddle = 0;
    } else {
        const int rq = (tid & 7) | (((tid >> 4) & 3) << 3), kq4 = ((tid >> 3) & 1) | ((tid >> 6) << 1);
        const int row = row0 + 4 * rq;
        L.kloc = 4 * kq4;
        L.p0 = op.base + xoff(row, op.rshift, op.rstride, 1);
        L.ok = row + 3 < op.rows ? 15 : 0;
        L.straddle = row < op.rows && row + 3 >= op.rows;
    }
}

// one 128 x 32 tile at k0.  Fast tiles: one k offset per thread (a 4-aligned group of k never crosses a k block).
#define YTILE_LOAD(op, L, row0, k0, v)                                                                    \
    do {                                                                                                  \
        if (op.vec && (k0) + YBK <= kend && !L.straddle) {                                                \
            if (op.kcontig) {                                                                             \
                const int64_t ko_ = xoff((k0) + L.kloc, op.kshift, op.kstride, 1);                        \
                v##0 = xld4(L.p0 + ko_, L.ok & 1); v##1 = xld4(L.p1 + ko_, L.ok & 2);                     \
                v##2 = xld4(L.p2 + ko_, L.ok & 4); v##3 = xld4(L.p3 + ko_, L.ok & 8);                     \
            } else {                                                                                      \
                const float* q_ = L.p0 + xoff((k0) + L.kloc, op.kshift, op.kstride, op.s_k);             \
                v##0 = xld4(q_, L.ok & 1); v##1 = xld4(q_ + op.s_k, L.ok & 2);                            \
                v##2 = xld4(q_ + 2 * op.s_k, L.ok & 4); v##3 = xld4(q_ + 3 * op.s_k, L.ok & 8);           \
            }                                                                                             \
        } else {                                                                                          \
            YLOAD(op, row0, k0, v);                                                                       \
        }                                                                                                 \
    } while (0)

__device__ __forceinline__ void ystore_lds(uint16_t* tile, bool kcontig, int tid, float4 v0, float4 v1, float4 v2, float4 v3) {
    if (kcontig) {
        const int kq4 = tid & 7, r = tid >> 3;
        uint16_t* base = tile + (kq4 >> 1) * YPLANE + (kq4 & 1) * 4;
        *reinterpret_cast<uint2*>(base + yslot(r) * 8) = make_uint2(ypack(v0.x, v0.y), ypack(v0.z, v0.w));
        *reinterpret_cast<uint2*>(base + yslot(r + 32) * 8) = make_uint2(ypack(v1.x, v1.y), ypack(v1.z, v1.w));
        *reinterpret_cast<uint2*>(base + yslot(r + 64) * 8) = make_uint2(ypack(v2.x, v2.y), ypack(v2.z, v2.w));
        *reinterpret_cast<uint2*>(base + yslot(r + 96) * 8) = make_uint2(ypack(v3.x, v3.y), ypack(v3.z, v3.w));
    } else {
        const int rq = (tid & 7) | (((tid >> 4) & 3) << 3), kq4 = ((tid >> 3) & 1) | ((tid >> 6) << 1);
        uint16_t* base = tile + (kq4 >> 1) * YPLANE + (kq4 & 1) * 4;
        const int row = 4 * rq;
        *reinterpret_cast<uint2*>(base + yslot(row + 0) * 8) = make_uint2(ypack(v0.x, v1.x), ypack(v2.x, v3.x));
        *reinterpret_cast<uint2*>(base + yslot(row + 1) * 8) = make_uint2(ypack(v0.y, v1.y), ypack(v2.y, v3.y));
        *reinterpret_cast<uint2*>(base + yslot(row + 2) * 8) = make_uint2(ypack(v0.z, v1.z), ypack(v2.z, v3.z));
        *reinterpret_cast<uint2*>(base + yslot(row + 3) * 8) = make_uint2(ypack(v0.w, v1.w), ypack(v2.w, v3.w));
    }
}

__global__ __launch_bounds__(256, 2) void skg_gemmx_bf16_kernel(const skg_gemmx_group g) {
    __shared__ __attribute__((aligned(16))) uint16_t smem[4 * YTILE];      // A0 | B0 | A1 | B1
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int vecbits = g.vec[gi];
    const int S = d.split_k > 1 ? d.split_k : 1;
    const int nbn = (d.N + XBN - 1) / XBN;
    int b = blockIdx.x - g.start[gi];
    const int slice = b % S; b /= S;
    const int tn = b % nbn, tm = b / nbn;
    const int m0 = tm * XBM, n0 = tn * XBN;
    const int nkt = (d.K + YBK - 1) / YBK;
    const int per = (nkt + S - 1) / S;
    const int kt0 = slice * per, kt1 = min(nkt, kt0 + per);
    const int kend = min(d.K, kt1 * YBK);

    XOperand A, B;
    A.base = d.A; A.s_row = d.a_sm; A.s_k = d.a_sk; A.rshift = 0; A.kshift = 0; A.rstride = 0; A.kstride = 0;
    A.rows = d.M; A.kcontig = d.a_sk == 1; A.vec = vecbits & 1;
    B.base = d.B; B.s_row = d.b_sn; B.s_k = d.b_sk; B.rshift = d.b_nshift; B.kshift = d.b_kshift;
    B.rstride = d.b_nstride; B.kstride = d.b_kstride; B.rows = d.N; B.kcontig = d.b_sk == 1; B.vec = (vecbits >> 1) & 1;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lk = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool do_rowsum = d.a_rowsum != nullptr && tn == 0;
    float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;

    // fragment addresses (bf16 elements) inside a tile, per k-step ks: + (2 ks) * YPLANE
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = lk * YPLANE + yslot(wm * 64 + i * 32 + li) * 8;
        fb[i] = lk * YPLANE + yslot(wn * 64 + i * 32 + li) * 8;
    }

    float4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;
    va0 = va1 = va2 = va3 = vb0 = vb1 = vb2 = vb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    XLane LA, LB;
    yprep(A, m0, tid, LA);
    yprep(B, n0, tid, LB);
    if (kt0 < kt1) {
        YTILE_LOAD(A, LA, m0, kt0 * YBK, va);
        YTILE_LOAD(B, LB, n0, kt0 * YBK, vb);
        ystore_lds(smem, A.kcontig, tid, va0, va1, va2, va3);
        ystore_lds(smem + YTILE, B.kcontig, tid, vb0, vb1, vb2, vb3);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const uint16_t* As = smem + cur * 2 * YTILE;
        const uint16_t* Bs = As + YTILE;
        const bool more = kt + 1 < kt1;
        if (do_rowsum) {                                  // fp32 row sums of the tile now in LDS (registers still hold it)
            if (A.kcontig) {
                rs0 += (va0.x + va0.y) + (va0.z + va0.w); rs1 += (va1.x + va1.y) + (va1.z + va1.w);
                rs2 += (va2.x + va2.y) + (va2.z + va2.w); rs3 += (va3.x + va3.y) + (va3.z + va3.w);
            } else {
                rs0 += (va0.x + va1.x) + (va2.x + va3.x); rs1 += (va0.y + va1.y) + (va2.y + va3.y);
                rs2 += (va0.z + va1.z) + (va2.z + va3.z); rs3 += (va0.w + va1.w) + (va2.w + va3.w);
            }
        }
        if (more) {
            YTILE_LOAD(A, LA, m0, (kt + 1) * YBK, va);
            YTILE_LOAD(B, LB, n0, (kt + 1) * YBK, vb);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[2], bq[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(As + 2 * ks * YPLANE + fa[i]);
                bq[i] = *reinterpret_cast<const bf16x8*>(Bs + 2 * ks * YPLANE + fb[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            uint16_t* An = smem + (cur ^ 1) * 2 * YTILE;
            ystore_lds(An, A.kcontig, tid, va0, va1, va2, va3);
            ystore_lds(An + YTILE, B.kcontig, tid, vb0, vb1, vb2, vb3);
        }
        __syncthreads();
    }

    // ---- epilogue.  acc[mi][ni][4*gq + t] = row m0 + wm*64 + mi*32 + 8*gq + 4*lk + t, column n0 + wn*64 + ni*32 + li.
    const bool split = S > 1;
    const int64_t MN = (int64_t)d.M * d.N;
    float* ws = split ? d.split_ws + (int64_t)slice * (MN + d.M) : nullptr;
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
                }
            }
    if (do_rowsum) {                                       // uniform per workgroup
        float* part = reinterpret_cast<float*>(smem);      // [8][128] partial sums; the k loop ended on a barrier
        if (A.kcontig) {
            float* q = part + (tid & 7) * 128 + (tid >> 3);
            q[0] = rs0; q[32] = rs1; q[64] = rs2; q[96] = rs3;
        } else {
            const int rq = (tid & 7) | (((tid >> 4) & 3) << 3), kq4 = ((tid >> 3) & 1) | ((tid >> 6) << 1);
            float* q = part + kq4 * 128 + 4 * rq;
            q[0] = rs0; q[1] = rs1; q[2] = rs2; q[3] = rs3;
        }
        __syncthreads();
        if (tid < XBM && m0 + tid < d.M) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += part[q * 128 + tid];
            if (split) ws[MN + m0 + tid] = s;
            else d.a_rowsum[m0 + tid] = d.accumulate ? d.a_rowsum[m0 + tid] + s : s;
        }
    }
}

// Adds the split-K slices in slice order and applies the epilogue.  One thread per output element (coalesced along n).
__global__ __launch_bounds__(256) void skg_gemmx_reduce_kernel(const skg_gemmx_group g) {
    int gi = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMMX_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) gi = t;
    const skg_gemmx_desc& d = g.d[gi];
    const int64_t MN = (int64_t)d.M * d.N;
    const int64_t total = MN + (d.a_rowsum ? d.M : 0);
    const int64_t i = (int64_t)(blockIdx.x - g.start[gi]) * 256 + threadIdx.x;
    if (i >= total) return;
    float v = 0.f;
    for (int s = 0; s < d.split_k; ++s) v += d.split_ws[(int64_t)s * (MN + d.M) + i];
    if (i >= MN) {                                         // bias gradient (row sums of A)
        const int m = (int)(i - MN);
        d.a_rowsum[m] = d.accumulate ? d.a_rowsum[m] + v : v;
        return;
    }
    const int row = (int)(i / d.N), col = (int)(i % d.N);
    if (d.bias) v += d.bias[col];
    if (d.relu) v = fmaxf(v, 0.f);
    float* p = d.C + xoff(col, d.c_nshift, d.c_nstride, 1) + (int64_t)row * d.ldc;
    if (d.accumulate) v += *p;
    if (d.mask && !(d.mask[(int64_t)row * d.ldmask + col] > 0.f)) v = 0.f;      // accumulate first, mask last
    *p = v;
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
    return 0;
}

static bool xmul4(int64_t v) { return (v & 3) == 0; }

extern "C" int64_t skg_gemmx_ws_floats(const skg_gemmx_desc* d) {
    if (!d) return SKG_E_ARG;
    return d->split_k > 1 ? (int64_t)d->split_k * ((int64_t)d->M * d->N + d->M) : 0;
}

static int skg_gemmx_launch(const skg_gemmx_desc* descs_host, int n, void* stream, bool bf16) {
    if (!descs_host || n < 1 || n > SKG_GEMMX_GROUP_MAX) return SKG_E_ARG;
    skg_gemmx_group g, r;
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
        if (skg_aligned16(d.A) && (d.a_sk == 1 ? xmul4(d.a_sm) : xmul4(d.a_sk))) vec |= 1;
        if (skg_aligned16(d.B) && (d.b_sk == 1 ? xmul4(d.b_sn) : xmul4(d.b_sk)) &&
            (d.b_kshift == 0 || xmul4(d.b_kstride)) && (d.b_nshift == 0 || xmul4(d.b_nstride)))
            vec |= 2;
        if ((((uintptr_t)d.C) & 7u) == 0 && (d.ldc & 1) == 0 && (d.c_nshift == 0 || (d.c_nstride & 1) == 0)) vec |= 4;
        g.d[g.n] = d; g.vec[g.n] = vec; g.start[g.n] = (int)blocks; ++g.n;
        blocks += nb;
        if (S > 1) {
            const int64_t total = (int64_t)d.M * d.N + (d.a_rowsum ? d.M : 0);
            const int64_t nr = (total + 255) / 256;
            if (rblocks + nr > 0x7fffffffLL) return SKG_E_LIMIT;
            r.d[r.n] = d; r.vec[r.n] = vec; r.start[r.n] = (int)rblocks; ++r.n;
            rblocks += nr;
        }
    }
    if (g.n == 0) return 0;
    for (int i = g.n; i <= SKG_GEMMX_GROUP_MAX; ++i) g.start[i] = (int)blocks;
    if (bf16) hipLaunchKernelGGL(skg_gemmx_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
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

extern "C" int skg_gemmx_bf16(const skg_gemmx_desc* descs_host, int n, void* stream) {
    return skg_gemmx_launch(descs_host, n, stream, true);
}
