// skg_gemm.hip -- fp32 dense layer on the CDNA4 matrix cores:  C = epilogue(A x W^T + bias).
//
// Replaces every nn.Linear of the interaction head and, with row-/column-stacked branch weights, the 16-branch
// MultiBranchFusion / MessageMBF GEMMs (reference heads/adamixer_transH_spatial_r50_head.py:469-474, 509-527).
//
// Design (gfx950): 128x128 block tile, 4 wavefronts (2x2), each owning 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs); K
// is consumed 16 at a time through a double-buffered LDS tile; weights keep the nn.Linear layout [N, K] (both operands
// "row = output index, k contiguous").  Three main loops share the tile map and the fused epilogues:
//   * MODE 2, fp16x2 (taken when the descriptor carries w_split): every operand value travels as h + m, two fp16
//     numbers (22 significant bits); v_mfma_f32_32x32x16_f16 accumulates h.m + m.h + h.h in fp32 -- 3 MFMA passes per
//     16 k against 8 of the fp32 MFMA, fp32-grade results (~2^-22 relative per product).  W is pre-split into
//     fragment-ordered planes (skg_split_weights_f16x2), A is split in registers on its way to LDS.  Tiles that leave
//     the fp16 range or hold inf / nan are recomputed by the exact loop.
//   * MODE 1, exact: v_mfma_f32_32x32x2_f32 (bit-for-bit an fmaf chain; peak 157 TFLOP/s).  Every lane fetches four
//     consecutive k with ONE ds_read_b128 and feeds them to four MFMAs; A and B use the same k permutation, which is
//     all a dot product needs.  Tiles are staged straight into LDS by global_load_lds_dwordx4 from wave-uniform scalar
//     bases + unsigned per-lane offsets; the LDS image is lane-linear (16 rows x 64 B per wave instruction) and the
//     16-byte k-chunk index is XOR-ed with (row >> 2) & 3 on the global SOURCE side and on the fragment reads, which
//     makes the reads conflict-free.  All fragment reads of a tile are issued before the DMA of the next tile (hipcc
//     drains vmcnt before any ds_read while an LDS-DMA is in flight), one barrier per tile.  Needs K % 16 == 0 and no
//     row gather; also instantiated with 64x64 tiles for grids that would leave most CUs idle.
//   * MODE 0, exact, register staged (any K % 4 == 0, gathered A rows, grouped launches, MODE 2's fallback): rows padded
//     to 20 dwords, loads unconditional and masked only when written to LDS.
//   * Block -> tile map: XCD groups (see skg_gemm_map) keep a <= 2 MiB W slice in each XCD's private L2 and have the A
//     panel fetched by NG XCDs instead of 8.
//   * Epilogue: accumulators are transposed through LDS so that every lane owns 4 consecutive columns of a row; bias,
//     ReLU, gathered multiplier tables (MBF fc_1 * fc_2), residual and the adjacency row-dot are fused with 16-byte
//     accesses; interior tiles take a guard-free path.  Split-K (plain epilogues) and a grouped launch cover the
//     small-M layers.
#include "skg_common.h"
#include <string.h>
#include <type_traits>

// 2^e for e in [-126, 127], exactly
__device__ __forceinline__ float skg_exp2i(int e) { return __uint_as_float((uint32_t)(127 + e) << 23); }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#ifndef SKG_BK
#define SKG_BK 16
#endif
#ifndef SKG_MINW
#define SKG_MINW 2
#endif
#ifndef SKG_USE_GLDS
#define SKG_USE_GLDS 1
#endif
#define BK SKG_BK
#ifndef SKG_XNST
#define SKG_XNST 4                      // register stages of the split-operand loop (tiles in flight)
#endif
#ifndef SKG_NB4
#define SKG_NB4 3                       // buffers of the direct-to-LDS latency loop's ring (MODE 4, 32 KiB each)
#endif
#define LDS_LD (BK + 4)                 // + 4 dwords of padding: conflict-free ds_read_b128 at strides 20 and 36
#define A_TILE (BM * LDS_LD)
#define B_TILE (BN * LDS_LD)

// Grids of up to this many tiles skip the XCD-group map: it rounds the grid up to whole groups (up to 4x the workgroups,
// the surplus returning at once), and for a launch this small the workgroup dispatch rate (~100 per microsecond chip-wide,
// measured) is what the launch costs, not L2 reuse.
#define SKG_DIRECT_MAP_TILES 1024
// g = N-tiles per group (W slice <= 2 MiB), NG = number of groups rounded up to a power of two.
#ifndef SKG_LOCKSTEP_MAP
#define SKG_LOCKSTEP_MAP 1
#endif
// K so long that not even ONE tile's W slice fits (box_head layer 1: K = 12544, 6.4 MB per 128 columns): nothing stays
// resident, but tiles that run SIDE BY SIDE on an XCD and walk K together share every k-slice they have in common.  The
// direct map deals the N-tiles of one row panel to eight different XCDs -- each reads the whole A from HBM (7.5x the
// algorithmic bytes, measured) --; the lockstep map gives an XCD whole row panels: its 32 CUs hold four panels x all
// N-tiles (up to 8), each A k-slice is fetched once per XCD and reused by the panel's N-tiles, each W k-slice by the four
// panels.  Returns false for the direct map.
__host__ __device__ __forceinline__ bool skg_gemm_use_map(int64_t nbm, int nbn, int K, int T, int& g, int& NG) {
    g = (8192 / T) / (K > 0 ? K : 1);                    // tiles of 64*T columns whose W slice fits in ~2 MiB
    const bool lockstep = SKG_LOCKSTEP_MAP && g < 1 && nbn <= 8;
    if (lockstep) g = nbn;
    if (g < 1) g = 1;
    if (g > nbn) g = nbn;
    const int ng = (nbn + g - 1) / g;
    NG = 1;
    while (NG < ng) NG <<= 1;
    if (NG >= 8) return false;
    const int64_t tiles = nbm * nbn;
    if (lockstep) {                                       // worth it at any grid size, unless the rounding to whole groups
        const int64_t mapped = 8LL * g * ((nbm + 8 / NG - 1) / (8 / NG));       // adds more than 1/8 of idle workgroups
        return mapped * 8 <= tiles * 9;
    }
    return tiles > SKG_DIRECT_MAP_TILES;
}

__host__ __device__ __forceinline__ int64_t skg_gemm_blocks(int M, int N, int K, int T) {
    const int64_t nbm = (M + 64 * T - 1) / (64 * T), nbn = (N + 64 * T - 1) / (64 * T);
    int g, NG;
    if (!skg_gemm_use_map(nbm, (int)nbn, K, T, g, NG)) return nbm * nbn;
    const int XG = 8 / NG;
    return 8LL * g * ((nbm + XG - 1) / XG);
}

// Pins a wave-uniform pointer in SGPRs so that `base + per-lane 32-bit offset` selects the saddr + voffset form.
__device__ __forceinline__ const char* skg_uniform_ptr(const char* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// EPI >= 0: epilogue fixed at compile time; EPI < 0: taken from the descriptor (grouped launches of small GEMMs).
// T = tile scale: block tile 64T x 64T, wave tile 32T x 32T = T x T MFMA tiles (T = 2: 128 x 128, the throughput
// shape; T = 1: 64 x 64 for small M, four times the workgroups for the same problem).
// MODE = main loop: 0 register-staged fp32 MFMA, 1 DMA-staged fp32 MFMA, 2 fp16x2-split operands on the fp16 MFMA.
template <int EPI_T, int MODE, int T>
__device__ __forceinline__ void skg_gemm_tile(const skg_gemm_desc& d, int block_id, float* smem) {
    static_assert(MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5 || T == 2, "only the DMA-staged and the latency loops have a 64 x 64 variant");
    static_assert((MODE != 3 && MODE != 4 && MODE != 5) || T == 1, "the latency loops are 64 x 64 tiles");
    constexpr bool GLDS = MODE == 1;
    const int Kmap = d.K;
    constexpr int TBM = 64 * T, TBN = 64 * T;
    const int EPI = EPI_T >= 0 ? EPI_T : d.epilogue;
    // split-K (small-M layers, e.g. box_head at batch 1: M = 40, K = 12544): slice s of the K range goes to blocks
    // [s * tiles, (s + 1) * tiles); raw partial sums land in d.split_ws[s] and skg_splitk_reduce_kernel applies the
    // bias / ReLU epilogue in a fixed slice order (deterministic).
    int kt_begin = 0, kt_end = (d.K + BK - 1) / BK, split_slice = 0;
    if (d.split_k > 1) {
        const int tiles = (int)skg_gemm_blocks(d.M, d.N, Kmap, T);
        split_slice = block_id / tiles;
        block_id -= split_slice * tiles;
        const int per = (kt_end + d.split_k - 1) / d.split_k;
        kt_begin = split_slice * per;
        kt_end = kt_begin + per < kt_end ? kt_begin + per : kt_end;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = (tid >> 6) & 3;              // (MODE 5: eight waves, two per 32 x 32 sub-tile -- wave w and w + 4)
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, lh = lane >> 5;

    // ---- block -> tile map (speed only; correctness never depends on block placement).
    // Blocks are dealt round-robin over the 8 XCDs (observed), each XCD has a private 4 MiB L2.  The N-tiles are cut
    // into NG groups whose W slice (g tiles x 128 x K x 4 B) fits in ~2 MiB; XG = 8 / NG XCDs serve one group and
    // split the M-tiles between them.  Inside an XCD consecutive blocks walk the g N-tiles of ONE M-tile, so the A
    // panel is fetched from HBM by NG XCDs (not 8) and W stays L2-resident.
    const int nbn = (d.N + TBN - 1) / TBN;
    const int nbm = (d.M + TBM - 1) / TBM;
    int bm, bn;
    {
        int g, NG;
        if (!skg_gemm_use_map(nbm, nbn, Kmap, T, g, NG)) {
            bn = block_id % nbn;
            bm = block_id / nbn;
        } else {
            const int XG = 8 / NG;
            const int xcd = block_id & 7, slot = block_id >> 3;
            const int ng = xcd % NG, mg = xcd / NG;
            bn = ng * g + slot % g;
            bm = (slot / g) * XG + mg;
            if (bn >= nbn || bm >= nbm) return;
        }
    }
    const int m0 = bm * TBM, n0 = bn * TBN;

    f32x16 acc[T][T];
#pragma unroll
    for (int mi = 0; mi < T; ++mi)
#pragma unroll
        for (int ni = 0; ni < T; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = kt_end;
    // ---- exact fp32 loop, register staged: MODE 0, and the per-tile fallback of MODE 2
    auto loop_exact = [&]() {
    if constexpr (T == 2) {
    // ---- global -> register staging map: thread owns rows (lr, lr+64) x 4 consecutive k
    constexpr int TPR = BK / 4;                  // threads per tile row (float4 each)
    constexpr int RPP = 256 / TPR;               // rows per staging pass
    constexpr int NPASS = BM / RPP;
    const int lr = tid / TPR;
    const int lc = (tid % TPR) * 4;
    const float* pa[NPASS];
    const float* pw[NPASS];
    bool va[NPASS], vw[NPASS];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        const int r = m0 + lr + RPP * i;
        int src = -1;
        if (r < d.M) src = d.a_rows ? d.a_rows[r] : r;
        va[i] = src >= 0;
        pa[i] = d.A + (int64_t)(va[i] ? src : 0) * d.lda;       // always a readable row; masked after the load
        const int c = n0 + lr + RPP * i;
        vw[i] = c < d.N;
        pw[i] = d.W + (int64_t)(vw[i] ? c : 0) * d.ldw;
    }

    float4 ra[NPASS], rw[NPASS];

    // Loads are unconditional (addresses are clamped to readable rows / k); rows past M or N and the K tail are
    // zeroed with selects when the tile is written to LDS, AFTER the MFMAs of the current tile.  (A select next to the
    // load makes hipcc branch around it and wait vmcnt(0) per load, which serialises the whole prefetch.)
    auto gload = [&](int kt) {
        const int k = kt * BK + lc;
        const int kk = (k < d.K) ? k : 0;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(pa[i] + kk);
            rw[i] = *reinterpret_cast<const float4*>(pw[i] + kk);
        }
    };
    auto lstore = [&](int buf, int kt) {
        float* a_s = smem + buf * A_TILE;
        float* b_s = smem + 2 * A_TILE + buf * B_TILE;
        const bool kin = kt * BK + lc < d.K;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const bool ma = kin && va[i], mw = kin && vw[i];
            const float4 x = ra[i], w = rw[i];
            *reinterpret_cast<float4*>(a_s + (lr + RPP * i) * LDS_LD + lc) =
                make_float4(ma ? x.x : 0.f, ma ? x.y : 0.f, ma ? x.z : 0.f, ma ? x.w : 0.f);
            *reinterpret_cast<float4*>(b_s + (lr + RPP * i) * LDS_LD + lc) =
                make_float4(mw ? w.x : 0.f, mw ? w.y : 0.f, mw ? w.z : 0.f, mw ? w.w : 0.f);
        }
    };

    gload(kt_begin);
    lstore(kt_begin & 1, kt_begin);
    __syncthreads();

    for (int kt = kt_begin; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* a_s = smem + cur * A_TILE + (wr * 64 + li) * LDS_LD + 4 * lh;
        const float* b_s = smem + 2 * A_TILE + cur * B_TILE + (wc * 64 + li) * LDS_LD + 4 * lh;
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(a_s + i * 32 * LDS_LD + ks * 8);
                b[i] = *reinterpret_cast<const float4*>(b_s + i * 32 * LDS_LD + ks * 8);
            }
            const float av[2][4] = {{a[0].x, a[0].y, a[0].z, a[0].w}, {a[1].x, a[1].y, a[1].z, a[1].w}};
            const float bv[2][4] = {{b[0].x, b[0].y, b[0].z, b[0].w}, {b[1].x, b[1].y, b[1].z, b[1].w}};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][t], bv[ni][t], acc[mi][ni], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);              // keep the staging writes (and their vmcnt wait) below the MFMAs
        if (kt + 1 < nk) lstore(cur ^ 1, kt + 1);
        __syncthreads();
    }

    }
    };
    if constexpr (MODE == 0) {
        loop_exact();
    } else if constexpr (MODE == 3 || MODE == 5) {
        // ---- latency loop for a FEW images (64 x 64 tile, 64 k per step, exact fp32 MFMA).  At one to a handful of
        // graphs a launch has a few hundred workgroups at most: every workgroup walks its K range alone, and what a
        // step costs is one memory round trip (~1 us) whatever it computes.  The 16-k steps of the throughput loops make
        // K = 1024 sixty-four such round trips (34 us per GEMM of one image's grid, measured); here a step carries 64 k:
        // sixteen round trips, each hidden behind 32 MFMAs per wave (0.85 us) -- the loads of step t+1 are in flight in
        // registers while step t computes.  One LDS buffer, rows padded to 68 dwords (conflict-free ds_read_b128),
        // two barriers per step; gathers, K % 4 and split-K as in the register-staged loop.
        // MODE 5 (round 4): the same loop on EIGHT waves.  A single image leaves each CU ONE workgroup -- one wave per SIMD,
        // whose 32 dependent MFMAs of a step wait for its own fragment reads, LDS stores and barriers: 55-60 % MFMA duty
        // measured, 25-32 us for K = 1024 where the MFMA chain alone is 13.6 us.  Here waves w and w + 4 share a 32 x 32
        // sub-tile and split every 64-k step between them (k-quads 0-3 | 4-7 of each row): two independent accumulation
        // chains per SIMD, half the staging work per thread; the upper waves hand their accumulators over through LDS
        // after the last step (sum = lower + upper, a fixed order) and retire before the epilogue.
        constexpr int BK3 = 64, LD3 = BK3 + 4;
        constexpr int NW3 = MODE == 5 ? 8 : 4;       // waves of the workgroup
        constexpr int RP3 = NW3 * 4;                 // rows per staging pass
        constexpr int NP3 = 64 / RP3;                // passes
        const int kh = MODE == 5 ? (tid >> 8) : 0;   // k half this wave multiplies
        float* a_s3 = smem;
        float* b_s3 = smem + 64 * LD3;
        const int lr = tid >> 4;                     // RP3 rows per pass
        const int lc = (tid & 15) * 4;               // k quad inside the 64-k step
        const float* pa[NP3];
        const float* pw[NP3];
        bool va[NP3], vw[NP3];
#pragma unroll
        for (int i = 0; i < NP3; ++i) {
            const int r = m0 + lr + RP3 * i;
            int src = -1;
            if (r < d.M) src = d.a_rows ? d.a_rows[r] : r;
            va[i] = src >= 0;
            pa[i] = d.A + (int64_t)(va[i] ? src : 0) * d.lda;
            const int c = n0 + lr + RP3 * i;
            vw[i] = c < d.N;
            pw[i] = d.W + (int64_t)(vw[i] ? c : 0) * d.ldw;
        }
        int s_begin = 0, s_end = (d.K + BK3 - 1) / BK3;
        if (d.split_k > 1) {
            const int per = (s_end + d.split_k - 1) / d.split_k;
            s_begin = split_slice * per;
            s_end = s_begin + per < s_end ? s_begin + per : s_end;
        }
        // Two register stages: the loads of step st + 2 are issued while step st is multiplied, so a tile has two steps
        // (~1 us each: 32 MFMAs of 64 cycles per wave) to arrive.  With one stage the round trip to L2 / HBM stuck out of
        // every step by ~0.7 us (2 us per step measured at one image).
        float4 ra[2][NP3], rw[2][NP3];
        auto gload3 = [&](auto S, int st) {
            constexpr int sg = decltype(S)::value;
#ifdef SKG_K3_NOLOAD                                  // (timing builds, tools/gemm_k3_knockout.sh: results are wrong)
            if (st > s_begin + 2) return;
#endif
            const int k = st * BK3 + lc;
            const int kk = (k < d.K) ? k : 0;        // clamped, masked when written to LDS (K % 4 == 0: whole quads)
#pragma unroll
            for (int i = 0; i < NP3; ++i) {
                ra[sg][i] = *reinterpret_cast<const float4*>(pa[i] + kk);
                rw[sg][i] = *reinterpret_cast<const float4*>(pw[i] + kk);
            }
        };
        auto lstore3 = [&](auto S, int st) {
            constexpr int sg = decltype(S)::value;
#ifdef SKG_K3_NOSTORE
            if (st > s_begin) return;
#endif
            const bool kin = st * BK3 + lc < d.K;
#pragma unroll
            for (int i = 0; i < NP3; ++i) {
                const bool ma = kin && va[i], mw = kin && vw[i];
                const float4 x = ra[sg][i], w = rw[sg][i];
                *reinterpret_cast<float4*>(a_s3 + (lr + RP3 * i) * LD3 + lc) =
                    make_float4(ma ? x.x : 0.f, ma ? x.y : 0.f, ma ? x.z : 0.f, ma ? x.w : 0.f);
                *reinterpret_cast<float4*>(b_s3 + (lr + RP3 * i) * LD3 + lc) =
                    make_float4(mw ? w.x : 0.f, mw ? w.y : 0.f, mw ? w.z : 0.f, mw ? w.w : 0.f);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        if (s_begin < s_end) {
            gload3(S0{}, s_begin);
            gload3(S1{}, s_begin + 1);               // (past the slice: a readable tile that is never stored)
            lstore3(S0{}, s_begin);
            gload3(S0{}, s_begin + 2);
        }
        __syncthreads();
        // one step: tile st is in LDS, tile st + 1 in stage NX, tile st + 2 in flight into the other stage
        auto step3 = [&](auto NX, int st) {
            const float* ap = a_s3 + (wr * 32 + li) * LD3 + 4 * lh + (MODE == 5 ? kh * (BK3 / 2) : 0);
            const float* bp = b_s3 + (wc * 32 + li) * LD3 + 4 * lh + (MODE == 5 ? kh * (BK3 / 2) : 0);
#pragma unroll
            for (int ks = 0; ks < BK3 / 8 / (MODE == 5 ? 2 : 1); ++ks) {
                const float4 a4 = *reinterpret_cast<const float4*>(ap + ks * 8);
                const float4 b4 = *reinterpret_cast<const float4*>(bp + ks * 8);
#ifdef SKG_K3_NOMFMA
                acc[0][0][ks] += a4.x * b4.x + a4.y * b4.y + a4.z * b4.z + a4.w * b4.w;
#else
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[0][0], 0, 0, 0);
#endif
            }
#ifndef SKG_K3_NOBAR
            __syncthreads();                          // every wave is done reading this step's tile
#endif
            if (st + 1 < s_end) {
                lstore3(NX, st + 1);
                gload3(NX, st + 3);
#ifndef SKG_K3_NOBAR
                __syncthreads();
#endif
            }
        };
        int st = s_begin;
        for (; st + 1 < s_end; st += 2) {
            step3(S1{}, st);
            step3(S0{}, st + 1);
        }
        if (st < s_end) step3(S1{}, st);
        if constexpr (MODE == 5) {
            // upper k-halves -> LDS -> added to the lower ones; the upper waves retire (a finished wave leaves the workgroup's
            // barrier count: the epilogue's barriers are among waves 0-3)
            float* red = smem + ((tid >> 6) & 3) * (16 * 64) + lane;
            __syncthreads();
            if (kh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[r * 64] = acc[0][0][r];
            }
            __syncthreads();
            if (kh) return;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += red[r * 64];
            __syncthreads();                          // (the epilogue's transposition reuses this memory)
        }
    } else if constexpr (MODE == 4) {
        // ---- latency loop, staged straight into LDS (round 4): the 64 x 64 x 64 steps of MODE 3 without its VGPR stage.  MODE 3
        // loads a step's 32 KiB into registers, writes them to ONE LDS buffer between two barriers and only then multiplies:
        // at one workgroup per CU (a single image: 208 workgroups) global loads, LDS writes, fragment reads and MFMAs of a
        // step run one after the other -- 1.6-1.9 us per step for 0.93 us of MFMA (profiles/r04_b1_eval_timeline.txt).  Here
        // global_load_lds_dwordx4 fills a ring of SKG_NB4 buffers (tiles t + 1 .. in flight across ONE barrier per step,
        // counted s_waitcnt vmcnt), nothing is converted or re-written, and the fragment reads are inline asm with counted
        // lgkmcnt waits (for a C++ LDS read hipcc waits vmcnt(0) on every DMA in flight: the ring would drain every step).
        // LDS image of an operand tile: [64 rows][64 k] fp32, 256-byte rows; the 16-byte slot s of row r holds k-quad
        // s ^ (r & 15) -- the swizzle lives on the per-lane SOURCE address (the DMA's destination is lane-linear) and makes the
        // ds_read_b128 of a fragment (32 rows, one k-quad) conflict-free.  Same k order inside a step as MODE 3 (k-quad
        // 2 ks + lh, element j = MFMA j): bit-identical sums.  Needs K % 64 == 0 and no row gather (host: skg_gemm_f32).
        constexpr int BK4 = 64, NB4 = SKG_NB4, OPB = 64 * BK4 * 4, BUFB = 2 * OPB;
        static_assert(NB4 == 2 || NB4 == 3, "the waits below count NB4 - 1 tiles in flight");
        const int wu = __builtin_amdgcn_readfirstlane(wid);     // provably wave-uniform: the DMA's LDS base stays scalar
        uint32_t oa[4], ow[4];                                  // per-lane byte offsets of the four 16-byte pieces per operand
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 16 * wid + 4 * i + (lane >> 4);       // wave-instruction i covers rows 16 wid + 4 i .. + 3
            const int q = (lane & 15) ^ (r & 15);
            const int ar = m0 + r < d.M ? r : 0, wrw = n0 + r < d.N ? r : 0;      // rows past M / N: a valid row, never stored
            oa[i] = (uint32_t)(((int64_t)ar * d.lda + 4 * q) * 4);
            ow[i] = (uint32_t)(((int64_t)wrw * d.ldw + 4 * q) * 4);
        }
        const char* a_base = reinterpret_cast<const char*>(d.A + (int64_t)m0 * d.lda);
        const char* w_base = reinterpret_cast<const char*>(d.W + (int64_t)n0 * d.ldw);
        int s_begin = 0, s_end = d.K / BK4;
        if (d.split_k > 1) {
            const int per = (s_end + d.split_k - 1) / d.split_k;
            s_begin = split_slice * per < s_end ? split_slice * per : s_end;
            s_end = s_begin + per < s_end ? s_begin + per : s_end;
        }
        const int nt = s_end - s_begin;
        char* ring = reinterpret_cast<char*>(smem);
        auto issue = [&](int t, int buf) {
            const char* ab = skg_uniform_ptr(a_base + (int64_t)(s_begin + t) * BK4 * 4);
            const char* wb = skg_uniform_ptr(w_base + (int64_t)(s_begin + t) * BK4 * 4);
            char* dst = ring + buf * BUFB + (4 * wu) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + oa[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + ow[i]),
                                                 (__attribute__((address_space(3))) void*)(dst + OPB + i * 1024), 16, 0, 0);
            }
        };
        // fragment addresses inside a buffer: row (wr | wc) * 32 + li, k-quad (2 ks + lh) ^ (row & 15)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
        uint32_t fa[8], fb[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int ra_ = wr * 32 + li, rb_ = wc * 32 + li;
            fa[ks] = (uint32_t)(ra_ * 256 + (((2 * ks + lh) ^ (ra_ & 15)) << 4));
            fb[ks] = (uint32_t)(OPB + rb_ * 256 + (((2 * ks + lh) ^ (rb_ & 15)) << 4));
        }
        auto step4 = [&](uint32_t buf) {
            f32x4 a4[8], b4[8];
#define S4_READ(KS)                                                                                                   \
            asm volatile("ds_read_b128 %0, %1" : "=v"(a4[KS]) : "v"(buf + fa[KS]));                                  \
            asm volatile("ds_read_b128 %0, %1" : "=v"(b4[KS]) : "v"(buf + fb[KS]));
#define S4_MFMA(KS)                                                                                                   \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[KS].x, b4[KS].x, acc[0][0], 0, 0, 0);                \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[KS].y, b4[KS].y, acc[0][0], 0, 0, 0);                \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[KS].z, b4[KS].z, acc[0][0], 0, 0, 0);                \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[KS].w, b4[KS].w, acc[0][0], 0, 0, 0);
            // eight k-quads per step; the reads of quads ks + 4 .. are issued while quads ks .. multiply (the LGKM counter
            // holds 15: at most 8 + 6 reads are outstanding)
            S4_READ(0) S4_READ(1) S4_READ(2) S4_READ(3)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a4[0]), "+v"(b4[0]));
            S4_READ(4)
            S4_MFMA(0)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a4[1]), "+v"(b4[1]));
            S4_READ(5)
            S4_MFMA(1)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a4[2]), "+v"(b4[2]));
            S4_READ(6)
            S4_MFMA(2)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a4[3]), "+v"(b4[3]));
            S4_READ(7)
            S4_MFMA(3)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(a4[4]), "+v"(b4[4]));
            S4_MFMA(4)
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a4[5]), "+v"(b4[5]));
            S4_MFMA(5)
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a4[6]), "+v"(b4[6]));
            S4_MFMA(6)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4[7]), "+v"(b4[7]));
            S4_MFMA(7)
#undef S4_READ
#undef S4_MFMA
        };
#pragma unroll
        for (int t = 0; t < NB4 - 1; ++t)
            if (t < nt) issue(t, t);
        int cur = 0;
        for (int t = 0; t < nt; ++t) {
            const int ahead = nt - 1 - t;                       // tiles issued after tile t (eight DMA instructions each)
            if (NB4 == 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");             // tile t landed for everybody; nobody reads buffer cur - 1 any more
            const int prev = cur == 0 ? NB4 - 1 : cur - 1;
            if (t + NB4 - 1 < nt) issue(t + NB4 - 1, prev);
            step4(lds0 + cur * BUFB);
            cur = cur + 1 == NB4 ? 0 : cur + 1;
        }
        __syncthreads();                                        // the epilogue's transposition reuses the ring
    } else if constexpr (MODE == 2) {
        // ---- fp32-grade result from the fp16 matrix pipe (3 MFMA passes instead of the 8 of the fp32 MFMA per 16 k).
        // Every operand value x is carried as h + m with h = fp16(x), m = fp16(x - h): 22 significant bits, i.e.
        // |x - h - m| <= 2^-23 |x| (2^-25 absolute once m is subnormal; the fp16 MFMA takes subnormals exactly).
        // a.b is accumulated from h.m, m.h and h.h in the fp32 accumulator of v_mfma_f32_32x32x16_f16; the dropped
        // m.m is <= 2^-22 |a.b|.  W comes pre-split from skg_split_weights_f16x2 -- scaled by a power of two so that its
        // m planes stay normal, un-scaled in the epilogue (d.w_scale) -- as MFMA-fragment-ordered 1 KiB planes
        // [n-tile 32][k-tile 16][h|m][k-half][row][8 k].  A stays fp32 in HBM (gathers and epilogue outputs
        // unchanged), is split in registers and written to LDS in the same plane format; fragment reads are
        // lane-linear ds_read_b128.  fp16 overflows at 65504: a tile whose accumulators come out non-finite (overflow,
        // inf / nan inputs) is recomputed by the exact loop, so those cases behave as in fp32 arithmetic.
        char* sm = reinterpret_cast<char*>(smem);
        // Power-of-two scale of every A row: row r travels as 2^-e_r * A[r, :] and the result row is multiplied back by
        // 2^e_r with the weight's scale in the epilogue, both exact.  e_r comes from d.a_exp (skg_row_exponents_f32: the
        // row's true maximum lands in [2^11, 2^12)) or, without it, from the workgroup's own estimate below: the h + m pair keeps its 22 significant bits at any activation magnitude
        // (un-scaled, m turns subnormal below |x| = 2^-3 and the error floor is 2^-25 absolute), and an outlier row does
        // not cost the other rows their precision.  Rows holding inf / nan keep e = 0; their tile takes the exact loop.
        constexpr int NC = 2, NCB = NC * 1024;
        constexpr int BUF = 8 * NCB, WOFF = 4 * NCB;                  // per buffer: A planes | W planes, 4 row tiles each
        const int q = tid & 3, r = tid >> 2;                          // staging: rows r, r + 64; k quad q
        const float* pa[2];
        bool va[2];
        float asc[2];
        uint32_t aw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = m0 + r + 64 * i;
            int src = -1;
            if (row < d.M) src = d.a_rows ? d.a_rows[row] : row;
            va[i] = src >= 0 || row >= d.M;                           // rows past M: any readable data, never stored
            asc[i] = (d.a_exp && row < d.M) ? skg_exp2i(-min(max(d.a_exp[row], -126), 126)) : 1.f;
            pa[i] = d.A + (int64_t)(src >= 0 ? src : 0) * d.lda + q * 4;
            aw[i] = (uint32_t)((((r >> 5) + 2 * i) * NC) * 1024 + (q >> 1) * 512 + (r & 31) * 16 + (q & 1) * 8);
        }
        const int nnt = (d.N + 31) >> 5, nkt = d.K >> 4;
        int nt = (n0 >> 5) + wid;
        if (nt >= nnt) nt = 0;                                        // columns past N: never stored
        const uint32_t lane16 = lane * 16;
        const char* pw = reinterpret_cast<const char*>(d.w_split) + (int64_t)nt * nkt * NCB + lane16;
        const uint32_t ww = (uint32_t)(WOFF + wid * NCB) + lane16;
        // register stages: NST tiles are in flight between global memory and LDS; plain loads keep the compiler's
        // vmcnt bookkeeping exact and barriers do not drain them
        constexpr int NST = SKG_XNST;
        f32x4 ra[NST][2];
        u32x4 rw[NST][NC];
        auto load_a = [&](auto S, int kt) {
            constexpr int st = decltype(S)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[st][i] = *reinterpret_cast<const f32x4*>(pa[i] + kt * 16);
        };
        auto load_w = [&](auto S, int kt) {
            constexpr int st = decltype(S)::value;
#pragma unroll
            for (int c = 0; c < NC; ++c) rw[st][c] = *reinterpret_cast<const u32x4*>(pw + (int64_t)kt * NCB + c * 1024);
        };
        auto load_tile = [&](auto S, int kt) { load_a(S, kt); load_w(S, kt); };
        auto store_tile = [&](auto S, int buf) {
            constexpr int st = decltype(S)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                h16x2 hp[2], mp[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float x0 = va[i] ? ra[st][i][2 * e] * asc[i] : 0.f, x1 = va[i] ? ra[st][i][2 * e + 1] * asc[i] : 0.f;
                    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;                       // round to nearest even
                    hp[e] = h16x2{h0, h1};
                    mp[e] = h16x2{(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1)};    // x - h is exact
                }
                char* dst = sm + buf * BUF + aw[i];
                *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(uint32_t, hp[0]), __builtin_bit_cast(uint32_t, hp[1]));
                *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(__builtin_bit_cast(uint32_t, mp[0]), __builtin_bit_cast(uint32_t, mp[1]));
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) *reinterpret_cast<u32x4*>(sm + buf * BUF + ww + c * 1024) = rw[st][c];
        };
        // Software pipeline of one wave, per k-tile t (step index j = t - kt_begin):
        //   fragments of tile t+1: LDS buffer (j+1)&1 -> fragment set (j+1)&1       (written in step j-1, barrier since)
        //   12 MFMAs on tile t from fragment set j&1                                 (read in step j-1)
        //   tile t+2: register stage (j+2)%NST -> split -> LDS buffer j&1            (its readers finished in step j-1)
        //   tile t+2+NST: global -> the stage just freed
        //   one barrier.
        // Nothing a step issues is needed before the NEXT step, so LDS and global latencies hide under the MFMAs.
        f16x8 fa[2][2][NC], fb[2][2][NC];
        auto read_frags = [&](auto F, int buf) {
            constexpr int fs = decltype(F)::value;
            const char* a_f = sm + buf * BUF + (2 * wr * NC) * 1024 + lane16;
            const char* b_f = sm + buf * BUF + WOFF + (2 * wc * NC) * 1024 + lane16;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    fa[fs][i][c] = *reinterpret_cast<const f16x8*>(a_f + (i * NC + c) * 1024);
                    fb[fs][i][c] = *reinterpret_cast<const f16x8*>(b_f + (i * NC + c) * 1024);
                }
        };
        auto step = [&](auto J, int t) {
            constexpr int j = decltype(J)::value;                       // step index modulo NST (NST even)
            read_frags(std::integral_constant<int, (j + 1) & 1>{}, (j + 1) & 1);
            // small terms first (h.m, m.h), then h.h; the four accumulators take turns
            constexpr int PA[3] = {0, 1, 0}, PB[3] = {1, 0, 0};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[j & 1][mi][PA[p]], fb[j & 1][ni][PB[p]],
                                                                             acc[mi][ni], 0, 0, 0);
            using SG = std::integral_constant<int, (j + 2) % NST>;
            store_tile(SG{}, j & 1);                                    // past the end: clamped tiles into dead buffers
            load_w(SG{}, t + 2 + NST < nk ? t + 2 + NST : nk - 1);      // unconditional: no branch in the loop body
            load_a(SG{}, t + 2 + NST < nk ? t + 2 + NST : nk - 1);
            __syncthreads();
        };
        static_assert(NST == 4, "stage / buffer / fragment-set indices below assume 4 register stages");
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        auto clampk = [&](int t) { return t < nk ? t : nk - 1; };
        load_tile(I0{}, clampk(kt_begin));                              // (an empty split-K slice has kt_begin >= nk)
        load_tile(I1{}, clampk(kt_begin + 1));
        load_tile(I2{}, clampk(kt_begin + 2));
        load_tile(I3{}, clampk(kt_begin + 3));
        if (!d.a_exp) {
            // No exponents given: every workgroup estimates its rows' scale from the first 64 k it walks (the four tiles
            // now in registers; 16 values per thread, 4 threads per row).  The estimate can only be LOW (it is the maximum
            // of a part of the row): values up to 2^7 times larger still fit fp16, beyond that the tile overflows and is
            // recomputed by the exact loop -- never a wrong result, and no separate pass over the operand.  A row whose
            // first 64 k are all zero stays un-scaled.
            int* rowexp = reinterpret_cast<int*>(sm + 2 * BUF);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                uint32_t mx = 0u;
#pragma unroll
                for (int st = 0; st < NST; ++st)
#pragma unroll
                    for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(ra[st][i][e]) & 0x7fffffffu);
                if (!va[i]) mx = 0u;
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, 1, 64));
                mx = max(mx, (uint32_t)__shfl_xor((int)mx, 2, 64));
                const int E = (int)(mx >> 23);
                const int ex = (mx == 0u || E == 255) ? 0 : min(max(E - 127 - 9, -126), 126);
                asc[i] = skg_exp2i(-ex);
                if (q == 0) rowexp[r + 64 * i] = ex;
            }
        }
        store_tile(I0{}, 0);
        store_tile(I1{}, 1);
        load_tile(I0{}, clampk(kt_begin + 4));
        load_tile(I1{}, clampk(kt_begin + 5));
        __syncthreads();
        read_frags(I0{}, 0);
        __syncthreads();                                                // buffer 0 is rewritten by the first step
        // groups of four steps (static indices), ONE loop exit: with early exits between the steps hipcc keeps the
        // accumulators in fresh registers per step and copies all 64 of them back every tile
        int kt = kt_begin;
        for (; kt + 3 < nk; kt += 4) {
            step(I0{}, kt);
            step(I1{}, kt + 1);
            step(I2{}, kt + 2);
            step(I3{}, kt + 3);
        }
        if (kt < nk) step(I0{}, kt);
        if (kt + 1 < nk) step(I1{}, kt + 1);
        if (kt + 2 < nk) step(I2{}, kt + 2);
        // un-scale; x * 0 is nan exactly when x is inf or nan: one flag per block decides the exact re-run
        float bad = 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    bad = fmaf(acc[mi][ni][rr], 0.f, bad);
                    acc[mi][ni][rr] *= d.w_scale;
                }
        {
            // acc[mi][ni][4 g + t] is row m0 + wr*64 + mi*32 + 8 g + 4 (lane >> 5) + t: four consecutive rows per g.
            // (rowexp was written before the first barrier of the loop; the tiles never reach that part of the LDS)
            const int* rowexp = reinterpret_cast<const int*>(sm + 2 * BUF);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lrow = wr * 64 + mi * 32 + 8 * g + 4 * (lane >> 5);
                    float un[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int row = m0 + lrow + t;
                        const int ex = d.a_exp ? (row < d.M ? d.a_exp[row] : 0) : rowexp[lrow + t];
                        un[t] = skg_exp2i(min(max(ex, -126), 126));
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[mi][ni][4 * g + t] *= un[t];
                }
        }
        if (__syncthreads_or(bad != bad)) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) acc[mi][ni][rr] = 0.f;
            loop_exact();
        }
    } else {
        // ---- direct-to-LDS staging (global_load_lds_dwordx4): no VGPR round trip, no ds_write, no masking.
        // Requirements (checked by the host): K % 16 == 0, no A-row gather, 128 rows x ld x 4 B < 4 GiB.  Rows past M / N are clamped:
        // they only feed accumulators whose outputs are never stored.  The LDS image is lane-linear per wave
        // instruction (16 rows x 64 B); bank conflicts of the ds_read_b128 fragment reads are removed by XOR-ing the
        // 16-byte k-chunk index with (row >> 2) & 3 on the global SOURCE side and on the read side.
        constexpr int GT = TBM * BK;                                  // floats per operand tile (unpadded)
        const int rl = lane >> 2;                                     // row inside a 16-row piece
        const int ch = lane & 3;                                      // 16-byte chunk the lane WRITES
        // per-lane byte offsets (32-bit) from wave-uniform bases that advance by one tile per iteration: the address
        // arithmetic of the DMA stays on the scalar unit
        uint32_t oa[T], ow[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {
            const int r = wid * (16 * T) + i * 16 + rl;               // tile row staged by this lane
            const int sw = (r >> 2) & 3;
            const int ar = m0 + r, wrw = n0 + r;
            // rows past M / N are clamped to the tile's FIRST row (offset 0, never negative: the offsets are unsigned)
            oa[i] = (uint32_t)(((int64_t)(ar < d.M ? ar - m0 : 0) * d.lda + 4 * (ch ^ sw)) * 4);
            ow[i] = (uint32_t)(((int64_t)(wrw < d.N ? wrw - n0 : 0) * d.ldw + 4 * (ch ^ sw)) * 4);
        }
        const char* a_base = reinterpret_cast<const char*>(d.A + (int64_t)m0 * d.lda);
        const char* w_base = reinterpret_cast<const char*>(d.W + (int64_t)n0 * d.ldw);
        const int wid_u = __builtin_amdgcn_readfirstlane(wid);      // provably wave-uniform: LDS DMA bases stay scalar
        auto stage = [&](int buf, int kt) {
            float* a_s = smem + buf * GT + (wid_u * 16 * T) * BK;
            float* b_s = smem + 2 * GT + buf * GT + (wid_u * 16 * T) * BK;
            const char* ab = skg_uniform_ptr(a_base + (int64_t)kt * BK * 4);
            const char* wb = skg_uniform_ptr(w_base + (int64_t)kt * BK * 4);
#pragma unroll
            for (int i = 0; i < T; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ab + oa[i]),
                                                 (__attribute__((address_space(3))) void*)(a_s + i * 16 * BK), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wb + ow[i]),
                                                 (__attribute__((address_space(3))) void*)(b_s + i * 16 * BK), 16, 0, 0);
            }
        };
        const int swr = (li >> 2) & 3;                                // (row >> 2) & 3 of the rows this lane READS
        // (a split-K slice can be empty -- more slices than k-tiles: stage a valid tile, the loop below does not run)
        stage(kt_begin & 1, kt_begin < nk ? kt_begin : nk - 1);
        __syncthreads();                                              // drains vmcnt: the first tile has landed
        for (int kt = kt_begin; kt < nk; ++kt) {
            const int cur = kt & 1;
            const float* a_s = smem + cur * GT + (wr * 32 * T + li) * BK;
            const float* b_s = smem + 2 * GT + cur * GT + (wc * 32 * T + li) * BK;
            // all fragment reads of this tile first: hipcc waits vmcnt(0) before any ds_read while an LDS-DMA is in
            // flight, so the next tile's DMA is issued only after them and lands under the MFMAs
            float4 a[BK / 8][T], b[BK / 8][T];
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                const int co = 4 * ((2 * ks + lh) ^ swr);
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    a[ks][i] = *reinterpret_cast<const float4*>(a_s + i * 32 * BK + co);
                    b[ks][i] = *reinterpret_cast<const float4*>(b_s + i * 32 * BK + co);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < BK / 8; ++ks) {
                float av[T][4], bv[T][4];
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    av[i][0] = a[ks][i].x; av[i][1] = a[ks][i].y; av[i][2] = a[ks][i].z; av[i][3] = a[ks][i].w;
                    bv[i][0] = b[ks][i].x; bv[i][1] = b[ks][i].y; bv[i][2] = b[ks][i].z; bv[i][3] = b[ks][i].w;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mi = 0; mi < T; ++mi)
#pragma unroll
                        for (int ni = 0; ni < T; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][t], bv[ni][t], acc[mi][ni], 0, 0, 0);
            }
            __syncthreads();                                          // vmcnt(0) + barrier: next tile landed, this one free
        }
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5),
    // i.e. a lane owns ONE column of 16 rows: stored directly that is 4-byte accesses.  Instead each wave transposes
    // its accumulators through LDS (the staging tiles are dead now; 32 rows x 64 columns per pass, rows padded to 68
    // dwords) and walks them row-wise: every lane then handles 4 consecutive columns of one row, so bias / multiplier
    // tables / residual loads and the stores are 16-byte accesses covering 256 contiguous bytes per row.
    constexpr int EST_LD = 32 * T + 4;              // padded row: 16-byte aligned, conflict-free transposition
    constexpr int LPR = 8 * T;                      // lanes per row (4 columns each)
    constexpr int RPI = 64 / LPR;                   // rows per pass of the wave
    constexpr int NIT = 32 / RPI;                   // passes per 32-row MFMA tile
    float* est = smem + wid * (32 * EST_LD);
    const bool vec_ok = ((d.ldc & 3) == 0) && skg_aligned16_dev(d.C) && skg_aligned16_dev(d.bias) &&
                        (EPI != SKG_EPI_MUL_RELU ||
                         (skg_aligned16_dev(d.P) && skg_aligned16_dev(d.Q) && skg_aligned16_dev(d.mbias) &&
                          skg_aligned16_dev(d.C_raw) && ((d.ldp | d.ldq | d.ldc_raw) & 3) == 0)) &&
                        (EPI != SKG_EPI_BIAS_RES_RELU || (skg_aligned16_dev(d.res) && (d.ldres & 3) == 0)) &&
                        (EPI != SKG_EPI_RELU_DOT || skg_aligned16_dev(d.dot_w));
    // ---- fast path: tile entirely inside the matrix, everything 16-byte aligned, no split-K.  Column-dependent
    // operands (bias, multiplier bias, dot weights) are loaded once per lane; rows advance by pointer increments.
    const bool interior = vec_ok && d.split_k <= 1 && (m0 + TBM <= d.M) && (n0 + TBN <= d.N);
    if (interior) {
        const int c4 = (lane % LPR) * 4, r0l = lane / LPR;
        const int col = n0 + wc * 32 * T + c4;
        float4 bia4 = make_float4(0.f, 0.f, 0.f, 0.f), mb4 = bia4, dw4 = bia4;
        if (d.bias) bia4 = *reinterpret_cast<const float4*>(d.bias + col);
        if (EPI == SKG_EPI_MUL_RELU && d.mbias) mb4 = *reinterpret_cast<const float4*>(d.mbias + col);
        if (EPI == SKG_EPI_RELU_DOT) dw4 = *reinterpret_cast<const float4*>(d.dot_w + col);
        // row indices of the multiplier gathers / the row scatter for all rows this lane will visit, fetched up front:
        // inside the row loop each one would sit in front of a dependent 16-byte gather, behind the previous row's stores
        // (fp16x2 loop only: its staging registers are free here; the exact kernels would lose a wave per SIMD to them)
        constexpr bool PRE = MODE == 2;
        int pix[T][NIT], qix[T][NIT], oix[T][NIT];
        auto row_indices = [&](int mi, int it) {
            const int row = m0 + wr * 32 * T + mi * 32 + r0l + it * RPI;
            pix[mi][it] = (EPI == SKG_EPI_MUL_RELU && d.P && d.p_idx) ? d.p_idx[row] : row;
            qix[mi][it] = (EPI == SKG_EPI_MUL_RELU && d.Q && d.q_idx) ? d.q_idx[row] : row;
            oix[mi][it] = (EPI != SKG_EPI_RELU_DOT && d.out_rows) ? d.out_rows[row] : row;
        };
        if constexpr (PRE) {
#pragma unroll
            for (int mi = 0; mi < T; ++mi)
#pragma unroll
                for (int it = 0; it < NIT; ++it) row_indices(mi, it);
        }
#pragma unroll
        for (int mi = 0; mi < T; ++mi) {
#pragma unroll
            for (int ni = 0; ni < T; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    est[((r & 3) + 8 * (r >> 2) + 4 * lh) * EST_LD + ni * 32 + li] = acc[mi][ni][r];
            const int rowb = m0 + wr * 32 * T + mi * 32 + r0l;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = rowb + it * RPI;
                if constexpr (!PRE) row_indices(mi, it);
                const float4 a4 = *reinterpret_cast<const float4*>(est + (it * RPI + r0l) * EST_LD + c4);
                float4 v = make_float4(a4.x + bia4.x, a4.y + bia4.y, a4.z + bia4.z, a4.w + bia4.w);
                if (EPI == SKG_EPI_RELU_DOT) {
                    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    float sdot = (v.x * dw4.x + v.y * dw4.y) + (v.z * dw4.z + v.w * dw4.w);
                    if (d.C) *reinterpret_cast<float4*>(d.C + (int64_t)row * d.ldc + col) = v;
#pragma unroll
                    for (int off = LPR / 2; off > 0; off >>= 1) sdot += __shfl_xor(sdot, off, 64);
                    if ((lane % LPR) == 0) d.dot_partial[(int64_t)(bn * 2 + wc) * d.M + row] = sdot;
                    continue;
                }
                const int orow = oix[mi][it];
                if (EPI == SKG_EPI_MUL_RELU) {
                    if (d.C_raw) *reinterpret_cast<float4*>(d.C_raw + (int64_t)row * d.ldc_raw + col) = v;
                    if (orow < 0) continue;
                    float4 m = mb4;
                    if (d.P) {
                        const int pi = pix[mi][it];
                        const float4 t = *reinterpret_cast<const float4*>(d.P + (int64_t)pi * d.ldp + col);
                        m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
                    }
                    if (d.Q) {
                        const int qi = qix[mi][it];
                        const float4 t = *reinterpret_cast<const float4*>(d.Q + (int64_t)qi * d.ldq + col);
                        m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w;
                    }
                    *reinterpret_cast<float4*>(d.C + (int64_t)orow * d.ldc + col) =
                        make_float4(fmaxf(v.x * m.x, 0.f), fmaxf(v.y * m.y, 0.f), fmaxf(v.z * m.z, 0.f), fmaxf(v.w * m.w, 0.f));
                    continue;
                }
                if (orow < 0) continue;
                if (EPI == SKG_EPI_BIAS_RELU || EPI == SKG_EPI_BIAS_RES_RELU)
                    v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                if (EPI == SKG_EPI_BIAS_RES_RELU) {
                    const float4 t = *reinterpret_cast<const float4*>(d.res + (int64_t)row * d.ldres + col);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                *reinterpret_cast<float4*>(d.C + (int64_t)orow * d.ldc + col) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < T; ++mi) {
#pragma unroll
        for (int ni = 0; ni < T; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                est[((r & 3) + 8 * (r >> 2) + 4 * lh) * EST_LD + ni * 32 + li] = acc[mi][ni][r];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = it * 64 + lane;
            const int rl = idx / LPR, c4 = (idx % LPR) * 4;
            const int row = m0 + wr * 32 * T + mi * 32 + rl;
            const int col = n0 + wc * 32 * T + c4;
            const float4 a4 = *reinterpret_cast<const float4*>(est + rl * EST_LD + c4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const bool rin = row < d.M;
            if (d.split_k > 1) {                                      // raw partial sums, [slice][M][N], no epilogue
                if (rin) {
                    float* wsp = d.split_ws + ((int64_t)split_slice * d.M + row) * d.N + col;
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (col + c < d.N) wsp[c] = v[c];
                }
                continue;
            }
            const bool full = vec_ok && (col + 3 < d.N);
            float bia[4] = {0.f, 0.f, 0.f, 0.f};
            if (d.bias) {
                if (full) {
                    const float4 t = *reinterpret_cast<const float4*>(d.bias + col);
                    bia[0] = t.x; bia[1] = t.y; bia[2] = t.z; bia[3] = t.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) bia[c] = (col + c < d.N) ? d.bias[col + c] : 0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += bia[c];

            if (EPI == SKG_EPI_RELU_DOT) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    v[c] = fmaxf(v[c], 0.f);
                    if (rin && col + c < d.N) s += v[c] * d.dot_w[col + c];
                }
                if (d.C && rin) {
                    if (full) *reinterpret_cast<float4*>(d.C + (int64_t)row * d.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (col + c < d.N) d.C[(int64_t)row * d.ldc + col + c] = v[c];
                }
#pragma unroll
                for (int off = LPR / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);      // LPR lanes share a row
                if ((lane % LPR) == 0 && rin) d.dot_partial[(int64_t)(bn * 2 + wc) * d.M + row] = s;
                continue;
            }
            if (!rin || col >= d.N) continue;
            const int orow = d.out_rows ? d.out_rows[row] : row;
            if (EPI == SKG_EPI_MUL_RELU) {
                if (d.C_raw) {
                    if (full) *reinterpret_cast<float4*>(d.C_raw + (int64_t)row * d.ldc_raw + col) = make_float4(v[0], v[1], v[2], v[3]);
                    else
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (col + c < d.N) d.C_raw[(int64_t)row * d.ldc_raw + col + c] = v[c];
                }
                if (orow < 0) continue;
                const int pi = d.p_idx ? d.p_idx[row] : row;
                const int qi = d.Q ? (d.q_idx ? d.q_idx[row] : row) : 0;
                float m[4] = {0.f, 0.f, 0.f, 0.f};
                if (full) {
                    if (d.mbias) { const float4 t = *reinterpret_cast<const float4*>(d.mbias + col); m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w; }
                    if (d.P) { const float4 t = *reinterpret_cast<const float4*>(d.P + (int64_t)pi * d.ldp + col); m[0] += t.x; m[1] += t.y; m[2] += t.z; m[3] += t.w; }
                    if (d.Q) { const float4 t = *reinterpret_cast<const float4*>(d.Q + (int64_t)qi * d.ldq + col); m[0] += t.x; m[1] += t.y; m[2] += t.z; m[3] += t.w; }
                    *reinterpret_cast<float4*>(d.C + (int64_t)orow * d.ldc + col) =
                        make_float4(fmaxf(v[0] * m[0], 0.f), fmaxf(v[1] * m[1], 0.f), fmaxf(v[2] * m[2], 0.f), fmaxf(v[3] * m[3], 0.f));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (col + c >= d.N) continue;
                        float mm = d.mbias ? d.mbias[col + c] : 0.f;
                        if (d.P) mm += d.P[(int64_t)pi * d.ldp + col + c];
                        if (d.Q) mm += d.Q[(int64_t)qi * d.ldq + col + c];
                        d.C[(int64_t)orow * d.ldc + col + c] = fmaxf(v[c] * mm, 0.f);
                    }
                }
                continue;
            }
            if (orow < 0) continue;
            if (EPI == SKG_EPI_BIAS_RELU || EPI == SKG_EPI_BIAS_RES_RELU) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            if (EPI == SKG_EPI_BIAS_RES_RELU) {
                if (full) {
                    const float4 t = *reinterpret_cast<const float4*>(d.res + (int64_t)row * d.ldres + col);
                    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (col + c < d.N) v[c] += d.res[(int64_t)row * d.ldres + col + c];
                }
            }
            if (full) *reinterpret_cast<float4*>(d.C + (int64_t)orow * d.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            else
#pragma unroll
                for (int c = 0; c < 4; ++c) if (col + c < d.N) d.C[(int64_t)orow * d.ldc + col + c] = v[c];
        }
    }
}

#define SKG_SMEM3 (2 * 64 * 68)              // latency loop: one 64 x 64(+4) tile per operand (34 KB)
#define SKG_SMEM4 (SKG_NB4 * 2 * 64 * 64)
template <int EPI, int MODE, int T>
__global__ __launch_bounds__(MODE == 5 ? 512 : 256, (MODE == 5 || (MODE == 4 && SKG_NB4 > 2)) ? 1 : SKG_MINW) void skg_gemm_kernel(const skg_gemm_desc d) {
    __shared__ __attribute__((aligned(1024))) float smem[MODE == 4 ? SKG_SMEM4 : (MODE == 3 || MODE == 5) ? SKG_SMEM3 : (T == 2 ? 2 * (A_TILE + B_TILE) : 4 * 32 * 36)];
    skg_gemm_tile<EPI, MODE, T>(d, blockIdx.x, smem);
}

// Several independent small GEMMs in ONE launch (node-row GEMMs with M = sum n_h or sum n fill a fraction of the 256
// CUs each; grouped they run side by side).  Block ranges: [start[g], start[g+1]).
struct skg_gemm_group_args {
    skg_gemm_desc d[SKG_GEMM_GROUP_MAX];
    int start[SKG_GEMM_GROUP_MAX + 1];
    int n;
};

__global__ __launch_bounds__(256, SKG_MINW) void skg_gemm_group_kernel(const skg_gemm_group_args g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_gemm_tile<-1, 0, 2>(g.d[k], blockIdx.x - g.start[k], smem);
}

// the same with the fp16x2 loop (every descriptor of the group carries a weight twin)
__global__ __launch_bounds__(256, SKG_MINW) void skg_gemm_group_split_kernel(const skg_gemm_group_args g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_gemm_tile<-1, 2, 2>(g.d[k], blockIdx.x - g.start[k], smem);
}

// ... with 64 x 64 tiles on the DMA-staged loop: the node-row GEMMs of a FEW images (M = sum n_h / sum n of one to a
// handful of graphs) and the grid-row GEMMs of a single image.  A 128 x 128 block walking all of K alone on its CU takes
// ~1.4 us per k-tile whatever M is (88 us at K = 1024: measured, 42 % of the batch-1 forward); four times the blocks,
// plus split-K for the one-tile-high problems, put the whole chip on them.
__global__ __launch_bounds__(256, SKG_MINW) void skg_gemm_group_small_kernel(const skg_gemm_group_args g) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 32 * 36];
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_gemm_tile<-1, 1, 1>(g.d[k], blockIdx.x - g.start[k], smem);
}

__global__ __launch_bounds__(256, SKG_NB4 > 2 ? 1 : SKG_MINW) void skg_gemm_group_direct_kernel(const skg_gemm_group_args g) {
    __shared__ __attribute__((aligned(1024))) float smem[SKG_SMEM4];
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_gemm_tile<-1, 4, 1>(g.d[k], blockIdx.x - g.start[k], smem);
}

__global__ __launch_bounds__(256, SKG_MINW) void skg_gemm_group_latency_kernel(const skg_gemm_group_args g) {
    __shared__ __attribute__((aligned(16))) float smem[SKG_SMEM3];
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_gemm_tile<-1, 3, 1>(g.d[k], blockIdx.x - g.start[k], smem);
}

__global__ __launch_bounds__(512, 1) void skg_gemm_group_khalves_kernel(const skg_gemm_group_args g) {
    __shared__ __attribute__((aligned(16))) float smem[SKG_SMEM3];
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_gemm_tile<-1, 5, 1>(g.d[k], blockIdx.x - g.start[k], smem);
}

// Split-K reduction: adds the slices in slice order (deterministic) and applies the plain epilogues.
__device__ __forceinline__ void skg_splitk_reduce_one(const skg_gemm_desc& d, int64_t i) {
    const int64_t total = (int64_t)d.M * d.N;
    if (i >= total) return;
    const int row = (int)(i / d.N), col = (int)(i % d.N);
    // four independent chains keep four loads in flight (one chain walks the slices at one HBM round trip each); the
    // order of the additions is fixed, so the result is deterministic
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int s = 0;
    for (; s + 3 < d.split_k; s += 4) {
        v0 += d.split_ws[(int64_t)s * total + i];
        v1 += d.split_ws[(int64_t)(s + 1) * total + i];
        v2 += d.split_ws[(int64_t)(s + 2) * total + i];
        v3 += d.split_ws[(int64_t)(s + 3) * total + i];
    }
    for (; s < d.split_k; ++s) v0 += d.split_ws[(int64_t)s * total + i];
    float v = (v0 + v1) + (v2 + v3);
    if (d.bias) v += d.bias[col];
    if (d.epilogue == SKG_EPI_BIAS_RELU || d.epilogue == SKG_EPI_BIAS_RES_RELU) v = fmaxf(v, 0.f);
    if (d.epilogue == SKG_EPI_BIAS_RES_RELU) v += d.res[(int64_t)row * d.ldres + col];
    const int orow = d.out_rows ? d.out_rows[row] : row;
    if (orow >= 0) d.C[(int64_t)orow * d.ldc + col] = v;
}

__global__ __launch_bounds__(256) void skg_splitk_reduce_kernel(const skg_gemm_desc d) {
    skg_splitk_reduce_one(d, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void skg_splitk_reduce_group_kernel(const skg_gemm_group_args g) {
    int k = 0;
#pragma unroll
    for (int t = 1; t < SKG_GEMM_GROUP_MAX; ++t)
        if (t < g.n && (int)blockIdx.x >= g.start[t]) k = t;
    skg_splitk_reduce_one(g.d[k], (int64_t)(blockIdx.x - g.start[k]) * 256 + threadIdx.x);
}

// One thread per (row, k quad) of the padded weight: writes 2 x 8 bytes in the plane format of the MODE 2 loop.
__global__ __launch_bounds__(256) void skg_split_weights_kernel(const float* __restrict__ W, int N, int K, int64_t ldw,
                                                                float scale, char* __restrict__ out) {
    const int nkt = (K + 15) >> 4, nnt = (N + 31) >> 5;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)nnt * 32 * nkt * 4;
    if (i >= total) return;
    const int qk = (int)(i % (nkt * 4)), row = (int)(i / (nkt * 4));
    const int kt = qk >> 2, q = qk & 3;
    h16x2 hp[2], mp[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float x[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = kt * 16 + q * 4 + 2 * e + t;
            x[t] = (row < N && k < K) ? W[(int64_t)row * ldw + k] * scale : 0.f;      // power-of-two scale: exact
        }
        const _Float16 h0 = (_Float16)x[0], h1 = (_Float16)x[1];
        hp[e] = h16x2{h0, h1};
        mp[e] = h16x2{(_Float16)(x[0] - (float)h0), (_Float16)(x[1] - (float)h1)};
    }
    char* dst = out + ((int64_t)(row >> 5) * nkt + kt) * 2048 + (q >> 1) * 512 + (row & 31) * 16 + (q & 1) * 8;
    *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(uint32_t, hp[0]), __builtin_bit_cast(uint32_t, hp[1]));
    *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(__builtin_bit_cast(uint32_t, mp[0]), __builtin_bit_cast(uint32_t, mp[1]));
}

extern "C" int64_t skg_split_weights_bytes(int N, int K) {
    if (N < 0 || K < 0) return SKG_E_ARG;
    return (int64_t)((N + 31) >> 5) * ((K + 15) >> 4) * 2048;
}

extern "C" int skg_split_weights_f16x2(const float* W, int N, int K, int64_t ldw, float scale, void* out, void* stream) {
    if (N < 0 || K < 0) return SKG_E_ARG;
    if (N == 0 || K == 0) return 0;
    int ex = 0;
    if (!W || !out || ldw < K || !skg_aligned16(out) || !(scale > 0.f) || frexpf(scale, &ex) != 0.5f) return SKG_E_ARG;
    const int64_t total = (int64_t)((N + 31) >> 5) * 32 * ((K + 15) >> 4) * 4;
    hipLaunchKernelGGL(skg_split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       W, N, K, ldw, scale, reinterpret_cast<char*>(out));
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ row exponents of an operand
// exp_out[r] = floor(log2(max |A[r, :]|)) - 11 (0 for an all-zero row and for rows holding inf / nan), clamped to
// [-126, 126]: the power of two the split-operand loop divides row r by.  One wave per row, 16-byte loads.
__global__ __launch_bounds__(256) void skg_row_exponents_kernel(const float* __restrict__ A, int64_t lda,
                                                                const int32_t* __restrict__ a_rows, int M, int K,
                                                                int32_t* __restrict__ exp_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int src = a_rows ? a_rows[row] : row;
    uint32_t mx = 0u;
    if (src >= 0) {
        const float* p = A + (int64_t)src * lda;
        for (int k = 4 * lane; k < K; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(p + k);
            mx = max(max(mx, __float_as_uint(v.x) & 0x7fffffffu), __float_as_uint(v.y) & 0x7fffffffu);
            mx = max(max(mx, __float_as_uint(v.z) & 0x7fffffffu), __float_as_uint(v.w) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
    if (lane == 0) {
        const int E = (int)(mx >> 23);
        exp_out[row] = (mx == 0u || E == 255) ? 0 : min(max(E - 127 - 11, -126), 126);
    }
}

extern "C" int skg_row_exponents_f32(const float* A, int64_t lda, const int32_t* a_rows, int M, int K, int32_t* exp_out,
                                     void* stream) {
    if (M < 0 || K <= 0 || (K & 3) || (lda & 3)) return SKG_E_ARG;
    if (M == 0) return 0;
    if (!A || !exp_out) return SKG_E_ARG;
    if (!skg_aligned16(A)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_row_exponents_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, A, lda, a_rows, M, K,
                       exp_out);
    return skg_launch_status();
}

// Which 64 x 64 loop the small launches take: 4 = latency loop staged straight into LDS (64 k per step; needs K % 64 == 0,
// otherwise 3), 3 = latency loop (64 k per step, register staged), 1 = DMA-staged 16-k steps.  A developer switch for A/B measurements (tools/small_batch_loop.py); both give bit-identical results only
// with themselves (the summation order over k differs).
// Round 4, measured on the single-image forward (profiles/r04_b1_eval_timeline_direct_lds_loop.txt): the direct-to-LDS loop (4)
// is NOT faster than the register-staged one (3) -- single launches of 208 workgroups 24-30 us against 25-32, the 624-workgroup
// group 76 against 64 us (its 96 KiB ring leaves one workgroup per CU), B = 1 0.517 against 0.498 ms.  A step of this tile is
// a chain of 32 dependent fp32 MFMAs (0.93 us) plus a ~0.4 us bubble at the barrier whatever stages the tile: the launch is
// bound by that chain at one wave per SIMD, not by the staging.  Mode 3 stays the default; 4 is selectable.
// (round 5: the switches live in the calling thread's current context -- skg_tuning, skg_common.h -- not in the library)
#define g_small_mode skg_tune_small_mode()
#define g_khalves_blocks skg_tune_khalves_blocks()
// A launch (or group) counts as small -- 64 x 64 tiles -- below this many 128 x 128 tiles (split-K slices included): 384.
#define g_small_tiles skg_tune_small_tiles()

// 64 x 64 tiles when the 128 x 128 grid would leave most CUs idle (small M: low-batch inference); needs the DMA path.
static int skg_gemm_tile_scale(const skg_gemm_desc* d) {
    const bool glds = SKG_USE_GLDS && BK == 16 && (d->K % BK) == 0 && !d->a_rows &&
                      (int64_t)BM * d->lda * 4 < 0xffffffffLL && (int64_t)BN * d->ldw * 4 < 0xffffffffLL;
    const int64_t tiles128 = (int64_t)((d->M + 127) / 128) * ((d->N + 127) / 128) * (d->split_k > 1 ? d->split_k : 1);
    if (d->w_split && (d->K % 16) == 0 && d->w_scale > 0.f) return 2;
    // M <= 64: a 128-row tile would spend half of its MFMAs on padding rows whatever the grid size (box_head at one image)
    return (glds && (tiles128 < g_small_tiles || d->M <= 64)) ? 1 : 2;
}

extern "C" int skg_gemm_dot_partials(const skg_gemm_desc* dh) {
    if (!dh) return SKG_E_ARG;
    const int T = skg_gemm_tile_scale(dh);
    return 2 * ((dh->N + 64 * T - 1) / (64 * T));
}

static int skg_gemm_validate(const skg_gemm_desc& d) {
    if (d.split_k > 1 && (!d.split_ws || (d.epilogue != SKG_EPI_BIAS && d.epilogue != SKG_EPI_BIAS_RELU &&
                                          d.epilogue != SKG_EPI_BIAS_RES_RELU) || d.split_k > 64))
        return SKG_E_ARG;
    if (d.M < 0 || d.N <= 0 || d.K <= 0 || !d.A || !d.W) return SKG_E_ARG;
    if ((d.K & 3) || (d.lda & 3) || (d.ldw & 3)) return SKG_E_ALIGN;
    if (!skg_aligned16(d.A) || !skg_aligned16(d.W)) return SKG_E_ALIGN;
    if (d.lda < d.K && !d.a_rows && d.M > 1) return SKG_E_ARG;
    switch (d.epilogue) {
        case SKG_EPI_BIAS: case SKG_EPI_BIAS_RELU:
            if (!d.C) return SKG_E_ARG; break;
        case SKG_EPI_MUL_RELU:
            if (!d.C || (!d.P && !d.Q)) return SKG_E_ARG; break;
        case SKG_EPI_RELU_DOT:
            if (!d.dot_w || !d.dot_partial) return SKG_E_ARG; break;
        case SKG_EPI_BIAS_RES_RELU:
            if (!d.C || !d.res) return SKG_E_ARG; break;
        default: return SKG_E_ARG;
    }
    return 0;
}

// ---- mid-size launches on the free-layout GEMM.  Between "a few hundred workgroups of 64 x 64" (one image: the latency loop)
// and "thousands of 128 x 128 tiles" (the DMA-staged loop, two workgroups per CU) lies the regime of 2-8 images: 100-400
// tiles of 128 x 128, at most one workgroup per CU.  There skg_gemmx_f32's loop (three register stages of prefetch, double
// LDS buffer, staged epilogue) is the fastest of the three: M = 1600, N = K = 1024 split in two 50 us against 80-92 us on
// either eval loop; the three fc_2 products of four images as one launch ~2.5x faster than the grouped eval kernel.  A launch
// goes there when it has at least g_route_tiles tiles of 128 x 128 (split slices counted) but is still "small", carries
// no weight twin / row gather / dot epilogue, and its fused epilogue can run in the staged epilogue (skg_gemmx_can_fuse).
// The summation order over k differs from the eval loops (as theirs do from each other): results agree to rounding.
#define g_route_tiles skg_tune_route_tiles()

static bool skg_route_desc(const skg_gemm_desc& d, skg_gemmx_desc& x, skg_gemmx_fused& f) {
    if (d.w_split || d.a_rows || d.a_exp || d.epilogue == SKG_EPI_RELU_DOT) return false;
    memset(&x, 0, sizeof(x)); memset(&f, 0, sizeof(f));
    x.A = d.A; x.a_sm = d.lda; x.a_sk = 1;
    x.B = d.W; x.b_sn = d.ldw; x.b_sk = 1;
    x.C = d.C; x.ldc = d.ldc; x.M = d.M; x.N = d.N; x.K = d.K;
    x.bias = d.bias; x.relu = d.epilogue == SKG_EPI_BIAS_RELU ? 1 : 0;
    x.split_k = d.split_k > 1 ? d.split_k : 0; x.split_ws = d.split_ws;       // same [slice][M][N] workspace layout
    f.out_rows = d.out_rows;
    if (d.epilogue == SKG_EPI_MUL_RELU) {
        f.kind = SKG_EPI_MUL_RELU;
        f.P = d.P; f.p_idx = d.p_idx; f.ldp = d.ldp; f.Q = d.Q; f.q_idx = d.q_idx; f.ldq = d.ldq; f.mbias = d.mbias;
        f.C_raw = d.C_raw; f.ldc_raw = d.ldc_raw;
    } else if (d.epilogue == SKG_EPI_BIAS_RES_RELU) {
        f.kind = SKG_EPI_BIAS_RES_RELU; f.res = d.res; f.ldres = d.ldres;
    }
    return skg_gemmx_can_fuse(&x, &f) != 0;
}

// 64 x 64 tiles for a whole group?  Every member must be able to take the DMA-staged loop, none may carry a weight twin,
// and together they must be small (the same bound as a single launch: fewer than 384 tiles of 128 x 128).
static bool skg_gemm_group_small(const skg_gemm_desc* descs, int n) {
    int64_t tiles128 = 0;
    for (int i = 0; i < n; ++i) {
        const skg_gemm_desc& d = descs[i];
        if (d.M == 0) continue;
        const bool glds = SKG_USE_GLDS && BK == 16 && (d.K % BK) == 0 && !d.a_rows &&
                          (int64_t)BM * d.lda * 4 < 0xffffffffLL && (int64_t)BN * d.ldw * 4 < 0xffffffffLL;
        if (!glds || (d.w_split && d.w_scale > 0.f)) return false;
        tiles128 += (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128);
    }
    return tiles128 < g_small_tiles;
}

extern "C" int skg_gemm_group_tile(const skg_gemm_desc* descs_host, int n) {
    if (!descs_host || n < 1 || n > SKG_GEMM_GROUP_MAX) return SKG_E_ARG;
    return skg_gemm_group_small(descs_host, n) ? 1 : 2;
}

extern "C" int skg_gemm_group_f32(const skg_gemm_desc* descs_host, int n, void* stream) {
    if (!descs_host || n < 1 || n > SKG_GEMM_GROUP_MAX) return SKG_E_ARG;
    skg_gemm_group_args g, r;
    g.n = r.n = 0;
    int64_t blocks = 0, rblocks = 0;
    bool split = true;
    for (int i = 0; i < n; ++i) {
        const int rc = skg_gemm_validate(descs_host[i]);
        if (rc) return rc;
    }
    const bool small = skg_gemm_group_small(descs_host, n);
    if (n <= SKG_GEMMX_GROUP_MAX) {                        // mid-size group: the free-layout GEMM (see g_route_tiles)
        skg_gemmx_desc xs[SKG_GEMMX_GROUP_MAX];
        skg_gemmx_fused fs[SKG_GEMMX_GROUP_MAX];
        int64_t tiles128 = 0;
        bool ok = true;
        for (int i = 0; i < n && ok; ++i) {
            const skg_gemm_desc& d = descs_host[i];
            ok = skg_route_desc(d, xs[i], fs[i]);
            tiles128 += (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * (d.split_k > 1 ? d.split_k : 1);
        }
        if (ok && tiles128 >= g_route_tiles && tiles128 < 4 * g_small_tiles) return skg_gemmx_f32_fused(xs, fs, n, stream);
    }
    for (int i = 0; i < n; ++i) {
        const skg_gemm_desc& d = descs_host[i];
        if (d.split_k > 1 && !small) return SKG_E_ARG;                     // split-K only with the 64 x 64 tiles
        if (d.M == 0) continue;
        split = split && d.w_split && (d.K % 16) == 0 && d.w_scale > 0.f;
        const int64_t nb = skg_gemm_blocks(d.M, d.N, d.K, small ? 1 : 2) * (d.split_k > 1 ? d.split_k : 1);
        if (blocks + nb > 0x7fffffffLL) return SKG_E_LIMIT;
        g.d[g.n] = d;
        g.start[g.n] = (int)blocks;
        blocks += nb;
        ++g.n;
        if (d.split_k > 1) {
            r.d[r.n] = d;
            r.start[r.n] = (int)rblocks;
            rblocks += ((int64_t)d.M * d.N + 255) / 256;
            ++r.n;
        }
    }
    if (g.n == 0) return 0;
    for (int i = g.n; i <= SKG_GEMM_GROUP_MAX; ++i) g.start[i] = (int)blocks;
    bool direct = small && g_small_mode == 4;               // every member with whole 64-k steps: the direct-to-LDS loop
    for (int i = 0; i < g.n && direct; ++i) direct = (g.d[i].K % 64) == 0;
    const bool halves = small && (g_small_mode == 5 || (g_small_mode == 6 && blocks <= g_khalves_blocks));
    if (direct) hipLaunchKernelGGL(skg_gemm_group_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else if (halves) hipLaunchKernelGGL(skg_gemm_group_khalves_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, g);
    else if (small && g_small_mode >= 3) hipLaunchKernelGGL(skg_gemm_group_latency_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else if (small) hipLaunchKernelGGL(skg_gemm_group_small_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else if (split) hipLaunchKernelGGL(skg_gemm_group_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(skg_gemm_group_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
    if (r.n) {
        for (int i = r.n; i <= SKG_GEMM_GROUP_MAX; ++i) r.start[i] = (int)rblocks;
        hipLaunchKernelGGL(skg_splitk_reduce_group_kernel, dim3((unsigned)rblocks), dim3(256), 0, (hipStream_t)stream, r);
    }
    return skg_launch_status();
}

extern "C" int skg_gemm_f32(const skg_gemm_desc* dh, void* stream) {
    if (!dh) return SKG_E_ARG;
    const skg_gemm_desc d = *dh;
    const int rc = skg_gemm_validate(d);
    if (rc) return rc;
    if (d.M == 0) return 0;
    const bool glds = SKG_USE_GLDS && BK == 16 && (d.K % BK) == 0 && !d.a_rows &&
                      (int64_t)BM * d.lda * 4 < 0xffffffffLL && (int64_t)BN * d.ldw * 4 < 0xffffffffLL;
    const bool split = d.w_split && (d.K % 16) == 0 && d.w_scale > 0.f;
    const int T = split ? 2 : skg_gemm_tile_scale(&d);
    if (!split && T == 1 && d.M > 64) {                    // mid-size launch: the free-layout GEMM (see g_route_tiles)
        const int64_t tiles128 = (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * (d.split_k > 1 ? d.split_k : 1);
        skg_gemmx_desc x; skg_gemmx_fused f;
        if (tiles128 >= g_route_tiles && skg_route_desc(d, x, f)) return skg_gemmx_f32_fused(&x, &f, 1, stream);
    }
    const int64_t nblk = skg_gemm_blocks(d.M, d.N, d.K, T) * (d.split_k > 1 ? d.split_k : 1);
    if (nblk > 0x7fffffffLL) return SKG_E_LIMIT;
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t s = (hipStream_t)stream;
#define SKG_LAUNCH(E)                                                                              \
    if (split) hipLaunchKernelGGL((skg_gemm_kernel<E, 2, 2>), grid, block, 0, s, d);               \
    else if (glds && T == 1 && g_small_mode == 4 && (d.K % 64) == 0) hipLaunchKernelGGL((skg_gemm_kernel<E, 4, 1>), grid, block, 0, s, d); \
    else if (glds && T == 1 && (g_small_mode == 5 || (g_small_mode == 6 && nblk <= g_khalves_blocks))) hipLaunchKernelGGL((skg_gemm_kernel<E, 5, 1>), grid, dim3(512), 0, s, d); \
    else if (glds && T == 1 && g_small_mode >= 3) hipLaunchKernelGGL((skg_gemm_kernel<E, 3, 1>), grid, block, 0, s, d); \
    else if (glds && T == 1) hipLaunchKernelGGL((skg_gemm_kernel<E, 1, 1>), grid, block, 0, s, d); \
    else if (glds) hipLaunchKernelGGL((skg_gemm_kernel<E, 1, 2>), grid, block, 0, s, d);           \
    else hipLaunchKernelGGL((skg_gemm_kernel<E, 0, 2>), grid, block, 0, s, d);
    switch (d.epilogue) {
        case SKG_EPI_BIAS:          SKG_LAUNCH(SKG_EPI_BIAS) break;
        case SKG_EPI_BIAS_RELU:     SKG_LAUNCH(SKG_EPI_BIAS_RELU) break;
        case SKG_EPI_MUL_RELU:      SKG_LAUNCH(SKG_EPI_MUL_RELU) break;
        case SKG_EPI_RELU_DOT:      SKG_LAUNCH(SKG_EPI_RELU_DOT) break;
        case SKG_EPI_BIAS_RES_RELU: SKG_LAUNCH(SKG_EPI_BIAS_RES_RELU) break;
    }
#undef SKG_LAUNCH
    if (d.split_k > 1) {
        const int64_t total = (int64_t)d.M * d.N;
        hipLaunchKernelGGL(skg_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d);
    }
    return skg_launch_status();
}
