// skg_train_plan.hip -- native launch plan of the fused TRAINING step's dense part (include/skghoi.h, skg_train_plan).
//
// The reference's training step is `zero_grad -> net -> backward -> step` (utils.py:213-229) over eager PyTorch: autograd
// issues ~1000 launches for GraphHead.forward (HEAD:769-993) and its backward.  The first MI355X version cut that to ~130
// launches but issued them from Python (descriptor objects, ~100 tensor allocations, ctypes marshalling): ~3.4 ms of
// host time per batch-4 step in front of 2.1 ms of GPU work.  Here the whole launch sequence of the forward (after the
// step's one host synchronisation) and of the backward is ONE C call each: descriptors are plain structs on the stack,
// every activation lives at a fixed offset of one caller-provided workspace, split-K scratch is a single region reused
// by consecutive launches of the stream.  The sequence itself -- which kernels, in which order, on which operands -- is
// the one skghoi_amd/train_fused.py (TrainJob) describes; the algebra is DESIGN.md section 3.
#include <hip/hip_runtime.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include "skghoi.h"
#include "skg_common.h"

// ---- optional per-launch timing of the plan's dense products (skg_train_plan.timer; bench.py's roofline record) ----------
struct skg_train_timer {
    std::mutex m;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;        // 2 * cap events, timing enabled
    double* flops = nullptr;
};

namespace {

struct Mat {                      // a row-major fp32 matrix view
    float* p; int64_t ld; int rows, cols;
    Mat() : p(nullptr), ld(0), rows(0), cols(0) {}
    Mat(float* p_, int r, int c) : p(p_), ld(c), rows(r), cols(c) {}
    Mat(float* p_, int r, int c, int64_t ld_) : p(p_), ld(ld_), rows(r), cols(c) {}
    Mat from_col(int c0) const { return Mat(p + c0, rows, cols - c0, ld); }
    Mat row_range(int r0, int n) const { return Mat(p + (int64_t)r0 * ld, n, cols, ld); }
};
static inline Mat cm(const float* p, int r, int c) { return Mat(const_cast<float*>(p), r, c); }

constexpr int BLK_SHIFT = 6;                 // fc_3 weights are stored branch-major [16][1024][64]
constexpr int64_t BLK_STRIDE = 1024 * 64;

struct Ws;
struct Ctx {
    const skg_train_plan* P;
    hipStream_t stream;
    bool dry;                    // sizing pass: nothing is launched, the scratch high-water mark is recorded
    int64_t scratch_need;        // floats
    float* scratch;              // split-K partials (one region, reused by consecutive launches of the stream)
    int rc;
    double flops;                // 2 M N K over every dense product issued (sizing pass: what the plan WILL issue)
    const float* no_twin;        // workspace buffer whose twin nobody writes (dadj: produced next to dWt, one column wide)
    const struct Ws* w;          // the workspace layout: where the SIDE twins live (twin_of)
    // second branch (skg_branch: a stream, two events, a split-K scratch region of its own) or NULL: one stream
    struct skg_branch* br;
};
// What a two-branch plan call needs besides the caller's stream; owned by a context (skg_ctx_train_forward_f32 / the worker).
struct skg_branch {
    hipStream_t stream;
    hipEvent_t fork, join;       // no timing, device-scope release
    bool forked;                 // a backward's node chain is out on the branch, not joined yet: a staged (data-parallel) backward
                                 // forks in the call that holds stage 6 and joins in the one that completes the next arena chunk
};
// Runs the launches of its scope on the branch's stream with the branch's scratch region.
struct OnBranch {
    Ctx& c; hipStream_t s0; float* sc0;
    OnBranch(Ctx& c_, float* scratch2) : c(c_), s0(c_.stream), sc0(c_.scratch) { c.stream = c.br->stream; c.scratch = scratch2; }
    ~OnBranch() { c.stream = s0; c.scratch = sc0; }
};

// bf16 twin of a tensor the plan's products read or write, or NULL (defined behind the workspace layout); *ld receives the
// twin's pitch where it is a padded copy with a pitch of its own (0: indexed like the fp32 tensor)
static uint16_t* twin_of(const Ctx& c, const float* p, int64_t* ld);

// announces the twin ranges to the per-row kernels for the duration of one plan call (skg_common.h, skg_tls_twin)
struct TwinScope {
    skg_twin_map saved;
    explicit TwinScope(const skg_train_plan* P) : saved(skg_tls_twin) {
        skg_twin_map m = {{nullptr, nullptr}, {nullptr, nullptr}, {0, 0}};
        if (P->bf16 && P->ws16) {
            m.base[0] = P->ws; m.base16[0] = P->ws16; m.n[0] = P->ws_floats;
            if (P->pf16) { m.base[1] = P->pair_features; m.base16[1] = P->pf16; m.n[1] = (int64_t)(P->Mp > 0 ? P->Mp : 1) * 2048; }
        }
        skg_tls_twin = m;
    }
    ~TwinScope() { skg_tls_twin = saved; }
};

// ---- skg_gemmx descriptor builders (skghoi_amd/gemmx.py: forward / input_grad / weight_grad)
static skg_gemmx_desc op_zero() { skg_gemmx_desc d; memset(&d, 0, sizeof(d)); return d; }

static skg_gemmx_desc FWD(const Mat& x, const Mat& W, const Mat& out, const float* bias, bool relu, int M = -1, int K = -1,
                          int N = -1, bool blocks = false) {
    skg_gemmx_desc d = op_zero();
    d.A = x.p; d.a_sm = x.ld; d.a_sk = 1;
    d.B = W.p; d.b_sk = 1;
    d.C = out.p; d.ldc = out.ld;
    d.M = M < 0 ? x.rows : M; d.K = K < 0 ? x.cols : K; d.N = N < 0 ? out.cols : N;
    d.bias = bias; d.relu = relu ? 1 : 0;
    if (!blocks) d.b_sn = W.ld;
    else { d.b_kshift = BLK_SHIFT; d.b_kstride = BLK_STRIDE; d.b_sn = 1 << BLK_SHIFT; }
    return d;
}
static skg_gemmx_desc IG(const Mat& dz, const Mat& W, const Mat& dx, const Mat* mask, bool accumulate, int M = -1,
                         int N_in = -1, int K_out = -1, bool blocks = false) {
    skg_gemmx_desc d = op_zero();
    d.A = dz.p; d.a_sm = dz.ld; d.a_sk = 1;
    d.B = W.p; d.b_sn = 1;
    d.C = dx.p; d.ldc = dx.ld;
    d.M = M < 0 ? dz.rows : M; d.N = N_in < 0 ? dx.cols : N_in; d.K = K_out < 0 ? dz.cols : K_out;
    if (mask) { d.mask = mask->p; d.ldmask = mask->ld; }
    d.accumulate = accumulate ? 1 : 0;
    if (!blocks) d.b_sk = W.ld;
    else { d.b_nshift = BLK_SHIFT; d.b_nstride = BLK_STRIDE; d.b_sk = 1 << BLK_SHIFT; }
    return d;
}
static skg_gemmx_desc WG(const Mat& dz, const Mat& x, const Mat& dW, float* db, bool accumulate, int rows = -1,
                         int n_out = -1, int k_in = -1, bool blocks = false) {
    skg_gemmx_desc d = op_zero();
    d.A = dz.p; d.a_sm = 1; d.a_sk = dz.ld;
    d.B = x.p; d.b_sn = 1; d.b_sk = x.ld;
    d.C = dW.p;
    d.M = n_out < 0 ? dz.cols : n_out; d.N = k_in < 0 ? x.cols : k_in; d.K = rows < 0 ? dz.rows : rows;
    d.a_rowsum = db; d.accumulate = accumulate ? 1 : 0;
    if (!blocks) d.ldc = dW.ld;
    else { d.c_nshift = BLK_SHIFT; d.c_nstride = BLK_STRIDE; d.ldc = 1 << BLK_SHIFT; }
    return d;
}

// Split-K factor of one product: measured on MI355X at the training shapes (tools/gemmx_split_sweep.py; re-swept on the whole
// batch-4 step in round 3: 200 ... 800 / 500 ... 1400 workgroups and a per-LAUNCH instead of per-product count all came out
// level or slower) -- the exact fp32 loop was best at ~1000 workgroups, the bf16 loop at ~450; slices keep >= 128 k.
// With the staged epilogue and the four-column reduce a workgroup costs less and a slice's round trip relatively more:
// re-swept on the step, bf16 128 / 192 / 256 / 320 / 448 / 640 -> 1.503 / 1.489 / 1.486 / 1.555 / 1.546 / 1.61 ms,
// fp32 300 / 500 / 750 / 1000 / 1400 -> 2.676 / 2.672 / 2.703 / 2.696 / 2.81 ms.
// End of round 4 (direct-to-LDS kernel on most launches): bf16 64 / 96 / 128 / 160 / 208 / 256 / 320 -> 1.332 / 1.260 / 1.274 /
// 1.258 / 1.279 (other box) / 1.265 / 1.378 ms; fp32 350 / 500 / 700 level (2.49 / 2.48 / 2.49).
// (the knobs travel in the plan -- skg_train_plan.split_target / split_max -- not in the library: no process-wide state)
static int pick_split(const skg_train_plan* P, const skg_gemmx_desc& o, int bk) {
    int64_t tiles = (int64_t)((o.M + 127) / 128) * ((o.N + 127) / 128);
    int kt = (o.K + bk - 1) / bk;
    const int target = P->split_target > 0 ? P->split_target : (bk == 16 ? 500 : 160);
    const int smax = P->split_max > 0 ? P->split_max : 64;
    int cap = kt * bk / 128; if (cap > smax) cap = smax;
    if (tiles == 0 || cap < 2) return 1;
    int sk = (int)((double)target / (double)tiles + 0.5);
    if (sk > cap) sk = cap;
    return sk < 1 ? 1 : sk;
}

static void launch(Ctx& c, skg_gemmx_desc* ops, int n) {
    if (c.rc) return;
    const bool bf16 = c.P->bf16 != 0;
    const int bk = bf16 ? 32 : 16;
    skg_gemmx_desc live[16];
    int m = 0;
    for (int i = 0; i < n; ++i)
        if (ops[i].M > 0 && ops[i].N > 0) live[m++] = ops[i];
    for (int i0 = 0; i0 < m; i0 += SKG_GEMMX_GROUP_MAX) {
        int cnt = m - i0 < SKG_GEMMX_GROUP_MAX ? m - i0 : SKG_GEMMX_GROUP_MAX;
        int64_t used = 0, used_ctr = 0;
        for (int i = 0; i < cnt; ++i) {
            skg_gemmx_desc& d = live[i0 + i];
            int sk = pick_split(c.P, d, bk);
            d.split_k = sk > 1 ? sk : 0;
            d.split_ctr = nullptr;
            if (sk > 1 && c.P->counters) {                 // reduced inside the product launch: one counter per tile
                const int64_t tiles = (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128);
                if (used_ctr + tiles <= c.P->n_counters) { d.split_ctr = c.P->counters + used_ctr; used_ctr += tiles; }
            }
            if (bf16 && !c.dry) {
                d.A16 = twin_of(c, d.A, &d.a16_ld); d.B16 = twin_of(c, d.B, &d.b16_ld);
                int64_t ldc16 = 0;
                uint16_t* c16 = twin_of(c, d.C, &ldc16);
                d.C16 = (c16 && !ldc16 && (d.C < c.P->params || d.C >= c.P->params + c.P->params_floats)) ? c16 : nullptr;
            }
            if (sk > 1) {
                int64_t need = (int64_t)sk * ((int64_t)d.M * d.N + d.M);
                need = (need + 7) & ~(int64_t)7;
                d.split_ws = c.scratch ? c.scratch + used : nullptr;
                used += need;
            }
        }
        if (used > c.scratch_need) c.scratch_need = used;
        for (int i = 0; i < cnt; ++i) c.flops += 2.0 * live[i0 + i].M * (double)live[i0 + i].N * live[i0 + i].K;
        if (c.dry) continue;
        skg_train_timer* tm = c.P->timer;
        int slot = -1;
        double fl = 0.0;
        if (tm) {
            for (int i = 0; i < cnt; ++i) fl += 2.0 * live[i0 + i].M * (double)live[i0 + i].N * live[i0 + i].K;
            std::lock_guard<std::mutex> g(tm->m);
            if (tm->n < tm->cap) { slot = tm->n++; tm->flops[slot] = fl; }
        }
        if (slot >= 0) (void)hipEventRecord(tm->ev[2 * slot], c.stream);
        int rc = bf16 ? skg_gemmx_bf16(live + i0, cnt, c.stream) : skg_gemmx_f32(live + i0, cnt, c.stream);
        if (slot >= 0) (void)hipEventRecord(tm->ev[2 * slot + 1], c.stream);
        if (rc) { c.rc = rc; return; }
    }
}
#define CK(call) do { if (!c.rc && !c.dry) { int rc__ = (call); if (rc__) c.rc = rc__; } } while (0)

// ---- tiny kernels of the plan ------------------------------------------------------------------------------------------
__global__ void b3sum_kernel(const float* __restrict__ b3, float* __restrict__ out) {      // [4][16][1024] -> [4][1024]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 1024) return;
    int m = i >> 10, col = i & 1023;
    float s = 0.f;
    for (int b = 0; b < 16; ++b) s += b3[((int64_t)m * 16 + b) * 1024 + col];
    out[i] = s;
}
__global__ void b3bcast_kernel(const float* __restrict__ db3, float* __restrict__ out) {    // [4][1024] -> [4][16][1024]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 4 * 16 * 1024) return;
    int m = i >> 14, col = i & 1023;
    out[i] = db3[m * 1024 + col];
}
// out[c] = sum_r X[r, c] + sum_r Y[r, c]  (gradient of attention_head.fc_1's bias: added once per row of both uses)
// (sixteen row groups per workgroup of 1024 threads: four groups walked ~80 node rows in 40 dependent round trips, 17 us)
__global__ __launch_bounds__(1024) void colsum2_kernel(const float* __restrict__ X, int rx, const float* __restrict__ Y, int ry,
                                                       float* __restrict__ out) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, col = blockIdx.x * 64 + lane;
    const int part = threadIdx.x >> 6;
    float s = 0.f, t = 0.f;
    for (int r = part; r < rx; r += 16) s += X[(int64_t)r * 1024 + col];
    for (int r = part; r < ry; r += 16) t += Y[(int64_t)r * 1024 + col];
    red[part][lane] = s + t;
    __syncthreads();
    if (part == 0) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q += 4) a += (red[q][lane] + red[q + 1][lane]) + (red[q + 2][lane] + red[q + 3][lane]);
        out[col] = a;
    }
}

// Gradient of the adjacency Linear(1024 -> 1) (HEAD:644, 897): dw[c] = sum_r dadj[r] Wt[r, c], db = sum_r dadj[r].  As a product
// of the step (M = 1) it kept its whole launch -- the attention fc_3's dX and dW, 13.4 GFLOP -- off the direct-to-LDS kernel
// (a row-contiguous twin is read in 8-row pieces); here: partial sums over 16-row chunks (4 x ceil(rows / 16) workgroups, one
// column per thread), then the chunks in a fixed order.  fp32 operands, fixed order.
#define ADJW_ROWS 16                         // rows per partial chunk: 16 independent loads per thread (64 made the kernel a
                                             // chain of load latencies: 17.9 us for 13 MB; the finish summed 50 chunks serially: 15 us)
__global__ __launch_bounds__(256) void adjw_partial_kernel(const float* __restrict__ dadj, const float* __restrict__ Wt,
                                                           int rows, float* __restrict__ part) {
    const int col = (blockIdx.x & 3) * 256 + threadIdx.x, chunk = blockIdx.x >> 2;
    const int r0 = chunk * ADJW_ROWS;
    float s = 0.f, sb = 0.f;
    if (r0 + ADJW_ROWS <= rows) {
        float d[ADJW_ROWS], v[ADJW_ROWS];
#pragma unroll
        for (int i = 0; i < ADJW_ROWS; ++i) { d[i] = dadj[r0 + i]; v[i] = Wt[(int64_t)(r0 + i) * 1024 + col]; }
#pragma unroll
        for (int i = 0; i < ADJW_ROWS; ++i) { s += d[i] * v[i]; sb += d[i]; }
    } else {
        for (int r = r0; r < rows; ++r) {
            const float d = dadj[r];
            s += d * Wt[(int64_t)r * 1024 + col];
            sb += d;
        }
    }
    part[(int64_t)chunk * 1028 + col] = s;
    if (col == 0) part[(int64_t)chunk * 1028 + 1024] = sb;
}
// out[col] = sum over the chunks in a FIXED order: sixteen chunk groups per workgroup (chunks g, g + 16, ...), then the groups
__global__ __launch_bounds__(1024) void adjw_finish_kernel(const float* __restrict__ part, int chunks, float* __restrict__ dw,
                                                           float* __restrict__ db) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, col = blockIdx.x * 64 + lane;   // 17 workgroups: 1024 columns + the bias
    float s = 0.f;
    if (col <= 1024)
        for (int k = g; k < chunks; k += 16) s += part[(int64_t)k * 1028 + col];
    red[g][lane] = s;
    __syncthreads();
    if (g == 0 && col <= 1024) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q += 4) a += (red[q][lane] + red[q + 1][lane]) + (red[q + 2][lane] + red[q + 3][lane]);
        if (col < 1024) dw[col] = a; else db[0] = a;
    }
}

// dF [Mg, 4096] = [attention | obj_to_sub | sub_to_obj | global]: the read-out writes blocks 0 and 3 at the rows of kept pairs;
// the in-loop stage overwrites blocks 1 and 2 on every row and ADDS to block 0.  What has to be zero beforehand is therefore
// blocks 0 and 3 of the grid rows WITHOUT a pair (the self pairs, grid_pair < 0: 20 of an image's 400 rows) -- this kernel --
// not the whole 52 MB (+ 26 MB twin) a memset node cleared every step (15 us).  One workgroup per grid row.
__global__ __launch_bounds__(256) void zero_selfpair_rows_kernel(float* __restrict__ dF, uint16_t* __restrict__ dF16,
                                                                 const int32_t* __restrict__ grid_pair, int Mg) {
    const int g = blockIdx.x;
    if (grid_pair[g] >= 0) return;
    const int c = threadIdx.x * 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(dF + (int64_t)g * 4096 + c) = z;
    *reinterpret_cast<float4*>(dF + (int64_t)g * 4096 + 3072 + c) = z;
    if (dF16) {
        *reinterpret_cast<uint2*>(dF16 + (int64_t)g * 4096 + c) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(dF16 + (int64_t)g * 4096 + 3072 + c) = make_uint2(0u, 0u);
    }
}

// ---- workspace ---------------------------------------------------------------------------------------------------------
struct Ws {
    // forward, kept for the backward
    float *E1, *enc, *G1, *s1, *s2, *Xhn, *GH, *GO, *Sp, *A1h, *A1o, *C1o, *C1h, *F, *T, *Tos, *Tso, *Tg, *Wt, *adj_raw,
          *U, *V, *adj, *alpha, *beta, *M1, *M2, *Hp, *h_node, *st_h, *Op, *node, *st_o, *B1h, *B1o, *Tp, *b3sum;
    // backward temporaries
    float *dPF, *dTp, *dTg, *dF, *dB1h, *dB1o, *dG1, *dh_node, *dnode, *dHp, *dHm, *dOp, *dOm, *dU, *dV, *dTos, *dTso,
          *da, *dadj, *dWt, *dT, *dA1h, *dA1o, *dC1o, *dC1h, *dS, *ds2, *ds1, *dXhn, *d_enc, *dE1, *db3;
    // SIDE twins (bf16 step only; only the twin-workspace half of these regions is used): bf16 copies of the operands the
    // CALLER owns (pooled box features, spatial codes, pooled features['3'], d logits) and PADDED copies of the three weights
    // whose rows are not 16-byte multiples (fc_head / fc_tail [1024, 1074] -> pitch 1088, spatial_head.0 [128, 46] -> 48),
    // written by one launch in forward part 0 (d logits: backward stage 0): with them EVERY product of the step has both
    // twins and runs on the direct-to-LDS kernel (round 4: 7 of 21 launches stayed on the register-staged loop)
    float *x0t, *sp48t, *gft, *fhw_t, *ftw_t, *sp0w_t, *dlog_t;
    bool side;                   // the regions above exist (bf16 step, not switched off)
    float* scratch;
    float* scratch2;             // the second branch's split-K scratch (set by the entry points: scratch + the sizing pass's need)
    int64_t total;               // floats, scratch excluded
};
constexpr int FH_PITCH = 1088, SP0_PITCH = 48;

static bool side_twins_enabled() {                      // developer A/B switch
    static int on = -1;
    if (on < 0) { const char* e = getenv("SKG_SIDE_TWINS"); on = (e && atoi(e) == 0) ? 0 : 1; }
    return on != 0;
}

static void layout_ws(const skg_train_plan* P, float* base, Ws& w) {
    int64_t off = 0;
    auto take = [&](int64_t rows, int64_t cols) -> float* {
        int64_t n = rows * cols; if (n < 8) n = 8;
        n = (n + 7) & ~(int64_t)7;                          // 32-byte steps: the bf16 twin of every buffer is 16-byte aligned
        float* p = base ? base + off : nullptr;
        off += n;
        return p;
    };
    const int64_t NA = P->NA, Mg = P->Mg, Mp = P->Mp > 0 ? P->Mp : 1, Mh = P->Mh, Mn = P->Mn, Bf = P->Bf;
    w.E1 = take(NA, 1024); w.enc = take(NA, 1024); w.G1 = take(Bf, 1024); w.s1 = take(Mg, 128); w.s2 = take(Mg, 256);
    w.Xhn = take(Mh + Mn, 1088); w.GH = take(Mh, 1024); w.GO = take(Mn, 1024); w.Sp = take(Mg, 1024);
    w.A1h = take(Mh, 1024); w.A1o = take(Mn, 1024); w.C1o = take(Mn, 1024); w.C1h = take(Mh, 1024);
    w.F = take(Mg, 4096); w.T = take(Mg, 1024); w.Tos = take(Mg, 1024); w.Tso = take(Mg, 1024); w.Tg = take(Mp, 1024);
    w.Wt = take(Mg, 1024); w.adj_raw = take(Mg, 1); w.U = take(Mh, 1024); w.V = take(Mn, 1024);
    w.adj = take(Mg, 1); w.alpha = take(Mg, 1); w.beta = take(Mg, 1);
    w.M1 = take(Mh, 1024); w.M2 = take(Mn, 1024); w.Hp = take(Mh, 1024); w.h_node = take(Mh, 1024); w.st_h = take(Mh, 2);
    w.Op = take(Mn, 1024); w.node = take(Mn, 1024); w.st_o = take(Mn, 2); w.B1h = take(Mh, 1024); w.B1o = take(Mn, 1024);
    w.Tp = take(Mp, 1024); w.b3sum = take(4, 1024);
    w.dPF = take(Mp, 2048); w.dTp = take(Mp, 1024); w.dTg = take(Mp, 1024);
    w.dF = take(Mg, 4096); w.dG1 = take(Bf, 1024);      // (adjacent: zero-filled by ONE memset in backward stage 1)
    w.dB1h = take(Mh, 1024); w.dB1o = take(Mn, 1024); w.dh_node = take(Mh, 1024);
    w.dnode = take(Mn, 1024); w.dHp = take(Mh, 1024); w.dHm = take(Mh, 1024); w.dOp = take(Mn, 1024); w.dOm = take(Mn, 1024);
    w.dU = take(Mh, 1024); w.dV = take(Mn, 1024); w.dTos = take(Mg, 1024); w.dTso = take(Mg, 1024); w.da = take(4, Mg);
    w.dadj = take(Mg, 1); w.dWt = take(Mg, 1024); w.dT = take(Mg, 1024); w.dA1h = take(Mh, 1024); w.dA1o = take(Mn, 1024);
    w.dC1o = take(Mn, 1024); w.dC1h = take(Mh, 1024); w.dS = take(Mg, 1024); w.ds2 = take(Mg, 256); w.ds1 = take(Mg, 128);
    w.dXhn = take(Mh + Mn, 1088); w.d_enc = take(NA, 1024); w.dE1 = take(NA, 1024); w.db3 = take(4, 1024);
    w.x0t = w.sp48t = w.gft = w.fhw_t = w.ftw_t = w.sp0w_t = w.dlog_t = nullptr;
    w.side = P->bf16 && side_twins_enabled();
    if (w.side) {
        w.x0t = take(NA, P->x0_k); w.sp48t = take(Mg, SKG_SPATIAL_LD); w.gft = take(Bf, P->Cf);
        w.fhw_t = take(1024, FH_PITCH); w.ftw_t = take(1024, FH_PITCH); w.sp0w_t = take(128, SP0_PITCH);
        w.dlog_t = take(Mp, (P->ld_logits + 7) & ~7);      // (pitch of 8: V-COCO's 25 logit columns are stored 28 apart)
    }
    w.total = off;
    w.scratch = base ? base + off : nullptr;
    w.scratch2 = nullptr;
}

static uint16_t* twin_of(const Ctx& c, const float* p, int64_t* ld) {
    const skg_train_plan* P = c.P;
    *ld = 0;
    if (!P->ws16 || !p) return nullptr;
    if (p >= P->ws && p < P->ws + P->ws_floats) return (c.no_twin && p == c.no_twin) ? nullptr : P->ws16 + (p - P->ws);
    const int64_t npf = (int64_t)(P->Mp > 0 ? P->Mp : 1) * 2048;
    if (P->pf16 && p >= P->pair_features && p < P->pair_features + npf) return P->pf16 + (p - P->pair_features);
    const Ws* w = c.w;
    if (w && w->side && w->x0t) {                          // side twins: caller-owned operands, padded weight copies
        auto side = [&](const float* region) { return P->ws16 + (region - P->ws); };
        const float* fh = P->params + P->seg_off[SKG_SEG_FH_W];
        const float* ft = P->params + P->seg_off[SKG_SEG_FT_W];
        const float* s0 = P->params + P->seg_off[SKG_SEG_SP0_W];
        if (p == fh) { *ld = FH_PITCH; return side(w->fhw_t); }
        if (p == ft) { *ld = FH_PITCH; return side(w->ftw_t); }
        if (p == s0) { *ld = SP0_PITCH; return side(w->sp0w_t); }
        if (p >= P->x0 && p < P->x0 + (int64_t)P->NA * P->x0_k) return side(w->x0t) + (p - P->x0);
        if (p >= P->sp48 && p < P->sp48 + (int64_t)P->Mg * SKG_SPATIAL_LD) return side(w->sp48t) + (p - P->sp48);
        if (p >= P->gfeat && p < P->gfeat + (int64_t)P->Bf * P->Cf) return side(w->gft) + (p - P->gfeat);
        if (P->dlogits && P->Mp > 0 && p == P->dlogits) { *ld = (P->ld_logits + 7) & ~7; return side(w->dlog_t); }
    }
    if (P->params16 && p >= P->params && p < P->params + P->params_floats) return P->params16 + (p - P->params);
    return nullptr;
}

// dst[r, c] = bf16(src[r, c]) for c < cols, 0 up to the twin's pitch: up to 8 segments in one launch (two elements per thread)
struct TwinSeg { const float* src; uint16_t* dst; int64_t ld_src, ld_dst; int rows, cols; int64_t first; };
struct TwinSegs { TwinSeg s[8]; int n; int64_t total; };
// (work items of FOUR elements: a 16-byte load where the source row allows it -- pitch and base multiples of 4 floats, the
//  quad inside the row --, scalar loads otherwise; the twin's pitch is a multiple of 4 by construction)
__global__ __launch_bounds__(256) void twin_segments_kernel(const TwinSegs g) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < g.total; i += (int64_t)gridDim.x * 256) {
        int k = 0;
#pragma unroll
        for (int t = 1; t < 8; ++t)
            if (t < g.n && i >= g.s[t].first) k = t;
        const TwinSeg& sg = g.s[k];
        const int64_t local = i - sg.first, qpr = sg.ld_dst >> 2;
        const int64_t row = local / qpr;
        const int col = (int)(local - row * qpr) * 4;
        const float* sp = sg.src + row * sg.ld_src + col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col + 3 < sg.cols && (sg.ld_src & 3) == 0 && skg_aligned16_dev(sg.src)) v = *reinterpret_cast<const float4*>(sp);
        else {
            if (col < sg.cols) v.x = sp[0];
            if (col + 1 < sg.cols) v.y = sp[1];
            if (col + 2 < sg.cols) v.z = sp[2];
            if (col + 3 < sg.cols) v.w = sp[3];
        }
        skg_store_twin4(sg.dst + row * sg.ld_dst + col, v);
    }
}
static void twin_segments(Ctx& c, TwinSeg* segs, int n) {
    if (c.rc || c.dry || n == 0) return;
    TwinSegs g; g.n = 0; g.total = 0;
    for (int i = 0; i < n; ++i) {
        if (segs[i].rows <= 0 || !segs[i].src || !segs[i].dst) continue;
        segs[i].first = g.total;
        g.total += (int64_t)segs[i].rows * (segs[i].ld_dst >> 2);
        g.s[g.n++] = segs[i];
    }
    if (!g.n) return;
    int64_t blocks = (g.total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(twin_segments_kernel, dim3((unsigned)blocks), dim3(256), 0, c.stream, g);
}

// ---- parameter / gradient segments -------------------------------------------------------------------------------------
struct Seg {
    const skg_train_plan* P; float* base;
    float* at(int s) const { return base + P->seg_off[s]; }
};

static int check_plan(const skg_train_plan* P) {
    if (!P || !P->params || !P->meta) return SKG_E_ARG;
    if (P->NA <= 0 || P->Mg <= 0 || P->Mh <= 0 || P->Mn <= 0 || P->A <= 0 || P->K <= 0 || P->Bf <= 0 || P->Cf <= 0 ||
        P->x0_k <= 0 || P->Mp < 0)
        return SKG_E_ARG;
    if (P->ld_logits < P->K + 1 || (P->ld_logits & 3)) return SKG_E_ARG;
    return 0;
}

// ---- forward -----------------------------------------------------------------------------------------------------------
// part 0: what needs neither the TransH tables nor the label counts (parameter-side sums, box_head, the global branch's
//         fc_1, two spatial layers) -- the step driver enqueues it BEFORE its host synchronisation;  part 1: the rest.
static void forward(Ctx& c, const Ws& w, int part) {
    const skg_train_plan* P = c.P;
    Seg W{P, const_cast<float*>(P->params)};
    const int NA = P->NA, Mg = P->Mg, Mp = P->Mp, Mh = P->Mh, Mn = P->Mn, A = P->A, K = P->K, Bf = P->Bf, Cf = P->Cf;
    const int kx = P->x0_k;
    Mat x0 = cm(P->x0, NA, kx), gfeat = cm(P->gfeat, Bf, Cf), sp48 = cm(P->sp48, Mg, SKG_SPATIAL_LD);
    Mat E1(w.E1, NA, 1024), enc(w.enc, NA, 1024), G1(w.G1, Bf, 1024), s1(w.s1, Mg, 128), s2(w.s2, Mg, 256);
    Mat bh1_w(W.at(SKG_SEG_BH1_W), 1024, kx), bh3_w(W.at(SKG_SEG_BH3_W), 1024, 1024);
    Mat sp0_w(W.at(SKG_SEG_SP0_W), 128, 46), sp2_w(W.at(SKG_SEG_SP2_W), 256, 128), sp4_w(W.at(SKG_SEG_SP4_W), 1024, 256);
    Mat W1[4] = {Mat(W.at(SKG_SEG_W1_0), 1024, 2048), Mat(W.at(SKG_SEG_W1_1), 1024, 1024),
                 Mat(W.at(SKG_SEG_W1_2), 1024, 1024), Mat(W.at(SKG_SEG_W1_3), 1024, Cf)};
    float* b1[4] = {W.at(SKG_SEG_B1_0), W.at(SKG_SEG_B1_1), W.at(SKG_SEG_B1_2), W.at(SKG_SEG_B1_3)};
    Mat W3[4] = {Mat(W.at(SKG_SEG_W3_0), 1024, 1024), Mat(W.at(SKG_SEG_W3_1), 1024, 1024),
                 Mat(W.at(SKG_SEG_W3_2), 1024, 1024), Mat(W.at(SKG_SEG_W3_3), 1024, 1024)};
    enum { ATT = 0, OS = 1, SO = 2, GL = 3 };
    if (part == 0) {
        if (!c.dry)
            hipLaunchKernelGGL(b3sum_kernel, dim3(16), dim3(256), 0, c.stream, W.at(SKG_SEG_B3), w.b3sum);
        // bf16 step: the twin of the whole parameter arena, once per step (whatever changed the parameters since the last
        // step -- the optimizer, load_state_dict, a caller writing p.data -- the products read what the arena holds NOW)
        if (P->bf16 && P->params16 && P->params_floats > 0)
            CK(skg_twin_bf16(P->params, P->params16, P->params_floats & ~(int64_t)3, c.stream));
        if (w.side && P->ws16 && !c.dry) {
            auto side = [&](const float* region) { return P->ws16 + (region - P->ws); };
            TwinSeg sg[6] = {{P->x0, side(w.x0t), kx, kx, NA, kx, 0}, {P->sp48, side(w.sp48t), SKG_SPATIAL_LD, SKG_SPATIAL_LD, Mg, SKG_SPATIAL_LD, 0},
                             {P->gfeat, side(w.gft), Cf, Cf, Bf, Cf, 0},
                             {W.at(SKG_SEG_FH_W), side(w.fhw_t), 1074, FH_PITCH, 1024, 1074, 0},
                             {W.at(SKG_SEG_FT_W), side(w.ftw_t), 1074, FH_PITCH, 1024, 1074, 0},
                             {W.at(SKG_SEG_SP0_W), side(w.sp0w_t), 46, SP0_PITCH, 128, 46, 0}};
            twin_segments(c, sg, 6);
        }
        // ---- box_head (HEAD:812), fc_1 of the global branch (HEAD:971), the first two spatial layers (HEAD:888)
        skg_gemmx_desc l1[2] = {FWD(x0, bh1_w, E1, W.at(SKG_SEG_BH1_B), true),
                                FWD(sp48, sp0_w, s1, W.at(SKG_SEG_SP0_B), true, -1, 46)};
        launch(c, l1, 2);
        skg_gemmx_desc l2[3] = {FWD(E1, bh3_w, enc, W.at(SKG_SEG_BH3_B), true), FWD(gfeat, W1[GL], G1, b1[GL], false),
                                FWD(s1, sp2_w, s2, W.at(SKG_SEG_SP2_B), true)};
        launch(c, l2, 3);
        return;
    }
    const float* b3 = w.b3sum;
    const int Mp1 = Mp > 0 ? Mp : 1;
    // ---- fc_head / fc_tail on unique node rows (HEAD:884-885)
    Mat Xhn(w.Xhn, Mh + Mn, 1088), GH(w.GH, Mh, 1024), GO(w.GO, Mn, 1024), Sp(w.Sp, Mg, 1024);
    Mat fh_w(W.at(SKG_SEG_FH_W), 1024, 1074), ft_w(W.at(SKG_SEG_FT_W), 1024, 1074);
    Mat A1h(w.A1h, Mh, 1024), A1o(w.A1o, Mn, 1024), C1o(w.C1o, Mn, 1024), C1h(w.C1h, Mh, 1024);
    Mat Wa1 = W1[ATT];
    Mat F(w.F, Mg, 4096), T(w.T, Mg, 1024), Tos(w.Tos, Mg, 1024), Tso(w.Tso, Mg, 1024), Tg(w.Tg, Mp1, 1024);
    const float* W2 = W.at(SKG_SEG_W2); const float* b2 = W.at(SKG_SEG_B2);
    // The NODE chain (entity rows -> fc_head / fc_tail -> the four fc_1 projections: a few hundred rows, launches bound by
    // their latency) shares nothing with the SPATIAL chain (last spatial layer -> fc_2 of all four MBFs on the grid rows:
    // the step's largest forward product) until the fc_1 * fc_2 products.  With a second branch (bf16 step) the node chain
    // runs there, beside the spatial chain on the caller's stream, and joins in front of the products.
    auto node_chain = [&]() {
        CK(skg_concat_entity_f32(w.enc, 1024, P->enc_row_hn, P->ent, P->img_hn, P->ent_row_hn, Mh + Mn, w.Xhn, 1088, c.stream));
        skg_gemmx_desc l[2] = {FWD(Xhn.row_range(0, Mh), fh_w, GH, W.at(SKG_SEG_FH_B), true, -1, 1074),
                               FWD(Xhn.row_range(Mh, Mn), ft_w, GO, W.at(SKG_SEG_FT_B), true, -1, 1074)};
        launch(c, l, 2);
        // fc_1 projections on node rows (HEAD:894-896 separable over [human | object]; HEAD:514, 524)
        skg_gemmx_desc l4[4] = {FWD(GH, Wa1, A1h, nullptr, false, -1, 1024), FWD(GO, Wa1.from_col(1024), A1o, nullptr, false, -1, 1024),
                                FWD(GO, W1[OS], C1o, b1[OS], false), FWD(GH, W1[SO], C1h, b1[SO], false)};
        launch(c, l4, 4);
    };
    const bool two = c.br != nullptr && P->bf16 && !c.dry;
    if (two) {
        CK((int)hipEventRecord(c.br->fork, c.stream));
        CK((int)hipStreamWaitEvent(c.br->stream, c.br->fork, 0));
        {
            OnBranch on(c, w.scratch2);
            node_chain();
            CK((int)hipEventRecord(c.br->join, c.stream));
        }
        skg_gemmx_desc l[1] = {FWD(s2, sp4_w, Sp, W.at(SKG_SEG_SP4_B), true)};           // the last spatial layer
        launch(c, l, 1);
    } else {
        // (one stream; the sizing pass walks this form: the same products, the scratch high-water mark is per launch)
        CK(skg_concat_entity_f32(w.enc, 1024, P->enc_row_hn, P->ent, P->img_hn, P->ent_row_hn, Mh + Mn, w.Xhn, 1088, c.stream));
        skg_gemmx_desc l[3] = {FWD(Xhn.row_range(0, Mh), fh_w, GH, W.at(SKG_SEG_FH_B), true, -1, 1074),
                               FWD(Xhn.row_range(Mh, Mn), ft_w, GO, W.at(SKG_SEG_FT_B), true, -1, 1074),
                               FWD(s2, sp4_w, Sp, W.at(SKG_SEG_SP4_B), true)};
        launch(c, l, 3);
        skg_gemmx_desc l4[4] = {FWD(GH, Wa1, A1h, nullptr, false, -1, 1024), FWD(GO, Wa1.from_col(1024), A1o, nullptr, false, -1, 1024),
                                FWD(GO, W1[OS], C1o, b1[OS], false), FWD(GH, W1[SO], C1h, b1[SO], false)};
        launch(c, l4, 4);
    }
    // ---- fc_2 on the grid rows; the raw fc_2 output F = [F2 | F_os | F_so | F_g] (ld 4096) is kept for the backward
    if (P->bf16) {
        skg_gemmx_desc l[1] = {FWD(Sp, cm(W2, 4096, 1024), F, b2, false)};            // all four fc_2 as ONE N = 4096 product
        launch(c, l, 1);
        if (two) CK((int)hipStreamWaitEvent(c.stream, c.br->join, 0));                 // the node chain's fc_1 tables
        {   // the four fc_1 * fc_2 -> ReLU products as one launch
            skg_rows_mul_args m[4] = {
                {w.A1h, P->grid_h, 1024, w.A1o, P->grid_o, 1024, b1[ATT], w.F, nullptr, 4096, Mg, 1024, w.T, 1024},
                {w.C1o, P->grid_o, 1024, nullptr, nullptr, 1024, nullptr, w.F + 1024, nullptr, 4096, Mg, 1024, w.Tos, 1024},
                {w.C1h, P->grid_h, 1024, nullptr, nullptr, 1024, nullptr, w.F + 2048, nullptr, 4096, Mg, 1024, w.Tso, 1024},
                {w.G1, P->pair_img, 1024, nullptr, nullptr, 1024, nullptr, w.F + 3072, P->pair_grid, 4096, Mp, 1024, w.Tg, 1024}};
            CK(skg_rows_mul_relu_multi(m, Mp > 0 ? 4 : 3, c.stream));
        }
    } else {
        // exact fp32: the eval kernel with the fc_1 * fc_2 -> ReLU product fused in its epilogue, raw output stored too
        skg_gemm_desc ds[4];
        for (int i = 0; i < 4; ++i) {
            skg_gemm_desc& d = ds[i]; memset(&d, 0, sizeof(d));
            d.A = w.Sp; d.lda = 1024; d.W = W2 + (int64_t)1024 * 1024 * i; d.ldw = 1024; d.bias = b2 + 1024 * i;
            d.ldc = 1024; d.M = Mg; d.N = 1024; d.K = 1024; d.epilogue = SKG_EPI_MUL_RELU;
            d.C_raw = w.F + 1024 * i; d.ldc_raw = 4096;
            if (i == ATT) { d.C = w.T; d.P = w.A1h; d.p_idx = P->grid_h; d.ldp = 1024; d.Q = w.A1o; d.q_idx = P->grid_o;
                            d.ldq = 1024; d.mbias = b1[ATT]; }
            else if (i == OS) { d.C = w.Tos; d.P = w.C1o; d.p_idx = P->grid_o; d.ldp = 1024; }
            else if (i == SO) { d.C = w.Tso; d.P = w.C1h; d.p_idx = P->grid_h; d.ldp = 1024; }
            else { d.C = w.Tg; d.P = w.G1; d.p_idx = P->grid_img; d.ldp = 1024; d.out_rows = P->grid_pair; }
            c.flops += 2.0 * Mg * 1024.0 * 1024.0;
        }
        // up to ~6000 grid rows the four products together are a mid-size group (csrc/skg_gemm.hip, g_route_tiles): ONE launch on
        // the free-layout GEMM fills the CUs that each 200-tile product alone leaves idle; larger steps keep one launch each
        if ((int64_t)((Mg + 127) / 128) * 8 * 4 < 1536) {
            CK(skg_gemm_group_f32(ds, 4, c.stream));
        } else {
            for (int i = 0; i < 4; ++i) CK(skg_gemm_f32(&ds[i], c.stream));
        }
    }
    // ---- attention fc_3 + ReLU, adjacency logits (HEAD:896-897)
    Mat Wt(w.Wt, Mg, 1024);
    {
        skg_gemmx_desc l[1] = {FWD(T, W3[ATT], Wt, b3 + 1024 * ATT, true, -1, -1, -1, true)};
        launch(c, l, 1);
    }
    CK(skg_rowdot_f32(w.Wt, 1024, W.at(SKG_SEG_ADJ_W), Mg, 1024, w.adj_raw, c.stream));
    // ---- softmax-weighted aggregation before the linear fc_3 (HEAD:907-922); the adjacency bias cancels in the softmax
    CK(skg_graph_aggregate_train_f32(w.adj_raw, 1, Mg, 0.0f, P->meta, A, P->hum_img, P->node_img, Mh, Mn, w.Tos, w.Tso,
                                     1024, 1024, w.U, w.V, 1024, w.adj, w.alpha, w.beta, c.stream));
    // ---- message fc_3 + ReLU, residual, LayerNorm (HEAD:909-914, 916-925)
    Mat U(w.U, Mh, 1024), V(w.V, Mn, 1024), M1(w.M1, Mh, 1024), M2(w.M2, Mn, 1024);
    {
        skg_gemmx_desc l[2] = {FWD(U, W3[OS], M1, b3 + 1024 * OS, true, -1, -1, -1, true),
                               FWD(V, W3[SO], M2, b3 + 1024 * SO, true, -1, -1, -1, true)};
        launch(c, l, 2);
    }
    {
        skg_add_layernorm_args ln[2] = {
            {w.GH, 1024, w.M1, 1024, W.at(SKG_SEG_NH_W), W.at(SKG_SEG_NH_B), Mh, w.Hp, w.h_node, w.st_h},
            {w.GO, 1024, w.M2, 1024, W.at(SKG_SEG_NO_W), W.at(SKG_SEG_NO_B), Mn, w.Op, w.node, w.st_o}};
        CK(skg_add_layernorm_multi(ln, 2, 1e-5f, c.stream));
    }
    // ---- read-out on the kept pairs (HEAD:966-973)
    Mat h_node(w.h_node, Mh, 1024), node(w.node, Mn, 1024), B1h(w.B1h, Mh, 1024), B1o(w.B1o, Mn, 1024);
    {
        skg_gemmx_desc l[2] = {FWD(h_node, Wa1, B1h, nullptr, false, -1, 1024),
                               FWD(node, Wa1.from_col(1024), B1o, nullptr, false, -1, 1024)};
        launch(c, l, 2);
    }
    if (Mp > 0)
        CK(skg_rows_mul_relu_f32(w.B1h, P->pair_h, 1024, w.B1o, P->pair_o, 1024, b1[ATT], w.F, P->pair_grid, 4096, Mp,
                                 1024, w.Tp, 1024, c.stream));
    Mat Tp(w.Tp, Mp1, 1024), PF(P->pair_features, Mp1, 2048);
    {
        skg_gemmx_desc l[2] = {FWD(Tp, W3[ATT], PF, b3 + 1024 * ATT, true, Mp, -1, 1024, true),
                               FWD(Tg, W3[GL], PF.from_col(1024), b3 + 1024 * GL, true, Mp, -1, 1024, true)};
        launch(c, l, 2);
    }
    // ---- classifier: predictor | suppressor as one product (HEAD:410-411); the caller zero-filled `logits`
    {
        Mat logits(P->logits, Mp1, P->ld_logits);
        skg_gemmx_desc l[1] = {FWD(PF, Mat(W.at(SKG_SEG_CLS_W), K + 1, 2048), logits, W.at(SKG_SEG_CLS_B), false, Mp, -1, K + 1)};
        launch(c, l, 1);
    }
}

// ---- backward ----------------------------------------------------------------------------------------------------------
// stages [first, last): 0 classifier + read-out ... 11 box_head; the gradient arena prefix that is final after stage s
// is skghoi_amd/train_fused.py's Stacked.milestone_end (the data-parallel exchange runs behind it).
static void backward(Ctx& c, const Ws& w, int first, int last) {
    const skg_train_plan* P = c.P;
    Seg W{P, const_cast<float*>(P->params)}, G{P, P->grads};
    const int NA = P->NA, Mg = P->Mg, Mp = P->Mp, Mh = P->Mh, Mn = P->Mn, A = P->A, K = P->K, Bf = P->Bf, Cf = P->Cf;
    const int kx = P->x0_k, Mp1 = Mp > 0 ? Mp : 1;
    enum { ATT = 0, OS = 1, SO = 2, GL = 3 };
    Mat W1[4] = {Mat(W.at(SKG_SEG_W1_0), 1024, 2048), Mat(W.at(SKG_SEG_W1_1), 1024, 1024),
                 Mat(W.at(SKG_SEG_W1_2), 1024, 1024), Mat(W.at(SKG_SEG_W1_3), 1024, Cf)};
    Mat dW1[4] = {Mat(G.at(SKG_SEG_W1_0), 1024, 2048), Mat(G.at(SKG_SEG_W1_1), 1024, 1024),
                  Mat(G.at(SKG_SEG_W1_2), 1024, 1024), Mat(G.at(SKG_SEG_W1_3), 1024, Cf)};
    float* b1[4] = {W.at(SKG_SEG_B1_0), W.at(SKG_SEG_B1_1), W.at(SKG_SEG_B1_2), W.at(SKG_SEG_B1_3)};
    float* db1[4] = {G.at(SKG_SEG_B1_0), G.at(SKG_SEG_B1_1), G.at(SKG_SEG_B1_2), G.at(SKG_SEG_B1_3)};
    Mat W3[4] = {Mat(W.at(SKG_SEG_W3_0), 1024, 1024), Mat(W.at(SKG_SEG_W3_1), 1024, 1024),
                 Mat(W.at(SKG_SEG_W3_2), 1024, 1024), Mat(W.at(SKG_SEG_W3_3), 1024, 1024)};
    Mat dW3[4] = {Mat(G.at(SKG_SEG_W3_0), 1024, 1024), Mat(G.at(SKG_SEG_W3_1), 1024, 1024),
                  Mat(G.at(SKG_SEG_W3_2), 1024, 1024), Mat(G.at(SKG_SEG_W3_3), 1024, 1024)};
    float* db3 = w.db3;
    Mat dlogits = cm(P->dlogits, Mp1, P->ld_logits);
    Mat PF(P->pair_features, Mp1, 2048), Tp(w.Tp, Mp1, 1024), Tg(w.Tg, Mp1, 1024);
    Mat dPF(w.dPF, Mp1, 2048), dTp(w.dTp, Mp1, 1024), dTg(w.dTg, Mp1, 1024);
    Mat Wa1 = W1[ATT], dWa1 = dW1[ATT];
    Mat dB1h(w.dB1h, Mh, 1024), dB1o(w.dB1o, Mn, 1024), dG1(w.dG1, Bf, 1024);
    Mat dh_node(w.dh_node, Mh, 1024), dnode(w.dnode, Mn, 1024), h_node(w.h_node, Mh, 1024), node(w.node, Mn, 1024);
    Mat dHp(w.dHp, Mh, 1024), dHm(w.dHm, Mh, 1024), dOp(w.dOp, Mn, 1024), dOm(w.dOm, Mn, 1024);
    Mat dU(w.dU, Mh, 1024), dV(w.dV, Mn, 1024), U(w.U, Mh, 1024), V(w.V, Mn, 1024);
    Mat dWt(w.dWt, Mg, 1024), dT(w.dT, Mg, 1024), T(w.T, Mg, 1024), Wt(w.Wt, Mg, 1024), dadj(w.dadj, Mg, 1);
    Mat dA1h(w.dA1h, Mh, 1024), dA1o(w.dA1o, Mn, 1024), dC1o(w.dC1o, Mn, 1024), dC1h(w.dC1h, Mh, 1024);
    Mat GH(w.GH, Mh, 1024), GO(w.GO, Mn, 1024), Sp(w.Sp, Mg, 1024), dS(w.dS, Mg, 1024), dF(w.dF, Mg, 4096);
    Mat s1(w.s1, Mg, 128), s2(w.s2, Mg, 256), ds1(w.ds1, Mg, 128), ds2(w.ds2, Mg, 256), sp48 = cm(P->sp48, Mg, SKG_SPATIAL_LD);
    Mat sp0_w(W.at(SKG_SEG_SP0_W), 128, 46), sp2_w(W.at(SKG_SEG_SP2_W), 256, 128), sp4_w(W.at(SKG_SEG_SP4_W), 1024, 256);
    Mat g_sp0(G.at(SKG_SEG_SP0_W), 128, 46), g_sp2(G.at(SKG_SEG_SP2_W), 256, 128), g_sp4(G.at(SKG_SEG_SP4_W), 1024, 256);
    Mat Xhn(w.Xhn, Mh + Mn, 1088), dXhn(w.dXhn, Mh + Mn, 1088);
    Mat fh_w(W.at(SKG_SEG_FH_W), 1024, 1074), ft_w(W.at(SKG_SEG_FT_W), 1024, 1074);
    Mat g_fh(G.at(SKG_SEG_FH_W), 1024, 1074), g_ft(G.at(SKG_SEG_FT_W), 1024, 1074);
    Mat gfeat = cm(P->gfeat, Bf, Cf), x0 = cm(P->x0, NA, kx);
    Mat E1(w.E1, NA, 1024), dE1(w.dE1, NA, 1024), d_enc(w.d_enc, NA, 1024);
    Mat bh1_w(W.at(SKG_SEG_BH1_W), 1024, kx), bh3_w(W.at(SKG_SEG_BH3_W), 1024, 1024);
    Mat g_bh1(G.at(SKG_SEG_BH1_W), 1024, kx), g_bh3(G.at(SKG_SEG_BH3_W), 1024, 1024);
    // Behind stage 5 the backward is two chains that share nothing: the SPATIAL chain on the grid rows (fc_2 of all four MBFs
    // -- the step's largest product --, then the spatial head) and the NODE chain on a few hundred rows (the fc_1 projections,
    // fc_head / fc_tail, box_head: launches bound by their latency).  With a second branch (bf16 step) the node chain runs
    // there from stage 6 on, beside the spatial chain on the caller's stream; the call joins them before it returns.
    // A staged backward (data parallel: one call per run of stages up to the next arena chunk) forks in the call that holds
    // stage 6 -- the chunk that stage completes is all spatial-chain and earlier output, its event needs no join -- and joins
    // at the end of the first call that ran node-chain stages, in front of that call's chunk event.
    const bool two = c.br != nullptr && P->bf16 && !c.dry;
    bool no_branch = false;
    bool& forked = two ? c.br->forked : no_branch;
    auto on_node = [&](auto&& fn) {
        if (two && forked) { OnBranch on(c, w.scratch2); fn(); }
        else fn();
    };
    for (int st = first; st < last && !c.rc; ++st) switch (st) {
    case 0: {
        // ---- classifier
        if (w.side && P->ws16 && Mp > 0 && !c.dry) {
            TwinSeg sg[1] = {{P->dlogits, P->ws16 + (w.dlog_t - P->ws), P->ld_logits, (P->ld_logits + 7) & ~7, Mp, P->ld_logits, 0}};
            twin_segments(c, sg, 1);
        }
        Mat clsW(W.at(SKG_SEG_CLS_W), K + 1, 2048), g_clsW(G.at(SKG_SEG_CLS_W), K + 1, 2048);
        skg_gemmx_desc l[2] = {IG(dlogits, clsW, dPF, &PF, false, Mp, 2048, K + 1),
                               WG(dlogits, PF, g_clsW, G.at(SKG_SEG_CLS_B), false, Mp, K + 1, 2048)};
        launch(c, l, 2);
    } break;
    case 1: {
        // ---- read-out fc_3 (both branches): dT = dPF W3 cut by the product's ReLU; dW3 = dPF^T T
        skg_gemmx_desc l[4] = {IG(dPF, W3[ATT], dTp, &Tp, false, Mp, 1024, 1024, true),
                               IG(dPF.from_col(1024), W3[GL], dTg, &Tg, false, Mp, 1024, 1024, true),
                               WG(dPF, Tp, dW3[ATT], db3 + 1024 * ATT, false, Mp, 1024, 1024, true),
                               WG(dPF.from_col(1024), Tg, dW3[GL], db3 + 1024 * GL, false, Mp, 1024, 1024, true)};
        launch(c, l, 4);
        // ---- read-out fc_1 * fc_2 products: dF at the pairs' grid rows (self-pair rows stay zero), dm in place
        if (!c.dry) {
            uint16_t* dF16 = (P->bf16 && P->ws16) ? P->ws16 + (w.dF - P->ws) : nullptr;
            if (Mp > 0) {
                hipLaunchKernelGGL(zero_selfpair_rows_kernel, dim3(Mg), dim3(256), 0, c.stream, w.dF, dF16, P->grid_pair, Mg);
                if (A < Bf) {          // images without a graph: their dG1 rows are not written by the segment sum below
                    hipError_t e = hipMemsetAsync(w.dG1, 0, sizeof(float) * (size_t)Bf * 1024, c.stream);
                    if (e == hipSuccess && dF16)
                        e = hipMemsetAsync(P->ws16 + (w.dG1 - P->ws), 0, sizeof(uint16_t) * (size_t)Bf * 1024, c.stream);
                    if (e != hipSuccess) { c.rc = (int)e; break; }
                }
            } else {
                const size_t nz = (size_t)((w.dG1 - w.dF) + (int64_t)Bf * 1024);
                hipError_t e = hipMemsetAsync(w.dF, 0, sizeof(float) * nz, c.stream);
                if (e == hipSuccess && dF16) e = hipMemsetAsync(dF16, 0, sizeof(uint16_t) * nz, c.stream);
                if (e != hipSuccess) { c.rc = (int)e; break; }
            }
        }
        if (Mp > 0) {
            skg_mul_bwd_args m[2] = {       // (disjoint column blocks of dF: one launch)
                {w.dTp, 1024, w.F, P->pair_grid, 4096, w.B1h, P->pair_h, 1024, w.B1o, P->pair_o, 1024, b1[ATT], Mp, w.dF, 4096, 0},
                {w.dTg, 1024, w.F + 3072, P->pair_grid, 4096, w.G1, P->pair_img, 1024, nullptr, nullptr, 0, nullptr, Mp,
                 w.dF + 3072, 4096, 0}};
            CK(skg_mul_bwd_multi(m, 2, c.stream));
        }
        CK(skg_segment_sum_f32(w.dTp, 1024, P->meta, A, P->hum_img, P->node_img, Mh, Mn, 1, w.dB1h, w.dB1o, 0, c.stream));
        CK(skg_segment_sum_f32(w.dTg, 1024, P->meta, A, nullptr, nullptr, 0, 0, 2, w.dG1, nullptr, 0, c.stream));
        // ---- read-out fc_1 on the normalised nodes: dh_node, dnode; dW1[att] from both halves
        skg_gemmx_desc l2[4] = {IG(dB1h, Wa1, dh_node, nullptr, false, -1, 1024), IG(dB1o, Wa1.from_col(1024), dnode, nullptr, false, -1, 1024),
                                WG(dB1h, h_node, dWa1, nullptr, false, -1, -1, 1024),
                                WG(dB1o, node, dWa1.from_col(1024), nullptr, false, -1, -1, 1024)};
        launch(c, l2, 4);
    } break;
    case 2: {
        // ---- LayerNorm + residual: dHp continues to the node, dHp cut by the message's ReLU goes to fc_3
        skg_layernorm_bwd_args ln[2] = {
            {w.dh_node, 1024, w.Hp, w.st_h, W.at(SKG_SEG_NH_W), Mh, w.dHp, w.M1, w.dHm, G.at(SKG_SEG_NH_W), G.at(SKG_SEG_NH_B)},
            {w.dnode, 1024, w.Op, w.st_o, W.at(SKG_SEG_NO_W), Mn, w.dOp, w.M2, w.dOm, G.at(SKG_SEG_NO_W), G.at(SKG_SEG_NO_B)}};
        CK(skg_layernorm_bwd_multi(ln, 2, c.stream));
    } break;
    case 3: {
        // ---- message fc_3
        skg_gemmx_desc l[4] = {IG(dHm, W3[OS], dU, nullptr, false, -1, 1024, -1, true), IG(dOm, W3[SO], dV, nullptr, false, -1, 1024, -1, true),
                               WG(dHm, U, dW3[OS], db3 + 1024 * OS, false, -1, -1, -1, true),
                               WG(dOm, V, dW3[SO], db3 + 1024 * SO, false, -1, -1, -1, true)};
        launch(c, l, 4);
    } break;
    case 4: {
        // ---- aggregation + softmax, adjacency Linear(1024 -> 1) over relu(fc_3(T)), attention fc_3
        float* da = w.da;
        CK(skg_aggregate_bwd_f32(w.dU, w.dV, w.Tos, w.Tso, w.alpha, w.beta, P->grid_h, P->grid_o, Mg, P->meta, P->hum_img,
                                 P->node_img, Mh, Mn, w.dTos, w.dTso, da, da + Mg, da + 2 * (int64_t)Mg, da + 3 * (int64_t)Mg,
                                 c.stream));
        CK(skg_adjacency_bwd_f32(da + 2 * (int64_t)Mg, da + 3 * (int64_t)Mg, W.at(SKG_SEG_ADJ_W), w.Wt, Mg, w.dadj, w.dWt,
                                 c.stream));
        skg_gemmx_desc l[3] = {IG(dWt, W3[ATT], dT, &T, false, -1, 1024, -1, true),
                               WG(dWt, T, dW3[ATT], db3 + 1024 * ATT, true, -1, -1, -1, true),
                               WG(dadj, Wt, Mat(G.at(SKG_SEG_ADJ_W), 1, 1024), G.at(SKG_SEG_ADJ_B), false)};
        if (P->bf16 && w.side) {
            // the one-row product on two small kernels of its own (partials in the split-K scratch, free between launches)
            const int chunks = (Mg + ADJW_ROWS - 1) / ADJW_ROWS;
            const int64_t need = (int64_t)chunks * 1028;
            if (need > c.scratch_need) c.scratch_need = need;
            c.flops += 2.0 * Mg * 1024.0;
            if (!c.dry && !c.rc) {
                hipLaunchKernelGGL(adjw_partial_kernel, dim3(4 * chunks), dim3(256), 0, c.stream, w.dadj, w.Wt, Mg, c.scratch);
                hipLaunchKernelGGL(adjw_finish_kernel, dim3(17), dim3(1024), 0, c.stream, c.scratch, chunks,
                                   G.at(SKG_SEG_ADJ_W), G.at(SKG_SEG_ADJ_B));
            }
            launch(c, l, 2);
        } else {
            launch(c, l, 3);
        }
    } break;
    case 5: {
        // ---- in-loop fc_1 * fc_2 products
        {   // three products into disjoint column blocks of dF, then their three neighbourhood sums: one launch each
            skg_mul_bwd_args m[3] = {
                {w.dT, 1024, w.F, nullptr, 4096, w.A1h, P->grid_h, 1024, w.A1o, P->grid_o, 1024, b1[ATT], Mg, w.dF, 4096, 1},
                {w.dTos, 1024, w.F + 1024, nullptr, 4096, w.C1o, P->grid_o, 1024, nullptr, nullptr, 0, nullptr, Mg, w.dF + 1024,
                 4096, 0},
                {w.dTso, 1024, w.F + 2048, nullptr, 4096, w.C1h, P->grid_h, 1024, nullptr, nullptr, 0, nullptr, Mg, w.dF + 2048,
                 4096, 0}};
            CK(skg_mul_bwd_multi(m, 3, c.stream));
            skg_segment_sum_args g[3] = {{w.dT, 1024, 0, w.dA1h, w.dA1o, 0}, {w.dTos, 1024, 0, nullptr, w.dC1o, 0},
                                         {w.dTso, 1024, 0, w.dC1h, nullptr, 0}};
            CK(skg_segment_sum_multi(g, 3, P->meta, P->hum_img, P->node_img, Mh, Mn, c.stream));
        }
        // the multiplier bias of attention_head's fc_1 is added once per row: its gradient is the sum over all rows
        if (!c.dry)
            hipLaunchKernelGGL(colsum2_kernel, dim3(16), dim3(1024), 0, c.stream, w.dA1h, Mh, w.dB1h, Mh, db1[ATT]);
    } break;
    case 6: {
        if (two && !forked) {                              // the node chain's stream behind everything issued so far
            CK((int)hipEventRecord(c.br->fork, c.stream));
            CK((int)hipStreamWaitEvent(c.br->stream, c.br->fork, 0));
            forked = !c.rc;
        }
        // ---- fc_2 of all four MBFs: ONE product for the input gradient (K = 4096), one for the weights
        Mat W2(W.at(SKG_SEG_W2), 4096, 1024), g_W2(G.at(SKG_SEG_W2), 4096, 1024);
        skg_gemmx_desc l[2] = {IG(dF, W2, dS, &Sp, false, -1, 1024), WG(dF, Sp, g_W2, G.at(SKG_SEG_B2), false)};
        launch(c, l, 2);
    } break;
    case 7: {
        // ---- fc_1 projections on node rows (gradients accumulate on top of the residual path) + spatial layer 3
        skg_gemmx_desc l[6] = {IG(dA1h, Wa1, dHp, nullptr, true, -1, 1024), IG(dA1o, Wa1.from_col(1024), dOp, nullptr, true, -1, 1024),
                               WG(dA1h, GH, dWa1, nullptr, true, -1, -1, 1024),
                               WG(dA1o, GO, dWa1.from_col(1024), nullptr, true, -1, -1, 1024),
                               IG(dS, sp4_w, ds2, &s2, false), WG(dS, s2, g_sp4, G.at(SKG_SEG_SP4_B), false)};
        if (two && forked) { on_node([&] { launch(c, l, 4); }); launch(c, l + 4, 2); }
        else launch(c, l, 6);
    } break;
    case 8: {
        skg_gemmx_desc l[6] = {IG(dC1h, W1[SO], dHp, &GH, true), IG(dC1o, W1[OS], dOp, &GO, true),
                               WG(dC1h, GH, dW1[SO], db1[SO], false), WG(dC1o, GO, dW1[OS], db1[OS], false),
                               IG(ds2, sp2_w, ds1, &s1, false), WG(ds2, s1, g_sp2, G.at(SKG_SEG_SP2_B), false)};
        if (two && forked) { on_node([&] { launch(c, l, 4); }); launch(c, l + 4, 2); }
        else launch(c, l, 6);
    } break;
    case 9: {
        // ---- fc_head / fc_tail, the first spatial layer and the global branch's fc_1 (HEAD:971)
        skg_gemmx_desc l[7] = {IG(dHp, fh_w, dXhn.row_range(0, Mh), nullptr, false, -1, 1074),
                               IG(dOp, ft_w, dXhn.row_range(Mh, Mn), nullptr, false, -1, 1074),
                               WG(dHp, Xhn.row_range(0, Mh), g_fh, G.at(SKG_SEG_FH_B), false, -1, -1, 1074),
                               WG(dOp, Xhn.row_range(Mh, Mn), g_ft, G.at(SKG_SEG_FT_B), false, -1, -1, 1074),
                               WG(ds1, sp48, g_sp0, G.at(SKG_SEG_SP0_B), false, -1, -1, 46),
                               WG(dG1, gfeat, dW1[GL], db1[GL], false), skg_gemmx_desc()};
        int n = 6;
        if (P->dgfeat || c.dry) l[n++] = IG(dG1, W1[GL], Mat(P->dgfeat, Bf, Cf), nullptr, false);   // (sizing: assume it)
        if (two && forked) {
            on_node([&] {
                launch(c, l, 4);
                CK(skg_entity_rows_bwd_f32(w.dXhn, 1088, P->hum_of, P->node_of, Mh, NA, w.enc, w.d_enc, c.stream));
            });
            launch(c, l + 4, n - 4);
        } else {
            launch(c, l, n);
            CK(skg_entity_rows_bwd_f32(w.dXhn, 1088, P->hum_of, P->node_of, Mh, NA, w.enc, w.d_enc, c.stream));
        }
    } break;
    case 10: {
        // ---- box_head layer 2
        skg_gemmx_desc l[2] = {IG(d_enc, bh3_w, dE1, &E1, false), WG(d_enc, E1, g_bh3, G.at(SKG_SEG_BH3_B), false)};
        on_node([&] { launch(c, l, 2); });
    } break;
    case 11: {
        // ---- box_head layer 1; every fc_3 branch gets its MBF's bias gradient (the bias is added once per row)
        skg_gemmx_desc l[2] = {WG(dE1, x0, g_bh1, G.at(SKG_SEG_BH1_B), false), skg_gemmx_desc()};
        int n = 1;
        if (P->dx0 || c.dry) l[n++] = IG(dE1, bh1_w, Mat(P->dx0, NA, kx), nullptr, false);
        on_node([&] { launch(c, l, n); });
        if (!c.dry)
            hipLaunchKernelGGL(b3bcast_kernel, dim3(256), dim3(256), 0, c.stream, w.db3, G.at(SKG_SEG_B3));
    } break;
    default: break;
    }
    if (forked && last >= 8) {                             // node-chain stages ran: the caller's stream behind the branch
        hipError_t e = hipEventRecord(c.br->join, c.br->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c.stream, c.br->join, 0);
        if (e != hipSuccess && !c.rc) c.rc = (int)e;
        forked = false;
    }
}

}  // namespace

extern "C" {

int64_t skg_train_ws_floats(const skg_train_plan* P) {
    int rc = check_plan(P);
    if (rc) return rc;
    Ws w; layout_ws(P, nullptr, w);
    Ctx c{P, nullptr, true, 0, nullptr, 0, 0.0, nullptr, &w, nullptr};
    forward(c, w, 0); forward(c, w, 1); backward(c, w, 0, SKG_TRAIN_BWD_STAGES);
    // (twice the split-K scratch: a two-branch call gives each branch a region of its own)
    return w.total + (P->two_branch ? 2 : 1) * c.scratch_need + 4;
}

}  // extern "C"

namespace {
// a branch is used when the plan asks for one, the caller's context supplied it and the workspace holds a second scratch region
static skg_branch* usable_branch(const skg_train_plan* P, skg_branch* br, Ws& w, int64_t scratch_need) {
    if (!br || !P->two_branch || !P->bf16 || P->counters) return nullptr;      // (tile counters are per stream: one branch only)
    if (w.total + 2 * scratch_need > P->ws_floats) return nullptr;
    w.scratch2 = w.scratch + scratch_need;
    return br;
}

int forward_entry(const skg_train_plan* P, int part, void* stream, skg_branch* br) {
    int rc = check_plan(P);
    if (rc) return rc;
    if (!P->ws || !P->x0 || !P->gfeat || !P->sp48 || part < 0 || part > 1) return SKG_E_ARG;
    if (part == 1 && (!P->ent || !P->logits || !P->pair_features)) return SKG_E_ARG;
    Ws w; layout_ws(P, P->ws, w);
    if (w.total > P->ws_floats) return SKG_E_LIMIT;
    // bound the scratch: the sizing pass told the caller how much the largest launch needs
    Ctx d{P, nullptr, true, 0, nullptr, 0, 0.0, nullptr, &w, nullptr};
    forward(d, w, part);
    if (w.total + d.scratch_need > P->ws_floats) return SKG_E_LIMIT;
    Ctx c{P, (hipStream_t)stream, false, 0, w.scratch, 0, 0.0, w.dadj, &w, usable_branch(P, br, w, d.scratch_need)};
    TwinScope twins(P);
    forward(c, w, part);
    if (!c.rc) { hipError_t e = hipGetLastError(); if (e != hipSuccess) c.rc = (int)e; }
    return c.rc;
}
}  // namespace

extern "C" {

int skg_train_forward_f32(const skg_train_plan* P, int part, void* stream) { return forward_entry(P, part, stream, nullptr); }

// what both the direct and the worker-thread entry check BEFORE anything is enqueued or queued: arguments, stage range and
// the workspace bound (a sizing pass over the requested stages)
static int validate_backward(const skg_train_plan* P, int first_stage, int last_stage) {
    int rc = check_plan(P);
    if (rc) return rc;
    if (!P->ws || !P->grads || !P->dlogits || !P->pair_features || first_stage < 0 || last_stage > SKG_TRAIN_BWD_STAGES ||
        first_stage > last_stage)
        return SKG_E_ARG;
    Ws w; layout_ws(P, P->ws, w);
    Ctx d{P, nullptr, true, 0, nullptr, 0, 0.0, nullptr, &w, nullptr};
    backward(d, w, first_stage, last_stage);
    if (w.total + d.scratch_need > P->ws_floats) return SKG_E_LIMIT;
    return 0;
}

static int backward_entry(const skg_train_plan* P, int first_stage, int last_stage, void* stream, skg_branch* br) {
    int rc = validate_backward(P, first_stage, last_stage);
    if (rc) return rc;
    Ws w; layout_ws(P, P->ws, w);
    int64_t need = 0;
    if (br && P->two_branch) {                             // (over ALL stages: the calls of a staged backward agree on scratch2)
        Ctx d{P, nullptr, true, 0, nullptr, 0, 0.0, nullptr, &w, nullptr};
        backward(d, w, 0, SKG_TRAIN_BWD_STAGES);
        need = d.scratch_need;
    }
    Ctx c{P, (hipStream_t)stream, false, 0, w.scratch, 0, 0.0, w.dadj, &w, usable_branch(P, br, w, need)};
    TwinScope twins(P);
    backward(c, w, first_stage, last_stage);
    if (!c.rc) { hipError_t e = hipGetLastError(); if (e != hipSuccess) c.rc = (int)e; }
    return c.rc;
}

int skg_train_backward_f32(const skg_train_plan* P, int first_stage, int last_stage, void* stream) {
    return backward_entry(P, first_stage, last_stage, stream, nullptr);
}

}  // extern "C"

/* ---- the backward enqueued from a worker thread.  Issuing the ~60 launches of a backward costs the calling thread
 * ~0.2 ms; a Python step loop that is bound by its own thread (the batch-4 step: 1.45 ms of host work per step, measured
 * with time.thread_time) hands the call to this thread and keeps preparing the next batch meanwhile.  The worker, its one
 * job slot and the job's progress belong to a CONTEXT (skg_context: one per trainer / device / host thread, created by the
 * caller); the context-free entry points use a default context of the process.  One job at a time per context; the plan
 * is copied; the worker selects the submitting thread's device.  With stage events the worker records events[s - first]
 * on the stream behind stage s and publishes the stage count: a data-parallel caller waits for "stage s issued"
 * (skg_ctx_train_backward_stage_wait) and orders its gradient collective behind that event, so the exchange runs chunk
 * by chunk behind a backward that is issued in ONE call.                                                                 */
struct skg_context {
    std::mutex m;
    std::condition_variable cv;
    std::thread worker;
    bool started = false, pending = false, quit = false;
    skg_train_plan plan;
    int first = 0, last = 0, device = 0, rc = 0;
    int issued = 0;                      // stages [first, issued) of the current / last job are on the stream
    void* stream = nullptr;
    hipEvent_t events[SKG_TRAIN_BWD_STAGES];       // caller's events (measurement), by stage - first
    hipEvent_t own_dev[SKG_TRAIN_BWD_STAGES];      // the context's own stage events: no timing, DEVICE-scope release
    hipEvent_t own_sys[SKG_TRAIN_BWD_STAGES];      // ... the same with the default SYSTEM-scope release: what orders a collective
                                                   // of a world > 1 behind a stage (a peer GPU, or a registered-buffer transport,
                                                   // may read the chunk directly: the stage's writes must have left this GPU's L2)
    hipEvent_t* own = own_dev;                     // the set the current job records
    uint32_t own_mask = 0;                         // stages (absolute) behind which own[stage] is recorded
    bool own_made = false, own_sys_made = false;
    bool with_events = false;
    skg_tuning tuning = {0, 0, 0, 0};              // the eval GEMM's switches for the threads this context is current on
    skg_branch branch = {nullptr, nullptr, nullptr, false};   // second branch of two-branch plan calls (stream + fork / join events)
    bool branch_made = false;
    skg_branch* get_branch() {                // (the submitting / calling thread's device is current)
        if (branch_made) return &branch;
        hipError_t e = hipStreamCreateWithFlags(&branch.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&branch.fork, hipEventDisableTiming | hipEventReleaseToDevice);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&branch.join, hipEventDisableTiming | hipEventReleaseToDevice);
        if (e != hipSuccess) return nullptr;
        branch_made = true;
        return &branch;
    }
    skg_exchange ex = {};                          // the arena chunks this job all-reduces (ex.comm) and / or updates (ex.adamw) itself
    bool has_ex = false;
    hipStream_t aux = nullptr;                     // single process + optimizer inside the backward: the updates' stream
    hipEvent_t aux_done = nullptr;
    // behind the stage that completes chunk i (its own event was recorded just before): the chunk's collective, then the
    // optimizer's update of the chunk's parameters -- both on the exchange stream, concurrent with the stages still to run
    int exchange_after(int e, hipStream_t stream) {
        if (!has_ex) return 0;
        for (int i = 0; i < ex.n_chunks; ++i) {
            if (ex.stage[i] != e) continue;
            int fi = 0;                                // the first chunk with table entries also bumps the step counters
            while (ex.adamw && fi < ex.n_chunks - 1 && ex.adamw_first[fi + 1] <= ex.adamw_first[fi]) ++fi;
            if (!ex.comm && i == ex.n_chunks - 1)      // no collective to wait for: the last update follows its stage in-stream
                return skg_adamw_slice(ex.adamw, ex.adamw_first[i], ex.adamw_first[i + 1], ex, i == fi, stream);
            // the step's last chunk has nothing left to hide behind: its collective (and its update) go on the step's own stream
            const bool in_stream = ex.comm && i == ex.n_chunks - 1;
            hipStream_t xs = in_stream ? stream : (ex.comm ? skg_comm_stream(ex.comm) : aux);
            int r = 0;
            if (in_stream)
                r = skg_comm_chunk_in_stream(ex.comm, stream, ex.arena + (i ? ex.end[i - 1] : 0), ex.end[i] - (i ? ex.end[i - 1] : 0));
            else if (ex.comm)
                r = skg_comm_chunk(ex.comm, own[e], ex.arena + (i ? ex.end[i - 1] : 0), ex.end[i] - (i ? ex.end[i - 1] : 0));
            else
                r = (int)hipStreamWaitEvent(xs, own[e], 0);
            if (!r && ex.adamw) r = skg_adamw_slice(ex.adamw, ex.adamw_first[i], ex.adamw_first[i + 1], ex, i == fi, xs);
            return r;
        }
        return 0;
    }
    // `stream` behind everything the exchange stream holds for this job
    int exchange_close(hipStream_t stream) {
        if (!has_ex) return 0;
        if (ex.comm) return 0;                       // (the last chunk ran on `stream` itself, behind the exchange stream's tail)
        hipError_t err = hipEventRecord(aux_done, aux);
        if (err == hipSuccess) err = hipStreamWaitEvent(stream, aux_done, 0);
        return (int)err;
    }
    void loop() {
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return pending || quit; });
            if (quit && !pending) return;
            lk.unlock();
            int r = (int)hipSetDevice(device);
            (void)skg_ctx_make_current(this);              // the eval GEMM's switches of THIS context (fp32 plan: the fc_2 products)
            if (!with_events) {
                if (!r) r = backward_entry(&plan, first, last, stream, plan.two_branch ? get_branch() : nullptr);
                lk.lock();
                issued = last;
            } else {
                // runs of stages up to the next stage that somebody waits for: one plan call per run, one event behind it
                int s = first;
                while (s < last && !r) {
                    int e = s;
                    while (e + 1 < last && !events[e - first] && !((own_mask >> e) & 1u)) ++e;
                    r = backward_entry(&plan, s, e + 1, stream, plan.two_branch ? get_branch() : nullptr);
                    if (!r && events[e - first]) r = (int)hipEventRecord(events[e - first], (hipStream_t)stream);
                    if (!r && ((own_mask >> e) & 1u)) r = (int)hipEventRecord(own[e], (hipStream_t)stream);
                    if (!r) r = exchange_after(e, (hipStream_t)stream);
                    if (r) break;
                    lk.lock(); issued = e + 1; lk.unlock();
                    cv.notify_all();
                    s = e + 1;
                }
                if (!r) r = exchange_close((hipStream_t)stream);                 // `stream` behind the last collective / update
                // a failure between two chunks leaves the peers inside a collective this rank will never join, on a
                // communicator nobody watches: abort it, so that they get an error instead of a hang
                if (r && has_ex && ex.comm) (void)skg_comm_abort(ex.comm);
                lk.lock();
            }
            rc = r; pending = false;
            lk.unlock();
            cv.notify_all();
        }
    }
};

namespace {
skg_context* g_default_ctx = nullptr;
void default_ctx_after_fork_in_child() {                       // threads do not survive fork(): a child starts without a worker
    if (g_default_ctx) g_default_ctx = new skg_context;        // (the parent's object, and whatever its mutex held, is abandoned)
}
skg_context* default_ctx() {
    static bool once = [] {
        g_default_ctx = new skg_context;                       // never destroyed: its thread outlives static destruction
        pthread_atfork(nullptr, nullptr, default_ctx_after_fork_in_child);
        return true;
    }();
    (void)once;
    return g_default_ctx;
}
inline skg_context* ctx_or_default(skg_context* c) { return c ? c : default_ctx(); }
}  // namespace

extern "C" {

skg_context* skg_context_create(void) { return new (std::nothrow) skg_context; }

int skg_ctx_set_tuning(skg_context* c, const skg_tuning* t) {
    if (!c || !t) return SKG_E_ARG;
    if (t->small_mode != 0 && t->small_mode != 1 && (t->small_mode < 3 || t->small_mode > 6)) return SKG_E_ARG;
    std::unique_lock<std::mutex> lk(c->m);
    c->tuning = *t;
    return 0;
}
int skg_ctx_get_tuning(skg_context* c, skg_tuning* out) {
    if (!c || !out) return SKG_E_ARG;
    std::unique_lock<std::mutex> lk(c->m);
    *out = c->tuning;
    return 0;
}
static thread_local skg_context* tls_current_ctx = nullptr;
skg_context* skg_ctx_make_current(skg_context* c) {
    skg_context* old = tls_current_ctx;
    tls_current_ctx = c;
    skg_tls_tuning = c ? &c->tuning : nullptr;
    return old;
}

void skg_context_destroy(skg_context* c) {
    if (!c) return;
    if (tls_current_ctx == c) (void)skg_ctx_make_current(nullptr);
    {
        std::unique_lock<std::mutex> lk(c->m);
        c->cv.wait(lk, [&] { return !c->pending; });           // a job in flight finishes issuing first
        c->quit = true;
    }
    c->cv.notify_all();
    if (c->started && c->worker.joinable()) c->worker.join();
    if (c->own_made)
        for (int s = 0; s < SKG_TRAIN_BWD_STAGES; ++s) (void)hipEventDestroy(c->own_dev[s]);
    if (c->own_sys_made)
        for (int s = 0; s < SKG_TRAIN_BWD_STAGES; ++s) (void)hipEventDestroy(c->own_sys[s]);
    if (c->branch_made) {
        (void)hipStreamSynchronize(c->branch.stream); (void)hipStreamDestroy(c->branch.stream);
        (void)hipEventDestroy(c->branch.fork); (void)hipEventDestroy(c->branch.join);
    }
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->aux_done) (void)hipEventDestroy(c->aux_done);
    delete c;
}

static int submit_backward(skg_context* ctx, const skg_train_plan* P, int first_stage, int last_stage, void* stream,
                           void* const* stage_events_host, uint32_t stage_mask, const skg_exchange* ex) {
    int rc = validate_backward(P, first_stage, last_stage);    // rejected here, at submit -- not at the join
    if (rc) return rc;
    if (ex) {
        if ((!ex->comm && !ex->adamw) || !ex->arena || ex->n_chunks < 1 || ex->n_chunks > SKG_TRAIN_BWD_STAGES) return SKG_E_ARG;
        if (ex->adamw) {
            if (!(ex->bias1 > 0.0) || !(ex->bias2 > 0.0) || !(ex->eps >= 0.0) || ex->adamw_n_steps < 0 ||
                (ex->adamw_n_steps > 0 && !ex->adamw_steps) || ex->adamw_first[0] < 0)
                return SKG_E_ARG;
            for (int i = 0; i < ex->n_chunks; ++i)
                if (ex->adamw_first[i + 1] < ex->adamw_first[i]) return SKG_E_ARG;
        }
        stage_mask = 0;
        for (int i = 0; i < ex->n_chunks; ++i) {
            if (ex->stage[i] < first_stage || ex->stage[i] >= last_stage || (i && ex->stage[i] <= ex->stage[i - 1]) ||
                ex->end[i] < (i ? ex->end[i - 1] : 0))
                return SKG_E_ARG;
            if (i + 1 < ex->n_chunks) stage_mask |= 1u << ex->stage[i];      // (no event behind the last chunk: it stays in-stream)
        }
    }
    skg_context* a = ctx_or_default(ctx);
    std::unique_lock<std::mutex> lk(a->m);
    if (a->pending || a->quit) return SKG_E_LIMIT;             // one job at a time per context: join first
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    // events with SYSTEM-scope release where the chunk behind a stage goes to OTHER GPUs: the library's communicator of a world
    // > 1, or a caller that says so (bit 31 of stage_mask: its process group has more than one rank).  World size 1 and the
    // single-process optimizer chunks keep the device-scope set (measured: a dozen system-scope records cost a backward 0.09 ms).
    const bool sys_scope = (ex && ex->comm && skg_comm_world(ex->comm) > 1) || (!ex && (stage_mask & 0x80000000u));
    stage_mask &= 0x7fffffffu;
    if (stage_mask && sys_scope && !a->own_sys_made) {
        for (int s = 0; s < SKG_TRAIN_BWD_STAGES; ++s) {
            e = hipEventCreateWithFlags(&a->own_sys[s], hipEventDisableTiming);
            if (e != hipSuccess) {
                for (int t = 0; t < s; ++t) (void)hipEventDestroy(a->own_sys[t]);
                return (int)e;
            }
        }
        a->own_sys_made = true;
    }
    if (stage_mask && !sys_scope && !a->own_made) {
        // DEVICE-scope release: the default (system scope) writes the L2 back and invalidates it at every record -- a dozen
        // of those inside a backward cost its kernels their L2-resident operands (measured: backward 0.95-0.97 ms with them, 0.86-0.88 ms with device-scope events)
        for (int s = 0; s < SKG_TRAIN_BWD_STAGES; ++s) {
            e = hipEventCreateWithFlags(&a->own_dev[s], hipEventDisableTiming | hipEventReleaseToDevice);
            if (e != hipSuccess) {
                for (int t = 0; t < s; ++t) (void)hipEventDestroy(a->own_dev[t]);
                return (int)e;
            }
        }
        a->own_made = true;
    }
    a->own_mask = stage_mask;
    a->own = sys_scope ? a->own_sys : a->own_dev;
    a->has_ex = ex != nullptr;
    if (ex) a->ex = *ex; else memset(&a->ex, 0, sizeof(a->ex));
    if (ex && !ex->comm && !a->aux) {
        e = hipStreamCreateWithFlags(&a->aux, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a->aux_done, hipEventDisableTiming);
        if (e != hipSuccess) return (int)e;
    }
    a->branch.forked = false;
    a->plan = *P; a->first = first_stage; a->last = last_stage; a->stream = stream; a->device = dev; a->rc = 0;
    a->issued = first_stage;
    a->with_events = stage_events_host != nullptr || stage_mask != 0 || ex != nullptr;
    for (int s = 0; s < SKG_TRAIN_BWD_STAGES; ++s)
        a->events[s] = (stage_events_host && s < last_stage - first_stage) ? (hipEvent_t)stage_events_host[s] : nullptr;
    a->pending = true;
    if (!a->started) {
        a->started = true;
        a->worker = std::thread(&skg_context::loop, a);
        if (a == g_default_ctx) a->worker.detach();
    }
    lk.unlock();
    a->cv.notify_all();
    return 0;
}

int skg_ctx_train_forward_f32(skg_context* ctx, const skg_train_plan* P, int part, void* stream) {
    skg_context* a = ctx_or_default(ctx);
    skg_branch* br = nullptr;
    if (P && P->two_branch) {
        std::unique_lock<std::mutex> lk(a->m);
        br = a->get_branch();
    }
    return forward_entry(P, part, stream, br);
}

int skg_ctx_train_backward_async_f32(skg_context* ctx, const skg_train_plan* P, int first_stage, int last_stage,
                                     void* stream, void* const* stage_events_host, uint32_t stage_mask) {
    return submit_backward(ctx, P, first_stage, last_stage, stream, stage_events_host, stage_mask, nullptr);
}

int skg_ctx_train_backward_exchange_f32(skg_context* ctx, const skg_train_plan* P, int first_stage, int last_stage,
                                        void* stream, void* const* stage_events_host, const skg_exchange* ex) {
    if (!ex) return SKG_E_ARG;
    return submit_backward(ctx, P, first_stage, last_stage, stream, stage_events_host, 0u, ex);
}

int skg_sizeof_exchange(void) { return (int)sizeof(skg_exchange); }     // (bindings check their mirror of the struct)

int skg_ctx_train_backward_stage_wait(skg_context* ctx, int stage) {
    skg_context* a = ctx_or_default(ctx);
    std::unique_lock<std::mutex> lk(a->m);
    a->cv.wait(lk, [&] { return a->issued > stage || !a->pending; });
    if (a->issued > stage) return 0;
    return a->rc ? a->rc : SKG_E_ARG;                          // the job ended without reaching that stage
}

int skg_ctx_stream_wait_stage(skg_context* ctx, int stage, void* waiting_stream) {
    skg_context* a = ctx_or_default(ctx);
    if (stage < 0 || stage >= SKG_TRAIN_BWD_STAGES) return SKG_E_ARG;
    std::unique_lock<std::mutex> lk(a->m);
    if (!((a->own_mask >> stage) & 1u) || a->issued <= stage) return SKG_E_ARG;
    hipEvent_t ev = a->own[stage];
    lk.unlock();
    return (int)hipStreamWaitEvent((hipStream_t)waiting_stream, ev, 0);
}

/* Where the context's current (or last) job stands, without blocking: stages issued so far | 0x100 while a job is pending.
 * For failure records (a rank stuck in a collective: which stage had gone out). */
int skg_ctx_train_backward_progress(skg_context* ctx) {
    skg_context* a = ctx_or_default(ctx);
    std::unique_lock<std::mutex> lk(a->m);
    return a->issued | (a->pending ? 0x100 : 0);
}

int skg_ctx_train_backward_join(skg_context* ctx) {
    skg_context* a = ctx_or_default(ctx);
    std::unique_lock<std::mutex> lk(a->m);
    a->cv.wait(lk, [&] { return !a->pending; });
    const int rc = a->rc;
    a->rc = 0;
    return rc;
}

int skg_train_backward_async_f32(const skg_train_plan* P, int first_stage, int last_stage, void* stream) {
    return skg_ctx_train_backward_async_f32(nullptr, P, first_stage, last_stage, stream, nullptr, 0u);
}

int skg_train_backward_join(void) { return skg_ctx_train_backward_join(nullptr); }

skg_train_timer* skg_train_timer_create(int capacity) {
    if (capacity <= 0) return nullptr;
    skg_train_timer* t = new (std::nothrow) skg_train_timer;
    if (!t) return nullptr;
    t->cap = capacity;
    t->ev = new hipEvent_t[2 * (size_t)capacity];
    t->flops = new double[capacity];
    for (int i = 0; i < 2 * capacity; ++i)
        if (hipEventCreate(&t->ev[i]) != hipSuccess) { t->cap = i / 2; break; }
    return t;
}

void skg_train_timer_destroy(skg_train_timer* t) {
    if (!t) return;
    for (int i = 0; i < 2 * t->cap; ++i) (void)hipEventDestroy(t->ev[i]);
    delete[] t->ev; delete[] t->flops;
    delete t;
}

int skg_train_timer_read(skg_train_timer* t, double* out3_host) {
    if (!t || !out3_host) return SKG_E_ARG;
    std::lock_guard<std::mutex> g(t->m);
    double ms = 0.0, fl = 0.0;
    for (int i = 0; i < t->n; ++i) {
        hipError_t e = hipEventSynchronize(t->ev[2 * i + 1]);
        float d = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&d, t->ev[2 * i], t->ev[2 * i + 1]);
        if (e != hipSuccess) { t->n = 0; return (int)e; }
        ms += d; fl += t->flops[i];
    }
    out3_host[0] = ms; out3_host[1] = (double)t->n; out3_host[2] = fl;
    t->n = 0;
    return 0;
}

/* 2 M N K summed over every dense product of the step: which = 0 forward (both parts), 1 backward, 2 both */
double skg_train_flops(const skg_train_plan* P, int which) {
    if (check_plan(P)) return -1.0;
    Ws w; layout_ws(P, nullptr, w);
    Ctx c{P, nullptr, true, 0, nullptr, 0, 0.0, nullptr, &w, nullptr};
    if (which == 0 || which == 2) { forward(c, w, 0); forward(c, w, 1); }
    if (which == 1 || which == 2) backward(c, w, 0, SKG_TRAIN_BWD_STAGES);
    return c.flops;
}

/* offset (floats) of a saved activation inside the workspace: 0 = pair-independent debug reads (tests) */
int64_t skg_train_ws_offset(const skg_train_plan* P, int which) {
    if (check_plan(P)) return -1;
    Ws w; layout_ws(P, reinterpret_cast<float*>(16), w);       // non-null dummy base: offsets = pointer differences
    float* base = reinterpret_cast<float*>(16);
    switch (which) {
    case 0: return w.enc - base;
    case 1: return w.h_node - base;
    case 2: return w.node - base;
    case 3: return w.adj - base;
    case 4: return w.F - base;
    default: return -1;
    }
}

}  // extern "C"
