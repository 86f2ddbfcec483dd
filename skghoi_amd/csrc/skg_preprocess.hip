// skg_preprocess.hip -- InteractionHead.preprocess on the device, one workgroup per image.
//
// Reference: heads/adamixer_transH_spatial_r50_head.py:92-151 (score filter, torchvision batched_nms, argsort,
// top-k humans/objects, humans first) and the published torchvision algorithm for batched_nms / nms
// (coordinate trick + greedy suppression at IoU > thr; restated in oracle/tv_boxes.py).
//
// Integer/compare work with fp32 IoU: compiled with -ffp-contract=off so every product and sum rounds exactly like
// the unfused CPU arithmetic -- the selected indices are bit-exact against the oracle.
#include "skg_common.h"

#define PRE_THREADS 256

__device__ __forceinline__ uint32_t skg_orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // monotone: a < b  <=>  ord(a) < ord(b)
}

// the label of a candidate from its staged int32 copy; a label that did not fit is re-read (it multiplies the NMS offset
// exactly as the reference's int64 would)
__device__ __forceinline__ int64_t labels64_of(int staged, const int64_t* __restrict__ labels, int64_t at) {
    return staged != -2147483647 - 1 ? (int64_t)staged : labels[at];
}

__global__ __launch_bounds__(PRE_THREADS) void skg_preprocess_kernel(
    const float* __restrict__ boxes, const float* __restrict__ scores, const int64_t* __restrict__ labels,
    const int32_t* __restrict__ det_off, int human_idx, float score_thresh, float nms_thresh, int max_human,
    int max_object, const int32_t* __restrict__ nverbs, int num_obj_classes, float prior_pow,
    int32_t* __restrict__ out_index, int32_t* __restrict__ out_count) {
    __shared__ unsigned long long skey[SKG_MAX_DET_PER_IMAGE];
    __shared__ float4 sbox[SKG_MAX_DET_PER_IMAGE];      // boxes shifted by label * (max_coord + 1), sorted order
    __shared__ float sarea[SKG_MAX_DET_PER_IMAGE];
    __shared__ unsigned char ssup[SKG_MAX_DET_PER_IMAGE];
    __shared__ unsigned char shum[SKG_MAX_DET_PER_IMAGE];
    __shared__ float sred[PRE_THREADS / 64];
    __shared__ int ssel[SKG_MAX_NODES];
    __shared__ int sact;
    // the image's raw candidates, read ONCE: boxes, scores, labels and the verb counts of the classes all leave in one round
    // trip (independent loads), every later phase -- sorted-order boxes, the counts at the end -- reads these copies.  The
    // kernel is a chain of dependent phases on one workgroup: each trip to global memory it does not make is ~2 us of a
    // single-image forward (it made six; now two: the offsets, then everything else).
    __shared__ float4 rbox[SKG_MAX_DET_PER_IMAGE];
    __shared__ float rscore[SKG_MAX_DET_PER_IMAGE];
    __shared__ int rlab[SKG_MAX_DET_PER_IMAGE];
    __shared__ int snv[256];                           // nverbs of classes 0..255 (more classes: read from global)

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int base = det_off[b];
    const int n0 = det_off[b + 1] - base;
    const int ld_out = max_human + max_object;

    // More candidates than the LDS arrays hold (or a corrupt offset table): report it instead of writing past them.
    // Uniform per workgroup and ahead of every barrier.  The host side turns count[0] == -1 into an error.
    if (n0 < 0 || n0 > SKG_MAX_DET_PER_IMAGE) {
        for (int t = tid; t < ld_out; t += PRE_THREADS) out_index[(int64_t)b * ld_out + t] = -1;
        if (tid == 0) {
            out_count[4 * b + 0] = -1; out_count[4 * b + 1] = -1; out_count[4 * b + 2] = -1; out_count[4 * b + 3] = n0;
        }
        return;
    }

    int npow = 1;
    while (npow < n0) npow <<= 1;

    // ---- keys: descending score, ties by ascending input index; inactive (score < thresh or NaN) sort last
    float lmax = -INFINITY;
    int lact = 0;
    for (int t = tid; t < 256 && t < num_obj_classes; t += PRE_THREADS) snv[t] = nverbs[t];
    for (int i = tid; i < npow; i += PRE_THREADS) {
        unsigned long long key = ~0ull;
        if (i < n0) {
            const float s = scores[base + i];
            const float4 bx = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)(base + i));
            const int64_t lab64 = labels[base + i];
            rscore[i] = s; rbox[i] = bx;
            // (labels outside int32 cannot name a class: they keep "no verbs" and never equal human_idx below)
            rlab[i] = (lab64 >= -2147483647LL && lab64 <= 2147483647LL) ? (int)lab64 : -2147483647 - 1;
            if (s >= score_thresh) {
                key = ((unsigned long long)(~skg_orderable(s)) << 32) | (unsigned)i;
                lmax = fmaxf(lmax, fmaxf(fmaxf(bx.x, bx.y), fmaxf(bx.z, bx.w)));
                ++lact;
            }
        }
        skey[i] = key;
    }
    // max coordinate over the active boxes (boxes[active].max(), torchvision batched_nms)
    lmax = skg_wave_max(lmax);
    if ((tid & 63) == 0) sred[tid >> 6] = lmax;
    __syncthreads();
    const float max_coord = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
    // number of active candidates
    if (tid == 0) sact = 0;
    __syncthreads();
    if (lact) atomicAdd(&sact, lact);
    __syncthreads();
    const int nact = sact;

    // ---- bitonic sort of the keys (ascending)
    for (int k = 2; k <= npow; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow; i += PRE_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = skey[i], c = skey[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { skey[i] = c; skey[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }

    // ---- shifted boxes in sorted order
    const float shift_unit = max_coord + 1.0f;
    for (int i = tid; i < nact; i += PRE_THREADS) {
        const int idx = (int)(skey[i] & 0xffffffffu);
        const float4 bx = rbox[idx];
        const int64_t lab = labels64_of(rlab[idx], labels, base + idx);
        const float off = (float)lab * shift_unit;
        const float4 sb = make_float4(bx.x + off, bx.y + off, bx.z + off, bx.w + off);
        sbox[i] = sb;
        sarea[i] = (sb.z - sb.x) * (sb.w - sb.y);
        ssup[i] = 0;
        shum[i] = (lab == (int64_t)human_idx) ? 1 : 0;
    }
    __syncthreads();

    // ---- greedy NMS over the sorted candidates; selection of the first max_human humans / max_object others.
    // Up to 256 candidates (the usual case: a detector hands over ~100 boxes per image) ONE wavefront runs the greedy
    // loop on its own: every lane keeps up to four candidates' boxes and suppression bits in registers, the kept box is
    // an LDS broadcast read, "is candidate i suppressed" a lane read -- no barrier per candidate (the four-wave loop
    // below pays one per candidate: 40 candidates took 12 us).  Same comparisons in the same order either way.
    __shared__ int scnt[2];
    int nh = 0, no = 0;
    if (nact <= 256) {
        if (tid < 64) {
            float4 mb[4]; float ma[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = tid + 64 * u;
                mb[u] = j < nact ? sbox[j] : make_float4(0.f, 0.f, 0.f, 0.f);
                ma[u] = j < nact ? sarea[j] : 0.f;
            }
            int supmask = 0;
            for (int i = 0; i < nact; ++i) {
                if (nh >= max_human && no >= max_object) break;                  // uniform
                const bool sup = (__shfl(supmask, i & 63, 64) >> (i >> 6)) & 1;  // uniform
                if (sup) continue;
                const bool hum = shum[i] != 0;
                if (hum) { if (nh < max_human) { if (tid == 0) ssel[nh] = i; ++nh; } }
                else     { if (no < max_object) { if (tid == 0) ssel[max_human + no] = i; ++no; } }
                const float4 bi = sbox[i];
                const float ai = sarea[i];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = tid + 64 * u;
                    if (j <= i || j >= nact || ((supmask >> u) & 1)) continue;
                    const float4 bj = mb[u];
                    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
                    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
                    const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
                    const float inter = w * h;
                    const float ovr = inter / (ai + ma[u] - inter);
                    if (ovr > nms_thresh) supmask |= 1 << u;
                }
            }
            if (tid == 0) { scnt[0] = nh; scnt[1] = no; }
        }
        __syncthreads();
        nh = scnt[0]; no = scnt[1];
    } else {
        for (int i = 0; i < nact; ++i) {
            if (nh >= max_human && no >= max_object) break;          // uniform
            const bool sup = ssup[i] != 0;                           // uniform (LDS broadcast)
            if (!sup) {
                const bool hum = shum[i] != 0;
                if (hum) { if (nh < max_human) { if (tid == 0) ssel[nh] = i; ++nh; } }
                else     { if (no < max_object) { if (tid == 0) ssel[max_human + no] = i; ++no; } }
                const float4 bi = sbox[i];
                const float ai = sarea[i];
                for (int j = i + 1 + tid; j < nact; j += PRE_THREADS) {
                    if (ssup[j]) continue;
                    const float4 bj = sbox[j];
                    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
                    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
                    const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
                    const float inter = w * h;
                    const float ovr = inter / (ai + sarea[j] - inter);
                    if (ovr > nms_thresh) ssup[j] = 1;
                }
            }
            __syncthreads();
        }
        __syncthreads();
    }

    // ---- outputs: indices (humans first), counts, and L = number of non-zero prior cells (HEAD:315, 721-767):
    //      every human whose prior score is non-zero pairs with every other node, each pair carrying the verbs of the
    //      partner's class:  L = #(humans with s^p != 0) * (sum of verbs over all nodes - verbs of the human class).
    const int n = nh + no;
    if (tid < 2) scnt[tid] = 0;
    __syncthreads();
    int lv = 0, lz = 0;
    for (int t = tid; t < ld_out; t += PRE_THREADS) {
        int v = -1;
        if (t < n) {
            const int s = (t < nh) ? ssel[t] : ssel[max_human + (t - nh)];
            v = (int)(skey[s] & 0xffffffffu);
            const int64_t lab = labels64_of(rlab[v], labels, base + v);
            lv += (lab >= 0 && lab < num_obj_classes) ? (lab < 256 ? snv[lab] : nverbs[lab]) : 0;
            if (t < nh && powf(rscore[v], prior_pow) != 0.f) ++lz;
        }
        out_index[(int64_t)b * ld_out + t] = v;
    }
    if (lv) atomicAdd(&scnt[0], lv);
    if (lz) atomicAdd(&scnt[1], lz);
    __syncthreads();
    if (tid == 0) {
        int L = 0;
        if (nh > 0 && n > 1) {
            const int64_t hl = human_idx;
            const int vh = (hl >= 0 && hl < num_obj_classes) ? nverbs[hl] : 0;
            L = scnt[1] * (scnt[0] - vh);
        }
        out_count[4 * b + 0] = nh;
        out_count[4 * b + 1] = n;
        out_count[4 * b + 2] = L;
        out_count[4 * b + 3] = nact;
    }
}

extern "C" int skg_preprocess_f32(const float* boxes, const float* scores, const int64_t* labels,
                                  const int32_t* det_off, int B, int human_idx, float score_thresh, float nms_thresh,
                                  int max_human, int max_object, const int32_t* nverbs, int num_obj_classes,
                                  float prior_pow, int32_t* out_index, int32_t* out_count, void* stream) {
    if (B < 0 || !det_off || !out_index || !out_count || !nverbs) return SKG_E_ARG;
    if (B == 0) return 0;
    if (!boxes || !scores || !labels) return SKG_E_ARG;
    if (max_human < 0 || max_object < 0 || max_human + max_object > SKG_MAX_NODES) return SKG_E_LIMIT;
    if (!skg_aligned16(boxes)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_preprocess_kernel, dim3(B), dim3(PRE_THREADS), 0, (hipStream_t)stream, boxes, scores,
                       labels, det_off, human_idx, score_thresh, nms_thresh, max_human, max_object, nverbs,
                       num_obj_classes, prior_pow, out_index, out_count);
    return skg_launch_status();
}

__global__ void skg_pack_detections_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                           const int64_t* __restrict__ labels, const int32_t* __restrict__ det_off,
                                           const int32_t* __restrict__ index, int index_ld,
                                           const int32_t* __restrict__ sel_off, float* __restrict__ out_boxes,
                                           float* __restrict__ out_scores, int64_t* __restrict__ out_labels) {
    const int b = blockIdx.x;
    const int n = sel_off[b + 1] - sel_off[b];
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int src = det_off[b] + index[(int64_t)b * index_ld + t];
        const int dst = sel_off[b] + t;
        *reinterpret_cast<float4*>(out_boxes + 4 * (int64_t)dst) =
            *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)src);
        out_scores[dst] = scores[src];
        out_labels[dst] = labels[src];
    }
}

extern "C" int skg_pack_detections_f32(const float* boxes, const float* scores, const int64_t* labels,
                                       const int32_t* det_off, const int32_t* index, int index_ld,
                                       const int32_t* sel_off, int B, float* out_boxes, float* out_scores,
                                       int64_t* out_labels, void* stream) {
    if (B < 0 || !det_off || !index || !sel_off) return SKG_E_ARG;
    if (B == 0) return 0;
    if (!boxes || !scores || !labels || !out_boxes || !out_scores || !out_labels) return SKG_E_ARG;
    if (!skg_aligned16(boxes) || !skg_aligned16(out_boxes)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_pack_detections_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, boxes, scores, labels,
                       det_off, index, index_ld, sel_off, out_boxes, out_scores, out_labels);
    return skg_launch_status();
}
