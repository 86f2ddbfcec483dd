// skg_post.hip -- prior scores + HOI scoring/compaction, and the TransH hyperplane scores.
//
// Reference: heads/adamixer_transH_spatial_r50_head.py:721-767 (compute_prior_scores: Python list comprehensions over
// pairs x verbs), HEAD:237-337 (postprocess: nonzero(prior[0]) -> sigmoid(logit) * prior_h * prior_o * sigmoid(weight)),
// heads/TransH/TransH.py:56-106.  The object->verbs mapping is a CSR table; cells are emitted in the order
// torch.nonzero yields them (pair-major, verb ascending).
#include "skg_common.h"

__device__ __forceinline__ float skg_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

// Number of scored cells of kept pair `pl` of image `mt` (HEAD:747-760): the verbs valid for the object's class, if
// the human's prior score is non-zero.
__device__ __forceinline__ int skg_pair_cells(const skg_image_meta& mt, int pl, const int64_t* __restrict__ x_keep,
                                              const int64_t* __restrict__ y_keep, const float* __restrict__ scores,
                                              const int64_t* __restrict__ labels, const int32_t* __restrict__ verb_off,
                                              int num_obj_classes, float prior_pow) {
    const int64_t gp = (int64_t)mt.pair_off + pl;
    const int bh = mt.box_off + (int)x_keep[gp];
    const int bo = mt.box_off + (int)y_keep[gp];
    const int64_t lab = labels[bo];
    const int nv = (lab >= 0 && lab < num_obj_classes) ? verb_off[(int)lab + 1] - verb_off[(int)lab] : 0;
    return (powf(scores[bh], prior_pow) != 0.f) ? nv : 0;
}

// Grid: (active image, chunk of 256 kept pairs).  A workgroup first counts the cells of the image's earlier pairs (a few
// loads per pair) to find where its own chunk starts in the packed result -- one workgroup per image walking all chunks
// in turn took 40 us at a single 20 x 20 image -- then scans its chunk and emits.
__global__ __launch_bounds__(256) void skg_postprocess_kernel(
    const float* __restrict__ logits, int64_t ld_logits, int K, const float* __restrict__ boxes,
    const float* __restrict__ scores, const int64_t* __restrict__ labels, const skg_image_meta* __restrict__ meta,
    const int64_t* __restrict__ x_keep,
    const int64_t* __restrict__ y_keep, const int32_t* __restrict__ verb_off, const int32_t* __restrict__ verb_list,
    int num_obj_classes, float prior_pow, int64_t L_total_arg, const int32_t* __restrict__ L_total_dev,
    int64_t* __restrict__ out_index,
    int64_t* __restrict__ out_pred, float* __restrict__ out_scores, float* __restrict__ out_prior,
    float* __restrict__ out_weights, int64_t* __restrict__ out_object, float* __restrict__ out_boxes_h,
    float* __restrict__ out_boxes_o) {
    __shared__ int swave[4];
    __shared__ int spre[4];
    // stride between the two rows of out_prior: from device memory when the launch is replayed from a captured graph
    // (the number of scored cells changes from call to call, the launch arguments cannot)
    const int64_t L_total = L_total_dev ? (int64_t)max(*L_total_dev, 1) : L_total_arg;
    const int a = blockIdx.x;
    const skg_image_meta mt = meta[a];
    const int P = mt.n_h * (mt.n - 1);
    const int p0 = blockIdx.y * 256;
    if (p0 >= P) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- cells of the pairs in front of this chunk
    int before = 0;
    for (int q = tid; q < p0; q += 256)
        before += skg_pair_cells(mt, q, x_keep, y_keep, scores, labels, verb_off, num_obj_classes, prior_pow);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) spre[wv] = before;
    __syncthreads();
    const int sbase = spre[0] + spre[1] + spre[2] + spre[3];
    // ---- this chunk
    const int pl = p0 + tid;
    int cnt = 0, cls = 0;
    float ph = 0.f, po = 0.f, wgt = 0.f;
    int64_t gp = 0;
    if (pl < P) {
        gp = (int64_t)mt.pair_off + pl;
        const int bh = mt.box_off + (int)x_keep[gp];
        const int bo = mt.box_off + (int)y_keep[gp];
        ph = powf(scores[bh], prior_pow);
        po = powf(scores[bo], prior_pow);
        const int64_t lab = labels[bo];
        cls = (int)lab;
        const int nv = (lab >= 0 && lab < num_obj_classes) ? verb_off[cls + 1] - verb_off[cls] : 0;
        cnt = (ph != 0.f) ? nv : 0;
        wgt = skg_sigmoid(logits[gp * ld_logits + K]);
        out_weights[gp] = wgt;
        out_object[gp] = lab;
        *reinterpret_cast<float4*>(out_boxes_h + 4 * gp) = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)bh);
        *reinterpret_cast<float4*>(out_boxes_o + 4 * gp) = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)bo);
    }
    // exclusive scan of cnt over the 256 pairs of this chunk
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) swave[wv] = inc;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < wv; ++k) wbase += swave[k];
    const int64_t o0 = (int64_t)mt.out_off + sbase + wbase + (inc - cnt);
    if (cnt > 0) {
        // four cells per trip: the verb ids, then the four logits, are in flight together (one cell per trip was two
        // dependent round trips to memory per verb: ~1.5 us each, 15 us for a pair row of ten verbs)
        const int v0 = verb_off[cls];
        for (int t = 0; t < cnt; t += 4) {
            int v[4];
            float lg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (t + u < cnt) ? verb_list[v0 + t + u] : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) lg[u] = (t + u < cnt) ? logits[gp * ld_logits + v[u]] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (t + u >= cnt) continue;
                const int64_t o = o0 + t + u;
                out_index[o] = pl;
                out_pred[o] = v[u];
                out_prior[o] = ph;
                out_prior[L_total + o] = po;
                const float s = skg_sigmoid(lg[u]);
                out_scores[o] = s * (ph * po) * wgt;
            }
        }
    }
}

extern "C" int skg_postprocess_f32(const float* logits, int64_t ld_logits, int K, const float* boxes,
                                   const float* scores, const int64_t* labels, const skg_image_meta* meta,
                                   int n_active, const int64_t* x_keep, const int64_t* y_keep,
                                   const int32_t* verb_off, const int32_t* verb_list, int num_obj_classes,
                                   float prior_pow, int64_t L_total, const int32_t* L_total_dev,
                                   int max_pairs_per_image, int64_t* out_index,
                                   int64_t* out_pred, float* out_scores, float* out_prior, float* out_weights,
                                   int64_t* out_object, float* out_boxes_h, float* out_boxes_o, void* stream) {
    if (n_active < 0 || K <= 0 || ld_logits <= K || L_total < 0 || num_obj_classes <= 0) return SKG_E_ARG;
    if (n_active == 0) return 0;
    if (!logits || !boxes || !scores || !labels || !meta || !x_keep || !y_keep || !verb_off || !verb_list ||
        !out_index || !out_pred || !out_scores || !out_prior || !out_weights || !out_object || !out_boxes_h ||
        !out_boxes_o)
        return SKG_E_ARG;
    if (!skg_aligned16(boxes) || !skg_aligned16(out_boxes_h) || !skg_aligned16(out_boxes_o)) return SKG_E_ALIGN;
    // pair chunks per image: SKG_MAX_NODES bounds n_h * (n - 1); chunks past an image's pairs return at once
    const int max_pairs = max_pairs_per_image > 0 ? max_pairs_per_image : (SKG_MAX_NODES / 2) * (SKG_MAX_NODES - 1);
    hipLaunchKernelGGL(skg_postprocess_kernel, dim3(n_active, (max_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, ld_logits, K,
                       boxes, scores, labels, meta, x_keep, y_keep, verb_off, verb_list, num_obj_classes, prior_pow,
                       L_total, L_total_dev, out_index, out_pred, out_scores, out_prior, out_weights, out_object, out_boxes_h,
                       out_boxes_o);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ TransH scores
// heads/TransH/TransH.py:56-106 with the index vectors of HEAD:570-572: head = human_idx for every triple, tail = the
// node position y, relation k.  The score of (pair (x, y), k) therefore depends on (k, y) only: one wavefront per
// (image, relation) computes it for every node and writes it to the pairs that end in that node.
__device__ __forceinline__ float skg_dot50(float a, float b) { return skg_wave_sum(a * b); }

__global__ __launch_bounds__(64) void skg_transh_scores_kernel(const float* __restrict__ ent,
                                                               const float* __restrict__ rel,
                                                               const float* __restrict__ nrm, int K, int human_idx,
                                                               const skg_image_meta* __restrict__ meta,
                                                               float* __restrict__ scores) {
    const int a = blockIdx.x / K, k = blockIdx.x % K;
    const skg_image_meta mt = meta[a];
    const int lane = threadIdx.x;
    const bool in = lane < SKG_TRANSH_DIM;
    const float* E = ent + (int64_t)a * SKG_TRANSH_ENT * SKG_TRANSH_DIM;
    const float wn = in ? nrm[((int64_t)a * K + k) * SKG_TRANSH_DIM + lane] : 0.f;
    const float rr = in ? rel[((int64_t)a * K + k) * SKG_TRANSH_DIM + lane] : 0.f;
    const float eh = in ? E[(int64_t)human_idx * SKG_TRANSH_DIM + lane] : 0.f;
    const float w = wn / fmaxf(sqrtf(skg_dot50(wn, wn)), 1e-12f);            // F.normalize(norm), TransH.py:76
    const float hp = eh - skg_dot50(eh, w) * w;                               // TransH.py:85
    const float hn = hp / fmaxf(sqrtf(skg_dot50(hp, hp)), 1e-12f);            // TransH.py:58
    const float rn = rr / fmaxf(sqrtf(skg_dot50(rr, rr)), 1e-12f);            // TransH.py:59
    const int n = mt.n, n_h = mt.n_h;
    for (int j = 0; j < n; ++j) {
        const float et = in ? E[(int64_t)j * SKG_TRANSH_DIM + lane] : 0.f;
        const float tp = et - skg_dot50(et, w) * w;
        const float tn = tp / fmaxf(sqrtf(skg_dot50(tp, tp)), 1e-12f);        // TransH.py:60
        const float dv = (hn + rn) - tn;                                      // TransH.py:68
        const float sc = sqrtf(skg_dot50(dv, dv));                            // TransH.py:70
        for (int i = lane; i < n_h; i += 64) {
            if (i == j) continue;
            const int64_t p = (int64_t)mt.pair_off + (int64_t)i * (n - 1) + (j < i ? j : j - 1);
            scores[p * K + k] = sc;
        }
    }
}

extern "C" int skg_transh_scores_f32(const float* ent, const float* rel, const float* nrm, int K, int human_idx,
                                     const skg_image_meta* meta, int n_active, float* scores, void* stream) {
    if (n_active < 0 || K <= 0 || human_idx < 0 || human_idx >= SKG_TRANSH_ENT) return SKG_E_ARG;
    if (n_active == 0) return 0;
    if (!ent || !rel || !nrm || !meta || !scores) return SKG_E_ARG;
    hipLaunchKernelGGL(skg_transh_scores_kernel, dim3(n_active * K), dim3(64), 0, (hipStream_t)stream, ent, rel, nrm, K,
                       human_idx, meta, scores);
    return skg_launch_status();
}

extern "C" int skg_abi_version(void) { return SKG_ABI_VERSION; }
extern "C" const char* skg_build_info(void) { return "libskghoi_hip gfx950 " __VERSION__ " " __DATE__; }

// ------------------------------------------------------------------------------------------------ GT association
// GraphHead.associate_with_ground_truth (HEAD:703-719) for every active image in one launch: labels[p, verb] = 1 where
// min(IoU(box_h, gt_h), IoU(box_o, gt_o)) >= thresh for a ground-truth pair with that verb.  IoU as torchvision's
// box_iou (inter / (a1 + a2 - inter), no eps); compiled without fma contraction -> same decisions as the CPU code.
// labels must be zero-filled by the caller; npos[a] = number of non-zero labels of image a (HEAD:936, 166).
__device__ __forceinline__ float skg_iou(const float4 a, const float4 b) {
    const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = w * h;
    return inter / (aa + ab - inter);
}

__global__ __launch_bounds__(256) void skg_associate_kernel(const float* __restrict__ boxes,
                                                            const skg_image_meta* __restrict__ meta,
                                                            const int64_t* __restrict__ x_keep,
                                                            const int64_t* __restrict__ y_keep,
                                                            const float* __restrict__ gt_h,
                                                            const float* __restrict__ gt_o,
                                                            const int64_t* __restrict__ gt_label,
                                                            const int32_t* __restrict__ gt_off, int K, float thresh,
                                                            float* __restrict__ labels, int32_t* __restrict__ npos) {
    __shared__ int sred[4];
    const int a = blockIdx.x;
    const skg_image_meta mt = meta[a];
    const int P = mt.n_h * (mt.n - 1);
    const int g0 = gt_off[a], g1 = gt_off[a + 1];
    for (int pl = threadIdx.x; pl < P; pl += 256) {
        const int64_t p = (int64_t)mt.pair_off + pl;
        const float4 bh = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)(mt.box_off + (int)x_keep[p]));
        const float4 bo = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)(mt.box_off + (int)y_keep[p]));
        for (int g = g0; g < g1; ++g) {
            const float ih = skg_iou(bh, *reinterpret_cast<const float4*>(gt_h + 4 * (int64_t)g));
            const float io = skg_iou(bo, *reinterpret_cast<const float4*>(gt_o + 4 * (int64_t)g));
            if (fminf(ih, io) >= thresh && !(ih != ih) && !(io != io)) {
                const int64_t v = gt_label[g];
                if (v >= 0 && v < K) labels[p * K + v] = 1.f;
            }
        }
    }
    __syncthreads();
    int cnt = 0;
    const int64_t base = (int64_t)mt.pair_off * K, tot = (int64_t)P * K;
    for (int64_t i = threadIdx.x; i < tot; i += 256) cnt += labels[base + i] != 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) npos[a] = sred[0] + sred[1] + sred[2] + sred[3];
}

extern "C" int skg_associate_f32(const float* boxes, const skg_image_meta* meta, int n_active, const int64_t* x_keep,
                                 const int64_t* y_keep, const float* gt_h, const float* gt_o, const int64_t* gt_label,
                                 const int32_t* gt_off, int K, float thresh, float* labels, int32_t* npos,
                                 void* stream) {
    if (n_active < 0 || K <= 0) return SKG_E_ARG;
    if (n_active == 0) return 0;
    if (!boxes || !meta || !x_keep || !y_keep || !gt_h || !gt_o || !gt_label || !gt_off || !labels || !npos)
        return SKG_E_ARG;
    if (!skg_aligned16(boxes) || !skg_aligned16(gt_h) || !skg_aligned16(gt_o)) return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_associate_kernel, dim3(n_active), dim3(256), 0, (hipStream_t)stream, boxes, meta, x_keep,
                       y_keep, gt_h, gt_o, gt_label, gt_off, K, thresh, labels, npos);
    return skg_launch_status();
}
