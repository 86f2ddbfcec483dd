// skg_eval.hip -- the evaluator behind the head's results on the device: 600-way HOI mapping, box-pair association at
// IoU 0.5, and 11-point AP per interaction class.
//
// Reference: utils.py:148-198 (`test`): per image, interactions = object_n_verb_to_interaction[object][verb]
// (hicodet/hicodet.py:139-153), then for every predicted interaction class pocket's BoxPairAssociation(min_iou = 0.5)
// against the ground-truth pairs of that class -- a detection matches the ground-truth pair of largest
// min(IoU_human, IoU_object) if that reaches min_iou, and among the detections matched to one ground-truth pair the
// highest-scoring one is the true positive -- and pocket's DetectionAPMeter(600, algorithm = '11P').  The reference does
// this per detection in Python on the CPU, one image per forward; pocket is a third-party package absent from the image
// (parity is unpinned at that boundary; oracle/eval_oracle.py restates the published semantics loop by loop).
//
//   skg_eval_associate_f32 : one workgroup per image.  Every lane takes detections, scans the image's ground-truth pairs of
//                            its class (a few dozen at most), and bids for its match with a 64-bit key (score, then lower
//                            index) by LDS atomicMax; the winners are the true positives.
//   skg_eval_ap11_f64      : one workgroup per class over the globally sorted detections (class ascending, score
//                            descending, stable): running true-positive count by block scans, eleven running maxima of
//                            the precision, float64 like pocket.
#include "skg_common.h"

#define EV_MAX_GT 2048            // ground-truth pairs per image the association keeps in LDS

__device__ __forceinline__ float ev_iou(const float4 a, const float4 b) {
    const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
    const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f), h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
    const float inter = w * h;
    return inter / (aa + ab - inter);
}

__device__ __forceinline__ uint32_t ev_orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void skg_eval_associate_kernel(
    const float* __restrict__ boxes_h, const float* __restrict__ boxes_o, const int64_t* __restrict__ object,
    const int32_t* __restrict__ pair_off, const int64_t* __restrict__ index, const int64_t* __restrict__ pred,
    const float* __restrict__ scores, const int32_t* __restrict__ cell_off, const int32_t* __restrict__ lut, int n_obj,
    int n_verb, const float* __restrict__ gt_h, const float* __restrict__ gt_o, const int64_t* __restrict__ gt_hoi,
    const int32_t* __restrict__ gt_off, float min_iou, int32_t* __restrict__ hoi_out, float* __restrict__ labels,
    int32_t* __restrict__ status) {
    __shared__ unsigned long long skey[EV_MAX_GT];
    const int a = blockIdx.x, tid = threadIdx.x;
    const int c0 = cell_off[a], c1 = cell_off[a + 1];
    const int g0 = gt_off[a], ng = gt_off[a + 1] - g0;
    if (ng > EV_MAX_GT) {                                    // uniform: reported, nothing written past the LDS array
        if (tid == 0) atomicMax(status, ng);
        for (int c = c0 + tid; c < c1; c += 256) { labels[c] = 0.f; hoi_out[c] = -1; }
        return;
    }
    for (int g = tid; g < ng; g += 256) skey[g] = 0ull;
    __syncthreads();
    const int p0 = pair_off[a];
    for (int c = c0 + tid; c < c1; c += 256) {
        const int64_t p = (int64_t)p0 + index[c];
        const int64_t ob = object[p], vb = pred[c];
        const int hoi = (ob >= 0 && ob < n_obj && vb >= 0 && vb < n_verb) ? lut[ob * n_verb + vb] : -1;
        hoi_out[c] = hoi;
        int match = -1;
        if (hoi >= 0) {
            const float4 dh = *reinterpret_cast<const float4*>(boxes_h + 4 * p);
            const float4 dob = *reinterpret_cast<const float4*>(boxes_o + 4 * p);
            float best = -1.f;
            for (int g = 0; g < ng; ++g) {
                if (gt_hoi[g0 + g] != hoi) continue;
                const float v = fminf(ev_iou(*reinterpret_cast<const float4*>(gt_h + 4 * (int64_t)(g0 + g)), dh),
                                      ev_iou(*reinterpret_cast<const float4*>(gt_o + 4 * (int64_t)(g0 + g)), dob));
                if (v > best) { best = v; match = g; }       // first maximum, NaN never wins
            }
            if (!(best >= min_iou)) match = -1;
        }
        // (score, lower index wins ties) as one ordered 64-bit key; 0 = no bid
        const unsigned long long key =
            ((unsigned long long)ev_orderable(scores[c]) << 32) | (unsigned)(0xffffffffu - (unsigned)(c - c0));
        if (match >= 0) atomicMax(&skey[match], key);
        labels[c] = match >= 0 ? __uint_as_float((unsigned)match + 1u) : 0.f;      // parked: match + 1 as raw bits
    }
    __syncthreads();
    for (int c = c0 + tid; c < c1; c += 256) {
        const unsigned m = __float_as_uint(labels[c]);
        if (m == 0u) continue;
        const unsigned long long key =
            ((unsigned long long)ev_orderable(scores[c]) << 32) | (unsigned)(0xffffffffu - (unsigned)(c - c0));
        labels[c] = (skey[m - 1u] == key) ? 1.f : 0.f;
    }
}

extern "C" int skg_eval_associate_f32(const float* boxes_h, const float* boxes_o, const int64_t* object,
                                      const int32_t* pair_off, const int64_t* index, const int64_t* pred,
                                      const float* scores, const int32_t* cell_off, int n_images, const int32_t* lut,
                                      int n_obj, int n_verb, const float* gt_h, const float* gt_o, const int64_t* gt_hoi,
                                      const int32_t* gt_off, float min_iou, int32_t* hoi_out, float* labels,
                                      int32_t* status, void* stream) {
    if (n_images < 0 || n_obj <= 0 || n_verb <= 0) return SKG_E_ARG;
    if (n_images == 0) return 0;
    if (!boxes_h || !boxes_o || !object || !pair_off || !index || !pred || !scores || !cell_off || !lut || !gt_h || !gt_o ||
        !gt_hoi || !gt_off || !hoi_out || !labels || !status)
        return SKG_E_ARG;
    if (!skg_aligned16(boxes_h) || !skg_aligned16(boxes_o) || !skg_aligned16(gt_h) || !skg_aligned16(gt_o))
        return SKG_E_ALIGN;
    hipLaunchKernelGGL(skg_eval_associate_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, boxes_h, boxes_o,
                       object, pair_off, index, pred, scores, cell_off, lut, n_obj, n_verb, gt_h, gt_o, gt_hoi, gt_off,
                       min_iou, hoi_out, labels, status);
    return skg_launch_status();
}

// ------------------------------------------------------------------------------------------------ 11-point AP
// labels_sorted: the 0/1 labels of all detections, sorted by class (ascending) and inside a class by score (descending,
// ties in arrival order); class_off[c .. c+1] = the class's range.  ap[c] = (1/11) sum_k max{prec_i : rec_i >= thr[k]},
// prec_i = tp_i / (i + 1), rec_i = tp_i / num_gt[c]; 0 for a class without ground truth or detections.
__global__ __launch_bounds__(256) void skg_eval_ap11_kernel(const float* __restrict__ labels_sorted,
                                                            const int64_t* __restrict__ class_off,
                                                            const int64_t* __restrict__ num_gt,
                                                            const double* __restrict__ thr, double* __restrict__ ap) {
    __shared__ double swave[4];
    __shared__ double sbase;
    __shared__ double smax[4][11];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t s0 = class_off[c], s1 = class_off[c + 1];
    const double ng = (double)num_gt[c];
    if (ng <= 0.0 || s1 <= s0) { if (tid == 0) ap[c] = 0.0; return; }
    double t[11], best[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) { t[k] = thr[k]; best[k] = 0.0; }
    if (tid == 0) sbase = 0.0;
    __syncthreads();
    for (int64_t i0 = s0; i0 < s1; i0 += 256) {
        const int64_t i = i0 + tid;
        const double y = i < s1 ? (double)labels_sorted[i] : 0.0;
        // inclusive scan of y over the 256 lanes of this tile
        double inc = y;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) swave[wv] = inc;
        __syncthreads();
        double tp = sbase + inc;
        for (int w = 0; w < wv; ++w) tp += swave[w];
        if (i < s1) {
            const double prec = tp / (double)(i - s0 + 1), rec = tp / ng;
#pragma unroll
            for (int k = 0; k < 11; ++k)
                if (rec >= t[k] && prec > best[k]) best[k] = prec;
        }
        __syncthreads();
        if (tid == 0) sbase += (swave[0] + swave[1]) + (swave[2] + swave[3]);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        double v = best[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
        if (lane == 0) smax[wv][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < 11; ++k) s += fmax(fmax(smax[0][k], smax[1][k]), fmax(smax[2][k], smax[3][k])) / 11.0;
        ap[c] = s;
    }
}

extern "C" int skg_eval_ap11_f64(const float* labels_sorted, const int64_t* class_off, const int64_t* num_gt, int n_classes,
                                 const double* thresholds11, double* ap, void* stream) {
    if (n_classes < 0) return SKG_E_ARG;
    if (n_classes == 0) return 0;
    if (!class_off || !num_gt || !thresholds11 || !ap) return SKG_E_ARG;
    hipLaunchKernelGGL(skg_eval_ap11_kernel, dim3(n_classes), dim3(256), 0, (hipStream_t)stream, labels_sorted, class_off,
                       num_gt, thresholds11, ap);
    return skg_launch_status();
}
