// skg_roialign.hip -- MultiScaleRoIAlign (the feature-cache producer in front of the interaction head).
//
// Reference call site: models/adamixer_transH_spatial_r50_models.py:158-162 (torchvision MultiScaleRoIAlign,
// featmap_names '0'..'3', output 7x7, sampling_ratio 2) used at heads/adamixer_transH_spatial_r50_head.py:387.
// torchvision is not in the image; the published algorithm is restated (oracle/roi_align_oracle.py is the CPU twin):
//   level  = clamp(floor(4 + log2(sqrt(area) / 224) + 1e-6), k_min, k_max) - k_min          (LevelMapper)
//   roi_align(aligned = False): roi scaled by the level's spatial scale, width/height clamped to >= 1, every output bin
//   averages sampling_ratio^2 bilinear samples; samples outside [-1, size] contribute 0.
// One thread per output element with the bin column fastest: neighbouring lanes read neighbouring feature pixels;
// the 16 taps of a bin hit L2.  HBM-bound on the [rois, C, 7, 7] write.
#include "skg_common.h"

struct skg_roi_levels {
    const float* feat[SKG_ROI_MAX_LEVELS];    // [B, C, H_l, W_l]
    int H[SKG_ROI_MAX_LEVELS], W[SKG_ROI_MAX_LEVELS];
    float scale[SKG_ROI_MAX_LEVELS];
    int n_levels, k_min, k_max, C;
    float canonical_scale;
    int canonical_level;
};

__device__ __forceinline__ float skg_bilinear(const float* __restrict__ f, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.f;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
    const float v1 = f[y_low * W + x_low], v2 = f[y_low * W + x_high];
    const float v3 = f[y_high * W + x_low], v4 = f[y_high * W + x_high];
    return hy * hx * v1 + hy * lx * v2 + ly * hx * v3 + ly * lx * v4;
}

__global__ __launch_bounds__(256) void skg_roi_align_kernel(const skg_roi_levels L, const float* __restrict__ boxes,
                                                            const int32_t* __restrict__ box_image, int n_rois,
                                                            int pooled, int sampling, float* __restrict__ out) {
    const int64_t total = (int64_t)n_rois * L.C * pooled * pooled;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int pw = (int)(idx % pooled);
        const int ph = (int)((idx / pooled) % pooled);
        const int c = (int)((idx / ((int64_t)pooled * pooled)) % L.C);
        const int n = (int)(idx / ((int64_t)pooled * pooled * L.C));
        const float4 b = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)n);
        // LevelMapper (torchvision.ops.poolers.LevelMapper)
        const float s = sqrtf((b.z - b.x) * (b.w - b.y));
        float lv = floorf((float)L.canonical_level + log2f(s / L.canonical_scale) + 1e-6f);
        lv = fminf(fmaxf(lv, (float)L.k_min), (float)L.k_max);
        const int l = (int)lv - L.k_min;
        const int H = L.H[l], W = L.W[l];
        const float sc = L.scale[l];
        const float* f = L.feat[l] + ((int64_t)box_image[n] * L.C + c) * H * W;
        const float x1 = b.x * sc, y1 = b.y * sc, x2 = b.z * sc, y2 = b.w * sc;
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bw = rw / (float)pooled, bh = rh / (float)pooled;
        const int gh = sampling > 0 ? sampling : (int)ceilf(rh / pooled);
        const int gw = sampling > 0 ? sampling : (int)ceilf(rw / pooled);
        const float cnt = fmaxf((float)(gh * gw), 1.f);
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
            const float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
                acc += skg_bilinear(f, H, W, y, x);
            }
        }
        out[idx] = acc / cnt;
    }
}

// ------------------------------------------------------------------------------------------------ backward
// d feat[l][b, c, y, x] += sum over the samples that tapped it of  weight * d out[n, c, ph, pw] / count  (torchvision's
// roi_align backward: no gradient with respect to the boxes).  One thread per output element, float atomics on the
// feature gradients (zeroed by the caller): the summation order is not fixed, as in torchvision's kernel.
__global__ __launch_bounds__(256) void skg_roi_align_bwd_kernel(const skg_roi_levels L, const float* __restrict__ boxes,
                                                                const int32_t* __restrict__ box_image, int n_rois,
                                                                int pooled, int sampling,
                                                                const float* __restrict__ dout) {
    const int64_t total = (int64_t)n_rois * L.C * pooled * pooled;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int pw = (int)(idx % pooled);
        const int ph = (int)((idx / pooled) % pooled);
        const int c = (int)((idx / ((int64_t)pooled * pooled)) % L.C);
        const int n = (int)(idx / ((int64_t)pooled * pooled * L.C));
        const float4 b = *reinterpret_cast<const float4*>(boxes + 4 * (int64_t)n);
        const float s = sqrtf((b.z - b.x) * (b.w - b.y));
        float lv = floorf((float)L.canonical_level + log2f(s / L.canonical_scale) + 1e-6f);
        lv = fminf(fmaxf(lv, (float)L.k_min), (float)L.k_max);
        const int l = (int)lv - L.k_min;
        const int H = L.H[l], W = L.W[l];
        const float sc = L.scale[l];
        float* f = const_cast<float*>(L.feat[l]) + ((int64_t)box_image[n] * L.C + c) * H * W;       // gradient map of the level
        const float x1 = b.x * sc, y1 = b.y * sc, x2 = b.z * sc, y2 = b.w * sc;
        const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
        const float bw = rw / (float)pooled, bh = rh / (float)pooled;
        const int gh = sampling > 0 ? sampling : (int)ceilf(rh / pooled);
        const int gw = sampling > 0 ? sampling : (int)ceilf(rw / pooled);
        const float g = dout[idx] / fmaxf((float)(gh * gw), 1.f);
        for (int iy = 0; iy < gh; ++iy) {
            float y = y1 + ph * bh + (iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                float x = x1 + pw * bw + (ix + 0.5f) * bw / (float)gw;
                float yy = y;
                if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;
                if (yy <= 0.f) yy = 0.f;
                if (x <= 0.f) x = 0.f;
                int y_low = (int)yy, x_low = (int)x, y_high, x_high;
                if (y_low >= H - 1) { y_high = y_low = H - 1; yy = (float)y_low; } else y_high = y_low + 1;
                if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
                const float ly = yy - y_low, lx = x - x_low, hy = 1.f - ly, hx = 1.f - lx;
                atomicAdd(f + y_low * W + x_low, hy * hx * g);
                atomicAdd(f + y_low * W + x_high, hy * lx * g);
                atomicAdd(f + y_high * W + x_low, ly * hx * g);
                atomicAdd(f + y_high * W + x_high, ly * lx * g);
            }
        }
    }
}

static int skg_roi_levels_fill(skg_roi_levels& L, const float* const* feats_host, const int32_t* H_host,
                               const int32_t* W_host, const float* scales_host, int n_levels, int C, int k_min, int k_max,
                               float canonical_scale, int canonical_level) {
    for (int l = 0; l < SKG_ROI_MAX_LEVELS; ++l) {
        const bool in = l < n_levels;
        L.feat[l] = in ? feats_host[l] : nullptr;
        L.H[l] = in ? H_host[l] : 0; L.W[l] = in ? W_host[l] : 0; L.scale[l] = in ? scales_host[l] : 0.f;
        if (in && (!L.feat[l] || L.H[l] <= 0 || L.W[l] <= 0)) return SKG_E_ARG;
    }
    L.n_levels = n_levels; L.k_min = k_min; L.k_max = k_max; L.C = C;
    L.canonical_scale = canonical_scale; L.canonical_level = canonical_level;
    return 0;
}

extern "C" int skg_roi_align_bwd_f32(float* const* dfeats_host, const int32_t* H_host, const int32_t* W_host,
                                     const float* scales_host, int n_levels, int C, int k_min, int k_max,
                                     float canonical_scale, int canonical_level, const float* boxes,
                                     const int32_t* box_image, int n_rois, int pooled, int sampling, const float* dout,
                                     void* stream) {
    if (n_levels < 1 || n_levels > SKG_ROI_MAX_LEVELS || C <= 0 || pooled <= 0 || n_rois < 0 || k_max - k_min + 1 != n_levels)
        return SKG_E_ARG;
    if (n_rois == 0) return 0;
    if (!dfeats_host || !H_host || !W_host || !scales_host || !boxes || !box_image || !dout) return SKG_E_ARG;
    if (!skg_aligned16(boxes)) return SKG_E_ALIGN;
    skg_roi_levels L;
    const int rc = skg_roi_levels_fill(L, dfeats_host, H_host, W_host, scales_host, n_levels, C, k_min, k_max, canonical_scale,
                                       canonical_level);
    if (rc) return rc;
    const int64_t total = (int64_t)n_rois * C * pooled * pooled;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(skg_roi_align_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L, boxes,
                       box_image, n_rois, pooled, sampling, dout);
    return skg_launch_status();
}

extern "C" int skg_roi_align_f32(const float* const* feats_host, const int32_t* H_host, const int32_t* W_host,
                                 const float* scales_host, int n_levels, int C, int k_min, int k_max,
                                 float canonical_scale, int canonical_level, const float* boxes,
                                 const int32_t* box_image, int n_rois, int pooled, int sampling, float* out,
                                 void* stream) {
    if (n_levels < 1 || n_levels > SKG_ROI_MAX_LEVELS || C <= 0 || pooled <= 0 || n_rois < 0 || k_max - k_min + 1 != n_levels)
        return SKG_E_ARG;
    if (n_rois == 0) return 0;
    if (!feats_host || !H_host || !W_host || !scales_host || !boxes || !box_image || !out) return SKG_E_ARG;
    if (!skg_aligned16(boxes)) return SKG_E_ALIGN;
    skg_roi_levels L;
    for (int l = 0; l < SKG_ROI_MAX_LEVELS; ++l) {
        const bool in = l < n_levels;
        L.feat[l] = in ? feats_host[l] : nullptr;
        L.H[l] = in ? H_host[l] : 0; L.W[l] = in ? W_host[l] : 0; L.scale[l] = in ? scales_host[l] : 0.f;
        if (in && (!L.feat[l] || L.H[l] <= 0 || L.W[l] <= 0)) return SKG_E_ARG;
    }
    L.n_levels = n_levels; L.k_min = k_min; L.k_max = k_max; L.C = C;
    L.canonical_scale = canonical_scale; L.canonical_level = canonical_level;
    const int64_t total = (int64_t)n_rois * C * pooled * pooled;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;                          // grid-stride the rest
    hipLaunchKernelGGL(skg_roi_align_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, L, boxes,
                       box_image, n_rois, pooled, sampling, out);
    return skg_launch_status();
}
