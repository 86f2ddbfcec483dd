// skg_layout.cpp -- host-side batch layout of a TRAINING step in one call (include/skghoi.h, skg_layout_pack_train).
//
// The reference walks the images of a batch in a Python loop (heads/adamixer_transH_spatial_r50_head.py:822-982) and builds
// its index tensors image by image.  The MI355X step lays every image out in concatenated row spaces (skg_image_meta) and
// gathers through small int32 tables; building those with numpy (skghoi_amd/layout.py: build + pack_int_arrays, ~40 array
// operations) took ~0.13 ms of a 1.4 ms step whose host thread is the bottleneck.  This is the same arithmetic as plain
// loops, written straight into the caller's (pinned) staging buffer: one call, a few microseconds.  No device work.
#include <stdint.h>
#include <string.h>
#include "skghoi.h"

namespace {
struct Cursor {
    int32_t* buf; int64_t cap; int64_t cur; bool fits;
    int32_t* take(int64_t n, int32_t& off, int32_t& len) {
        cur = (cur + 3) & ~(int64_t)3;                       // every slice 16-byte aligned
        off = (int32_t)cur; len = (int32_t)n;
        int32_t* p = (buf && cur + n <= cap) ? buf + cur : nullptr;
        if (!p && n > 0) fits = false;
        cur += n;
        return p;
    }
};
}  // namespace

extern "C" int skg_layout_pack_train(const int64_t* n_h, const int64_t* n, const int64_t* L, int B, const float* shapes_hw,
                                     int human_idx, int faithful_skip_offset, int zip_truncation, const int32_t* gt_count,
                                     int32_t* buf_host, int64_t cap_ints, skg_layout_info* info) {
    if (!n_h || !n || !info || B < 0 || (B > 0 && !shapes_hw)) return SKG_E_ARG;
    memset(info, 0, sizeof(*info));
    int64_t sum_all = 0;
    for (int b = 0; b < B; ++b) { if (n[b] < 0 || n_h[b] < 0) return SKG_E_ARG; sum_all += n[b]; }
    const int n_visit = zip_truncation ? (int)(B < sum_all ? B : sum_all) : B;          // HEAD:822
    // active images and the sizes of the row spaces
    int A = 0;
    int64_t sum_n = 0, sum_h = 0, sum_g = 0, sum_p = 0, sum_l = 0, gt_total = 0, max_n = 0;
    for (int b = 0; b < n_visit; ++b) {
        if (n_h[b] == 0 || n[b] <= 1) continue;                                          // HEAD:829 (skipped)
        ++A; sum_n += n[b]; sum_h += n_h[b]; sum_g += n_h[b] * n[b]; sum_p += n_h[b] * (n[b] - 1);
        sum_l += L ? L[b] : 0;
        gt_total += gt_count ? gt_count[b] : 0;
        if (n[b] > max_n) max_n = n[b];
    }
    if (sum_g >= (1LL << 31) || sum_all >= (1LL << 31)) return SKG_E_LIMIT;
    info->B = B; info->n_visit = n_visit; info->n_active = A;
    info->sum_all = sum_all; info->sum_n = sum_n; info->sum_h = sum_h; info->sum_g = sum_g; info->sum_p = sum_p; info->sum_l = sum_l;
    // the reference indexes an 80-row embedding with y and human_idx (HEAD:570-572, 690): IndexError there
    info->index_error = (A && (max_n > SKG_TRANSH_ENT || human_idx >= SKG_TRANSH_ENT || human_idx < 0)) ? 1 : 0;

    Cursor c{buf_host, cap_ints, 0, true};
    int32_t* meta_i = c.take((int64_t)A * 12, info->off[SKG_LAY_META], info->len[SKG_LAY_META]);
    int32_t* node_img = c.take(sum_n, info->off[SKG_LAY_NODE_IMG], info->len[SKG_LAY_NODE_IMG]);
    int32_t* hum_img = c.take(sum_h, info->off[SKG_LAY_HUM_IMG], info->len[SKG_LAY_HUM_IMG]);
    int32_t* node_enc = c.take(sum_n, info->off[SKG_LAY_NODE_ENC_ROW], info->len[SKG_LAY_NODE_ENC_ROW]);
    int32_t* hum_enc = c.take(sum_h, info->off[SKG_LAY_HUM_ENC_ROW], info->len[SKG_LAY_HUM_ENC_ROW]);
    int32_t* node_ent = c.take(sum_n, info->off[SKG_LAY_NODE_ENT_ROW], info->len[SKG_LAY_NODE_ENT_ROW]);
    int32_t* hum_ent = c.take(sum_h, info->off[SKG_LAY_HUM_ENT_ROW], info->len[SKG_LAY_HUM_ENT_ROW]);
    int32_t* enc_hn = c.take(sum_h + sum_n, info->off[SKG_LAY_ENC_ROW_HN], info->len[SKG_LAY_ENC_ROW_HN]);
    int32_t* img_hn = c.take(sum_h + sum_n, info->off[SKG_LAY_IMG_HN], info->len[SKG_LAY_IMG_HN]);
    int32_t* ent_hn = c.take(sum_h + sum_n, info->off[SKG_LAY_ENT_ROW_HN], info->len[SKG_LAY_ENT_ROW_HN]);
    const int64_t na1 = sum_all > 0 ? sum_all : 1;
    int32_t* hum_of = c.take(na1, info->off[SKG_LAY_HUM_OF], info->len[SKG_LAY_HUM_OF]);
    int32_t* node_of = c.take(na1, info->off[SKG_LAY_NODE_OF], info->len[SKG_LAY_NODE_OF]);
    int32_t* pair_img = c.take(sum_p, info->off[SKG_LAY_PAIR_IMG], info->len[SKG_LAY_PAIR_IMG]);
    int32_t* gt_off = c.take((int64_t)A + 1, info->off[SKG_LAY_GT_OFF], info->len[SKG_LAY_GT_OFF]);
    int32_t* active = c.take(A, info->off[SKG_LAY_ACTIVE], info->len[SKG_LAY_ACTIVE]);
    info->ints = c.cur > 4 ? c.cur : 4;
    if (!c.fits || !buf_host) return 0;                       // sizing call (or a buffer too small): info->ints says how much
    (void)gt_total;

    for (int64_t i = 0; i < na1; ++i) { hum_of[i] = -1; node_of[i] = -1; }
    int a = 0;
    int64_t box_off = 0, node_off = 0, hum_off = 0, grid_off = 0, pair_off = 0, out_off = 0, gt_acc = 0;
    int64_t enc_run = 0;                                      // Q9: skipped images do not advance the encoding offset
    gt_off[0] = 0;
    for (int b = 0; b < B; ++b) {
        const bool act = b < n_visit && !(n_h[b] == 0 || n[b] <= 1);
        if (act) {
            const int64_t nh = n_h[b], nn = n[b];
            const int64_t enc_off = faithful_skip_offset ? enc_run : box_off;
            skg_image_meta* m = reinterpret_cast<skg_image_meta*>(meta_i) + a;
            m->image = b; m->n_h = (int32_t)nh; m->n = (int32_t)nn; m->box_off = (int32_t)box_off;
            m->enc_off = (int32_t)enc_off; m->node_off = (int32_t)node_off; m->hum_off = (int32_t)hum_off;
            m->grid_off = (int32_t)grid_off; m->pair_off = (int32_t)pair_off; m->out_off = (int32_t)out_off;
            m->img_h = shapes_hw[2 * b]; m->img_w = shapes_hw[2 * b + 1];
            for (int64_t j = 0; j < nn; ++j) {
                node_img[node_off + j] = a; node_enc[node_off + j] = (int32_t)(enc_off + j); node_ent[node_off + j] = (int32_t)j;
                img_hn[sum_h + node_off + j] = a; enc_hn[sum_h + node_off + j] = (int32_t)(enc_off + j);
                ent_hn[sum_h + node_off + j] = (int32_t)j;
                if (enc_off + j < na1) node_of[enc_off + j] = (int32_t)(node_off + j);
            }
            for (int64_t i = 0; i < nh; ++i) {
                hum_img[hum_off + i] = a; hum_enc[hum_off + i] = (int32_t)(enc_off + i); hum_ent[hum_off + i] = human_idx;
                img_hn[hum_off + i] = a; enc_hn[hum_off + i] = (int32_t)(enc_off + i); ent_hn[hum_off + i] = human_idx;
                if (enc_off + i < na1) hum_of[enc_off + i] = (int32_t)(hum_off + i);
            }
            const int64_t pp = nh * (nn - 1);
            for (int64_t p = 0; p < pp; ++p) pair_img[pair_off + p] = b;        // (the image's index in the BATCH)
            active[a] = b;
            gt_acc += gt_count ? gt_count[b] : 0;
            gt_off[a + 1] = (int32_t)gt_acc;
            node_off += nn; hum_off += nh; grid_off += nh * nn; pair_off += pp; out_off += L ? L[b] : 0;
            enc_run += nn;
            ++a;
        }
        box_off += n[b];
    }
    return 0;
}
