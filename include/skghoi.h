/* skghoi.h -- C ABI of the MI355X-native SKGHOI interaction-head hot path (libskghoi_hip.so).
 *
 * The reference (lijingzhu1/SKGHOI) has no native boundary: its interaction head is eager PyTorch.  This header is the
 * boundary a maintainer binds instead (ctypes stub: INTEGRATION.md); every entry point names the reference code it
 * replaces (paths relative to the reference root, HEAD = heads/adamixer_transH_spatial_r50_head.py).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  Every pointer is a DEVICE pointer unless its name ends in
 *     _host.  All floating point is fp32, indices are int32 unless stated (int64 where the reference's result tensors
 *     are int64).
 *   - Nothing allocates, nothing synchronises: work is enqueued on `stream` (a hipStream_t passed as void*), workspaces
 *     are provided by the caller.
 *   - Return value: 0 = enqueued; < 0 = rejected argument (SKG_E_*); > 0 = hipError_t from the launch.
 *   - Matrices are row-major; "ld" = elements between consecutive rows.  Weights keep the nn.Linear layout [out, in].
 */
#ifndef SKGHOI_H
#define SKGHOI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SKG_ABI_VERSION 18
#define SKG_E_ARG   (-1)   /* null pointer / negative size / unsupported shape            */
#define SKG_E_ALIGN (-2)   /* pointer or leading dimension not 16-byte aligned            */
#define SKG_E_LIMIT (-3)   /* exceeds a compiled-in limit (boxes per image, verbs, ...)   */
#define SKG_E_UNSUPPORTED (-4) /* an optional run-time dependency is absent (RCCL)        */
#define SKG_E_COMM  (-5)   /* an RCCL call failed: skg_comm_last_error() has the text     */

#define SKG_MAX_DET_PER_IMAGE 1024   /* candidates per image the preprocess kernel accepts */
#define SKG_MAX_NODES         160    /* max_human + max_object                             */
#define SKG_SPATIAL_LD        48     /* 46 spatial features padded to a multiple of 8      */
#define SKG_TRANSH_DIM        50     /* HEAD:686                                           */
#define SKG_TRANSH_ENT        80     /* HEAD:690                                           */

int         skg_abi_version(void);
const char* skg_build_info(void);     /* "gfx950 <compiler> <date>" */

/* ---------------------------------------------------------------------------------------------------------------
 * Per-image metadata of the ACTIVE (non-skipped, HEAD:829) images of a batch, built on the host from (n_h, n) and
 * uploaded once.  One record per active image.                                                                     */
typedef struct {
    int32_t image;      /* index of the image in the batch                                            */
    int32_t n_h, n;     /* humans, nodes (humans first; HEAD:826, HEAD:823)                           */
    int32_t box_off;    /* first row of this image in the packed detections [sumN_all, ...]           */
    int32_t enc_off;    /* first row of this image's node encodings in box_head's output (HEAD:843;   */
                        /* the reference does not advance this over skipped images)        */
    int32_t node_off;   /* first graph-node row (active images only)                                  */
    int32_t hum_off;    /* first human row (active images only)                                       */
    int32_t grid_off;   /* first grid row  (n_h*n rows per image, self pairs included; HEAD:847-860)  */
    int32_t pair_off;   /* first kept pair (x != y)                                                   */
    int32_t out_off;    /* first scored (pair, verb) cell in the packed result arrays                 */
    float   img_h, img_w;
} skg_image_meta;

/* ---------------------------------------------------------------------------------------------------------------
 * Host-side layout of a training batch in ONE call (no device work): from the per-image counts the preprocess kernel
 * reports -- humans n_h[B], nodes n[B], scored cells L[B] (may be NULL) -- the skg_image_meta records of the active images
 * and every small index table the step gathers through, written as int32 slices (each 16-byte aligned) into the caller's
 * staging buffer buf_host (pinned memory: the step uploads it in one copy).  Replaces the per-image index bookkeeping of
 * the reference's image loop (HEAD:822-982).  shapes_hw[2 b], [2 b + 1] = image height, width.  Slices (info->off / len,
 * in int32 units): META (n_active x 12 = skg_image_meta), NODE_IMG / HUM_IMG (active-image index of every node / human),
 * NODE_ENC_ROW / HUM_ENC_ROW (row of box_head's output; faithful_skip_offset != 0 reproduces the reference's offset bug on
 * skipped images), NODE_ENT_ROW / HUM_ENT_ROW (TransH entity row: position y / human_idx), the three
 * [human rows | node rows] tables of the fc_head | fc_tail gather (ENC_ROW_HN, IMG_HN, ENT_ROW_HN), HUM_OF / NODE_OF (inverse
 * maps encoding row -> human / node row, -1 = none; max(sum_all, 1) entries), PAIR_IMG (batch index of every kept pair's
 * image), GT_OFF (n_active + 1 prefix sums of gt_count over the active images; zeros when gt_count is NULL), ACTIVE (batch
 * index of every active image).  zip_truncation != 0: at most sum(n) images are visited (HEAD:822 zips over the rows of
 * box_features).  Returns 0 and fills info (sizes of the row spaces, info->ints = int32 words needed); when buf_host is
 * NULL or cap_ints < info->ints nothing is written (sizing call).  info->index_error = 1 where the reference raises
 * IndexError (more than 80 nodes, human_idx outside the 80-row TransH table: HEAD:570-572, 690).                        */
enum { SKG_LAY_META = 0, SKG_LAY_NODE_IMG, SKG_LAY_HUM_IMG, SKG_LAY_NODE_ENC_ROW, SKG_LAY_HUM_ENC_ROW, SKG_LAY_NODE_ENT_ROW,
       SKG_LAY_HUM_ENT_ROW, SKG_LAY_ENC_ROW_HN, SKG_LAY_IMG_HN, SKG_LAY_ENT_ROW_HN, SKG_LAY_HUM_OF, SKG_LAY_NODE_OF,
       SKG_LAY_PAIR_IMG, SKG_LAY_GT_OFF, SKG_LAY_ACTIVE, SKG_LAY_SLICES };
typedef struct {
    int32_t B, n_visit, n_active, index_error;
    int64_t sum_all, sum_n, sum_h, sum_g, sum_p, sum_l, ints;
    int32_t off[SKG_LAY_SLICES], len[SKG_LAY_SLICES];
} skg_layout_info;
int skg_layout_pack_train(const int64_t* n_h_host, const int64_t* n_host, const int64_t* L_host, int B,
                          const float* shapes_hw_host, int human_idx, int faithful_skip_offset, int zip_truncation,
                          const int32_t* gt_count_host, int32_t* buf_host, int64_t cap_ints, skg_layout_info* info_host);

/* ---------------------------------------------------------------------------------------------------------------
 * InteractionHead.preprocess (HEAD:92-151): score >= thresh -> class-wise NMS (torchvision batched_nms, coordinate
 * trick, IoU > nms_thresh suppresses) -> descending score (ties: ascending input index) -> first max_human humans and
 * max_object others -> humans first.  One workgroup per image.
 *   det_off[B+1]  : row range of each image in boxes/scores/labels (train: GT boxes already prepended, HEAD:107-116)
 *   nverbs[num_obj_classes] : number of target classes per object class (len(object_class_to_target_class[c]))
 *   prior_pow     : exponent applied to detection scores by compute_prior_scores (HEAD:742), used only for out_count
 *   out_index[B, max_human+max_object] : selected rows (index local to the image), humans first
 *   out_count[B,4] : {n_h, n, L, #candidates}; L = number of non-zero prior cells = size of the image's scored result
 *                    (HEAD:315).  An image with more than SKG_MAX_DET_PER_IMAGE rows (or a negative row range) is not
 *                    processed: its out_count is {-1, -1, -1, rows} and its out_index row is all -1.
 */
int skg_preprocess_f32(const float* boxes, const float* scores, const int64_t* labels, const int32_t* det_off, int B,
                       int human_idx, float score_thresh, float nms_thresh, int max_human, int max_object,
                       const int32_t* nverbs, int num_obj_classes, float prior_pow,
                       int32_t* out_index, int32_t* out_count, void* stream);

/* Gathers the selected detections into packed arrays [sumN, ...] (HEAD:144-149).  sel_off[B+1] = prefix of n. */
int skg_pack_detections_f32(const float* boxes, const float* scores, const int64_t* labels, const int32_t* det_off,
                            const int32_t* index, int index_ld, const int32_t* sel_off, int B,
                            float* out_boxes, float* out_scores, int64_t* out_labels, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Pair enumeration (HEAD:847-860) fused with compute_spatial_ratio_encodings (ops.py:85-157) and the NaN scrub
 * (HEAD:866-868).  One wavefront per active image, boxes staged in LDS.
 *   grid_h[sumG], grid_o[sumG] : global human row / global graph-node row of each grid row
 *   grid_pair[sumG]            : kept-pair index of the grid row, or -1 for the self pair
 *   grid_img[sumG]             : batch index of the row's image (meta.image)
 *   pair_h[sumP], pair_o[sumP] : global human row / graph-node row of each kept pair
 *   pair_grid[sumP]            : grid row of each kept pair (row-major nonzero(x != y) order)
 *   x_keep[sumP], y_keep[sumP] : int64 local indices, the reference's x_keep / y_keep (HEAD:852)
 *   spatial[sumG, 48]          : 23 features + log(f + 1e-10); columns 46, 47 are zero                             */
int skg_pairs_spatial_f32(const float* boxes, const skg_image_meta* meta, int n_active,
                          int32_t* grid_h, int32_t* grid_o, int32_t* grid_pair, int32_t* grid_img,
                          int32_t* pair_grid, int64_t* x_keep, int64_t* y_keep, int32_t* pair_h, int32_t* pair_o,
                          float* spatial, int scrub_nan, void* stream);
/* The same with CAPACITY PADDING for launch plans captured once per bucket of shapes (skghoi_amd/small.py): every image owns
 * grid_cap grid rows and pair_cap pair rows (its meta offsets are strided accordingly) of which it uses n_h * n and
 * n_h * (n - 1).  The unused tails receive index entries that are safe to gather / scatter through (a valid row to read,
 * -1 = "not stored" in grid_pair) and zero spatial features.  grid_cap = pair_cap = 0: no padding (= skg_pairs_spatial_f32). */
int skg_pairs_spatial_padded_f32(const float* boxes, const skg_image_meta* meta, int n_active, int32_t* grid_h,
                                 int32_t* grid_o, int32_t* grid_pair, int32_t* grid_img, int32_t* pair_grid,
                                 int64_t* x_keep, int64_t* y_keep, int32_t* pair_h, int32_t* pair_o, float* spatial,
                                 int scrub_nan, int grid_cap, int pair_cap, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * MultiScaleRoIAlign (models/adamixer_transH_spatial_r50_models.py:158-162, called at HEAD:387): the producer of the
 * cached box features.  feats_host / H_host / W_host / scales_host are HOST arrays of n_levels entries (device
 * pointers to [B, C, H_l, W_l] maps, their sizes and spatial scales); boxes [n_rois, 4] in image pixels, box_image
 * [n_rois] = batch index; out [n_rois, C, pooled, pooled].  Level = clamp(floor(canonical_level + log2(sqrt(area) /
 * canonical_scale) + 1e-6), k_min, k_max) - k_min; roi_align with aligned = False, sampling_ratio samples per bin axis
 * (<= 0: adaptive).                                                                                                 */
#define SKG_ROI_MAX_LEVELS 8
int skg_roi_align_f32(const float* const* feats_host, const int32_t* H_host, const int32_t* W_host,
                      const float* scales_host, int n_levels, int C, int k_min, int k_max, float canonical_scale,
                      int canonical_level, const float* boxes, const int32_t* box_image, int n_rois, int pooled,
                      int sampling, float* out, void* stream);
/* Backward of skg_roi_align_f32 with respect to the feature maps (torchvision's roi_align backward; no gradient for the
 * boxes): dfeats_host[l] points at a ZEROED [B, C, H_l, W_l] gradient map per level, dout is [n_rois, C, pooled, pooled].
 * Float atomics: the order of the additions into a pixel is not fixed (as in torchvision's kernel).               */
int skg_roi_align_bwd_f32(float* const* dfeats_host, const int32_t* H_host, const int32_t* W_host,
                          const float* scales_host, int n_levels, int C, int k_min, int k_max, float canonical_scale,
                          int canonical_level, const float* boxes, const int32_t* box_image, int n_rois, int pooled,
                          int sampling, const float* dout, void* stream);

/* AdaptiveAvgPool2d(1) of features['3'] (HEAD:811): in [B, C, HW] -> out [B, C]. */
int skg_global_avgpool_f32(const float* in, int B, int C, int HW, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Dense layer  C = epilogue(A x W^T + bias)  on the fp32 MFMA (v_mfma_f32_32x32x2_f32), 128x128x16 LDS tiles.
 * Replaces every nn.Linear / MultiBranchFusion GEMM of the head (HEAD:469-474, 509-527, 635-641, 662-669, 694-701,
 * 410-411).  The 16 branches of an MBF are one GEMM against the row-stacked fc_1/fc_2 weights [1024, in] and the
 * column-stacked fc_3 weights [1024, 16*64] (sum over branches == K-concatenation).                                */
typedef enum {
    SKG_EPI_BIAS      = 0,   /* C = A W^T + bias                                                                  */
    SKG_EPI_BIAS_RELU = 1,   /* C = relu(A W^T + bias)                                                            */
    SKG_EPI_MUL_RELU  = 2,   /* v = A W^T + bias ; C[orow] = relu(v * (P[pi[row]] + Q[qi[row]] + mbias))          */
    SKG_EPI_RELU_DOT  = 3,   /* v = relu(A W^T + bias) ; dot_partial[nb][row] = sum_cols v * dot_w  (HEAD:896-897) */
    SKG_EPI_BIAS_RES_RELU = 4 /* C = res + relu(A W^T + bias)   (message + residual, HEAD:909-914, 916-925)        */
} skg_epilogue;

typedef struct {
    const float* A;   int64_t lda;     /* [M, K]; with a_rows: row r is A + a_rows[r]*lda (a_rows[r] < 0 -> zeros)  */
    const float* W;   int64_t ldw;     /* [N, K]                                                                    */
    const float* bias;                 /* [N] or NULL                                                               */
    float*       C;   int64_t ldc;     /* [M(out rows), N]; may be NULL for SKG_EPI_RELU_DOT                        */
    int32_t M, N, K;                   /* K % 4 == 0, lda/ldw % 4 == 0                                              */
    int32_t epilogue;                  /* skg_epilogue                                                              */
    const int32_t* a_rows;             /* optional row gather of A                                                  */
    const int32_t* out_rows;           /* optional row scatter of C (out_rows[r] < 0 -> row not stored)             */
    /* SKG_EPI_MUL_RELU */
    const float* P; const int32_t* p_idx; int64_t ldp;    /* P may be NULL only if Q is given                        */
    const float* Q; const int32_t* q_idx; int64_t ldq;    /* optional second table                                   */
    const float* mbias;                                   /* optional [N], added to the multiplier                   */
    float*       C_raw; int64_t ldc_raw;                  /* optional: also store v (pre-multiplication), by row     */
    /* SKG_EPI_RELU_DOT */
    const float* dot_w;                /* [N]                                                                       */
    float*       dot_partial;          /* [skg_gemm_dot_partials(desc), M]: one partial per column slab of a wave    */
    /* SKG_EPI_BIAS_RES_RELU */
    const float* res; int64_t ldres;   /* [M, N]                                                                    */
    /* split-K (BIAS / BIAS_RELU / BIAS_RES_RELU): K is cut in split_k slices computed by separate workgroups; raw partial sums
     * go to split_ws [split_k, M, N] and a second kernel adds them in slice order and applies the epilogue.          */
    int32_t split_k;                   /* 0 or 1 = off                                                              */
    float   w_scale;                   /* with w_split: 1 / scale given to skg_split_weights_f16x2 (a power of two)  */
    float*  split_ws;
    /* optional: W as two fp16 planes per element (h + m, 22 significant bits) in MFMA-fragment order, made by
     * skg_split_weights_f16x2.  When given (and K % 16 == 0, w_scale > 0) the product runs on the fp16 matrix pipe
     * from 2-way fp16 splits of both operands (h.h + h.m + m.h, fp32 accumulation; ~2^-22 relative per product, i.e.
     * fp32 grade); tiles whose result is not finite (fp16 range exceeded, inf / nan inputs) are recomputed with the
     * exact fp32 loop, for which W is still required. */
    const void* w_split;
    /* optional with w_split: one exponent per A row (after the a_rows gather), from skg_row_exponents_f32 for THIS operand.
     * Row r then travels as 2^-a_exp[r] * A[r, :] (its max |.| in [2^11, 2^12)) and the result row is multiplied back by
     * 2^a_exp[r] in the epilogue -- both exact: the split keeps its 22 significant bits whatever the magnitude of the
     * activations (without it the absolute error floor is 2^-25: fine for O(1) activations, 3 % at a gain of 2^-20),
     * and an outlier row does not cost the others their precision.  NULL = no scaling.                               */
    const int32_t* a_exp;
} skg_gemm_desc;
/* exp_out[r] = floor(log2(max_k |A[r, k]|)) - 11, clamped to [-126, 126]; 0 for all-zero rows and for rows holding
 * inf / nan (their tile is recomputed exactly anyway).  Rows are gathered through a_rows when given (negative = zero row). */
int skg_row_exponents_f32(const float* A, int64_t lda, const int32_t* a_rows, int M, int K, int32_t* exp_out, void* stream);

int skg_gemm_f32(const skg_gemm_desc* desc_host, void* stream);

/* Re-encodes an nn.Linear weight W [N, K] (fp32, leading dimension ldw) for skg_gemm_desc.w_split: every value of
 * scale * W becomes h + m (two fp16), stored as 1 KiB planes [ceil(N/32)][ceil(K/16)][h|m][k half][32 rows][8 k],
 * zero padded.  `scale` must be a power of two; choose it so that max |scale * W| lies in [2^13, 2^14) (m then stays
 * a normal fp16 for every weight that matters) and pass 1 / scale as skg_gemm_desc.w_scale.  `out` holds
 * skg_split_weights_bytes(N, K) bytes, 16-byte aligned. */
int64_t skg_split_weights_bytes(int N, int K);
int skg_split_weights_f16x2(const float* W, int N, int K, int64_t ldw, float scale, void* out, void* stream);

/* Number of dot_partial slabs a SKG_EPI_RELU_DOT launch of `desc_host` writes (the launcher picks 128 x 128 tiles,
 * or 64 x 64 tiles when the grid would leave most of the 256 CUs idle; the slab is the column range of one wave).
 * Depends on M, N, K, lda, ldw, a_rows and split_k only; < 0 on a null descriptor. */
int skg_gemm_dot_partials(const skg_gemm_desc* desc_host);

/* Up to SKG_GEMM_GROUP_MAX independent GEMMs in one launch (the node-row GEMMs of the graph: fc_head | fc_tail,
 * the four fc_1 projections, the two message fc_3, HEAD:884-885, 894-896, 514-524); epilogues BIAS / BIAS_RELU /
 * BIAS_RES_RELU / MUL_RELU / RELU_DOT are selected per descriptor at run time. */
#define SKG_GEMM_GROUP_MAX 4
int skg_gemm_group_f32(const skg_gemm_desc* descs_host, int n, void* stream);
/* Tile scale the grouped launch of these descriptors will use: 2 = 128 x 128 tiles, 1 = 64 x 64 tiles (every member fits
 * the DMA-staged loop -- K % 16 == 0, no row gather, no weight twin -- and the group is small: a few images).  With 64 x 64
 * tiles a member may carry split_k > 1 (epilogues BIAS / BIAS_RELU / BIAS_RES_RELU): one more launch then reduces all
 * split members in slice order.  SKG_EPI_RELU_DOT members write 2 * ceil(N / (64 * scale)) dot_partial slabs.       */
int skg_gemm_group_tile(const skg_gemm_desc* descs_host, int n);
/* Tuning switches of the eval GEMM (developer A/B knobs; 0 = the library's default).  They live in a CONTEXT the caller creates
 * (skg_context, below) and apply to the skg_gemm_* calls of the threads that made that context current
 * (skg_ctx_make_current) -- the library keeps no process-wide tuning state.
 *   small_mode   which 64 x 64 main loop the small launches take: 3 = 64-k steps, register staged (default); 1 = 16-k steps,
 *                DMA staged; 4 = 64-k steps staged straight into LDS; 5 / 6 = the eight-wave loop (all / up to khalves_blocks
 *                workgroups)
 *   small_tiles  a launch / group takes the 64 x 64 tiles below this many 128 x 128 tiles (default 384)
 *   route_tiles  "small" launches with at least this many 128 x 128 tiles (split-K slices counted) run on skg_gemmx_f32's
 *                register-pipelined 128 x 128 loop, fused epilogue included -- the regime of 2-8 images (default 200; a very
 *                large value turns the routing off)                                                                       */
typedef struct { int32_t small_mode, small_tiles, route_tiles, khalves_blocks; } skg_tuning;

/* ---------------------------------------------------------------------------------------------------------------
 * fc_head / fc_tail input rows (HEAD:884-885): out[r] = [ enc[enc_row[r], 0:1024] | ent[ent_img[r], ent_row[r], 0:50]
 * | zeros ] with out_ld = 1088.  ent = per-image TransH entity tables [n_img, 80, 50] (HEAD:574-580; SURVEY Q3:
 * heads use row human_idx, tails use the node's position).                                                         */
int skg_concat_entity_f32(const float* enc, int64_t ld_enc, const int32_t* enc_row, const float* ent,
                          const int32_t* ent_img, const int32_t* ent_row, int rows, float* out, int64_t out_ld,
                          void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * bf16 dense layer (training configuration "bf16"): C = act(A W^T + bias) with A [M,K], W [N,K] in bf16 (raw uint16
 * bit patterns), fp32 accumulation on v_mfma_f32_32x32x16_bf16, bias fp32, C fp32 or bf16.  K % 64 == 0,
 * lda / ldw % 8 == 0.  split_k > 1: fp32 partials in split_ws [split_k, M, N], reduced in slice order.              */
typedef struct {
    const uint16_t* A; int64_t lda;
    const uint16_t* W; int64_t ldw;
    const float* bias;
    void* C; int64_t ldc;
    int32_t M, N, K;
    int32_t relu, out_bf16, split_k;
    float* split_ws;
} skg_gemm_bf16_desc;
int skg_gemm_bf16(const skg_gemm_bf16_desc* desc_host, void* stream);
int skg_transpose_bf16(const void* in, int64_t ld_in, int rows, int cols, void* out, int64_t ld_out, void* stream);

/* out[c, r] = in[r, c]  (rows x cols -> cols x rows, ld_out >= rows).  The backward GEMMs of the training step reuse
 * skg_gemm_f32 (both operands k-contiguous): dA = dZ (W^T)^T needs W^T, dW = dZ^T A needs dZ^T and A^T. */
int skg_transpose_f32(const float* in, int64_t ld_in, int rows, int cols, float* out, int64_t ld_out, void* stream);

/* out[r] = relu((P[pi[r]] + Q[qi[r]] + mbias) * F[fi[r]])  over `cols` columns (read-out MBF fc_1*fc_2, HEAD:970).  */
int skg_rows_mul_relu_f32(const float* P, const int32_t* p_idx, int64_t ldp, const float* Q, const int32_t* q_idx,
                          int64_t ldq, const float* mbias, const float* F, const int32_t* f_idx, int64_t ldf,
                          int rows, int cols, float* out, int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Bipartite message aggregation (HEAD:897-925).  adjacency logit of grid row r = adj_bias + sum_k dot_partial[k][r].
 *   U[h]  = sum_j softmax_j(adj[i, :])[j] * T_os[(i, j)]     (human h = (image, i); obj_to_sub pre-fc_3 rows)
 *   V[o]  = sum_i softmax_i(adj[:, j])[i] * T_so[(i, j)]     (node  o = (image, j); sub_to_obj pre-fc_3 rows)
 * fc_3 is linear and the softmax weights sum to 1, so  sum_j a_ij fc_3(T_ij) == fc_3(sum_j a_ij T_ij)  (DESIGN.md).
 * One workgroup per destination row; the image's adjacency row/column lives in LDS.  adj_out[sumG] receives the
 * logits (HEAD:897).                                                                                               */
int skg_graph_aggregate_f32(const float* dot_partial, int n_partial, int64_t partial_ld, float adj_bias,
                            const skg_image_meta* meta, int n_active, const int32_t* hum_img, const int32_t* node_img,
                            int sum_h, int sum_n, const float* T_os, const float* T_so, int64_t ldt, int cols,
                            float* U, float* V, int64_t ldu, float* adj_out, void* stream);

/* out = LayerNorm(x) * gamma + beta over `cols` (= 1024) columns, eps 1e-5 (HEAD:658-659, 912-914, 923-925). */
int skg_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, int rows, int cols,
                      float eps, float* out, int64_t ldo, void* stream);
/* ... the two of a graph pass (norm_h over the human rows, norm_o over the node rows) in ONE launch; per row bit-identical
 * to skg_layernorm_f32 (round 5: one launch and one graph node less per forward).                                   */
int skg_layernorm2_f32(const float* x0, int64_t ldx0, const float* gamma0, const float* beta0, int rows0, float* out0,
                       int64_t ldo0, const float* x1, int64_t ldx1, const float* gamma1, const float* beta1, int rows1,
                       float* out1, int64_t ldo1, int cols, float eps, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * compute_prior_scores (HEAD:721-767) + InteractionHead.postprocess (HEAD:237-337), table driven.
 *   logits [sumP, ld_logits]: columns 0..K-1 = box_pair_predictor, column K = box_pair_suppressor
 *   verb_off[num_obj+1], verb_list[]: CSR of object_class_to_target_class, verbs ascending per class
 * Per active image, cells are emitted in nonzero(prior[0]) order (pair-major, verb ascending) at meta.out_off.
 *   out_index/out_pred int64 [L], out_scores f32 [L], out_prior f32 [2, L_total] (row 0 = human, row 1 = object),
 *   out_weights f32 [sumP], out_object int64 [sumP], out_boxes_h/out_boxes_o f32 [sumP,4]
 *   L_total_dev (optional): device int32 holding L_total (values < 1 count as 1); when given it overrides the L_total
 *   argument, so that a launch captured into a hipGraph can be replayed for batches with other cell counts.
 *   max_pairs_per_image: the largest n_h * (n - 1) of the batch (sizes the grid: one workgroup per 256 kept pairs of an
 *   image; <= 0: the compiled-in bound).                                                                           */
int skg_postprocess_f32(const float* logits, int64_t ld_logits, int K, const float* boxes, const float* scores,
                        const int64_t* labels, const skg_image_meta* meta, int n_active, const int64_t* x_keep,
                        const int64_t* y_keep, const int32_t* verb_off, const int32_t* verb_list, int num_obj_classes,
                        float prior_pow, int64_t L_total, const int32_t* L_total_dev, int max_pairs_per_image,
                        int64_t* out_index, int64_t* out_pred, float* out_scores, float* out_prior, float* out_weights,
                        int64_t* out_object, float* out_boxes_h, float* out_boxes_o, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * HOST function (no GPU work): the TransH tables of `n_images` consecutive processed images, drawn from PyTorch's CPU
 * generator exactly as the reference's per-image `TransH(...)` construction draws them (HEAD:574-580,
 * heads/TransH/TransH.py:20-28: three nn.Embedding normal_ inits, then three xavier_uniform_).
 * `torch_cpu_rng_state` = the bytes of torch.get_rng_state() (5056; other sizes are rejected), advanced in place --
 * hand it back with torch.set_rng_state().  ent [n_images, 80, 50]; rel / nrm [n_images, K, 50] only when
 * need_relations (their draws are skipped otherwise).  fused_affine: 1 if this PyTorch build evaluates
 * x * (to - from) + from of uniform_ with a fused multiply-add (the binding decides by comparing against torch). */
int skg_transh_draw_f32(void* torch_cpu_rng_state, int64_t state_bytes, int n_images, int K, int need_relations,
                        int fused_affine, float* ent, float* rel, float* nrm);

/* HOST function: the host RNG of a TRAINING forward in the reference's order, for `n_images` processed images: per image
 * the six TransH table fills (HEAD:574-580; ent, rel, nrm all kept) followed by torch.randperm(n_neg[a]) (HEAD:938-939:
 * the sampled negatives), of which the first n_take[a] entries are written to perm_out (concatenated over the images).
 * The generator state ends where the reference's calls leave it (randperm draws n - 1 words).                          */
int skg_transh_draw_train_f32(void* torch_cpu_rng_state, int64_t state_bytes, int n_images, int K, int fused_affine,
                              const int64_t* n_neg, const int64_t* n_take, float* ent, float* rel, float* nrm,
                              int64_t* perm_out);

/* ---------------------------------------------------------------------------------------------------------------
 * TransH hyperplane scores (heads/TransH/TransH.py:56-106) for every (kept pair, relation):
 *   w = norm(nrm[k]); h = ent[human_idx] - (ent[human_idx].w) w; t = ent[y] - (ent[y].w) w
 *   score[p, k] = || norm(h) + norm(rel[k]) - norm(t) ||_2        (F.normalize eps 1e-12)
 *   ent [n_active, 80, 50], rel / nrm [n_active, K, 50] (a fresh table set per image, HEAD:574-580);
 *   scores [sumP, K].  Training only (inference discards them).  One wavefront per (image, relation).  */
int skg_transh_scores_f32(const float* ent, const float* rel, const float* nrm, int K, int human_idx,
                          const skg_image_meta* meta, int n_active, float* scores, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GraphHead.associate_with_ground_truth (HEAD:703-719), training: labels[p, verb] = 1 where a ground-truth pair with
 * that verb overlaps the kept pair p with min(IoU_h, IoU_o) >= thresh.  gt_* are the targets of the active images
 * concatenated, gt_off[n_active + 1] their row ranges; labels [sumP, K] must be zero-filled; npos[n_active] receives
 * the number of positive (pair, verb) cells per image.  One workgroup per image.                                    */
int skg_associate_f32(const float* boxes, const skg_image_meta* meta, int n_active, const int64_t* x_keep,
                      const int64_t* y_keep, const float* gt_h, const float* gt_o, const int64_t* gt_label,
                      const int32_t* gt_off, int K, float thresh, float* labels, int32_t* npos, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM with free operand layouts -- the dense products of the TRAINING step (forward on branch-major MBF
 * weights, and the backward of every nn.Linear / MultiBranchFusion layer: autograd of HEAD:469-474, 509-527, 635-701,
 * 410-411) without operand transposes:
 *     C(m, n) (+)= mask( relu( sum_k A(m, k) B(k, n) + bias[n] ) )
 *   A(m, k) = A[m * a_sm + k * a_sk]                       exactly one of a_sm, a_sk is 1
 *   B(k, n) = B[koff(k) + noff(n)],  koff(k) = b_kshift ? (k >> b_kshift) * b_kstride + (k & (2^b_kshift - 1)) * b_sk
 *                                                       : k * b_sk        (noff likewise with b_nshift / b_nstride / b_sn)
 *             exactly one of b_sk, b_sn is 1; the power-of-two blocking addresses the branch-major storage of the 16
 *             MBF fc_3 weights ([16][1024][64]: blocks of 64 along the contraction in the forward, along n in dX)
 *   C(m, n) = C[(n >> c_nshift) * c_nstride + m * ldc + (n & (2^c_nshift - 1))]   (c_nshift = 0: C[m * ldc + n])
 *   forward        y  = x W^T + b   : A = x (a_sk = 1),    B(k, n) = W[n][k]   (b_sk = 1, b_sn = ld)
 *   input grad     dx = dz W        : A = dz (a_sk = 1),   B(k, n) = W[k][n]   (b_sn = 1, b_sk = ld), mask = x's ReLU
 *   weight grad    dW = dz^T x      : A(m, k) = dz[k][m]   (a_sm = 1, a_sk = ld), B(k, n) = x[k][n] (b_sn = 1);
 *                  a_rowsum[m] = sum_k A(m, k) is the bias gradient, formed on the way by the workgroups of column tile 0
 *   accumulate: add to C / a_rowsum instead of overwriting (weights shared by two call sites; gradients with several
 *   consumers)
 *   mask (optional) [M, >= N], ldmask: C is zeroed where mask <= 0 -- the ReLU of the layer that produced the tensor C is
 *   the gradient of; applied AFTER the accumulation (the sum of all contributions is what the ReLU cuts)
 *   split_k > 1: slices of K by separate workgroups into split_ws (skg_gemmx_ws_floats(desc) floats), reduced in slice
 *   order -- by the last-arriving workgroup of every tile inside the same launch when split_ctr is given, else by a
 *   second launch -- with the epilogue applied to the sum.
 * Up to SKG_GEMMX_GROUP_MAX independent products per call share ONE launch (plus one reduce launch if any is split
 * without counters). */
typedef struct {
    const float* A; int64_t a_sm, a_sk;
    const float* B; int64_t b_sn, b_sk;
    int32_t b_kshift, b_nshift; int64_t b_kstride, b_nstride;
    float*  C; int64_t ldc; int32_t c_nshift, accumulate; int64_t c_nstride;
    int32_t M, N, K, relu;
    const float* bias;
    const float* mask; int64_t ldmask;
    float*  a_rowsum;
    int32_t split_k, reserved;
    float*  split_ws;
    /* bf16 twins (optional, may be NULL).  A16 / B16: the operand rounded to bf16 (round to nearest even), stored with the
     * SAME element indexing as A / B (strides and blocking count elements); skg_gemmx_bf16 reads the twin instead of the
     * fp32 array wherever its fast loop runs -- the result is bit-identical (the fp32 array is rounded the same way on
     * its way to the matrix core), the loop moves half the bytes.  The fp32 arrays stay mandatory (ragged tiles, the bias
     * gradient's row sums).  C16: receives the bf16 rounding of every value stored to C, same indexing as C -- the twin
     * the next product reads.  skg_gemmx_f32 ignores A16 / B16 and honours C16.                                       */
    const uint16_t* A16;
    const uint16_t* B16;
    uint16_t* C16;
    /* split_k > 1 only, optional: device array of ceil(M / 128) * ceil(N / 128) counters, ZERO before the first launch that
     * names it.  With it the slices are reduced INSIDE the product launch: every slice's workgroup stores its partial tile to
     * split_ws and arrives at the tile's counter; the last arriver adds the slices in slice order (the same additions in the
     * same order as the reduce launch: results are bit-identical, whichever slice finishes last) and applies the epilogue.
     * No second launch; the counters are zero again when the launch ends (they wrap), so one array serves every later
     * launch of the same stream.  Launches that may run CONCURRENTLY need arrays of their own.  Honoured for products
     * whose C / bias / mask / split_ws admit 16-byte accesses (N % 4 == 0, ...) and M * N < 2^29; others keep the
     * second launch.                                                                                                  */
    uint32_t* split_ctr;
    /* Leading dimension (elements) of A16 / B16 where the twin is a PADDED copy with a pitch of its own -- the stride that
     * is not 1 (a_sm or a_sk; b_sn or b_sk) -- or 0: the fp32 array's.  For operands whose fp32 rows are not multiples of
     * 16 bytes (a [1024, 1074] weight: twin rows of 1088 elements).  Only the direct-to-LDS kernel (both twins present)
     * reads such a twin; not with blocked B.  A row-contiguous twin is read in 8-row pieces: its rows must be a multiple
     * of 8 or its pitch >= rows rounded up to 8 (what lies past the end is read, never used for a stored output).      */
    int64_t a16_ld, b16_ld;
} skg_gemmx_desc;
#define SKG_GEMMX_GROUP_MAX 8
int64_t skg_gemmx_ws_floats(const skg_gemmx_desc* desc_host);
int skg_gemmx_f32(const skg_gemmx_desc* descs_host, int n, void* stream);
/* The same products with both operands rounded to bf16 (RNE) on their way to the matrix core and fp32 accumulation
 * (v_mfma_f32_32x32x16_bf16); operands, results, bias gradient and epilogue stay fp32 in memory.  The dense layers of a
 * precision="bf16" training step (BASELINE config 3: the reference under torch.autocast(bfloat16)).                  */
int skg_gemmx_bf16(const skg_gemmx_desc* descs_host, int n, void* stream);
/* Statistics: launches since the last reset by main loop -- out3_host = {exact fp32, bf16 register-staged, bf16 direct-to-LDS
 * (both twins)}; reset != 0 zeroes them.  For tests and profiles (which products of a step reach the direct-to-LDS kernel);
 * nothing reads them back into a decision.                                                                            */
void skg_gemmx_path_counts(int64_t* out3_host, int reset);

/* ---------------------------------------------------------------------------------------------------------------
 * Non-GEMM stages of the fused TRAINING step (skg_train.hip): the forward pieces that keep what the backward needs and
 * the hand-written backward of the graph stages the reference leaves to autograd (HEAD:884-925, 966-973; losses
 * HEAD:153-205 with ops.py:159-211).  All feature rows are 1024 wide; every neighbourhood reduction runs per
 * destination row in index order (no atomics).                                                                      */
/* skg_graph_aggregate_f32 that also returns the softmax weights alpha / beta [sumG] (per human / per node senders).  */
int skg_graph_aggregate_train_f32(const float* dot_partial, int n_partial, int64_t partial_ld, float adj_bias,
                                  const skg_image_meta* meta, int n_active, const int32_t* hum_img,
                                  const int32_t* node_img, int sum_h, int sum_n, const float* T_os, const float* T_so,
                                  int64_t ldt, int cols, float* U, float* V, int64_t ldu, float* adj_out,
                                  float* alpha_out, float* beta_out, void* stream);
/* out[r] = X[r, :] . w  (adjacency Linear(1024 -> 1) without its bias, HEAD:897).                                   */
int skg_rowdot_f32(const float* X, int64_t ld, const float* w, int rows, int cols, float* out, void* stream);
/* xsum = a + b (node + message, HEAD:912-914, 923-925), y = LayerNorm(xsum) * gamma + beta, stats[r] = {mean, rstd}.  */
int skg_add_layernorm_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* gamma,
                          const float* beta, int rows, float eps, float* xsum, float* y, float* stats, void* stream);
/* LayerNorm backward from the saved xsum / stats: dx [rows, 1024], dgamma / dbeta [1024] (sums over the rows).
 * relu_src / dx_masked (optional, [rows, 1024]): dx_masked = dx where relu_src > 0 else 0 -- xsum = node + relu(message):
 * dx itself continues along the residual, dx_masked is the gradient in front of the message's ReLU.                    */
int skg_layernorm_bwd_f32(const float* dy, int64_t lddy, const float* x, const float* stats, const float* gamma,
                          int rows, float* dx, const float* relu_src, float* dx_masked, float* dgamma, float* dbeta,
                          void* stream);
/* Backward of t = relu(m * f), m = P[p_idx] + Q[q_idx] + mbias, f = F[f_idx] (MBF fc_1 * fc_2, HEAD:469-474): g = dt
 * (zero where t <= 0) is overwritten with dm = g * f;  dF[f_idx] = (or +=, accumulate) g * m.                          */
int skg_mul_bwd_f32(float* g, int64_t ldg, const float* F, const int32_t* f_idx, int64_t ldf, const float* P,
                    const int32_t* p_idx, int64_t ldp, const float* Q, const int32_t* q_idx, int64_t ldq,
                    const float* mbias, int rows, float* dF, int64_t lddf, int accumulate, void* stream);
/* Row sums per human / node / image (gradients of the gathered fc_1 tables).  mode 0: src = grid rows; mode 1: src =
 * kept pairs; mode 2: src = kept pairs, outH[meta[a].image] = sum over the pairs of active image a (rows of images
 * without pairs are not written).  outH [sumH | batch, 1024], outN [sumN, 1024]; either may be NULL.                   */
int skg_segment_sum_f32(const float* src, int64_t ld, const skg_image_meta* meta, int n_active, const int32_t* hum_img,
                        const int32_t* node_img, int sum_h, int sum_n, int mode, float* outH, float* outN,
                        int accumulate, void* stream);
/* Backward of the softmax-weighted aggregation (HEAD:907-922): from dU [sumH, 1024], dV [sumN, 1024] and the saved
 * Tos / Tso / alpha / beta:  dTos = alpha dU[h] (where Tos > 0), dTso = beta dV[o] (where Tso > 0), and the adjacency
 * gradient in two halves dadj_h + dadj_n [sumG] (softmax over a human's resp. a node's senders); da / db [sumG] scratch. */
int skg_aggregate_bwd_f32(const float* dU, const float* dV, const float* Tos, const float* Tso, const float* alpha,
                          const float* beta, const int32_t* grid_h, const int32_t* grid_o, int sum_g,
                          const skg_image_meta* meta, const int32_t* hum_img, const int32_t* node_img, int sum_h,
                          int sum_n, float* dTos, float* dTso, float* da, float* db, float* dadj_h, float* dadj_n,
                          void* stream);
/* dadj = dadj_h + dadj_n;  dWt[r, c] = dadj[r] * w[c] where Wt[r, c] > 0   (adjacency = relu(.) . w, HEAD:896-897).   */
int skg_adjacency_bwd_f32(const float* dadj_h, const float* dadj_n, const float* w, const float* Wt, int rows,
                          float* dadj, float* dWt, void* stream);
/* Backward of skg_concat_entity_f32: d_enc[e] = dX[hum_of[e]] + dX[sum_h + node_of[e]] (first 1024 columns; an index of
 * -1 = no reader), zeroed where enc[e] <= 0 (box_head's second ReLU, HEAD:639).                                       */
int skg_entity_rows_bwd_f32(const float* dX, int64_t ldx, const int32_t* hum_of, const int32_t* node_of, int sum_h,
                            int n_enc, const float* enc, float* d_enc, void* stream);
/* Both focal losses (HEAD:153-205, ops.py:159-211) forward + d/dlogits in one pass over what skg_postprocess_f32
 * emitted (training: prior_pow 1).  labels [sumP, K] from skg_associate_f32.  Outputs: cell_labels [L] (labels at the
 * scored cells), unary [sumP] (min(sum_v labels, 1)), partial [n_active, SKG_LOSS_CHUNKS, 4] = per image and workgroup
 * {sum of the cell losses, sum of the pair losses, number of positive cells, number of positive pairs} (to be added up;
 * the counts are the normalisers n_p of HEAD:162-165),
 * dlogits [sumP, ldl] (ZERO-FILLED by the caller): columns < K d(cell loss sum)/dlogit, column K d(pair loss sum).    */
#define SKG_LOSS_CHUNKS 64
int skg_hoi_loss_f32(const float* logits, int64_t ldl, int K, const skg_image_meta* meta, int n_active,
                     int64_t cells_total, const int64_t* index, const int64_t* pred, const float* scores,
                     const float* labels, float* cell_labels, float* unary, float* partial, float* dlogits,
                     void* stream);

/* Data parallel: the three per-rank normaliser counts {#non-zero labels among the scored cells, #pairs with a label, the same
 * again} (HEAD:167-172, 194-199, 223-228 before their all_reduce) from the PREPARED batch -- associated labels [sumP, K],
 * kept pairs, detections (prior of the human != 0 -> the verbs of the object's class are scored, HEAD:747-760) -- so the
 * all-reduce can start with the preparation and skg_loss_finish_f32 is called once, with norm_in.  counts [3] is
 * overwritten.  Equal to skg_loss_finish_f32's counts_out for the same batch.                                           */
int skg_count_positives_f32(const float* labels, int K, const float* det_scores, const int64_t* det_labels,
                            const skg_image_meta* meta, int n_active, const int64_t* x_keep, const int64_t* y_keep,
                            const int32_t* verb_off, const int32_t* verb_list, int num_obj_classes, float prior_pow,
                            float* counts, void* stream);

/* The three loss scalars from the partial sums the loss and sampling kernels leave (HEAD:162-177, 190-205, 228-234):
 * sums = column sums of partial [rows, 4]; n_p = norm_in [3] (data parallel: all_reduce_sum(counts) / world, HEAD:167-172)
 * or, when NULL, {sums[2], sums[3], sums[3]};  losses = {sums[0] / n_p[0], sums[1] / n_p[1], (sum(mpart[0..n_img)) /
 * max(m_pos, 1) + margin) / n_p[2]};  scale = {1 / n_p[0], 1 / n_p[1]} * grad_share -- the factor on the logit
 * gradients; grad_share = 1 / world makes the ranks' gradient SUM their mean (utils.py:202-205 averages), 1 otherwise;
 * counts_out (optional) = {sums[2], sums[3], sums[3]} -- a data-parallel caller without prepared counts
 * (skg_count_positives_f32) asks for them first (losses = NULL), all-reduces them and calls again.                      */
int skg_loss_finish_f32(const float* partial, int rows, const float* mpart, int n_img, int64_t m_pos, float margin,
                        float grad_share, const float* norm_in, float* losses, float* scale, float* counts_out,
                        void* stream);
/* out[r, c] = dl[r, c] * (c < K ? scale[0] * g0[0] : scale[1] * g1[0]): the gradient of the summed losses w.r.t. the
 * logits from skg_hoi_loss_f32's d(sum)/d(logits), the normalisers and the upstream gradients of the two focal terms.  */
int skg_scale_dlogits_f32(const float* dl, int64_t ld, int rows, int K, const float* scale, const float* g0,
                          const float* g1, float* out, void* stream);

/* TransH positive / negative sampling of the training step (HEAD:936-963) and the margin term (HEAD:207-235 as intended;
 * heads/MarginLoss.py:28-36): labels, scores [sumP, K]; pos_off [n_active + 1] = prefix of the positives per image and
 * max_pos_per_image their maximum (the host knows them from skg_associate_f32's npos); perm [sum m] = per image the first
 * m entries of randperm(#zero cells) (skg_transh_draw_train_f32); ws = skg_transh_sample_ws_ints(...) int32 of scratch.
 * pos_scores[i] = score of the i-th positive cell (row-major), pos_cells[i] its cell index (pair * K + verb, local to the
 * image), neg_scores[i] = score of the zero cell of rank perm[i]; partial[a] = sum_i max(pos_i - neg_i, -margin).     */
int64_t skg_transh_sample_ws_ints(int n_active, int max_pos_per_image);
int skg_transh_sample_f32(const float* labels, const float* scores, int K, const skg_image_meta* meta, int n_active,
                          const int32_t* pos_off, int max_pos_per_image, const int64_t* perm, float margin, int32_t* ws,
                          int32_t* pos_cells, float* pos_scores, float* neg_scores, float* partial, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Evaluation of the head's results on the device (utils.py:148-198): interaction = lut[object][verb]
 * (hicodet/hicodet.py:139-153; lut [n_obj, n_verb] int32, -1 = invalid), box-pair association per interaction class at
 * min_iou (pocket BoxPairAssociation: best min(IoU_h, IoU_o) ground-truth pair; per ground-truth pair the highest-scoring
 * match is the true positive, ties to the earlier detection), for a BATCH of images laid out like the head's packed
 * results: boxes_h / boxes_o / object [sumP], pair_off [n_images] (first pair of each image), index / pred / scores [L]
 * (index local to the image), cell_off [n_images + 1]; ground truth gt_h / gt_o [Ng, 4], gt_hoi [Ng], gt_off
 * [n_images + 1].  Outputs hoi_out [L], labels [L] (1 = true positive).  status (device int32, zero it first) receives
 * the largest per-image ground-truth count if one exceeds the kernel's 2048 (that image's labels are all zero then).  */
int skg_eval_associate_f32(const float* boxes_h, const float* boxes_o, const int64_t* object, const int32_t* pair_off,
                           const int64_t* index, const int64_t* pred, const float* scores, const int32_t* cell_off,
                           int n_images, const int32_t* lut, int n_obj, int n_verb, const float* gt_h, const float* gt_o,
                           const int64_t* gt_hoi, const int32_t* gt_off, float min_iou, int32_t* hoi_out, float* labels,
                           int32_t* status, void* stream);
/* pocket DetectionAPMeter(algorithm = '11P'), float64: labels_sorted = labels of all detections sorted by class
 * (ascending) and score (descending, stable); class_off [n_classes + 1]; num_gt [n_classes]; thresholds11 = the eleven
 * recall thresholds (torch.linspace(0, 1, 11, float64): passed in so that host and device compare identical doubles).
 * ap[c] = (1/11) sum_k max{precision_i : recall_i >= thresholds11[k]}; 0 without ground truth or detections.          */
int skg_eval_ap11_f64(const float* labels_sorted, const int64_t* class_off, const int64_t* num_gt, int n_classes,
                      const double* thresholds11, double* ap, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Order-independent 64-bit checksum of the live parameters (bit patterns weighted by position) over a table of
 * chunks: chunk c covers `count` fp32 words at `ptr` (16-byte aligned) whose first word has global index `first`.
 * The reference reads its nn.Linear / LayerNorm parameters afresh in every forward (HEAD:812-973, 410-411); the host
 * engine keeps re-laid copies and uses this sum -- enqueued ahead of the preprocess kernel and read back with its
 * counts -- to notice any in-place change, including writes that bypass autograd's version counters.  `out` (8-byte
 * aligned, device) receives SKG_CHECKSUM_PARTIALS partial sums; the checksum is their sum modulo 2^64 (formed by the
 * host: no atomics, fixed order).                                                                                   */
#define SKG_CHECKSUM_PARTIALS 1024
typedef struct {
    const void* ptr;
    uint32_t    count;     /* fp32 words in this chunk */
    uint32_t    first;     /* global word index of ptr[0] */
} skg_param_chunk;
int skg_param_checksum(const skg_param_chunk* chunks, int n_chunks, uint64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * AdamW over ALL parameters of the step in ONE launch (SURVEY 8(f)-4 "fused ... optimizer kernels"; the reference
 * uses torch.optim.AdamW, lr 1e-4, weight decay 1e-4, main:109-127).  chunks[c] = up to SKG_ADAMW_CHUNK consecutive
 * elements of one tensor: parameter, gradient, exp_avg, exp_avg_sq (fp32, the four at the same element offset).
 * Per element, torch's decoupled rule:  p *= 1 - lr * weight_decay;  m += (g - m)(1 - beta1);
 * v = beta2 v + (1 - beta2) g^2;  p -= (lr / bias1) * m / (sqrt(v) / sqrt(bias2) + eps),  bias_k = 1 - beta_k^step
 * (bias1, bias2 are passed in: the step count lives on the host; the scalar
 * factors are formed in double and rounded once, like torch's).  steps [n_steps] (optional): fp32 step counters of the
 * optimizer state (torch keeps one tensor per parameter), each incremented by one in the same launch.  HBM-bound: 28 bytes
 * per parameter.                                                                                                      */
#define SKG_ADAMW_CHUNK 16384
typedef struct {
    float* p; const float* g; float* m; float* v;
    uint32_t count, reserved;
} skg_adamw_chunk;
int skg_adamw_f32(const skg_adamw_chunk* chunks, int n_chunks, double lr, double beta1, double beta2, double eps,
                  double weight_decay, double bias1, double bias2, float* steps, int n_steps, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Native launch plan of the fused TRAINING step's dense part: GraphHead.forward in training mode (HEAD:769-993; the
 * classifier HEAD:410-411) and its backward -- what the reference leaves to eager PyTorch + autograd inside
 * `net(...)` / `loss.backward()` (utils.py:213-229) -- as ONE host call per phase.  The call enqueues the whole launch
 * sequence (every dense layer on skg_gemmx_f32 / skg_gemmx_bf16 / skg_gemm_f32, the graph stages of skg_train.hip) on
 * `stream`; nothing allocates or synchronises.
 *
 * Parameters and gradients live in two flat fp32 ARENAS with the same layout (skghoi_amd/train_fused.py, Stacked):
 * seg_off[s] = offset in floats of segment s.  Shapes: W1_m [1024, in_m] (in = 2048, 1024, 1024, Cf for m = attention_head,
 * obj_to_sub, sub_to_obj, attention_head_g: the 16 fc_1 branch weights as row blocks), b1_m [1024], W3_m [16][1024][64]
 * (branch-major fc_3), b3 [4][16][1024], W2 [4 x 1024, 1024] / b2 [4 x 1024] (fc_2 of the four MBFs back to back), CLS_W
 * [K + 1, 2048] (predictor rows, then the suppressor row) / CLS_B [K + 1], and the plain layers in their nn.Linear /
 * LayerNorm shapes (box_head BH1 [1024, x0_k], BH3; spatial_head SP0 [128, 46], SP2, SP4; fc_head FH / fc_tail FT
 * [1024, 1074]; adjacency ADJ [1, 1024]; norm_h NH, norm_o NO).
 * Row spaces (skg_image_meta): NA selected boxes, Mh humans, Mn nodes, Mg grid rows, Mp kept pairs of the A active images.
 * Index arrays are the ones skg_pairs_spatial_f32 and the host layout produce (skghoi_amd/layout.py).                  */
enum {
    SKG_SEG_W1_0 = 0, SKG_SEG_W1_1, SKG_SEG_W1_2, SKG_SEG_W1_3, SKG_SEG_B1_0, SKG_SEG_B1_1, SKG_SEG_B1_2, SKG_SEG_B1_3,
    SKG_SEG_W3_0, SKG_SEG_W3_1, SKG_SEG_W3_2, SKG_SEG_W3_3, SKG_SEG_B3, SKG_SEG_W2, SKG_SEG_B2, SKG_SEG_CLS_W, SKG_SEG_CLS_B,
    SKG_SEG_NH_W, SKG_SEG_NH_B, SKG_SEG_NO_W, SKG_SEG_NO_B, SKG_SEG_ADJ_W, SKG_SEG_ADJ_B, SKG_SEG_SP0_W, SKG_SEG_SP0_B,
    SKG_SEG_SP2_W, SKG_SEG_SP2_B, SKG_SEG_SP4_W, SKG_SEG_SP4_B, SKG_SEG_FH_W, SKG_SEG_FH_B, SKG_SEG_FT_W, SKG_SEG_FT_B,
    SKG_SEG_BH3_W, SKG_SEG_BH3_B, SKG_SEG_BH1_W, SKG_SEG_BH1_B, SKG_TRAIN_SEGS
};
#define SKG_TRAIN_BWD_STAGES 12
typedef struct skg_train_timer skg_train_timer;     /* optional per-launch timing of the plan's dense products, see below */
typedef struct {
    int32_t NA, Mg, Mp, Mh, Mn, A, K, Bf, Cf;   /* row-space sizes, verbs, feature maps [Bf, Cf] after the global pool   */
    int32_t x0_k;                               /* columns of the flattened pooled box features (256 * 7 * 7)            */
    int32_t bf16;                               /* 0: exact fp32 products; 1: operands rounded to bf16 (skg_gemmx_bf16)  */
    int32_t ld_logits;                          /* (K + 1) rounded up to a multiple of 4                                 */
    const float* params; float* grads;          /* the two arenas                                                        */
    int64_t seg_off[SKG_TRAIN_SEGS];
    const float* x0;                            /* [NA, x0_k] pooled box features (MultiScaleRoIAlign output, flattened)  */
    const float* gfeat;                         /* [Bf, Cf] globally pooled features['3'] (HEAD:811)                     */
    const float* sp48;                          /* [Mg, 48] spatial encodings (skg_pairs_spatial_f32)                    */
    const float* ent;                           /* [A, 80, 50] TransH entity tables of the step (part 1 only)            */
    const skg_image_meta* meta;
    const int32_t *enc_row_hn, *img_hn, *ent_row_hn, *hum_img, *node_img;      /* host layout (layout.pack_int_arrays) */
    const int32_t *grid_h, *grid_o, *grid_pair, *grid_img, *pair_grid, *pair_h, *pair_o;   /* skg_pairs_spatial_f32   */
    const int32_t *pair_img, *hum_of, *node_of;
    float* ws; int64_t ws_floats;               /* activations + scratch: skg_train_ws_floats(plan) floats               */
    float* pair_features;                       /* out [max(Mp, 1), 2048] (HEAD:966-973)                                 */
    float* logits;                              /* out [max(Mp, 1), ld_logits], ZERO-FILLED by the caller                */
    const float* dlogits;                       /* backward in: [max(Mp, 1), ld_logits]                                  */
    float* dx0; float* dgfeat;                  /* backward out, optional: gradients of x0 / gfeat                       */
    skg_train_timer* timer;                     /* NULL, or: HIP events around every skg_gemmx launch of the plan          */
    /* bf16 twins (bf16 = 1 only; all optional -- NULL: the products read and round the fp32 tensors).  ws16: twin of the
     * workspace, ws_floats elements, element i twins ws[i]: the products' outputs and the per-row kernels' outputs are also
     * stored rounded to bf16 (round to nearest even) there, and every product whose two operands have twins runs on the
     * direct-to-LDS kernel (skg_gemmx_t16_kernel).  params16: twin of the parameter arena (params_floats elements), written
     * by part 0 of the forward.  pf16: twin of pair_features.                                                          */
    uint16_t* ws16; uint16_t* params16; uint16_t* pf16;
    int64_t params_floats;
    /* optional: n_counters zeroed uint32 (zero before the first call that names them; every call leaves them zero): the
     * plan's split-K products are then reduced inside their own launch (skg_gemmx_desc.split_ctr) instead of by a second
     * launch each -- 16 launches less per batch-4 step.  Products whose tiles do not fit keep the reduce launch.       */
    uint32_t* counters; int64_t n_counters;
    /* split-K of the plan's products (developer knobs; 0 = the measured defaults: ~500 workgroups per product on the exact
     * fp32 loop, ~160 on the bf16 loops, at most 64 slices)                                                              */
    int32_t split_target, split_max;
    /* 1: two-branch schedule (bf16 step): the NODE chain of the step -- fc_head / fc_tail and the fc_1 projections in the
     * forward; from backward stage 6 on the fc_1 / fc_head / box_head gradients -- runs on a second stream beside the SPATIAL
     * chain on the grid rows (fc_2 of all four MBFs, the spatial head), which it shares nothing with; every call returns
     * with the caller's stream ordered behind both.  The second stream and its two events belong to a CONTEXT: honoured by
     * skg_ctx_train_forward_f32 and by backwards issued through a context; skg_train_ws_floats then asks for a second
     * split-K scratch region.  0: one stream (and always so for the context-free entry points).                        */
    int32_t two_branch, reserved2;
} skg_train_plan;
/* Floats of workspace the plan needs (activations kept for the backward, backward temporaries, split-K scratch);
 * < 0: rejected plan.  Only the sizes, bf16 and params (non-null) are read.                                              */
int64_t skg_train_ws_floats(const skg_train_plan* plan_host);
/* part 0: everything that needs neither the TransH tables nor the label counts (box_head, fc_1 of the global branch, two
 * spatial layers) -- enqueue it before the step's host synchronisation; part 1: the rest, down to the logits.           */
int skg_train_forward_f32(const skg_train_plan* plan_host, int part, void* stream);
/* Backward stages [first_stage, last_stage) of SKG_TRAIN_BWD_STAGES, from dlogits to the gradient arena (and dx0 /
 * dgfeat).  After stage s the arena prefix of the segments whose gradients that stage completes is final (the order of
 * the segments in the arena follows the stages: read-out layers first, box_head last), so a data-parallel caller can
 * exchange the arena chunk by chunk between calls.                                                                      */
int skg_train_backward_f32(const skg_train_plan* plan_host, int first_stage, int last_stage, void* stream);
/* ---- contexts.  Everything of the library that outlives a call -- today: the worker thread that issues a backward, its
 * one job slot and the job's progress -- belongs to a context the CALLER creates: one per trainer / device / host thread,
 * so that two hosts in one process (two trainers, two devices) never share state.  Every skg_ctx_* entry point accepts
 * NULL for the process's default context, which is what the context-free forms below use.  A context must be destroyed
 * by the process that created it (fork(): the child starts with a fresh default context and must not touch others).   */
typedef struct skg_context skg_context;
skg_context* skg_context_create(void);
void skg_context_destroy(skg_context* ctx);       /* waits for a job in flight, stops and joins the worker */
/* The context's tuning switches (skg_tuning above): set copies *t (SKG_E_ARG for an unknown small_mode), get reads them back.
 * skg_ctx_make_current makes `ctx` (NULL: none -- the defaults) the CALLING THREAD's current context and returns the previous
 * one: the skg_gemm_* calls of that thread then read its switches.  A context must not be destroyed while it is current on
 * another thread.                                                                                                        */
int skg_ctx_set_tuning(skg_context* ctx, const skg_tuning* t);
int skg_ctx_get_tuning(skg_context* ctx, skg_tuning* out);
skg_context* skg_ctx_make_current(skg_context* ctx);
/* skg_train_backward_f32 issued from the context's worker thread (one job at a time per context; the plan is copied, the
 * worker selects the caller's current device): returns at once -- 0, SKG_E_* for a plan / stage range / workspace that
 * skg_train_backward_f32 would reject (checked HERE, before the job is queued), SKG_E_LIMIT while a job is pending.
 * stage_mask: bit s set = the worker records the context's OWN event for stage s on `stream` right behind that stage (created
 * without timing and with DEVICE-scope release: an event of the default, system-scope kind writes the L2 back and invalidates
 * it at every record -- a dozen of those inside a backward cost 0.09 ms: backward 0.95-0.97 -> 0.86-0.88 ms); bit 31 set = the
 * chunks behind these stages go to OTHER GPUs (a process group of more than one rank): the events are then of the default,
 * system-scope kind -- a peer or a registered-buffer transport may read the chunk directly, so the stage's writes must have
 * left this GPU's L2 (skg_ctx_train_backward_exchange_f32 decides this itself from its communicator's world size).
 * skg_ctx_train_backward_stage_wait(ctx, s)
 * blocks the HOST until stage s has been enqueued (and its event, if any, recorded) and returns 0, or the job's error if it
 * ended before reaching s; skg_ctx_stream_wait_stage(ctx, s, other_stream) then makes `other_stream` wait for stage s on the
 * DEVICE: a data-parallel caller orders the collective of the gradient-arena prefix stage s completed behind it, while the
 * worker keeps issuing the later stages (reference: DistributedDataParallel's bucket all-reduces behind autograd,
 * utils.py:202-205).  stage_events_host: NULL, or last_stage - first_stage caller-owned hipEvent_t handles (entries may be
 * NULL) recorded behind their stages as well (measurement: a timing event behind the last stage).
 * skg_ctx_train_backward_join blocks until every launch of the job has been enqueued and returns what
 * skg_train_backward_f32 returned (0 when no job was pending).  Between submit and join the caller may enqueue work on
 * OTHER streams and do host work; it must not enqueue anything ordered after the gradients on `stream`, nor release a
 * buffer the plan names.  For step loops bound by their own host thread (a Python trainer at batch 4: ~0.2 ms of launch
 * calls).                                                                                                              */
int skg_ctx_train_backward_async_f32(skg_context* ctx, const skg_train_plan* plan_host, int first_stage, int last_stage,
                                     void* stream, void* const* stage_events_host, uint32_t stage_mask);
/* skg_train_forward_f32 with the context's second branch at hand (plan.two_branch; the calling thread issues both branches). */
int skg_ctx_train_forward_f32(skg_context* ctx, const skg_train_plan* plan_host, int part, void* stream);
int skg_ctx_stream_wait_stage(skg_context* ctx, int stage, void* waiting_stream);
int skg_ctx_train_backward_stage_wait(skg_context* ctx, int stage);
int skg_ctx_train_backward_join(skg_context* ctx);
/* Non-blocking: backward stages of the context's current / last job issued so far, | 0x100 while the job is pending.  For
 * failure records (a rank that does not come back from a step: how far its backward got).                              */
int skg_ctx_train_backward_progress(skg_context* ctx);
/* ---- data parallel: the gradient exchange inside the worker's backward (replaces utils.py:202-205, DistributedDataParallel's
 * bucketed NCCL all-reduce, for the head's gradient arena).  skg_comm is an RCCL communicator OF THIS LIBRARY -- RCCL is
 * bound at run time (skg_comm_load: dlopen of `librccl_path`, else of the librccl the process already has; SKG_E_UNSUPPORTED
 * when there is none) -- with a stream of its own.  Bootstrap like any NCCL program: rank 0 draws skg_comm_unique_id
 * (SKG_COMM_ID_BYTES bytes), the caller carries it to every rank (its process group's broadcast), every rank calls
 * skg_comm_create(id, rank, world) with ITS device current (collective: returns on all ranks or on none).
 * skg_ctx_train_backward_exchange_f32 = skg_ctx_train_backward_async_f32 whose worker, behind every stage ex->stage[i], orders
 * the communicator's stream behind that stage's device-scope event and all-reduces (sum, in place, fp32) arena[ex->end[i-1],
 * ex->end[i]) there, and behind the last stage orders `stream` behind the last collective: at the join the gradients on
 * `stream` are the ranks' SUM (the caller folded 1 / world into its logit gradients).  Chunks in ascending stage order.
 * skg_comm_all_reduce_chunks_f32: the same chunk sequence behind `stream`'s current tail, for a rank whose step did not take
 * the staged route (its peers' collectives must be met one for one).  skg_comm_exposed_ms: device time between
 * `after_event` (a recorded timing event, e.g. behind the backward's last launch) and the end of the step's last collective,
 * clipped at 0 (blocks until both have happened).  skg_comm_all_reduce_begin_f32 / _end: ONE further collective outside
 * the chunk sequence (the loss normalisers of the next batch, HEAD:167-172) -- begin orders the communicator's stream behind
 * `stream`'s tail and all-reduces p[0, n) there, end orders the stream it is given behind that collective; one pair
 * outstanding at a time.  Every rank must issue the communicator's collectives in the same order: callers keep them on
 * one host thread at a time (the worker between submit and join, the submitting thread otherwise).                      */
typedef struct skg_comm skg_comm;
#define SKG_COMM_ID_BYTES 128
typedef struct skg_exchange {
    skg_comm* comm;                            /* NULL: no collectives (single process; then `adamw` must be set) */
    float* arena;                              /* the gradient arena (plan.grads) */
    int32_t n_chunks;
    int32_t stage[SKG_TRAIN_BWD_STAGES];       /* stage that completes chunk i (ascending) */
    int64_t end[SKG_TRAIN_BWD_STAGES];         /* chunk i = arena[end[i - 1], end[i]) floats */
    /* The optimizer inside the backward (optional; adamw = NULL: none).  The arena is laid out in the order in which the
     * backward finishes gradients AND last reads the weights they belong to (a layer's input gradient and weight gradient
     * leave the same stage), so behind stage[i] -- behind chunk i's all-reduce when there is a communicator -- the parameters
     * of chunk i can be updated while the later stages still run: AdamW table entries [adamw_first[i], adamw_first[i + 1])
     * (the skg_adamw_f32 chunk table sorted by gradient address) are launched there, on the communicator's stream / on a
     * stream of the context, with the scalar factors of skg_adamw_f32; `stream` is ordered behind the last of them.  The
     * update of 29.6 M parameters moves 0.83 GB, 0.13 ms at HBM rate.  Measured at batch 4 on one MI355X the overlap does
     * NOT pay: the stream beside backward stages 7-11 slows them by more than it takes off the tail;
     * skghoi_amd's trainer keeps it opt-in.  adamw_steps [adamw_n_steps]: the state's step counters, bumped once.       */
    const skg_adamw_chunk* adamw;
    int32_t adamw_first[SKG_TRAIN_BWD_STAGES + 1];
    int32_t adamw_n_steps;
    float* adamw_steps;
    double lr, beta1, beta2, eps, weight_decay, bias1, bias2;
} skg_exchange;
int skg_sizeof_exchange(void);                                      /* sizeof(skg_exchange): for bindings that mirror it */
int skg_comm_load(const char* librccl_path);
int skg_comm_unique_id(void* id_out);
int skg_comm_create(const void* id, int rank, int world, skg_comm** out);
void skg_comm_destroy(skg_comm* comm);
/* ncclCommAbort: ends the communicator without its peers' cooperation, so that ranks already waiting in a collective this
 * rank will never join get an error instead of a hang.  The backward's worker calls it when a stage, an event or a collective
 * fails between two chunks; callers may after a failed step.  The object stays valid and dead (skg_comm_dead: every later
 * collective returns SKG_E_COMM) until skg_comm_destroy.                                                               */
int skg_comm_abort(skg_comm* comm);
int skg_comm_dead(const skg_comm* comm);
int skg_comm_world(const skg_comm* comm);
int skg_comm_rank(const skg_comm* comm);
int64_t skg_comm_collectives(const skg_comm* comm);                /* all-reduces issued since creation */
const char* skg_comm_last_error(void);                             /* text of this thread's last SKG_E_COMM / _UNSUPPORTED */
int skg_comm_all_reduce_chunks_f32(skg_comm* comm, float* arena, const int64_t* ends_host, int n_chunks, void* stream);
int skg_comm_exposed_ms(skg_comm* comm, void* after_event, float* ms_out);
int skg_comm_all_reduce_begin_f32(skg_comm* comm, float* p, int64_t n, void* stream);
int skg_comm_all_reduce_end(skg_comm* comm, void* stream);
int skg_ctx_train_backward_exchange_f32(skg_context* ctx, const skg_train_plan* plan_host, int first_stage, int last_stage,
                                        void* stream, void* const* stage_events_host, const skg_exchange* ex_host);
/* The same on the default context, without stage events. */
int skg_train_backward_async_f32(const skg_train_plan* plan_host, int first_stage, int last_stage, void* stream);
int skg_train_backward_join(void);
/* Measurement aid (bench.py's training roofline): a timer named by plan.timer makes every skg_gemmx launch the plan issues
 * (main kernel + its split-K reduce) sit between two HIP events on the plan's stream -- a few microseconds each, so it is
 * switched on for a handful of untimed steps only.  `capacity` = launches it can hold between two reads; further launches
 * go untimed.  skg_train_timer_read waits for the recorded launches and returns out3_host = {summed milliseconds, launches,
 * 2 M N K of those launches}, then starts over.  Owned by the caller: no global state.                               */
skg_train_timer* skg_train_timer_create(int capacity);
void skg_train_timer_destroy(skg_train_timer* timer);
int skg_train_timer_read(skg_train_timer* timer, double* out3_host);
/* dst[i] = bf16(src[i]), round to nearest even, n a multiple of 4: the twin of a whole fp32 buffer (the parameter arena). */
int skg_twin_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);
/* Arithmetic of the plan: 2 M N K summed over every dense product it issues (which = 0 forward, 1 backward, 2 both; the
 * backward is counted with dx0 / dgfeat requested).  For roofline records.                                              */
double skg_train_flops(const skg_train_plan* plan_host, int which);
/* Offset (floats) of a saved activation inside ws, for tests: 0 enc, 1 h_node, 2 node, 3 adjacency logits, 4 raw fc_2.   */
int64_t skg_train_ws_offset(const skg_train_plan* plan_host, int which);

#ifdef __cplusplus
}
#endif
#endif /* SKGHOI_H */
