// mirx_kernels.h -- launch wrappers implemented in the k_*.hip files (internal, not ABI).
#pragma once
#include "mirx_common.h"

namespace mirx {

// ---- k_prep.hip -------------------------------------------------------------------------
// Rows [n, dim] fp32 (device) -> padded fp32 master, bf16 copy, per-row bias (metric 1:
// -0.5*||g||^2, metric 0: 0) and an atomic running maximum of ||g|| (upper bound, fp32 bits).
hipError_t launch_ingest(const float *src, int64_t n, int dim, int dimp, float *g32,
                         uint16_t *g16, float *gbias, unsigned *gnorm_max_bits, int metric,
                         hipStream_t st);
// Queries [nq, dim] fp32 -> padded fp32 [nq_pad, dimp], bf16 [nq_pad, dimp], upper bound of
// ||q|| per query; rows nq..nq_pad-1 are zero.
hipError_t launch_prep_queries(const float *q, int64_t nq, int64_t nq_pad, int dim, int dimp,
                               float *q32p, uint16_t *q16, float *qnorm, hipStream_t st);
hipError_t launch_l2_normalize(float *x, int64_t n, int dim, hipStream_t st);
hipError_t launch_bn_relu_gap_l2norm(const float *x, const float *scale, const float *shift,
                                     int64_t n, int c, int hw, int normalize, float *y,
                                     hipStream_t st);
hipError_t launch_bn_relu_nchw(const float *x, int64_t x_batch_stride, const float *scale,
                               const float *shift, int64_t n, int c, int hw, float *y, hipStream_t st);
hipError_t launch_bn_relu_avgpool2(const float *x, int64_t x_batch_stride, const float *scale,
                                   const float *shift, int64_t n, int c, int h, int w, float *y,
                                   int64_t x_plane_stride, hipStream_t st, int64_t y_batch_stride = 0);
hipError_t launch_stem(const float *x, const float *w, const float *scale, const float *shift,
                       int64_t n, int h, int wd, float *y, hipStream_t st);
// k_stem_s3.hip: the same stem on three-term bf16 MFMAs; w3 = pre-split weights [2][11][3][32][16] bf16
// y_bs = output batch stride in floats (64 * (h/4) * (wd/4) for a packed tensor); out_range: range slots or null
hipError_t launch_stem_s3(const float *x, const uint16_t *w3, const float *scale, const float *shift, int64_t n, int h,
                          int wd, float *y, int64_t y_bs, float *out_range, hipStream_t st);

// k_stem_h2.hip: the stem on two fp16 terms (w2 = [2][11][2][32][16] fp16, per-output-channel scales in oscale[64];
// in_range = range slots of the input images) and the pass that fills such slots with the largest |x|
hipError_t launch_stem_h2(const float *x, const uint16_t *w2, const float *oscale, const float *scale, const float *shift,
                          int64_t n, int h, int wd, float *y, int64_t y_bs, const float *in_range, float *out_range,
                          hipStream_t st);
hipError_t launch_range_absmax(const float *x, int64_t per_image, int64_t n, float *row, hipStream_t st);
// raw 8-bit images, ToTensor + Normalize applied while staging (bit-identical to the fp32 path on the normalised tensor)
hipError_t launch_stem_h2_u8(const uint8_t *x, const float *mean, const float *stdv, const uint16_t *w2, const float *oscale,
                             const float *scale, const float *shift, int64_t n, int h, int wd, float *y, int64_t y_bs,
                             const float *in_range, float *out_range, hipStream_t st);
hipError_t launch_range_absmax_u8(const uint8_t *x, int64_t hw, int64_t n, const float *mean, const float *stdv, float *row,
                                  hipStream_t st);

// ---- k_conv1x1.hip ----------------------------------------------------------------------
hipError_t launch_conv1x1(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                          const float *wt, const float *bias, int64_t n, int hw, int cout, int relu_out, float *y,
                          hipStream_t st);

// ---- k_convnext.hip ---------------------------------------------------------------------
hipError_t launch_dwconv7_nhwc(const float *x, const float *wt, const float *bias, int64_t n, int c, int h, int wd, float *y,
                               hipStream_t st);
hipError_t launch_dwconv7(const float *x, const float *w, const float *bias, int64_t n, int c, int h, int wd,
                          float *y, hipStream_t st);
// gx[b][c] = sqrt(sum over the hw positions of x[b][p][c]^2)       (x = [n][hw][c], channels last)
hipError_t launch_grn_norm(const float *x, int64_t n, int hw, int c, float *gx, hipStream_t st);
hipError_t launch_grn_scale(const float *gx, const float *weight, int64_t n, int c, float eps, float *scale, float *smax,
                            hipStream_t st);
// x[b][p][c] = x[b][p][c] * scale[b][c] + shift[c], in place

// ---- k_exact.hip ------------------------------------------------------------------------
// out[i*ld + j] = fp64 ranking score of query qlist[i] (or i when qlist == null) vs row j.
hipError_t launch_scores_f64(const float *q32p, const int32_t *qlist, int nq, const float *g32,
                             int64_t n, int dimp, int metric, double *out, int64_t ld,
                             hipStream_t st);
// Per query row of `scores` [nq, ld]: top-k by (score desc, id asc), skipping exclude ids.
// Writes out_f64 / out_ids / out_val at row qlist[i] (or i) of the [*, k] outputs.
hipError_t launch_row_topk(const double *scores, int64_t ld, int64_t n, const int64_t *ids,
                           const int32_t *qlist, int nq, const int64_t *exclude, int k,
                           int metric, double *out_f64, int64_t *out_ids, float *out_val,
                           hipStream_t st);
// Full sort of every row: hits [nq, np2] (np2 = power of two >= n) built from scores, sorted.
hipError_t launch_rank_rows(const double *scores, int64_t ld, int64_t n, const int64_t *ids,
                            const int64_t *exclude, int nq, Hit *work, int64_t np2, int metric,
                            int64_t *out_ids, float *out_val, hipStream_t st);
hipError_t launch_topk_merge(const double *in_scores, const int64_t *in_ids, int nshard,
                             int64_t nq, int k, int metric, double *out_f64, float *out_val,
                             int64_t *out_ids, hipStream_t st);

// ---- k_conv1x1_s3.hip --------------------------------------------------------------------
hipError_t launch_conv1x1_s3(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                             const uint16_t *w3, const float *bias, int64_t n, int hw, int cout, int relu_out,
                             float *y, int64_t ybs, hipStream_t st);

// ---- k_conv1x1_h2.hip: the same convolution on two fp16 terms per operand; ranges travel in 64 "range slots" ----
hipError_t launch_conv1x1_h2(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                             const uint16_t *w2, const float *oscale, const float *bias, int64_t n, int hw, int cout,
                             int relu_out, float *y, int64_t ybs, const float *in_amax, float in_ks, float in_kb,
                             float *out_amax, float y_ks, float y_kb, float *y_inv_out, int64_t xps, int64_t yps,
                             hipStream_t st);

void set_conv1x1_small_max_wg(int v);      // mirx_set_tuning(MIRX_TUNE_CONV1X1_SMALL_MAX_WG)
// k_conv1x1_h2s.hip: the same contract for small launches (one wave per 32 x 32 tile, no LDS); bit-identical results
hipError_t launch_conv1x1_h2_small(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                                   const uint16_t *w2, const float *oscale, const float *bias, int64_t n, int hw, int cout,
                                   int relu_out, float *y, int64_t ybs, const float *in_amax, float in_ks, float in_kb,
                                   float *out_amax, float y_ks, float y_kb, float *y_inv_out, int64_t xps, int64_t yps,
                                   hipStream_t st);

// ---- k_linear_h2.hip: the token-major Linear on two fp16 terms per operand (3 MFMAs per product) ----
hipError_t launch_linear_h2(const float *x, int64_t m, int k, const uint16_t *w2, const float *bias, int n, int act,
                            const float *res, const float *gamma, float x_scale, float out_scale, float *y,
                            int tokens_per_image, const float *xb_dev, hipStream_t st, bool rows_out = false,
                            float *sq_out = nullptr);
hipError_t launch_grn_norm_partials(const float *part, int tpi, int64_t n_img, int c, float *gx, hipStream_t st);

// ---- k_linear_s3.hip ---------------------------------------------------------------------
// tokens_per_image == 0: y / res are [m][n]; > 0: token t is pixel t % tpi of image t / tpi and y / res are NCHW
hipError_t launch_linear_s3(const float *x, int64_t m, int k, const uint16_t *w3, const float *bias, int n, int act,
                            const float *res, const float *gamma, float *y, int tokens_per_image, hipStream_t st);

// ---- k_conv3x3.hip ----------------------------------------------------------------------
hipError_t launch_conv3x3_wino(const float *x, const float *u, int64_t n, int side, float *out, int64_t out_bs,
                               float *out_range, hipStream_t st);
// ---- k_conv3x3_s3.hip: the same convolution on three-term bf16 MFMAs (side 56 / 28 / 14) ----------
hipError_t launch_conv3x3_wino_s3(const float *x, const uint16_t *u3, int64_t n, int side, float *out, int64_t out_bs,
                                  hipStream_t st);
// ---- k_conv3x3_d3.hip: the same convolution as a direct implicit GEMM on three-term bf16 MFMAs ------
// w3 = [8 stages][9 taps][3 terms][32 oc][16 c] bf16
hipError_t launch_conv3x3_d3(const float *x, const uint16_t *w3, int64_t n, int side, float *out, int64_t out_bs,
                             hipStream_t st);

// ---- k_conv3x3_d2h.hip: the direct implicit GEMM on two fp16 terms per operand; ranges travel in range slots ------
// w2 = [8 stages][9 taps][2 terms][32 oc][16 c] fp16 (scaled per output channel), oscale = [32] fp32

// ---- k_dense_fused.hip: a whole dense layer of the 14 x 14 / 7 x 7 maps, bottleneck resident in LDS --------------------
hipError_t launch_dense_fused(float *buf, int64_t bs, int cin, const float *scale, const float *shift, const uint16_t *w2,
                              const float *oscale, const float *bias, const uint16_t *w3, const float *c3osc, int64_t n,
                              int side, float *range_row, float in_ks, float in_kb, float y_ks, float y_kb, hipStream_t st);

// ---- k_norm.hip: LayerNorm over rows, patchify (+ LayerNorm2d), attention for short query sets ------------------
hipError_t launch_layernorm_rows(const float *x, int64_t m, int c, const float *gamma, const float *beta, float eps,
                                 float *y, int tokens_per_image, hipStream_t st, void *yt = nullptr, float scale = 1.f,
                                 int patch_w = 0, int patch_h = 0);
// k_linear_t2.hip: the DMA-fed two-fp16-term Linear on "terms rows" and the fp32 -> terms conversion
hipError_t launch_rows_to_terms(const float *x, int64_t m, int k, int64_t ldx, float scale, void *xt, hipStream_t st);
hipError_t launch_linear_t2(const void *xt, int64_t m, int k, const void *wt, const float *bias, int n, int act,
                            const float *res, const float *gamma, float out_scale, float *y, void *yt, float y_scale,
                            void *workspace, size_t workspace_bytes, hipStream_t st);
size_t linear_t2_workspace_bytes(int64_t m, int k, int n);
hipError_t launch_patchify(const float *x, int64_t n, int c, int h, int w, int p, const float *gamma, const float *beta,
                           float eps, float *out, int kpad, hipStream_t st);
hipError_t launch_attention_small(const float *q, int64_t q_rs, const float *k, const float *v, int64_t kv_rs,
                                  const uint8_t *key_mask, int64_t batch, int heads, int head_dim, int nq, int nk,
                                  float scale, float *out, hipStream_t st);

// k_conv3x3_d2h.hip, second kernel: the input already split into fp16 terms by launch_conv1x1_h2(.., y_inv_out != null)
// pool_out != nullptr (sides 56 / 28 / 14, strip kernel only -- ask conv3x3_takes_small first): the launch also writes
// avgpool2x2(relu(v * pool_sc[oc] + pool_sh[oc])) of its 32 output channels to pool_out[image * pool_bs + oc * (side/2)^2 + ..]
hipError_t launch_conv3x3_d2p(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, int side, float *out,
                              int64_t out_bs, const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st,
                              const float *pool_sc = nullptr, const float *pool_sh = nullptr, float *pool_out = nullptr,
                              int64_t pool_bs = 0);
bool conv3x3_takes_small(int64_t n, int side);

// k_conv3x3_d2s.hip: the same contract for small launches (one wave per 32 output pixels, no LDS); bit-identical results
hipError_t launch_conv3x3_d2s(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, int side, float *out,
                              int64_t out_bs, const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st);
void set_conv3x3_small_max_wg(int v);      // mirx_set_tuning(MIRX_TUNE_CONV3X3_SMALL_MAX_WG)
int conv3x3_small_max_wg();

// ---- k_attention.hip --------------------------------------------------------------------
hipError_t launch_attention(const float *qkv, int64_t batch, int n, int heads, int head_dim, float scale, float *out,
                            hipStream_t st);
// ---- k_attention_s3.hip: the same attention on three-term bf16 MFMAs (head_dim 64) ------------------
hipError_t launch_attention_s3(const float *qkv, int64_t batch, int n, int heads, int head_dim, float scale, float *out,
                               hipStream_t st);
// ---- k_attention_h2.hip: the same attention on two fp16 terms per operand (head_dim 64, caller-supplied bounds) --
hipError_t launch_attention_h2(const float *qkv, int64_t batch, int n, int heads, int head_dim, float scale,
                               float qk_bound, float v_bound, float *out, hipStream_t st, void *out_terms = nullptr, float terms_scale = 1.f);

// ---- k_metrics.hip ----------------------------------------------------------------------
constexpr int MIRX_MAX_KAPPAS = 8;
struct RankMetricsArgs {
    const int64_t *ranks;          // [nq, row_stride >= n] gallery row ids, best first
    int64_t nq, n, row_stride;
    const int64_t *gallery_labels; // [n_labels] class id, or multi-hot bit mask (rel_kind 1)
    int64_t n_labels;
    const int64_t *query_labels;   // [nq]
    const int64_t *query_ids;      // [nq] id that is never relevant for the query, or null
    int drop_self;                 // 1: that entry is also taken out of the list (later ranks move up)
    double jaccard_threshold;
    int32_t kappas[MIRX_MAX_KAPPAS];
    int nk;
    double *out_ap;                // [nq]
    int64_t *out_cnt;              // [nq, nk]
    int64_t *out_nrel, *out_maxpos;
};
hipError_t launch_rank_metrics(const RankMetricsArgs &a, int rel_kind, int ap_kind, hipStream_t st);

// ---- k_gemm.hip -------------------------------------------------------------------------
struct GemmArgs {
    const uint16_t *g16;   // gallery bf16 [rows, dimp]
    const uint16_t *q16;   // queries bf16 [nq_pad, dimp]
    const float *gbias;    // per-row bias (metric 1) or null
    int64_t n_rows;        // valid gallery rows (filter mode) / sampled rows (group-max mode)
    int64_t row_stride;    // gallery row step between consecutive tile rows (1 = dense)
    int64_t nq_pad;        // multiple of the query tile
    int dimp;
    // filter mode: every (query, producer wave) owns a private region of CAND_SLOTS entries, so
    // the epilogue appends with plain stores; rows beyond a region go to a shared overflow list
    const float *tau;      // [nq_pad]
    int regions, slots;    // producer regions per query and slots per region (from gemm_plan)
    int nph;               // phases per query tile (filled by the launcher)
    int *region_cnt;       // [nq_pad, regions]  rows that passed, per region (zeroed per search)
    Cand *cand;            // [nq_pad, regions, slots]
    int *ovf_cnt;          // [nq_pad]           (zeroed per search)
    Cand *ovf;             // [nq_pad, CAND_OVF]
    void *spill;           // gemm_spill_bytes(): per-wave overflow area of the 256-query kernel's candidate rings
    // group-max mode
    float *groupmax;       // [nq_pad, ngroups]
    int ngroups;
};
int gemm_query_tile(int64_t nq);                 // query-tile width chosen for nq (64/128/256)
// Producer regions per query and slots per region of the filter GEMM for this problem.
void gemm_plan(int64_t n_rows, int64_t nq_pad, int bn, int *regions, int *slots);
size_t gemm_spill_bytes();                       // workspace the filter GEMM needs behind GemmArgs::spill
int gemm_groups_per_tile(int bn);                // group-max groups per 256-row gallery tile
hipError_t launch_gemm_filter(const GemmArgs &a, int bn, hipStream_t st);
hipError_t launch_gemm_groupmax(const GemmArgs &a, int bn, hipStream_t st);

// ---- k_finalize.hip ---------------------------------------------------------------------
struct FinalizeArgs {
    const float *q32p;       // [nq, dimp]
    const float *qnorm;      // [nq]
    const float *g32;        // [rows, dimp]
    const int64_t *ids;      // [rows]
    const unsigned *gnorm_max_bits;
    const float *tau;
    int regions, slots;
    const int *region_cnt;   // [nq, regions]
    const Cand *cand;        // [nq, regions, slots]
    const int *ovf_cnt;      // [nq]
    const Cand *ovf;         // [nq, CAND_OVF]
    const int64_t *exclude;  // or null
    int64_t n_rows;
    int dimp, k, metric, nq;
    double *out_f64;         // [nq, k]
    int64_t *out_ids;        // [nq, k]
    float *out_val;          // [nq, k]
    int32_t *fail_list;      // [nq]   queries for the exact scan (original query numbers)
    int *fail_count;         // [2]    [0] exact list length, [1] retry list length
    // second chance (pass 1 only; null in the retry pass): a query whose guard failed only because
    // tau was too high is re-filtered with tau2 = kth(s~) - 2*eps, which is complete by construction
    int32_t *retry_list;     // [nq]
    float *tau2;             // [nq]
    // retry pass: slot i of the candidate buffers / tau belongs to original query qmap[i]
    const int32_t *qmap;     // or null (identity)
    mirx_search_stats *stats;  // device copy
};
// q16r[i] = q16[list[i]] for i < n, zero rows and tau = +inf up to n_pad (retry pass input).
hipError_t launch_gather_queries(const uint16_t *q16, const float *tau2, const int32_t *list, int n,
                                 int64_t n_pad, int dimp, uint16_t *q16r, float *taur, hipStream_t st);
hipError_t launch_select_tau(const float *groupmax, int ngroups, int64_t nq, int64_t nq_pad,
                             int rank_j, float *tau, hipStream_t st);
hipError_t launch_finalize(const FinalizeArgs &a, hipStream_t st);

}  // namespace mirx
