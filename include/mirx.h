/*
 * mirx.h -- C ABI of libmirx.so: MI355X-native exhaustive retrieval + embedding-head kernels.
 *
 * This is the drop-in boundary underneath the reference's Python protocol.  The reference
 * (100 % Python, /root/reference) has no FFI of its own; each entry point below names the
 * reference call it stands in for, and INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference adds at that call site.
 *
 * Conventions
 *   - plain C types only; every pointer is a raw address, no torch/HIP C++ types;
 *   - return 0 on success, a negative MIRX_E* code on failure; mirx_last_error() gives the
 *     text for the calling thread; nothing throws across the ABI;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work of a
 *     call is enqueued on it and the call returns without waiting unless stated;
 *   - "device pointer" = memory of the index's device (hipMalloc / torch.cuda tensor
 *     data_ptr()); "host pointer" = ordinary host memory.  mirx_index_add accepts either
 *     (it asks the runtime); search inputs/outputs must be device pointers;
 *   - the caller owns every buffer it passes; the index owns its device copies;
 *   - an index is not re-entrant: add / search / destroy on one index must not overlap.
 *
 * Semantics of a search (pinned by oracle/search_ref.c):
 *   score   = fp64 accumulation of the fp32 inputs in the fixed "lane tree" order
 *   ranking = higher score first; equal scores -> lower id first
 *   metric MIRX_METRIC_IP      score = <q, g>               (cosine on unit rows)
 *   metric MIRX_METRIC_NEG_L2  ranks by -||q-g||^2, reports -||q-g||_2
 * The bf16 MFMA pass only proposes candidates; every returned hit is re-scored in fp64 and
 * a completeness guard proves that no better row was left out (otherwise the query falls
 * through to the exact scan), so results never depend on which tier answered.
 */
#ifndef MIRX_H
#define MIRX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIRX_VERSION 305

#define MIRX_OK 0
#define MIRX_EINVAL (-1)   /* bad argument (null pointer, dim mismatch, k out of range) */
#define MIRX_ENOMEM (-2)   /* device or host allocation failed */
#define MIRX_EHIP (-3)     /* a HIP runtime call failed; text in mirx_last_error() */
#define MIRX_ESTATE (-4)   /* call not valid in the index's current state */

#define MIRX_METRIC_IP 0
#define MIRX_METRIC_NEG_L2 1

/* search-tier selection (mirx_index_set_option MIRX_OPT_TIERS) */
#define MIRX_TIER_AUTO 0        /* bf16 MFMA candidates + guard, exact scan for the rest   */
#define MIRX_TIER_EXACT_ONLY 3  /* fp64 exact scan for every query (small galleries, tests) */

#define MIRX_OPT_TIERS 1
#define MIRX_OPT_SAMPLE_RANK 2   /* j: threshold = j-th largest sampled group maximum (default 8)  */
#define MIRX_OPT_FORCE_TAU 3     /* test hook: float bits of a fixed threshold; 0x7fc00000 = off  */
#define MIRX_OPT_PROFILE 4       /* 1: record HIP events around every stage of a search           */

/* stages timed when MIRX_OPT_PROFILE is on (mirx_index_last_timings) */
#define MIRX_STAGE_PREP 0        /* query conversion                                  */
#define MIRX_STAGE_SAMPLE 1      /* group-max GEMM on the row sample + threshold pick */
#define MIRX_STAGE_GEMM 2        /* the filter GEMM over the whole gallery            */
#define MIRX_STAGE_FINALIZE 3    /* candidate sort, guard, fp64 re-rank               */
#define MIRX_STAGE_EXACT 4       /* exact scan of rejected / routed queries           */
#define MIRX_NUM_STAGES 5

typedef struct mirx_index mirx_index;

/* Counters of the most recent search on an index (host-visible after the stream is idle). */
typedef struct mirx_search_stats {
    int64_t nq;               /* queries in the call                                        */
    int64_t tier1_answered;   /* answered by the bf16 MFMA pass + fp64 re-rank              */
    int64_t exact_answered;   /* fell through to (or were routed to) the fp64 exact scan    */
    int64_t candidates;       /* rows that passed the threshold, summed over queries        */
    int64_t reranked;         /* rows re-scored in fp64 by tier 1, summed over queries      */
    int64_t overflowed;       /* queries whose candidate list overflowed its capacity       */
    int64_t incomplete;       /* queries whose completeness guard failed                    */
} mirx_search_stats;

const char *mirx_last_error(void);
int mirx_version(void);

/*
 * Process-wide kernel-selection knobs.  They choose between kernels that return the SAME bits, so they change speed only
 * (the tests use them to run both kernels of a pair on one input).  No reference analogue: the reference leaves such
 * choices to cuDNN / MIOpen heuristics behind model(x) (model.py:71-84).
 *   MIRX_TUNE_CONV1X1_SMALL_MAX_WG  a 1x1 convolution of fewer than this many 128-channel x 128-pixel workgroups runs as
 *                                   one wave per 32 x 32 tile (small batches: test.py:1513 B = 64, milvus_retrieval.py:53-66
 *                                   B = 1); 0 = never.  Default 128.
 *   MIRX_TUNE_CONV3X3_SMALL_MAX_WG  the same for the dense layer's 3x3 convolution: fewer strip workgroups than this ->
 *                                   one wave per 32 output pixels; 0 = never.  Default 96.
 */
#define MIRX_TUNE_CONV1X1_SMALL_MAX_WG 1
#define MIRX_TUNE_CONV3X3_SMALL_MAX_WG 2
int mirx_set_tuning(int key, int64_t value);

/*
 * Index lifetime.  Replaces: Collection(name, schema) + create_index + load
 * (milvus/milvus_setup.py:139-222) -- an in-process, device-resident flat index.
 * `dim` is the embedding width (any value >= 1; rows are stored padded to 64).
 */
int mirx_index_create(int dim, int metric, int device, mirx_index **out);
void mirx_index_destroy(mirx_index *ix);

/*
 * Append n rows.  Replaces collection.insert([paths, labels, embeddings.tolist()])
 * (ingest_embeddings.py:399-411) for the embedding column; path/label metadata stay with the
 * Python Retriever.  rows: row-major [n, dim] fp32, host or device.  ids: n int64 (host or
 * device) or NULL for auto ids (previous size + i, Milvus auto_id analogue,
 * milvus_setup.py:170).  Synchronous: returns when the rows are resident.
 */
int mirx_index_add(mirx_index *ix, const float *rows, int64_t n, const int64_t *ids_or_null);
int mirx_index_reserve(mirx_index *ix, int64_t capacity_rows);
int64_t mirx_index_size(const mirx_index *ix);
int mirx_index_dim(const mirx_index *ix);
int mirx_index_set_option(mirx_index *ix, int option, int64_t value);

/* Copy rows [first, first+n) of the fp32 master back out (device or host destination). */
int mirx_index_get_rows(const mirx_index *ix, int64_t first, int64_t n, float *out_rows,
                        int64_t *out_ids_or_null);

/*
 * Exhaustive top-k.  Replaces collection.search(data=[q], anns_field="embedding", limit=k)
 * (milvus/milvus_retrieval.py:80-86, nih_zilliz_utils.py:263-270) and the brute force
 * `-torch.cdist` / `e @ e.t()` + fill_diagonal_(-inf) + topk of test.py:1080-1085,44.
 *   q            device, [nq, dim] fp32 row-major
 *   exclude_ids  device, nq int64 or NULL: gallery rows with this id are skipped for query i
 *                (the -inf diagonal of test.py:1081); use -1 for "nothing"
 *   out_scores   device, [nq, k] fp32: reported value (metric 0: dot; metric 1: -L2)
 *   out_ids      device, [nq, k] int64; slots past the number of eligible rows: id -1, -inf
 *   1 <= k <= 1024
 */
int mirx_index_search(mirx_index *ix, const float *q, int64_t nq, int k,
                      const int64_t *exclude_ids_or_null, float *out_scores, int64_t *out_ids,
                      void *stream);

/* fp64 ranking scores of the same hits ([nq, k] device doubles) for bit-exact parity tests. */
int mirx_index_search_f64(mirx_index *ix, const float *q, int64_t nq, int k,
                          const int64_t *exclude_ids_or_null, double *out_rank_scores,
                          int64_t *out_ids, void *stream);

/*
 * The same search in two calls, so that the host does not block while the first pass runs (the caller may enqueue the next
 * embed micro-batch, on this or another stream, between them):
 *   mirx_index_search_begin  enqueues query preparation, threshold sampling, the filter GEMM and finalize on `stream` and
 *       returns; the two counters that say whether any query needs the second-chance filter or the exact scan travel to
 *       pinned host memory behind an event.  Either output form may be NULL (not both).
 *   mirx_index_search_end    waits for that event (a host wait; the stream is not drained), runs the rare follow-up passes
 *       the counters ask for on the same stream, and returns.  The outputs are complete when work enqueued on the stream
 *       up to this call has finished; stats / timings refer to this search.
 * Between the two calls the index must not be searched, ranked or extended, and the query / output buffers must stay valid.
 * mirx_index_search == begin + end.  (No reference counterpart: MilvusRetriever.search is a blocking RPC,
 * milvus/milvus_retrieval.py:80-86.)
 */
int mirx_index_search_begin(mirx_index *ix, const float *q, int64_t nq, int k, const int64_t *exclude_ids_or_null,
                            float *out_scores_or_null, double *out_rank_scores_or_null, int64_t *out_ids, void *stream);
int mirx_index_search_end(mirx_index *ix);

/* Waits for `stream`, then copies the counters of the last search. */
int mirx_index_last_stats(mirx_index *ix, void *stream, mirx_search_stats *out);

/*
 * Device time of each stage of the last search in milliseconds, measured with HIP events
 * recorded on the search's own stream (needs MIRX_OPT_PROFILE = 1 before the search; stages
 * that did not run report 0; a search split into several internal passes reports sums).
 * out_ms: host array of MIRX_NUM_STAGES floats.
 */
int mirx_index_last_timings(mirx_index *ix, float *out_ms);

/*
 * Full ranking of every gallery row for each query (row = query): replaces
 * torch.argsort(dists, dim=0, descending=True) (test.py:1090,179) and the
 * top_k = num_entities search of query_nih_zilliz.py:53-63.  out_ids: device [nq, size]
 * int64 (excluded row last); out_scores_or_null: device [nq, size] fp32 reported values.
 * Gallery size limited to 65536 rows per call in this version.
 */
int mirx_index_rank_all(mirx_index *ix, const float *q, int64_t nq,
                        const int64_t *exclude_ids_or_null, int64_t *out_ids,
                        float *out_scores_or_null, void *stream);

/*
 * Merge per-shard top-k lists (multi-GPU: after the all-gather of SURVEY 8e).
 * in_scores/in_ids: device [nshard, nq, k] fp64 ranking scores / int64 ids; out: [nq, k].
 * Same order rule as search (score desc, id asc); id -1 entries sort last.
 */
int mirx_topk_merge(const double *in_scores, const int64_t *in_ids, int nshard, int64_t nq,
                    int k, int metric, double *out_rank_scores, float *out_scores,
                    int64_t *out_ids, void *stream);

/*
 * mirx_conv1x1_bn_relu with the operands carried as three bf16 terms each (x = xh + xm + xl, six bf16
 * MFMAs per product block, fp32 accumulation): fp32-grade results at 2.67x less matrix-pipe time, for
 * the layers with many input channels.  Same arguments as mirx_conv1x1_bn_relu except the weights:
 * w3 = device bf16 [cout / 128][cin / 16][3][128][16], the three terms of W[co, k] (already multiplied
 * by the folded norm2 scale) for output block co / 128 and input stage k / 16 (mirx.model._split3_weights),
 * and the output batch stride: image b is written at y + b * y_batch_stride as [cout, hw] (cout * hw for a
 * packed tensor; the channel-prefix of the next dense block's buffer for a transition).
 * cin % 16 == 0, cout % 128 == 0; any hw.
 */
int mirx_conv1x1_bn_relu_split3(const float *x, int64_t x_batch_stride, int cin, const float *scale1_or_null,
                                const float *shift1_or_null, const void *w3, const float *bias_or_null,
                                int64_t n, int hw, int cout, int relu_out, float *y, int64_t y_batch_stride,
                                void *stream);

/*
 * The two-fp16-term DenseNet path (224 x 224 inputs).  fp16 has 5 exponent bits, so the RANGE of every buffer travels with
 * it, PER IMAGE: a "range row" is device fp32 [n] (one float per image of the call, zeroed by the caller once per forward);
 * every mirx kernel that writes image b of a dense block's buffer folds the largest |value| it wrote into row[b] (unsigned
 * atomic max on the float bits), and every kernel that reads image b scales what it stages by a power of two derived from
 * row[b] alone.  An image's arithmetic therefore does not depend on its batch mates; a non-finite value makes THAT image's
 * outputs NaN (never a silently wrong finite value) and leaves the others untouched.
 *
 * mirx_conv1x1_bn_relu_split2h: mirx_conv1x1_bn_relu_split3 with TWO fp16 terms per operand (three MFMAs per product block
 * instead of six).  `in_range_or_null` = range row of x; the kernel stages act_in(x_b) * 2^s with
 *     bound_b = in_ks * row[b] + in_kb         (in_ks = max |scale1|, in_kb = max |shift1|; 1, 0 without prologue;
 *                                               row NULL: bound = in_kb, a caller-proved constant)
 * and s chosen so that bound_b * 2^s is in [2^14, 2^15).  w2 = device fp16 [cout / 128][cin / 16][2][128][16]: the two terms
 * of W[co, k] * ws[co], ws a power of two per output channel (mirx.model._split2h_weights); oscale = device fp32 [cout] =
 * 1 / ws.  `out_range_or_null`: range row of y.
 * x_plane_stride / y_plane_stride: floats between consecutive channel planes of x / y (0 = hw, packed planes).
 * Otherwise the contract of mirx_conv1x1_bn_relu_split3 (reference: the conv1 / transition conv calls inside
 * torchvision densenet121, model.py:53-60).
 */
int mirx_conv1x1_bn_relu_split2h(const float *x, int64_t x_batch_stride, int cin, const float *scale1_or_null,
                                 const float *shift1_or_null, const void *w2, const float *oscale,
                                 const float *bias_or_null, int64_t n, int hw, int cout, int relu_out, float *y,
                                 int64_t y_batch_stride, const float *in_range_or_null, float in_ks, float in_kb,
                                 float *out_range_or_null, int64_t x_plane_stride, int64_t y_plane_stride, void *stream);

/*
 * The dense layer with the 128-channel bottleneck handed over ALREADY SPLIT into its two fp16 terms (same bytes as fp32,
 * but the 3x3 conv then stages it by LDS DMA alone -- no register prefetch, no split, no LDS stores -- and the 1x1 conv
 * writes 16-byte runs instead of 4-byte channel-plane stores):
 *   mirx_conv1x1_bn_relu_split2h_terms: mirx_conv1x1_bn_relu_split2h for cout = 128 with relu, writing
 *       y_terms = device fp16 [n][8 groups][2 terms][hw][16]: group g holds the 16 channels
 *       64 (g >> 2) + 32 ((g >> 1) & 1) + 4 (g & 1) + {0..3, 8..11, 16..19, 24..27} (mirx.model.YTERMS_CHANNEL_ORDER; the
 *       consumer's weights use the same order), image b scaled by 2^t where |y_b| <= y_ks * bound_b + y_kb
 *       (y_ks = max_o sum_c |W[o, c]|, y_kb = max |bias|: a bound known before the kernel runs) is brought into
 *       [2^14, 2^15); y_inv_out = device fp32 [n] receives 2^-t of every image.
 *   mirx_conv3x3_direct_terms_nchw: the 3x3 conv (128 -> 32, pad 1) as a direct implicit GEMM on such y_terms and y_inv
 *       (device fp32 [n]); the padding ring of the staged strip comes from out-of-range buffer loads (zero); w2 = device fp16
 *       [8 stages][9 taps][2 terms][32 oc][16 c] in the permuted channel order, scaled per output channel by a power of two
 *       (mirx.model._conv3x3_weights_split2h), oscale = device fp32 [32] = 1 / that scale.  side 56 / 28 / 14, and 7 (four
 *       whole images per workgroup, each with its own zero ring).  `out_range_or_null`: range row of `out`.
 *   x_plane_stride / out_plane_stride: floats between consecutive channel planes of the dense block's buffer (0 = packed,
 *       hw resp. side^2).
 * (reference: the conv1 / conv2 calls inside torchvision densenet121, model.py:53-60)
 */
int mirx_conv1x1_bn_relu_split2h_terms(const float *x, int64_t x_batch_stride, int cin, const float *scale1,
                                       const float *shift1, const void *w2, const float *oscale, const float *bias,
                                       int64_t n, int hw, void *y_terms, const float *in_range, float in_ks, float in_kb,
                                       float y_ks, float y_kb, float *y_inv_out, int64_t x_plane_stride, void *stream);
int mirx_conv3x3_direct_terms_nchw(const void *y_terms, const void *w2, const float *oscale, int64_t n, int side, float *out,
                                   int64_t out_batch_stride, const float *y_inv, float *out_range_or_null,
                                   int64_t out_plane_stride, void *stream);
/*
 * mirx_conv3x3_direct_terms_nchw_pool (sides 56 / 28 / 14): the same launch ALSO writes the transition's pooled input for
 * its 32 new channels -- pooled[b, oc] = avgpool2x2(relu(out[b, oc] * pool_scale[oc] + pool_shift[oc])) with images
 * `pooled_batch_stride` floats apart and packed (side/2)^2 planes; pool_scale / pool_shift / pooled already point at this
 * layer's first channel.  The values are computed from the stored fp32 outputs by the same operations in the same order as
 * mirx_bn_relu_avgpool2: bit-identical to running that pass afterwards, without reading the block's map from HBM again
 * (the transition's mirx_bn_relu_avgpool2_into then covers only the block's first channels).  Exists in the strip kernel
 * only: mirx_conv3x3_small_launch(n, side) = 1 means a launch of n images takes the one-wave-per-block kernel (see
 * mirx_set_tuning) and this entry point refuses it.
 * (reference: torchvision _DenseLayer.conv2 followed, at the end of the block, by _Transition.norm / relu / pool, model.py:53-60)
 */
int mirx_conv3x3_direct_terms_nchw_pool(const void *y_terms, const void *w2, const float *oscale, int64_t n, int side, float *out,
                                        int64_t out_batch_stride, const float *y_inv, float *out_range_or_null,
                                        int64_t out_plane_stride, const float *pool_scale, const float *pool_shift,
                                        float *pooled, int64_t pooled_batch_stride, void *stream);
int mirx_conv3x3_small_launch(int64_t n, int side);

/*
 * mirx_dense_layer_fused: the dense layer of the 14 x 14 and 7 x 7 maps (dense blocks 3 and 4) in ONE launch --
 *     buf[b, cin : cin + 32] = conv2_3x3( relu( conv1_1x1( relu( buf[b, 0 : cin] * scale1 + shift1 ) ) + bias ) )
 * with the 128-channel bottleneck of a 196-pixel unit (one 14 x 14 image, four 7 x 7 images) kept in the CU's LDS: it never
 * reaches HBM.  Arguments as mirx_conv1x1_bn_relu_split2h_terms (w2, oscale, bias, in_ks, in_kb, y_ks, y_kb) and
 * mirx_conv3x3_direct_terms_nchw (c3w2 in the permuted channel order, c3oscale); buf = device fp32 [n, >= cin + 32, side^2]
 * (packed planes: plane_stride 0 or side^2; 16-byte aligned) at batch_stride floats per image (a multiple of 4); range_row = the buffer's range row (device fp32 [n]): read
 * for the input bound of every image, then raised to the largest |value| of its 32 new channels.  The 32 channels are
 * bit-identical to the two-launch form.  cin % 32 == 0, 128 <= cin <= 1024.
 * (reference: _DenseLayer of torchvision densenet121, model.py:53-60)
 */
int mirx_dense_layer_fused(float *buf, int64_t batch_stride, int64_t plane_stride, int cin, const float *scale1,
                           const float *shift1, const void *w2, const float *oscale, const float *bias, const void *c3w2,
                           const float *c3oscale, int64_t n, int side, float *range_row, float in_ks, float in_kb, float y_ks,
                           float y_kb, void *stream);

/* mirx_range_absmax: range_row[b] = max(range_row[b], largest |x| of image b), x = n images of `per_image` contiguous
 * fp32 each -- the range of the input images for mirx_stem_conv7_bn_relu_pool_split2h_into, the stem (conv 7x7 / 2 + norm0 +
 * relu0 + maxpool 3x3 / 2, one kernel) on two fp16 terms per operand: w2 = device fp16 [2][11][2][32][16], oscale = device
 * fp32 [64] (mirx.model._stem_weights_split2h); image b is written at y + b * y_batch_stride (the channel prefix of dense
 * block 1's buffer: no copy); in_range = range row of x, out_range_or_null = range row of y.  n <= 65535.
 * (reference: conv0 / norm0 / relu0 / pool0 of torchvision densenet121, model.py:53-60) */
int mirx_range_absmax(const float *x, int64_t per_image, int64_t n, float *range_row, void *stream);
/* The same two entry points for RAW 8-bit images x = device uint8 [n, 3, h, w] (hw = h * w): the reference's ToTensor +
 * Normalize (x = u / 255, then (x - mean3[c]) / std3[c] in fp32: test.py:1309-1332) is applied while the stem stages its input
 * patch (and by the range pass), through a 3 x 256 table built with exactly those operations -- the embeddings are
 * bit-identical to feeding the normalised fp32 tensor, at a quarter of the input bytes over PCIe and out of HBM.
 * mean3 / std3 = device fp32 [3]. */
int mirx_range_absmax_u8(const uint8_t *x, int64_t hw, int64_t n, const float *mean3, const float *std3, float *range_row,
                         void *stream);
int mirx_stem_conv7_bn_relu_pool_split2h_u8_into(const uint8_t *x, const float *mean3, const float *std3, const void *w2,
                                                 const float *oscale, const float *scale, const float *shift, int64_t n, int h,
                                                 int w, float *y, int64_t y_batch_stride, const float *in_range,
                                                 float *out_range_or_null, void *stream);
int mirx_stem_conv7_bn_relu_pool_split2h_into(const float *x, const void *w2, const float *oscale, const float *scale,
                                              const float *shift, int64_t n, int h, int w, float *y, int64_t y_batch_stride,
                                              const float *in_range, float *out_range_or_null, void *stream);

/*
 * Memory-bound glue of the token-major backbones, so that their forward runs without a library kernel:
 *
 * mirx_layernorm: y = (x - mean) / sqrt(var + eps) * gamma + beta over the last axis of x [m, c] (nn.LayerNorm inside
 *   timm ConvNeXtV2 / ViT and transformers SigLIP: model.py:96-100, 459-463, 553-557); biased variance, fp32.
 *   tokens_per_image == 0: y is [m, c] (may alias x); > 0: y is channels-first [m / tpi][c][tpi] (timm LayerNorm2d of the
 *   ConvNeXt stem; c <= 512, y != x).  c % 4 == 0, c <= 8192 (a row lives in one wavefront's registers).
 * mirx_patchify_nchw: non-overlapping patch x patch blocks of x [n, c, h, w] as rows
 *   out[((b * (h/patch) + py) * (w/patch) + px) * row_stride + (ch * patch + ky) * patch + kx], columns beyond
 *   c * patch * patch zeroed: a Conv2d(c, cout, kernel = stride = patch) is patchify + mirx_linear_split3 / _split2h with
 *   weight.flatten(1) (the ViT / SigLIP patch embedding, the ConvNeXt stem and downsample convolutions).  With
 *   ln_gamma / ln_beta every pixel is first normalised over its c channels (timm LayerNorm2d in front of the ConvNeXt
 *   downsample conv).
 * mirx_attention_small: softmax(scale q k^T) v for short query sets, one wavefront per (image, head, query), fp32:
 *   q[(b * n_queries + i) * q_row_stride + hd * head_dim + d], k / v likewise with kv_row_stride, key_mask[b * n_keys + j]
 *   != 0 keeps key j (NULL: all keys; a query whose keys are all masked gives zeros), out [batch, n_queries, heads *
 *   head_dim].  Used for the SigLIP text tower (64 tokens, padding mask: eval_medsiglip.py:164-186) and the SigLIP
 *   attention-pooling head (1 probe query).  head_dim 16 / 32 / 64 / 72.
 */
int mirx_layernorm(const float *x, int64_t m, int c, const float *gamma_or_null, const float *beta_or_null, float eps,
                   float *y, int tokens_per_image, void *stream);
int mirx_patchify_nchw(const float *x, int64_t n, int c, int h, int w, int patch, const float *ln_gamma_or_null,
                       const float *ln_beta_or_null, float eps, float *out, int row_stride, void *stream);
int mirx_attention_small(const float *q, int64_t q_row_stride, const float *k, const float *v, int64_t kv_row_stride,
                         const uint8_t *key_mask_or_null, int64_t batch, int heads, int head_dim, int n_queries, int n_keys,
                         float scale, float *out, void *stream);

/*
 * Linear layer of the token-major backbones (replaces the nn.Linear calls inside the timm / transformers
 * models the reference instantiates: model.py:448-494 DinoV2, model.py:87-118 ConvNeXtV2,
 * model.py:536-638 MedSigLIP vision tower), fp32-grade on the bf16 matrix pipe with three bf16 terms per operand:
 *     y[i, j] = epi( sum_k x[i, k] W[j, k] + bias[j] )
 *     act = 1: exact (erf) GELU;  residual != NULL:  y = residual + gamma[j] * v  (LayerScale + skip; gamma
 *     NULL = 1; y may alias residual).
 * x = device fp32 [m, k] row-major; w3 = device bf16 [ceil(n / 128)][k / 16][3][128][16] (the layout of
 * mirx_conv1x1_bn_relu_split3, mirx.model._split3_weights(W); weight rows beyond n are zero padding);
 * y = device fp32 [m, n].  k % 16 == 0, any n >= 1, any m >= 0.
 */
int mirx_linear_split3(const float *x, int64_t m, int k, const void *w3, const float *bias_or_null, int n, int act,
                       const float *residual_or_null, const float *gamma_or_null, float *y, void *stream);

/*
 * mirx_linear_split3 with TWO fp16 terms per operand (three MFMAs per product block instead of six; same measured
 * error, 1.7x the speed) for inputs whose range the caller can bound:
 *     y = epi( out_scale * sum_k (x[i, k] * x_scale) W2[j, k] + bias[j] ),   epilogues as mirx_linear_split3
 * w2 = device fp16 [ceil(n / 128)][k / 16][2][128][16]: the two terms of W * w_scale, w_scale a power of two
 * (mirx.model._linear_h2_weights); x_scale a power of two with |x * x_scale| <= 65504 for EVERY element (fp16
 * range: the caller's contract -- mirx.model uses this entry only behind a LayerNorm, whose output is bounded by
 * sqrt(C - 1) max|gamma| + max|beta|); out_scale = 1 / (x_scale * w_scale).  k % 16 == 0.
 */
int mirx_linear_split2h(const float *x, int64_t m, int k, const void *w2, const float *bias_or_null, int n, int act,
                        const float *residual_or_null, const float *gamma_or_null, float x_scale, float out_scale,
                        float *y, void *stream);

/*
 * The token-major Linear with BOTH operands pre-split ("terms rows"), the form the ViT / SigLIP towers run on: the same
 * arithmetic as mirx_linear_split2h (two fp16 terms per operand, three MFMAs per product block, fp32 accumulation), but the
 * activations arrive already split, so the kernel is a DMA-fed MFMA GEMM (k_linear_t2.hip).  Replaces nn.Linear inside timm
 * VisionTransformer blocks and transformers SiglipEncoderLayer (model.py:459-463, 553-557 of the reference build them).
 *
 * TERMS ROWS of a matrix a [rows][k] scaled by a power of two s: rows of ceil(k / 32) lines of 128 bytes,
 *     line g = fp16 hi(s a[32 g .. 32 g + 31]) | fp16 lo(..),   hi = fp16(s a), lo = fp16(s a - hi), zero beyond k
 * -- 4 bytes per element like fp32.  Contract: |s a| <= 65504 for every element (callers pass provable bounds).
 *
 * mirx_rows_to_terms:   xt = terms rows of `scale` * x, x = device fp32 [m][k] with `row_stride` floats between rows
 *                       (row_stride % 4 == 0, x 16-byte aligned).  xt: m * ceil32(k) * 4 bytes.
 * mirx_layernorm_terms: mirx_layernorm (token-major form) writing terms rows of `scale` * y instead of fp32.
 * mirx_linear_terms:    v = act( out_scale * sum_k xt[i, k] wt[j, k] + bias[j] ),   out_scale = 1 / (x scale * w scale)
 *                       act 0 none, 1 GELU (erf), 2 GELU (tanh);
 *                       y (fp32 [m][n]) = v, or residual + gamma[j] * v (act 0; gamma NULL = 1; y may alias residual);
 *                       or yt = terms rows of yt_scale * v (no residual) -- the next Linear's input, written in full lines.
 *                       wt = terms rows of W * w scale, ceil(n / 256) * 256 rows (zero beyond n).  n % 4 == 0.
 *                       workspace: a tile (256 tokens x 256 outputs) occupies one CU for its whole K loop, so a launch whose
 *                       tile count is not a multiple of the CU count would end in a mostly empty round.  With a device
 *                       buffer of mirx_linear_terms_workspace_bytes(m, k, n) bytes (0 = not needed; at most 64 MiB) the
 *                       tiles of that last round are cut along k into pieces that run side by side and are summed in piece
 *                       order by a second launch -- deterministic for a given (m, k, n).  NULL = whole tiles only.
 */
int mirx_rows_to_terms(const float *x, int64_t m, int k, int64_t row_stride, float scale, void *xt, void *stream);
/* LayerNorm over the channels of every pixel of channels-last maps x [n_img, h, w, c], written as the 2 x 2 patch rows of a
 * stride-2 convolution: y [n_img, h / 2, w / 2, 4 c] with feature order (ky, kx, channel) -- timm's LayerNorm2d + Conv2d(k = s = 2)
 * downsample of ConvNeXt becomes this + a Linear whose weight is conv.weight.permute(0, 2, 3, 1).reshape(cout, 4 c). */
int mirx_layernorm_patch2_nhwc(const float *x, int64_t n_img, int h, int w, int c, const float *gamma_or_null,
                               const float *beta_or_null, float eps, float *y, void *stream);
int mirx_layernorm_terms(const float *x, int64_t m, int c, const float *gamma_or_null, const float *beta_or_null, float eps,
                         float scale, void *yt, void *stream);
int mirx_linear_terms(const void *xt, int64_t m, int k, const void *wt, const float *bias_or_null, int n, int act,
                      const float *residual_or_null, const float *gamma_or_null, float out_scale, float *y_or_null,
                      void *yt_or_null, float yt_scale, void *workspace_or_null, int64_t workspace_bytes, void *stream);
int64_t mirx_linear_terms_workspace_bytes(int64_t m, int k, int n);

/*
 * Tail of a ConvNeXtV2 block (timm ConvNeXtBlock.forward, used by the reference's model.py:87-118): the second
 * point-wise Linear on the channels-last hidden map, written back channels-first with the block's skip added:
 *     y[b, j, p] = residual[b, j, p] + sum_k (x[b * tpi + p, k] * input_scale[b, k]) W[j, k] + bias[j]
 * x = device fp32 [n_img * tokens_per_image, k]; w3 as in mirx_linear_split3; residual / y = device fp32
 * [n_img, n, tokens_per_image] (NCHW); residual NULL = no skip; y may alias residual.
 * input_scale = device fp32 [n_img, k] or NULL: the GRN factor 1 + weight * gx / (mean gx + eps), applied while x
 * is staged (the GRN shift is constant per feature: the caller adds W . grn_bias to `bias`).
 */
int mirx_linear_split3_nchw(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w3,
                            const float *bias_or_null, int n, const float *residual_or_null,
                            const float *input_scale_or_null, float *y, void *stream);

/*
 * mirx_linear_split3_nchw on TWO fp16 terms per operand (mirx_linear_split2h's arithmetic) for inputs with a known bound:
 *   |x[i, j]| <= x_bound for every element (host scalar, e.g. the provable bound of a Linear fed by a LayerNorm, through GELU);
 *   input_scale (the GRN scale [n_img, k], multiplied into x while it is staged) comes with input_scale_max = device fp32[1]
 *   holding max |input_scale| -- it is computed per forward, so the kernel reads it and derives the power-of-two staging
 *   scale from x_bound * input_scale_max[0] itself (no host round trip; a non-finite bound makes every output NaN);
 *   w2 / w_inv = mirx.model._linear_h2_weights (the two fp16 terms of W * w_scale, and 1 / w_scale).
 * Reference: timm ConvNeXtBlock (mlp.fc2 after GRN, permute back, + shortcut) and the LayerNorm2d + 2x2/2 downsample conv,
 * as used by the reference's model.py:87-118.
 */
int mirx_linear_split2h_nchw(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w2,
                             const float *bias_or_null, int n, const float *residual_or_null,
                             const float *input_scale_or_null, float x_bound, const float *input_scale_max_or_null,
                             float w_inv, float *y, void *stream);
/* fc1 of a ConvNeXtV2 block with the global response norm's reduction folded in: y = gelu(mirx_linear_split2h(x)) (row-major
 * [n_img * tokens_per_image, n]) AND gx[b, j] = || y[b, :, j] ||_2 (what mirx_grn_norm_nhwc would compute from y in a pass of its
 * own): every workgroup returns the column sums of y^2 of its 128 token rows split by image in `partials` (device fp32,
 * ceil(n_img * tokens_per_image / 128) * 2 * n floats), a second small launch adds them in tile order.  tokens_per_image >= 128.
 * Bit-reproducible. */
int mirx_linear_split2h_gelu_grn(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w2,
                                 const float *bias_or_null, int n, float x_scale, float out_scale, float *y, float *partials,
                                 float *gx, void *stream);
/* The same block tail for a channels-last residual stream (the ConvNeXtV2 fast path): residual and y are row-major
 * [n_img * tokens_per_image, n] (y may alias the residual), input_scale / input_scale_max are required. */
int mirx_linear_split2h_grn_rows(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w2,
                                 const float *bias_or_null, int n, const float *residual_or_null, const float *input_scale,
                                 float x_bound, const float *input_scale_max, float w_inv, float *y, void *stream);

/*
 * Global response normalisation of ConvNeXtV2 (timm GlobalResponseNorm, channels last) as two HBM passes:
 *   mirx_grn_norm_nhwc:  gx[b, c] = || x[b, :, c] ||_2              x = device fp32 [n, hw, c], gx = [n, c]
 * with scale = 1 + weight * gx / (mean_c gx + 1e-6) and shift = bias formed by the caller on [n, c].
 * n <= 65535.  Fixed summation order (bit-reproducible).
 */
int mirx_grn_norm_nhwc(const float *x, int64_t n, int hw, int c, float *gx, void *stream);
/* The GRN scale vector in one launch: scale[b, j] = 1 + weight[j] * gx[b, j] / (mean_j gx[b, :] + eps)  (timm
 * GlobalResponseNorm: x * (1 + weight * Nx) + bias with Nx = Gx / (mean Gx + eps), eps = 1e-6), and scale_max[0] = the largest
 * |scale| over the batch -- the device-side bound mirx_linear_split2h_nchw reads; combined by atomic max, so the caller ZEROES
 * scale_max[0] first. */
int mirx_grn_scale(const float *gx, const float *weight, int64_t n, int c, float eps, float *scale, float *scale_max,
                   void *stream);

/*
 * 3x3 convolution of a DenseNet dense layer (128 -> 32 channels, stride 1, pad 1, no bias): conv2 of
 * torchvision's _DenseLayer (model.py:53), as Winograd F(2x2,3x3) on fp32 MFMA.  x: device NCHW fp32
 * [n, 128, side, side] (packed), side = 56, 28, 14 or 7.  u: device fp32 [16 stages][16][8][32] = the
 * transformed weights U_xi[oc, c] = (G g G^T)_xi of the layer, stage s holding channels 8s .. 8s+7 as
 * [xi = 4i + j][c][oc] (mirx.model prepares it once per layer).  The 32 output channels of image b are
 * written at out + b * out_batch_stride as [32, side, side] -- i.e. directly into the layer's slice of
 * the dense-block buffer.  Other sizes: MIRX_EINVAL (use the library convolution).
 */
int mirx_conv3x3_winograd_nchw(const float *x, const float *u, int64_t n, int side, float *out,
                               int64_t out_batch_stride, void *stream);

/*
 * mirx_conv3x3_winograd_nchw with the 16 Winograd-domain channel GEMMs on three-term bf16 MFMAs (fp32-grade,
 * see mirx_conv1x1_bn_relu_split3): same x / out / out_batch_stride; u3 = device bf16
 * [cin / 16][ij = 4 i + j][3 terms][32 oc][16 channels], the split of U = G g G^T
 * (mirx.model._winograd_weights_split3).  side in {56, 28, 14}; the 7 x 7 maps use mirx_conv3x3_winograd_nchw.
 */
int mirx_conv3x3_winograd_split3_nchw(const float *x, const void *u3, int64_t n, int side, float *out,
                                      int64_t out_batch_stride, void *stream);

/*
 * The same convolution (conv2 of a dense layer: 128 -> 32 channels, 3x3, pad 1) as a DIRECT implicit GEMM on
 * three-term bf16 MFMAs (no Winograd transform: K = 9 taps x 128 channels; fp32-grade): same x / out /
 * out_batch_stride; w3 = device bf16 [8 stages][9 taps = 3 ky + kx][3 terms][32 oc][16 channels]
 * (mirx.model._conv3x3_weights_split3).  side in {56, 28, 14}.
 */
int mirx_conv3x3_direct_split3_nchw(const float *x, const void *w3, int64_t n, int side, float *out,
                                    int64_t out_batch_stride, void *stream);

/*
 * Multi-head self-attention of the ViT backbones, fp32: out = softmax(q k^T * scale) v per (image,
 * head), scores never materialised.  Replaces the attention of timm's `vit_base_patch14_dinov2`
 * blocks (model.py:459-463; nih_multilabel_retrieval.py:175-221).  qkv: device [batch, n_tokens, 3,
 * heads, head_dim] fp32, exactly the output of the block's qkv Linear; out: device [batch, n_tokens,
 * heads, head_dim] fp32 (= [batch, n_tokens, C], no head transpose).  head_dim in {32, 64, 72, 96}
 * (64: ViT-B / DINOv2; 72: the SigLIP-So400m tower of MedSigLIP, model.py:536-638).
 */
int mirx_attention_qkv_f32(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim,
                           float scale, float *out, void *stream);

/*
 * mirx_attention_qkv_f32 with both GEMMs on three-term bf16 MFMAs (Q, K, V and the probabilities each carried
 * as xh + xm + xl; fp32-grade, see mirx_conv1x1_bn_relu_split3): same arguments, result layout and head_dim set.
 */
int mirx_attention_qkv_f32_split3(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim,
                                  float scale, float *out, void *stream);

/*
 * mirx_attention_qkv_f32 with both GEMMs on TWO fp16 terms per operand (three MFMAs per product block; see
 * mirx_linear_split2h).  qk_bound >= max |q|, |k| and v_bound >= max |v| over the packed projection are the caller's
 * contract (fp16 range; mirx.model derives them from the LayerNorm in front of the projection and its row norms);
 * the library turns them into exact power-of-two scales.  head_dim 64 (DINOv2 / ViT-B), or 32 / 72 / 96 through the
 * general kernel (72 = the SigLIP-So400m tower of MedSigLIP).
 */
int mirx_attention_qkv_f32_split2h(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim,
                                   float scale, float qk_bound, float v_bound, float *out, void *stream);
/* The same attention with the result written as terms rows of out_scale * out (include above: mirx_linear_terms), |out| <=
 * v_bound (a softmax-weighted average of V rows), so that the output projection reads it without a conversion pass.
 * out_terms: batch * n_tokens rows of ceil32(heads * head_dim) * 4 bytes. */
int mirx_attention_qkv_f32_split2h_terms(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim, float scale,
                                         float qk_bound, float v_bound, float out_scale, void *out_terms, void *stream);

/*
 * Metric tail over ranked lists, on the device: one pass per query over its
 * ranking `ranks[q, 0..n)` (gallery row ids, best first; rows `row_stride` apart) gives
 *   out_ap[q]      AP of the list: ap_kind 0 = the trapezoidal compute_ap of test.py:58-92 summed the
 *                  way compute_map does (test.py:95-146), ap_kind 1 = mean of precision at each relevant
 *                  rank (compute_map_multilabel test.py:941-985, fusion_eval/metrics.py:41-94);
 *                  NaN when the list holds no relevant id (the reference's "nempty" / skipped queries);
 *   out_cnt[q, j]  relevant ids within the first kappas[j] ranks (retrieval_accuracy test.py:38-54;
 *                  precision@kappa test.py:137-142; mP@k / R@k of fusion_eval/metrics.py:70-86);
 *   out_nrel[q]    relevant ids in the whole list; out_maxpos[q] largest 1-based relevant rank (0 = none).
 * Relevance of gallery row `id` for query q: rel_kind 0 -> gallery_labels[id] == query_labels[q];
 * rel_kind 1 -> labels are multi-hot bit masks (<= 64 classes) and
 * |a & b| / (|a | b| + 1e-8) > jaccard_threshold (test.py:956-965).  query_ids_or_null[q] is never
 * relevant (self-exclusion: binary_relevance[i] = 0, test.py:966).  Ids outside [0, n_labels) are
 * not relevant.  All pointers are device pointers except kappas (host, nk <= 8).
 */
int mirx_rank_metrics(const int64_t *ranks, int64_t nq, int64_t n, int64_t row_stride,
                      const int64_t *gallery_labels, int64_t n_labels, const int64_t *query_labels,
                      const int64_t *query_ids_or_null, int drop_self, int rel_kind,
                      double jaccard_threshold, int ap_kind, const int32_t *kappas, int nk,
                      double *out_ap, int64_t *out_cnt, int64_t *out_nrel, int64_t *out_maxpos,
                      void *stream);

/*
 * x <- x / max(||x||_2, 1e-12) row-wise, in place.  Replaces F.normalize(x, dim=1)
 * (model.py:83,116,493,634; milvus_retrieval.py:63).  x: device [n, dim] fp32.
 */
int mirx_l2_normalize(float *x, int64_t n, int dim, void *stream);

/*
 * Embedding head: y[b, c] = mean_{hw} relu(x[b, c, h, w] * scale[c] + shift[c]), then the
 * optional L2 normalisation of each row.  Replaces norm5 -> relu -> AdaptiveAvgPool2d(1)
 * -> flatten -> F.normalize of model.py:59-60,73-74,83 in one pass over the feature map.
 * x: device NCHW fp32 [n, c, hw]; scale/shift: device [c] (folded eval BatchNorm) or NULL
 * for identity; y: device [n, c] fp32.
 */
int mirx_bn_relu_gap_l2norm(const float *x, const float *scale, const float *shift, int64_t n,
                            int c, int hw, int normalize, float *y, void *stream);

/*
 * y = relu(x * scale[c] + shift[c]) for the first c channels of an NCHW fp32 tensor whose
 * images are `x_batch_stride` floats apart (a channel-prefix view of a wider dense-block
 * buffer); y is packed [n, c, hw].  Replaces norm1 -> relu1 (and transition norm -> relu) of
 * torchvision's _DenseLayer / _Transition (model.py:53) in ONE pass instead of two.
 */
int mirx_bn_relu_nchw(const float *x, int64_t x_batch_stride, const float *scale, const float *shift,
                      int64_t n, int c, int hw, float *y, void *stream);

/*
 * Transition front half: y = avgpool2x2(relu(x * scale[c] + shift[c])), x as above with
 * h x w pixels (h, w even), y packed [n, c, h/2, w/2].  The 1x1 transition conv commutes with
 * the average pool, so the caller runs it on the pooled map (4x fewer pixels).
 */
int mirx_bn_relu_avgpool2(const float *x, int64_t x_batch_stride, const float *scale,
                          const float *shift, int64_t n, int c, int h, int w, float *y, int64_t x_plane_stride,
                          void *stream);    /* x_plane_stride: floats between channel planes of x (0 = h * w; else % 4 == 0) */
/* ... the same with y = [n, >= c, h/2, w/2]: images `y_batch_stride` floats apart (the first c channels of a wider pooled map
 * whose other channels mirx_conv3x3_direct_terms_nchw_pool writes) */
int mirx_bn_relu_avgpool2_into(const float *x, int64_t x_batch_stride, const float *scale, const float *shift,
                               int64_t n, int c, int h, int w, float *y, int64_t y_batch_stride, int64_t x_plane_stride,
                               void *stream);

/*
 * DenseNet stem: conv 7x7 stride 2 pad 3 (3 -> 64 channels) + folded BatchNorm + ReLU +
 * max-pool 3x3 stride 2 pad 1, NCHW fp32 in, NCHW fp32 out [n, 64, H/4, W/4].
 * Replaces features.conv0/norm0/relu0/pool0 of torchvision densenet121 (model.py:53).
 * w: device [64, 3, 7, 7]; scale/shift: device [64].  H, W multiples of 4.
 */
int mirx_stem_conv7_bn_relu_pool(const float *x, const float *w, const float *scale,
                                 const float *shift, int64_t n, int h, int wd, float *y,
                                 void *stream);

/*
 * The same stem with the implicit GEMM on three-term bf16 MFMAs (fp32-grade, see mirx_conv1x1_bn_relu_split3).
 * w3 = device bf16 [2 blocks of 32 oc][11 steps][3 terms][32 oc][16 k]: step s, k = 8 g + i is weight
 * (c, ky) = divmod(2 s + g, 7), kx = 2 i for i < 4, 2 (i - 4) + 1 for i >= 4 (kx = 7 and row 21 are zero)
 * -- mirx.model._stem_weights_split3(conv0.weight).  n <= 65535.
 */
int mirx_stem_conv7_bn_relu_pool_split3(const float *x, const void *w3, const float *scale, const float *shift,
                                        int64_t n, int h, int wd, float *y, void *stream);

/*
 * Fused 1x1 convolution of a DenseNet dense layer / transition (fp32 MFMA):
 *     y[b, o, p] = act_out( sum_k wt[k, o] * act_in(x[b, k, p]) + bias[o] )
 * act_in(v) = relu(v * scale[k] + shift[k]) when scale != NULL (norm1 + relu1), identity otherwise;
 * act_out = relu when relu_out != 0 (relu2; norm2 = bias + a scale folded into wt by the caller).
 * Replaces norm1 -> relu1 -> conv1 -> norm2 -> relu2 of torchvision's _DenseLayer (model.py:53) in one
 * pass over the concatenated features.  x: channel-prefix view, image stride x_batch_stride floats,
 * cin % 32 == 0; wt: device [cin, cout] (the conv weight TRANSPOSED), cout % 128 == 0; bias: device
 * [cout] or NULL; y: device packed NCHW [n, cout, hw].
 */
int mirx_conv1x1_bn_relu(const float *x, int64_t x_batch_stride, int cin, const float *scale, const float *shift,
                         const float *wt, const float *bias, int64_t n, int hw, int cout, int relu_out, float *y,
                         void *stream);

/*
 * ConvNeXt block front end: y = permute_NHWC(depthwise_conv7x7(x) + bias), pad 3, stride 1.
 * Replaces conv_dw + x.permute(0, 2, 3, 1) of timm's ConvNeXtBlock (the convnextv2_base backbone
 * of model.py:96-100).  x: device NCHW fp32 [n, c, h, w]; w: device [c, 1, 7, 7]; bias: device [c]
 * or NULL; y: device NHWC fp32 [n, h, w, c].
 */
int mirx_dwconv7x7_nchw_to_nhwc(const float *x, const float *w, const float *bias, int64_t n, int c, int h,
                                int wd, float *y, void *stream);
/* The same convolution on a channels-last map (the ConvNeXtV2 fast path keeps its residual stream NHWC): x, y = device NHWC fp32
 * [n, h, w, c] (y != x); w_taps_first = device [49][c] (conv_dw.weight.view(c, 49).t(), prepared once per layer); bias [c] or
 * NULL.  No LDS: a lane owns one channel and walks a strip of 3 output rows with a 7-column window in registers. */
int mirx_dwconv7x7_nhwc(const float *x, const float *w_taps_first, const float *bias, int64_t n, int c, int h, int wd, float *y,
                        void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MIRX_H */
