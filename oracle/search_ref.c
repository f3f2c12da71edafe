/*
 * oracle/search_ref.c -- CPU restatement of the exhaustive search on the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package, libmirx.so) may
 * include, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker.
 *
 * What it restates (reference = /root/reference, read as text only):
 *   - test.py:1080       dists = -torch.cdist(embeds, embeds)            (metric 1)
 *   - test.py:296,893 / test_nonclip.py:151   embeds @ embeds.t()        (metric 0)
 *   - test.py:1081       fill_diagonal_(-inf)  -> per-query excluded id
 *   - test.py:44         output.topk(maxk, 1, True, True)
 *   - test.py:1090,179   torch.argsort(dists, dim=0, descending=True)   (full ranking)
 *   - model.py:83        F.normalize(x, dim=1)   (x / max(||x||_2, 1e-12))
 *
 * The reference computes these in fp32 with library kernels whose tie order is not
 * defined.  The oracle pins the semantics instead:
 *   score   = fp64 accumulation of the fp32 inputs, in the fixed "lane tree" order below
 *   ranking = higher score first, equal scores -> lower gallery id first
 * Metric 1 ranks by the NEGATIVE SQUARED L2 distance (monotone in -cdist); the value
 * reported to callers is -sqrt of it.
 *
 * Lane-tree order (this is the order the HIP re-rank kernel uses, so scores agree bit
 * for bit):  the row is cut into 16-byte chunks of 4 floats; chunk c belongs to lane
 * c % 64; every lane folds its chunks in increasing c, elements x,y,z,w in order, with
 * one fused multiply-add per element into an fp64 accumulator that starts at +0.0;
 * the 64 lane sums are then combined by a butterfly: for off = 32,16,8,4,2,1:
 * s[l] = s[l] + s[l ^ off].  Lane 0's value is the score.  dim must be a multiple of 4.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LANES 64

static double lane_tree(const float *a, const float *b, int dim, int metric)
{
    double s[LANES], t[LANES];
    int nchunk = dim / 4;
    for (int l = 0; l < LANES; ++l) {
        double acc = 0.0;
        for (int c = l; c < nchunk; c += LANES) {
            for (int e = 0; e < 4; ++e) {
                double x = (double)a[4 * c + e], y = (double)b[4 * c + e];
                if (metric == 0) {
                    acc = fma(x, y, acc);
                } else {
                    double d = x - y;
                    acc = fma(d, d, acc);
                }
            }
        }
        s[l] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < LANES; ++l) t[l] = s[l] + s[l ^ off];
        memcpy(s, t, sizeof s);
    }
    return metric == 0 ? s[0] : -s[0];
}

/* out[i*n + j] = ranking score of query i against gallery row j. */
void mirx_oracle_scores(const float *q, int64_t nq, const float *g, int64_t n, int dim,
                        int metric, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; ++i)
        for (int64_t j = 0; j < n; ++j)
            out[i * n + j] = lane_tree(q + i * dim, g + j * dim, dim, metric);
}

typedef struct { double s; int64_t id; } hit_t;

static int better(double sa, int64_t ia, double sb, int64_t ib)
{
    return sa > sb || (sa == sb && ia < ib);
}

static int cmp_hit(const void *pa, const void *pb)
{
    const hit_t *a = (const hit_t *)pa, *b = (const hit_t *)pb;
    if (better(a->s, a->id, b->s, b->id)) return -1;
    if (better(b->s, b->id, a->s, a->id)) return 1;
    return 0;
}

/*
 * Exhaustive top-k.  ids == NULL means gallery row j has id j.  exclude == NULL means no
 * exclusion; otherwise gallery rows whose id equals exclude[i] are skipped for query i
 * (the fill_diagonal_(-inf) of test.py:1081 in Q x N form).  Slots beyond the number of
 * eligible rows are filled with id -1 and score -inf.  out_scores holds the ranking
 * score (metric 0: dot product; metric 1: negative squared distance).
 */
void mirx_oracle_topk(const float *q, int64_t nq, const float *g, int64_t n, int dim,
                      int metric, int k, const int64_t *exclude, const int64_t *ids,
                      double *out_scores, int64_t *out_ids)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < nq; ++i) {
        hit_t *heap = (hit_t *)malloc(sizeof(hit_t) * (size_t)(k > 0 ? k : 1));
        int cnt = 0;
        for (int64_t j = 0; j < n; ++j) {
            int64_t id = ids ? ids[j] : j;
            if (exclude && exclude[i] == id) continue;
            double s = lane_tree(q + i * dim, g + j * dim, dim, metric);
            if (cnt < k) {
                heap[cnt].s = s; heap[cnt].id = id; ++cnt;
                if (cnt == k) qsort(heap, (size_t)k, sizeof(hit_t), cmp_hit);
            } else if (k > 0 && better(s, id, heap[k - 1].s, heap[k - 1].id)) {
                int p = k - 1;
                while (p > 0 && better(s, id, heap[p - 1].s, heap[p - 1].id)) {
                    heap[p] = heap[p - 1]; --p;
                }
                heap[p].s = s; heap[p].id = id;
            }
        }
        if (cnt < k) qsort(heap, (size_t)cnt, sizeof(hit_t), cmp_hit);
        for (int r = 0; r < k; ++r) {
            out_scores[i * k + r] = r < cnt ? heap[r].s : -INFINITY;
            out_ids[i * k + r] = r < cnt ? heap[r].id : -1;
        }
        free(heap);
    }
}

/*
 * Full ranking: out_ids[i*n + r] = gallery id at rank r for query i, excluded row last
 * (the reference puts it last because its score is -inf: test.py:1081,1090).
 */
void mirx_oracle_rank_all(const float *q, int64_t nq, const float *g, int64_t n, int dim,
                          int metric, const int64_t *exclude, int64_t *out_ids,
                          double *out_scores_or_null)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i = 0; i < nq; ++i) {
        hit_t *h = (hit_t *)malloc(sizeof(hit_t) * (size_t)(n > 0 ? n : 1));
        for (int64_t j = 0; j < n; ++j) {
            h[j].id = j;
            h[j].s = (exclude && exclude[i] == j)
                         ? -INFINITY
                         : lane_tree(q + i * dim, g + j * dim, dim, metric);
        }
        qsort(h, (size_t)n, sizeof(hit_t), cmp_hit);
        for (int64_t r = 0; r < n; ++r) {
            out_ids[i * n + r] = h[r].id;
            if (out_scores_or_null) out_scores_or_null[i * n + r] = h[r].s;
        }
        free(h);
    }
}

/*
 * F.normalize(x, dim=1) restated (model.py:83): sum of squares in the lane-tree order in
 * fp64, norm = sqrt, y = (float)((double)x / max(norm, 1e-12)).
 */
void mirx_oracle_l2_normalize(const float *x, int64_t n, int dim, float *y)
{
    float *zero = (float *)calloc((size_t)dim, sizeof(float));
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        /* metric 1 against the zero row gives -(sum of squares) */
        double ss = -lane_tree(x + i * dim, zero, dim, 1);
        double nrm = sqrt(ss);
        if (nrm < 1e-12) nrm = 1e-12;
        for (int d = 0; d < dim; ++d) y[i * dim + d] = (float)((double)x[i * dim + d] / nrm);
    }
    free(zero);
}

/* fp32 -> bf16 round-to-nearest-even of finite values, as the index stores its MFMA copy. */
uint16_t mirx_oracle_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
