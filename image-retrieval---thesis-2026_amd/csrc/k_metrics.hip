// k_metrics.hip -- the ranked-list metric tail on the device.
//
// One wavefront walks one query's ranking (best first) 64 ids at a time and produces, in one pass:
//   * AP, trapezoidal (compute_ap test.py:58-92 as summed by compute_map test.py:95-146) or
//     standard (sum of precision at each relevant rank / relevant count: compute_map_multilabel
//     test.py:941-985, fusion_eval/metrics.py:41-94),
//   * the number of relevant items within the first kappa ranks for up to 8 kappas
//     (retrieval_accuracy test.py:38-54; the precision@kappa of test.py:137-142 and the mP@k / R@k
//     of fusion_eval/metrics.py are ratios of these),
//   * the relevant count of the list and the largest 1-based relevant rank (the reference's
//     kq = min(max(pos), kappa) quirk needs it).
// Relevance is label equality (int64) or Jaccard(multi-hot bit masks) > threshold, evaluated as
// inter / (union + 1e-8) > thr in fp64 like test.py:958-961; an optional per-query id is never relevant
// (self-exclusion).  Sums are fp64, per-lane partials combined by a fixed butterfly: deterministic.
#include "mirx_kernels.h"

namespace mirx {

namespace {

template <int REL_KIND, int AP_KIND>
__global__ __launch_bounds__(256) void k_rank_metrics(RankMetricsArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= a.nq) return;
    const int64_t *row = a.ranks + q * a.row_stride;
    const int64_t qlab = a.query_labels[q];
    const int64_t self = a.query_ids ? a.query_ids[q] : -1;
    const double qbits = REL_KIND == 1 ? (double)__popcll((unsigned long long)qlab) : 0.0;
    double acc = 0.0;
    int64_t running = 0, maxpos = 0;
    int64_t cnt[MIRX_MAX_KAPPAS];
#pragma unroll
    for (int j = 0; j < MIRX_MAX_KAPPAS; ++j) cnt[j] = 0;

    int64_t dropped = 0;                                     // self entries already passed (drop_self)
    for (int64_t i = 0; i < a.n; i += 64) {
        const int64_t r = i + lane;
        bool rel = false, is_self = false;
        if (r < a.n) {
            const int64_t id = row[r];
            is_self = id == self;
            if (id >= 0 && id < a.n_labels && !is_self) {
                const int64_t glab = a.gallery_labels[id];
                if (REL_KIND == 0) {
                    rel = glab == qlab;
                } else {
                    const double inter = (double)__popcll((unsigned long long)(glab & qlab));
                    const double uni = qbits + (double)__popcll((unsigned long long)glab) - inter;
                    rel = inter / (uni + 1e-8) > a.jaccard_threshold;
                }
            }
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long smask = a.drop_self ? __ballot(is_self) : 0ull;
        const unsigned long long mask = __ballot(rel);
        // 0-based position in the list with the query's own entry taken out (fusion_eval/metrics.py:58-60)
        const int64_t reff = r - dropped - __popcll(smask & below);
        if (mask != 0) {
            if (rel) {
                const double j = (double)(running + __popcll(mask & below));                 // positives before
                const double rr = (double)reff;
                const double p1 = (j + 1.0) / (rr + 1.0);
                if (AP_KIND == 0) {
                    const double p0 = reff == 0 ? 1.0 : j / rr;
                    acc += (p0 + p1) * 0.5;
                } else {
                    acc += p1;
                }
            }
#pragma unroll
            for (int j = 0; j < MIRX_MAX_KAPPAS; ++j)
                if (j < a.nk) cnt[j] += __popcll(__ballot(rel && reff < (int64_t)a.kappas[j]));
            running += __popcll(mask);
            maxpos = __shfl(reff, 63 - __clzll((long long)mask), 64) + 1;
        }
        dropped += __popcll(smask);
    }
    acc = wave_butterfly_sum(acc);
    if (lane == 0) {
        a.out_ap[q] = running > 0 ? acc / (double)running : __longlong_as_double(0x7ff8000000000000ll);
        a.out_nrel[q] = running;
        a.out_maxpos[q] = maxpos;
        for (int j = 0; j < a.nk; ++j) a.out_cnt[q * a.nk + j] = cnt[j];
    }
}

}  // namespace

hipError_t launch_rank_metrics(const RankMetricsArgs &a, int rel_kind, int ap_kind, hipStream_t st) {
    if (a.nq <= 0) return hipSuccess;
    const dim3 grid((unsigned)((a.nq + 3) / 4));
#define MIRX_RM(R, P) hipLaunchKernelGGL((k_rank_metrics<R, P>), grid, dim3(256), 0, st, a)
    if (rel_kind == 0) { if (ap_kind == 0) MIRX_RM(0, 0); else MIRX_RM(0, 1); }
    else               { if (ap_kind == 0) MIRX_RM(1, 0); else MIRX_RM(1, 1); }
#undef MIRX_RM
    return hipGetLastError();
}

}  // namespace mirx
