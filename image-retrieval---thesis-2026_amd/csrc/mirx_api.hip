// mirx_api.hip -- the C ABI (include/mirx.h): index object, search orchestration.
//
// A search is a fixed sequence of launches on the caller's stream:
//   prep queries -> [group-max GEMM on a strided row sample -> thresholds] -> filter GEMM
//   -> finalize (sort candidates, completeness guard, fp64 re-rank, top-k)
//   -> exact scan (fp64 score tiles + row top-k) for the queries the guard rejected.
// The only host<->device synchronisation is one read of the rejected-query count.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <mutex>
#include <vector>

#include "mirx_kernels.h"

namespace mirx {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p);
            p = nullptr;
            bytes = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = need + need / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            e = hipMalloc(&p, need);
            want = need;
        }
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T> T *as() const { return reinterpret_cast<T *>(p); }
};

}  // namespace mirx

using namespace mirx;

struct mirx_index {
    int dim = 0, dimp = 0, metric = 0, device = 0;
    int64_t size = 0, cap = 0;
    float *g32 = nullptr;
    uint16_t *g16 = nullptr;
    int64_t *ids = nullptr;
    float *gbias = nullptr;
    unsigned *gnorm_max_bits = nullptr;   // device scalar
    // options
    int tiers = MIRX_TIER_AUTO;
    int sample_rank = 8;
    uint32_t force_tau_bits = 0x7fc00000u;
    int profile = 0;
    std::vector<hipEvent_t> ev_pool;      // lazily created, reused
    struct Span { int stage; hipEvent_t a, b; };
    std::vector<Span> spans;              // of the last search
    size_t ev_used = 0;
    // workspace
    DevBuf q32p, q16, qnorm, tau, cnt, cand, ovf_cnt, ovf, groupmax, fail_list, retry_list, tau2, q16r, taur,
        scores, stage, rankwork, spill;
    int *fail_count = nullptr;            // device
    mirx_search_stats *stats_dev = nullptr;
    int *fail_count_host = nullptr;       // pinned
    mirx_search_stats stats_host{};
    // a search whose first pass is enqueued and whose counters have not been read yet (search_begin / search_end)
    struct Pending {
        bool active = false;
        int64_t nb = 0;
        int k = 0;
        const int64_t *exb = nullptr;
        double *ofb = nullptr;
        int64_t *oib = nullptr;
        float *ovb = nullptr;
        hipStream_t st = nullptr;
        mirx::GemmArgs ga{};
        mirx::FinalizeArgs fa{};
    } pend;
    hipEvent_t pend_ev = nullptr;
};

namespace {

constexpr int64_t QUERY_BATCH = 8192;          // queries per internal pass
constexpr int64_t TIER1_MIN_ROWS = 32768;      // below this the exact scan answers directly
constexpr int TIER1_MAX_K = 64;
constexpr size_t EXACT_WS_BYTES = (size_t)1 << 30;   // fp64 score tile workspace per pass

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int is_device_pointer(const void *p, bool *dev) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();   // unregistered host memory reports an error: treat as host
        *dev = false;
        return MIRX_OK;
    }
    *dev = (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged);
    return MIRX_OK;
}

int grow(mirx_index *ix, int64_t want_rows) {
    if (want_rows <= ix->cap) return MIRX_OK;
    int64_t ncap = std::max<int64_t>(ix->cap * 2, want_rows);
    ncap = round_up(std::max<int64_t>(ncap, ROW_ALIGN), ROW_ALIGN);
    float *n32 = nullptr;
    uint16_t *n16 = nullptr;
    int64_t *nid = nullptr;
    float *nb = nullptr;
    const size_t b32 = (size_t)ncap * ix->dimp * sizeof(float);
    const size_t b16 = (size_t)ncap * ix->dimp * sizeof(uint16_t);
    hipError_t e;
    if ((e = hipMalloc(&n32, b32)) != hipSuccess || (e = hipMalloc(&n16, b16)) != hipSuccess ||
        (e = hipMalloc(&nid, (size_t)ncap * sizeof(int64_t))) != hipSuccess ||
        (e = hipMalloc(&nb, (size_t)ncap * sizeof(float))) != hipSuccess) {
        if (n32) (void)hipFree(n32);
        if (n16) (void)hipFree(n16);
        if (nid) (void)hipFree(nid);
        if (nb) (void)hipFree(nb);
        return fail(MIRX_ENOMEM, std::string("index grow: ") + hipGetErrorString(e));
    }
    // rows past `size` must read as zeros (tiles run over the padded tail)
    MIRX_HIP(hipMemset(n32, 0, b32));
    MIRX_HIP(hipMemset(n16, 0, b16));
    MIRX_HIP(hipMemset(nid, 0xFF, (size_t)ncap * sizeof(int64_t)));
    MIRX_HIP(hipMemset(nb, 0, (size_t)ncap * sizeof(float)));
    if (ix->size > 0) {
        MIRX_HIP(hipMemcpy(n32, ix->g32, (size_t)ix->size * ix->dimp * sizeof(float), hipMemcpyDeviceToDevice));
        MIRX_HIP(hipMemcpy(n16, ix->g16, (size_t)ix->size * ix->dimp * sizeof(uint16_t), hipMemcpyDeviceToDevice));
        MIRX_HIP(hipMemcpy(nid, ix->ids, (size_t)ix->size * sizeof(int64_t), hipMemcpyDeviceToDevice));
        MIRX_HIP(hipMemcpy(nb, ix->gbias, (size_t)ix->size * sizeof(float), hipMemcpyDeviceToDevice));
    }
    if (ix->g32) (void)hipFree(ix->g32);
    if (ix->g16) (void)hipFree(ix->g16);
    if (ix->ids) (void)hipFree(ix->ids);
    if (ix->gbias) (void)hipFree(ix->gbias);
    ix->g32 = n32;
    ix->g16 = n16;
    ix->ids = nid;
    ix->gbias = nb;
    ix->cap = ncap;
    return MIRX_OK;
}

struct StageTimer {
    mirx_index *ix;
    hipStream_t st;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(mirx_index *ix_, hipStream_t st_, int stage_) : ix(ix_), st(st_), stage(stage_) {
        if (!ix->profile) return;
        a = next();
        b = next();
        if (a && b) (void)hipEventRecord(a, st);
    }
    hipEvent_t next() {
        if (ix->ev_used == ix->ev_pool.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ix->ev_pool.push_back(e);
        }
        return ix->ev_pool[ix->ev_used++];
    }
    ~StageTimer() {
        if (a && b) {
            (void)hipEventRecord(b, st);
            ix->spans.push_back({stage, a, b});
        }
    }
};

// Exact scan for `nlist` queries given by a device list (or 0..nlist-1 when list == null).
int exact_pass(mirx_index *ix, const float *q32p, const int32_t *list_dev, int64_t nlist, int k,
               const int64_t *exclude, double *out_f64, int64_t *out_ids, float *out_val,
               hipStream_t st) {
    if (nlist <= 0) return MIRX_OK;
    const int64_t ld = round_up(std::max<int64_t>(ix->size, 1), 64);
    int64_t per = (int64_t)(EXACT_WS_BYTES / ((size_t)ld * sizeof(double)));
    per = std::max<int64_t>(4, std::min<int64_t>(per, 4096)) / 4 * 4;
    per = std::min<int64_t>(per, round_up(nlist, 4));
    MIRX_HIP(ix->scores.ensure((size_t)per * ld * sizeof(double)));
    for (int64_t b = 0; b < nlist; b += per) {
        const int cnt = (int)std::min<int64_t>(per, nlist - b);
        const int32_t *lst = list_dev ? list_dev + b : nullptr;
        // without a list the query index is the position: shift the base pointers instead
        const float *qb = list_dev ? q32p : q32p + b * ix->dimp;
        const int64_t *exb = exclude ? (list_dev ? exclude : exclude + b) : nullptr;
        double *of = list_dev ? out_f64 : out_f64 + b * k;
        int64_t *oi = list_dev ? out_ids : out_ids + b * k;
        float *ov = out_val ? (list_dev ? out_val : out_val + b * k) : nullptr;
        MIRX_HIP(launch_scores_f64(qb, lst, cnt, ix->g32, ix->size, ix->dimp, ix->metric,
                                   ix->scores.as<double>(), ld, st));
        MIRX_HIP(launch_row_topk(ix->scores.as<double>(), ld, ix->size, ix->ids, lst, cnt, exb, k,
                                 ix->metric, of, oi, ov, st));
    }
    return MIRX_OK;
}

// One internal batch (<= QUERY_BATCH queries), first half: everything that needs no host decision -- prepare the queries,
// sample the thresholds, filter GEMM, finalize -- then the two counters (queries to retry / to scan exactly) start their
// way to pinned host memory and an event marks that point.  Nothing here blocks the host.
int batch_front(mirx_index *ix, const float *qb, int64_t nb, int k, const int64_t *exb, float *ovb, double *ofb,
                int64_t *oib, bool tier1, hipStream_t st) {
    const int bn = gemm_query_tile(nb);
    const int64_t nb_pad = round_up(nb, bn);
    ix->pend.active = false;
    MIRX_HIP(ix->q32p.ensure((size_t)nb_pad * ix->dimp * sizeof(float)));
    MIRX_HIP(ix->q16.ensure((size_t)nb_pad * ix->dimp * sizeof(uint16_t)));
    MIRX_HIP(ix->qnorm.ensure((size_t)nb_pad * sizeof(float)));
    {
        StageTimer t(ix, st, MIRX_STAGE_PREP);
        MIRX_HIP(launch_prep_queries(qb, nb, nb_pad, ix->dim, ix->dimp, ix->q32p.as<float>(),
                                     ix->q16.as<uint16_t>(), ix->qnorm.as<float>(), st));
    }
    if (!tier1) {
        // (an empty gallery lands here too: every slot becomes (-1, -inf))
        StageTimer t(ix, st, MIRX_STAGE_EXACT);
        int rc = exact_pass(ix, ix->q32p.as<float>(), nullptr, nb, k, exb, ofb, oib, ovb, st);
        if (rc) return rc;
        ix->stats_host.exact_answered += nb;
        return MIRX_OK;
    }

    // ---- tier 1 ---------------------------------------------------------------------
    MIRX_HIP(ix->tau.ensure((size_t)nb_pad * sizeof(float)));
    int regions = 0, slots = 0;
    gemm_plan(ix->size, nb_pad, bn, &regions, &slots);
    MIRX_HIP(ix->cnt.ensure((size_t)nb_pad * regions * sizeof(int)));
    MIRX_HIP(ix->cand.ensure((size_t)nb_pad * regions * slots * sizeof(Cand)));
    MIRX_HIP(ix->ovf_cnt.ensure((size_t)nb_pad * sizeof(int)));
    MIRX_HIP(ix->ovf.ensure((size_t)nb_pad * CAND_OVF * sizeof(Cand)));
    MIRX_HIP(ix->spill.ensure(gemm_spill_bytes()));
    MIRX_HIP(ix->fail_list.ensure((size_t)nb_pad * sizeof(int32_t)));
    MIRX_HIP(ix->retry_list.ensure((size_t)nb_pad * sizeof(int32_t)));
    MIRX_HIP(ix->tau2.ensure((size_t)nb_pad * sizeof(float)));
    GemmArgs ga{};
    ga.g16 = ix->g16;
    ga.q16 = ix->q16.as<uint16_t>();
    ga.gbias = ix->metric == MIRX_METRIC_NEG_L2 ? ix->gbias : nullptr;
    ga.nq_pad = nb_pad;
    ga.dimp = ix->dimp;
    ga.tau = ix->tau.as<float>();
    ga.regions = regions;
    ga.slots = slots;
    ga.region_cnt = ix->cnt.as<int>();
    ga.cand = ix->cand.as<Cand>();
    ga.ovf_cnt = ix->ovf_cnt.as<int>();
    ga.ovf = ix->ovf.as<Cand>();
    ga.spill = ix->spill.p;
    if (ix->force_tau_bits != 0x7fc00000u) {
        // test hook: one fixed threshold for every query
        std::vector<float> t((size_t)nb_pad, INFINITY);
        float v;
        std::memcpy(&v, &ix->force_tau_bits, 4);
        std::fill(t.begin(), t.begin() + nb, v);
        MIRX_HIP(hipMemcpyAsync(ix->tau.p, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice, st));
        MIRX_HIP(hipStreamSynchronize(st));
    } else {
        // sample = every `stride`-th row, about size/32 rows, a multiple of the 256-row tile; at most
        // 16384 group maxima per query (the selection sorts them in LDS).  The threshold is the j-th
        // largest sampled group maximum with j ~ 256 / stride, i.e. about 256 expected candidates per
        // query whatever the gallery size (j = sample_rank = 8 at the usual stride of 32).
        const int gpt = gemm_groups_per_tile(bn);
        int64_t ms = round_up(std::max<int64_t>(ix->size / 32, 4096), ROW_ALIGN);
        ms = std::min<int64_t>(ms, (int64_t)(16384 / gpt) * ROW_ALIGN);
        const int64_t stride = std::max<int64_t>(ix->size / ms, 1);
        const int ngroups = (int)(ms / ROW_ALIGN) * gpt;
        const int rank_j = (int)std::max<int64_t>(1, std::min<int64_t>(ix->sample_rank, (32 * (int64_t)ix->sample_rank) / stride));
        MIRX_HIP(ix->groupmax.ensure((size_t)nb_pad * ngroups * sizeof(float)));
        GemmArgs gs = ga;
        gs.n_rows = ms;
        gs.row_stride = stride;
        gs.groupmax = ix->groupmax.as<float>();
        gs.ngroups = ngroups;
        StageTimer t(ix, st, MIRX_STAGE_SAMPLE);
        MIRX_HIP(launch_gemm_groupmax(gs, bn, st));
        MIRX_HIP(launch_select_tau(ix->groupmax.as<float>(), ngroups, nb, nb_pad, rank_j, ix->tau.as<float>(), st));
    }
    MIRX_HIP(hipMemsetAsync(ix->cnt.p, 0, (size_t)nb_pad * regions * sizeof(int), st));
    MIRX_HIP(hipMemsetAsync(ix->ovf_cnt.p, 0, (size_t)nb_pad * sizeof(int), st));
    MIRX_HIP(hipMemsetAsync(ix->fail_count, 0, 2 * sizeof(int), st));
    ga.n_rows = ix->size;
    ga.row_stride = 1;
    {
        StageTimer t(ix, st, MIRX_STAGE_GEMM);
        MIRX_HIP(launch_gemm_filter(ga, bn, st));
    }

    FinalizeArgs fa{};
    fa.q32p = ix->q32p.as<float>();
    fa.qnorm = ix->qnorm.as<float>();
    fa.g32 = ix->g32;
    fa.ids = ix->ids;
    fa.gnorm_max_bits = ix->gnorm_max_bits;
    fa.tau = ix->tau.as<float>();
    fa.regions = regions;
    fa.slots = slots;
    fa.region_cnt = ix->cnt.as<int>();
    fa.cand = ix->cand.as<Cand>();
    fa.ovf_cnt = ix->ovf_cnt.as<int>();
    fa.ovf = ix->ovf.as<Cand>();
    fa.exclude = exb;
    fa.n_rows = ix->size;
    fa.dimp = ix->dimp;
    fa.k = k;
    fa.metric = ix->metric;
    fa.nq = (int)nb;
    fa.out_f64 = ofb;
    fa.out_ids = oib;
    fa.out_val = ovb;
    fa.fail_list = ix->fail_list.as<int32_t>();
    fa.fail_count = ix->fail_count;
    fa.retry_list = ix->retry_list.as<int32_t>();
    fa.tau2 = ix->tau2.as<float>();
    fa.qmap = nullptr;
    fa.stats = ix->stats_dev;
    {
        StageTimer t(ix, st, MIRX_STAGE_FINALIZE);
        MIRX_HIP(launch_finalize(fa, st));
    }
    MIRX_HIP(hipMemcpyAsync(ix->fail_count_host, ix->fail_count, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    if (!ix->pend_ev) MIRX_HIP(hipEventCreateWithFlags(&ix->pend_ev, hipEventDisableTiming));
    MIRX_HIP(hipEventRecord(ix->pend_ev, st));
    ix->pend.active = true;
    ix->pend.nb = nb;
    ix->pend.k = k;
    ix->pend.exb = exb;
    ix->pend.ofb = ofb;
    ix->pend.oib = oib;
    ix->pend.ovb = ovb;
    ix->pend.st = st;
    ix->pend.ga = ga;
    ix->pend.fa = fa;
    return MIRX_OK;
}

// Second half: wait (host only, on the event -- the stream is not drained) for the two counters and run what they ask for:
// the second-chance filter for queries whose threshold was too high, the exact scan for what is still unanswered.  On the
// bench workload both counters are zero and this returns after the event.
int batch_back(mirx_index *ix) {
    if (!ix->pend.active) return MIRX_OK;
    ix->pend.active = false;
    hipStream_t st = ix->pend.st;
    const GemmArgs &ga = ix->pend.ga;
    const FinalizeArgs &fa = ix->pend.fa;
    const int k = ix->pend.k;
    MIRX_HIP(hipEventSynchronize(ix->pend_ev));
    const int nretry = ix->fail_count_host[1];
    if (nretry > 0) {
        // second chance: same filter GEMM over the gathered queries with tau2 = kth(s~) - 2*eps
        const int bn2 = gemm_query_tile(nretry);
        const int64_t nr_pad = round_up(nretry, bn2);
        int regions2 = 0, slots2 = 0;
        gemm_plan(ix->size, nr_pad, bn2, &regions2, &slots2);
        MIRX_HIP(ix->q16r.ensure((size_t)nr_pad * ix->dimp * sizeof(uint16_t)));
        MIRX_HIP(ix->taur.ensure((size_t)nr_pad * sizeof(float)));
        MIRX_HIP(ix->cnt.ensure((size_t)nr_pad * regions2 * sizeof(int)));
        MIRX_HIP(ix->cand.ensure((size_t)nr_pad * regions2 * slots2 * sizeof(Cand)));
        MIRX_HIP(launch_gather_queries(ix->q16.as<uint16_t>(), ix->tau2.as<float>(), ix->retry_list.as<int32_t>(),
                                       nretry, nr_pad, ix->dimp, ix->q16r.as<uint16_t>(), ix->taur.as<float>(),
                                       st));
        MIRX_HIP(hipMemsetAsync(ix->cnt.p, 0, (size_t)nr_pad * regions2 * sizeof(int), st));
        MIRX_HIP(hipMemsetAsync(ix->ovf_cnt.p, 0, (size_t)nr_pad * sizeof(int), st));
        GemmArgs g2 = ga;
        g2.q16 = ix->q16r.as<uint16_t>();
        g2.nq_pad = nr_pad;
        g2.tau = ix->taur.as<float>();
        g2.regions = regions2;
        g2.slots = slots2;
        g2.region_cnt = ix->cnt.as<int>();
        g2.cand = ix->cand.as<Cand>();
        {
            StageTimer t(ix, st, MIRX_STAGE_GEMM);
            MIRX_HIP(launch_gemm_filter(g2, bn2, st));
        }
        FinalizeArgs f2 = fa;
        f2.tau = ix->taur.as<float>();
        f2.regions = regions2;
        f2.slots = slots2;
        f2.region_cnt = ix->cnt.as<int>();
        f2.cand = ix->cand.as<Cand>();
        f2.nq = nretry;
        f2.retry_list = nullptr;          // no third chance: what fails now goes to the exact scan
        f2.tau2 = nullptr;
        f2.qmap = ix->retry_list.as<int32_t>();
        {
            StageTimer t(ix, st, MIRX_STAGE_FINALIZE);
            MIRX_HIP(launch_finalize(f2, st));
        }
        MIRX_HIP(hipMemcpyAsync(ix->fail_count_host, ix->fail_count, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        MIRX_HIP(hipEventRecord(ix->pend_ev, st));
        MIRX_HIP(hipEventSynchronize(ix->pend_ev));
    }
    const int nfail = ix->fail_count_host[0];
    if (nfail > 0) {
        StageTimer t(ix, st, MIRX_STAGE_EXACT);
        int rc = exact_pass(ix, ix->q32p.as<float>(), ix->fail_list.as<int32_t>(), nfail, k, ix->pend.exb, ix->pend.ofb,
                            ix->pend.oib, ix->pend.ovb, st);
        if (rc) return rc;
        ix->stats_host.exact_answered += nfail;
    }
    return MIRX_OK;
}

// wait_last = false: the last internal batch is left pending (mirx_index_search_begin); batch_back completes it.
int search_impl(mirx_index *ix, const float *q, int64_t nq, int k, const int64_t *exclude,
                float *out_val, double *out_f64_user, int64_t *out_ids, hipStream_t st, bool wait_last = true) {
    MIRX_CHECK(ix, "search: null index");
    MIRX_CHECK(!ix->pend.active, "search: a search begun with mirx_index_search_begin is still pending (call mirx_index_search_end)");
    MIRX_CHECK(k >= 1 && k <= MAX_K, "search: k must be in [1, 1024]");
    MIRX_CHECK(nq >= 0, "search: nq < 0");
    MIRX_CHECK(nq == 0 || (q && out_ids && (out_val || out_f64_user)), "search: null buffer");
    DeviceGuard dg(ix->device);
    if (!dg.ok) return fail(MIRX_EHIP, "search: cannot select the index device");
    MIRX_HIP(hipMemsetAsync(ix->stats_dev, 0, sizeof(mirx_search_stats), st));
    ix->stats_host = mirx_search_stats{};
    ix->stats_host.nq = nq;
    ix->spans.clear();
    ix->ev_used = 0;
    if (nq == 0) return MIRX_OK;

    const bool tier1 = ix->tiers == MIRX_TIER_AUTO && ix->size >= TIER1_MIN_ROWS &&
                       k + (exclude ? 1 : 0) <= TIER1_MAX_K;
    for (int64_t q0 = 0; q0 < nq; q0 += QUERY_BATCH) {
        const int64_t nb = std::min<int64_t>(QUERY_BATCH, nq - q0);
        const float *qb = q + q0 * ix->dim;
        const int64_t *exb = exclude ? exclude + q0 : nullptr;
        float *ovb = out_val ? out_val + q0 * k : nullptr;
        int64_t *oib = out_ids + q0 * k;
        // fp64 ranking scores are always produced (user buffer or workspace)
        double *ofb;
        if (out_f64_user) {
            ofb = out_f64_user + q0 * k;
        } else {
            MIRX_HIP(ix->stage.ensure((size_t)nb * k * sizeof(double)));
            ofb = ix->stage.as<double>();
        }
        int rc = batch_front(ix, qb, nb, k, exb, ovb, ofb, oib, tier1, st);
        if (rc) return rc;
        if (wait_last || q0 + QUERY_BATCH < nq) {
            rc = batch_back(ix);
            if (rc) return rc;
        }
    }
    return MIRX_OK;
}

}  // namespace

extern "C" {

const char *mirx_last_error(void) { return g_err.c_str(); }
int mirx_version(void) { return MIRX_VERSION; }

int mirx_set_tuning(int key, int64_t value) {
    switch (key) {
        case MIRX_TUNE_CONV1X1_SMALL_MAX_WG:
            MIRX_CHECK(value >= 0 && value <= (1 << 20), "set_tuning: conv1x1 small-launch limit out of range");
            set_conv1x1_small_max_wg((int)value);
            return MIRX_OK;
        case MIRX_TUNE_CONV3X3_SMALL_MAX_WG:
            MIRX_CHECK(value >= 0 && value <= (1 << 20), "set_tuning: conv3x3 small-launch limit out of range");
            set_conv3x3_small_max_wg((int)value);
            return MIRX_OK;
        default:
            return fail(MIRX_EINVAL, "set_tuning: unknown key");
    }
}

int mirx_index_create(int dim, int metric, int device, mirx_index **out) {
    MIRX_CHECK(out, "index_create: out is null");
    *out = nullptr;
    MIRX_CHECK(dim >= 1 && round_up(dim, DIM_ALIGN) <= MAX_DIMP, "index_create: dim out of range");
    MIRX_CHECK(metric == MIRX_METRIC_IP || metric == MIRX_METRIC_NEG_L2, "index_create: unknown metric");
    int ndev = 0;
    MIRX_HIP(hipGetDeviceCount(&ndev));
    MIRX_CHECK(device >= 0 && device < ndev, "index_create: no such device");
    DeviceGuard dg(device);
    if (!dg.ok) return fail(MIRX_EHIP, "index_create: cannot select device");
    mirx_index *ix = new (std::nothrow) mirx_index();
    if (!ix) return fail(MIRX_ENOMEM, "index_create: host allocation failed");
    ix->dim = dim;
    ix->dimp = (int)round_up(dim, DIM_ALIGN);
    ix->metric = metric;
    ix->device = device;
    hipError_t e;
    if ((e = hipMalloc(&ix->gnorm_max_bits, sizeof(unsigned))) != hipSuccess ||
        (e = hipMemset(ix->gnorm_max_bits, 0, sizeof(unsigned))) != hipSuccess ||
        (e = hipMalloc(&ix->fail_count, 2 * sizeof(int))) != hipSuccess ||
        (e = hipMalloc(&ix->stats_dev, sizeof(mirx_search_stats))) != hipSuccess ||
        (e = hipMemset(ix->stats_dev, 0, sizeof(mirx_search_stats))) != hipSuccess ||
        (e = hipHostMalloc(&ix->fail_count_host, 2 * sizeof(int))) != hipSuccess) {
        mirx_index_destroy(ix);
        return fail(MIRX_ENOMEM, std::string("index_create: ") + hipGetErrorString(e));
    }
    *out = ix;
    return MIRX_OK;
}

void mirx_index_destroy(mirx_index *ix) {
    if (!ix) return;
    DeviceGuard dg(ix->device);
    if (ix->g32) (void)hipFree(ix->g32);
    if (ix->g16) (void)hipFree(ix->g16);
    if (ix->ids) (void)hipFree(ix->ids);
    if (ix->gbias) (void)hipFree(ix->gbias);
    if (ix->gnorm_max_bits) (void)hipFree(ix->gnorm_max_bits);
    if (ix->fail_count) (void)hipFree(ix->fail_count);
    if (ix->stats_dev) (void)hipFree(ix->stats_dev);
    if (ix->fail_count_host) (void)hipHostFree(ix->fail_count_host);
    for (hipEvent_t e : ix->ev_pool) (void)hipEventDestroy(e);
    if (ix->pend_ev) (void)hipEventDestroy(ix->pend_ev);
    for (DevBuf *b : {&ix->q32p, &ix->q16, &ix->qnorm, &ix->tau, &ix->cnt, &ix->cand, &ix->ovf_cnt, &ix->ovf, &ix->groupmax,
                      &ix->fail_list, &ix->retry_list, &ix->tau2, &ix->q16r, &ix->taur, &ix->scores, &ix->stage,
                      &ix->rankwork, &ix->spill})
        b->release();
    delete ix;
}

int mirx_index_reserve(mirx_index *ix, int64_t capacity_rows) {
    MIRX_CHECK(ix && capacity_rows >= 0, "index_reserve: bad argument");
    DeviceGuard dg(ix->device);
    return grow(ix, capacity_rows);
}

int64_t mirx_index_size(const mirx_index *ix) { return ix ? ix->size : -1; }
int mirx_index_dim(const mirx_index *ix) { return ix ? ix->dim : -1; }

int mirx_index_set_option(mirx_index *ix, int option, int64_t value) {
    MIRX_CHECK(ix, "set_option: null index");
    switch (option) {
        case MIRX_OPT_TIERS:
            MIRX_CHECK(value == MIRX_TIER_AUTO || value == MIRX_TIER_EXACT_ONLY, "set_option: bad tier");
            ix->tiers = (int)value;
            return MIRX_OK;
        case MIRX_OPT_SAMPLE_RANK:
            MIRX_CHECK(value >= 1 && value <= 256, "set_option: sample rank out of range");
            ix->sample_rank = (int)value;
            return MIRX_OK;
        case MIRX_OPT_FORCE_TAU:
            ix->force_tau_bits = (uint32_t)value;
            return MIRX_OK;
        case MIRX_OPT_PROFILE:
            ix->profile = value ? 1 : 0;
            return MIRX_OK;
        default:
            return fail(MIRX_EINVAL, "set_option: unknown option");
    }
}

int mirx_index_add(mirx_index *ix, const float *rows, int64_t n, const int64_t *ids_or_null) {
    MIRX_CHECK(ix && n >= 0 && (rows || n == 0), "index_add: bad argument");
    if (n == 0) return MIRX_OK;
    MIRX_CHECK(ix->size + n <= 0x7FFFFF00ll, "index_add: more than 2^31 rows per index");
    DeviceGuard dg(ix->device);
    if (!dg.ok) return fail(MIRX_EHIP, "index_add: cannot select device");
    int rc = grow(ix, ix->size + n);
    if (rc) return rc;
    bool rows_dev = false, ids_dev = false;
    is_device_pointer(rows, &rows_dev);
    if (ids_or_null) is_device_pointer(ids_or_null, &ids_dev);
    const int64_t chunk = std::max<int64_t>(1, ((int64_t)256 << 20) / ((int64_t)ix->dim * 4));
    for (int64_t b = 0; b < n; b += chunk) {
        const int64_t m = std::min(chunk, n - b);
        const float *src = rows + b * ix->dim;
        if (!rows_dev) {
            MIRX_HIP(ix->stage.ensure((size_t)m * ix->dim * sizeof(float)));
            MIRX_HIP(hipMemcpy(ix->stage.p, src, (size_t)m * ix->dim * sizeof(float), hipMemcpyHostToDevice));
            src = ix->stage.as<float>();
        }
        const int64_t at = ix->size + b;
        MIRX_HIP(launch_ingest(src, m, ix->dim, ix->dimp, ix->g32 + at * ix->dimp, ix->g16 + at * ix->dimp,
                               ix->gbias + at, ix->gnorm_max_bits, ix->metric, nullptr));
        MIRX_HIP(hipStreamSynchronize(nullptr));
    }
    if (ids_or_null) {
        MIRX_HIP(hipMemcpy(ix->ids + ix->size, ids_or_null, (size_t)n * sizeof(int64_t),
                           ids_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    } else {
        std::vector<int64_t> auto_ids((size_t)n);
        for (int64_t i = 0; i < n; ++i) auto_ids[(size_t)i] = ix->size + i;
        MIRX_HIP(hipMemcpy(ix->ids + ix->size, auto_ids.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    ix->size += n;
    return MIRX_OK;
}

int mirx_index_get_rows(const mirx_index *ix, int64_t first, int64_t n, float *out_rows,
                        int64_t *out_ids_or_null) {
    MIRX_CHECK(ix && out_rows && first >= 0 && n >= 0 && first + n <= ix->size, "get_rows: bad range");
    DeviceGuard dg(ix->device);
    MIRX_HIP(hipMemcpy2D(out_rows, (size_t)ix->dim * sizeof(float), ix->g32 + first * ix->dimp,
                         (size_t)ix->dimp * sizeof(float), (size_t)ix->dim * sizeof(float), (size_t)n,
                         hipMemcpyDefault));
    if (out_ids_or_null)
        MIRX_HIP(hipMemcpy(out_ids_or_null, ix->ids + first, (size_t)n * sizeof(int64_t), hipMemcpyDefault));
    return MIRX_OK;
}

int mirx_index_search(mirx_index *ix, const float *q, int64_t nq, int k, const int64_t *exclude_ids_or_null,
                      float *out_scores, int64_t *out_ids, void *stream) {
    return search_impl(ix, q, nq, k, exclude_ids_or_null, out_scores, nullptr, out_ids,
                       reinterpret_cast<hipStream_t>(stream));
}

int mirx_index_search_f64(mirx_index *ix, const float *q, int64_t nq, int k,
                          const int64_t *exclude_ids_or_null, double *out_rank_scores, int64_t *out_ids,
                          void *stream) {
    return search_impl(ix, q, nq, k, exclude_ids_or_null, nullptr, out_rank_scores, out_ids,
                       reinterpret_cast<hipStream_t>(stream));
}

int mirx_index_search_begin(mirx_index *ix, const float *q, int64_t nq, int k, const int64_t *exclude_ids_or_null,
                            float *out_scores_or_null, double *out_rank_scores_or_null, int64_t *out_ids, void *stream) {
    return search_impl(ix, q, nq, k, exclude_ids_or_null, out_scores_or_null, out_rank_scores_or_null, out_ids,
                       reinterpret_cast<hipStream_t>(stream), /*wait_last=*/false);
}

int mirx_index_search_end(mirx_index *ix) {
    MIRX_CHECK(ix, "search_end: null index");
    DeviceGuard dg(ix->device);
    if (!dg.ok) return fail(MIRX_EHIP, "search_end: cannot select the index device");
    return batch_back(ix);
}

int mirx_index_last_stats(mirx_index *ix, void *stream, mirx_search_stats *out) {
    MIRX_CHECK(ix && out, "last_stats: null argument");
    DeviceGuard dg(ix->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MIRX_HIP(hipStreamSynchronize(st));
    mirx_search_stats d{};
    MIRX_HIP(hipMemcpy(&d, ix->stats_dev, sizeof(d), hipMemcpyDeviceToHost));
    *out = d;
    out->nq = ix->stats_host.nq;
    out->exact_answered = ix->stats_host.exact_answered;
    return MIRX_OK;
}

int mirx_index_last_timings(mirx_index *ix, float *out_ms) {
    MIRX_CHECK(ix && out_ms, "last_timings: null argument");
    DeviceGuard dg(ix->device);
    for (int i = 0; i < MIRX_NUM_STAGES; ++i) out_ms[i] = 0.0f;
    for (const auto &sp : ix->spans) {
        MIRX_HIP(hipEventSynchronize(sp.b));
        float ms = 0.0f;
        MIRX_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        out_ms[sp.stage] += ms;
    }
    return MIRX_OK;
}

int mirx_index_rank_all(mirx_index *ix, const float *q, int64_t nq, const int64_t *exclude_ids_or_null,
                        int64_t *out_ids, float *out_scores_or_null, void *stream) {
    MIRX_CHECK(ix && q && out_ids && nq >= 0, "rank_all: bad argument");
    MIRX_CHECK(ix->size <= 65536, "rank_all: gallery larger than 65536 rows");
    if (nq == 0 || ix->size == 0) return MIRX_OK;
    DeviceGuard dg(ix->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int64_t n = ix->size;
    int64_t np2 = 1;
    while (np2 < n) np2 <<= 1;
    const int64_t ld = round_up(n, 64);
    int64_t per = std::max<int64_t>(4, std::min<int64_t>(1024, (int64_t)((size_t)1 << 29) / (np2 * 16))) / 4 * 4;
    per = std::min<int64_t>(per, round_up(nq, 4));
    MIRX_HIP(ix->scores.ensure((size_t)per * ld * sizeof(double)));
    MIRX_HIP(ix->rankwork.ensure((size_t)per * np2 * sizeof(Hit)));
    MIRX_HIP(ix->q32p.ensure((size_t)per * ix->dimp * sizeof(float)));
    MIRX_HIP(ix->q16.ensure((size_t)per * ix->dimp * sizeof(uint16_t)));
    MIRX_HIP(ix->qnorm.ensure((size_t)per * sizeof(float)));
    for (int64_t b = 0; b < nq; b += per) {
        const int cnt = (int)std::min<int64_t>(per, nq - b);
        MIRX_HIP(launch_prep_queries(q + b * ix->dim, cnt, cnt, ix->dim, ix->dimp, ix->q32p.as<float>(),
                                     ix->q16.as<uint16_t>(), ix->qnorm.as<float>(), st));
        MIRX_HIP(launch_scores_f64(ix->q32p.as<float>(), nullptr, cnt, ix->g32, n, ix->dimp, ix->metric,
                                   ix->scores.as<double>(), ld, st));
        MIRX_HIP(launch_rank_rows(ix->scores.as<double>(), ld, n, ix->ids,
                                  exclude_ids_or_null ? exclude_ids_or_null + b : nullptr, cnt,
                                  ix->rankwork.as<Hit>(), np2, ix->metric, out_ids + b * n,
                                  out_scores_or_null ? out_scores_or_null + b * n : nullptr, st));
    }
    return MIRX_OK;
}

int mirx_topk_merge(const double *in_scores, const int64_t *in_ids, int nshard, int64_t nq, int k,
                    int metric, double *out_rank_scores, float *out_scores, int64_t *out_ids, void *stream) {
    MIRX_CHECK(in_scores && in_ids && out_ids && nshard >= 1 && nq >= 0 && k >= 1, "topk_merge: bad argument");
    MIRX_CHECK((int64_t)nshard * k <= 8192, "topk_merge: nshard * k must be <= 8192");
    MIRX_HIP(launch_topk_merge(in_scores, in_ids, nshard, nq, k, metric, out_rank_scores, out_scores,
                               out_ids, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_conv1x1_bn_relu_split3(const float *x, int64_t x_batch_stride, int cin, const float *scale1_or_null,
                                const float *shift1_or_null, const void *w3, const float *bias_or_null, int64_t n,
                                int hw, int cout, int relu_out, float *y, int64_t y_batch_stride, void *stream) {
    MIRX_CHECK(n >= 0 && hw >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 128 && cout % 128 == 0,
               "conv1x1_split3: cin must be a multiple of 16 and cout of 128");
    MIRX_CHECK((scale1_or_null == nullptr) == (shift1_or_null == nullptr), "conv1x1_split3: scale and shift go together");
    MIRX_CHECK(n == 0 || (x && w3 && y), "conv1x1_split3: null buffer");
    MIRX_CHECK(x_batch_stride >= (int64_t)cin * hw, "conv1x1_split3: batch stride smaller than the channel prefix");
    MIRX_CHECK(y_batch_stride >= (int64_t)cout * hw, "conv1x1_split3: output batch stride smaller than cout * hw");
    MIRX_HIP(launch_conv1x1_s3(x, x_batch_stride, cin, scale1_or_null, shift1_or_null,
                               reinterpret_cast<const uint16_t *>(w3), bias_or_null, n, hw, cout, relu_out, y,
                               y_batch_stride, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_conv1x1_bn_relu_split2h(const float *x, int64_t x_batch_stride, int cin, const float *scale1_or_null,
                                 const float *shift1_or_null, const void *w2, const float *oscale,
                                 const float *bias_or_null, int64_t n, int hw, int cout, int relu_out, float *y,
                                 int64_t y_batch_stride, const float *in_range_or_null, float in_ks, float in_kb,
                                 float *out_range_or_null, int64_t x_plane_stride, int64_t y_plane_stride, void *stream) {
    MIRX_CHECK(n >= 0 && hw >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 128 && cout % 128 == 0,
               "conv1x1_split2h: cin must be a multiple of 16 and cout of 128");
    if (!x_plane_stride) x_plane_stride = hw;
    if (!y_plane_stride) y_plane_stride = hw;
    MIRX_CHECK(x_plane_stride >= hw && y_plane_stride >= hw, "conv1x1_split2h: a plane stride is 0 (= hw) or at least hw");
    MIRX_CHECK((scale1_or_null == nullptr) == (shift1_or_null == nullptr), "conv1x1_split2h: scale and shift go together");
    MIRX_CHECK(n == 0 || (x && w2 && y && oscale), "conv1x1_split2h: null buffer");
    MIRX_CHECK(x_batch_stride >= (int64_t)cin * x_plane_stride, "conv1x1_split2h: batch stride smaller than the channel prefix");
    MIRX_CHECK(y_batch_stride >= (int64_t)cout * y_plane_stride, "conv1x1_split2h: output batch stride smaller than cout planes");
    MIRX_CHECK(in_ks >= 0.f && in_kb >= 0.f && (in_range_or_null || in_kb > 0.f),
               "conv1x1_split2h: in_ks / in_kb are non-negative; without range slots in_kb is the bound itself");
    MIRX_HIP(launch_conv1x1_h2(x, x_batch_stride, cin, scale1_or_null, shift1_or_null,
                               reinterpret_cast<const uint16_t *>(w2), oscale, bias_or_null, n, hw, cout, relu_out, y,
                               y_batch_stride, in_range_or_null, in_ks, in_kb, out_range_or_null, 0.f, 0.f, nullptr,
                               x_plane_stride, y_plane_stride, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}


int mirx_conv1x1_bn_relu_split2h_terms(const float *x, int64_t x_batch_stride, int cin, const float *scale1,
                                       const float *shift1, const void *w2, const float *oscale, const float *bias,
                                       int64_t n, int hw, void *y_terms, const float *in_range, float in_ks, float in_kb,
                                       float y_ks, float y_kb, float *y_inv_out, int64_t x_plane_stride, void *stream) {
    MIRX_CHECK(n >= 0 && hw >= 1 && cin >= 16 && cin % 16 == 0, "conv1x1_split2h_terms: cin must be a multiple of 16");
    MIRX_CHECK(n == 0 || (x && w2 && y_terms && oscale && scale1 && shift1 && bias && in_range && y_inv_out),
               "conv1x1_split2h_terms: null buffer");
    if (!x_plane_stride) x_plane_stride = hw;
    MIRX_CHECK(x_plane_stride >= hw, "conv1x1_split2h_terms: the plane stride is 0 (= hw) or at least hw");
    MIRX_CHECK(x_batch_stride >= (int64_t)cin * x_plane_stride, "conv1x1_split2h_terms: batch stride smaller than the channel prefix");
    MIRX_CHECK(in_ks >= 0.f && in_kb >= 0.f && y_ks >= 0.f && y_kb >= 0.f, "conv1x1_split2h_terms: bounds are non-negative");
    MIRX_HIP(launch_conv1x1_h2(x, x_batch_stride, cin, scale1, shift1, reinterpret_cast<const uint16_t *>(w2), oscale, bias, n,
                               hw, 128, 1, reinterpret_cast<float *>(y_terms), 0, in_range, in_ks, in_kb, nullptr, y_ks, y_kb,
                               y_inv_out, x_plane_stride, 0, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_conv3x3_direct_terms_nchw(const void *y_terms, const void *w2, const float *oscale, int64_t n, int side, float *out,
                                   int64_t out_batch_stride, const float *y_inv, float *out_range_or_null,
                                   int64_t out_plane_stride, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535, "conv3x3_terms: batch must be in [0, 65535]");
    MIRX_CHECK(side == 56 || side == 28 || side == 14 || side == 7, "conv3x3_terms: side must be 56, 28, 14 or 7");
    MIRX_CHECK(n == 0 || (y_terms && w2 && oscale && out && y_inv), "conv3x3_terms: null buffer");
    if (!out_plane_stride) out_plane_stride = (int64_t)side * side;
    MIRX_CHECK(out_plane_stride >= (int64_t)side * side, "conv3x3_terms: the plane stride is 0 (= side^2) or at least side^2");
    MIRX_CHECK(out_batch_stride >= 32 * out_plane_stride, "conv3x3_terms: output batch stride too small");
    MIRX_HIP(launch_conv3x3_d2p(reinterpret_cast<const uint16_t *>(y_terms), reinterpret_cast<const uint16_t *>(w2), oscale, n,
                                side, out, out_batch_stride, y_inv, out_range_or_null, out_plane_stride,
                                reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_conv3x3_direct_terms_nchw_pool(const void *y_terms, const void *w2, const float *oscale, int64_t n, int side, float *out,
                                        int64_t out_batch_stride, const float *y_inv, float *out_range_or_null,
                                        int64_t out_plane_stride, const float *pool_scale, const float *pool_shift,
                                        float *pooled, int64_t pooled_batch_stride, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535, "conv3x3_terms_pool: batch must be in [0, 65535]");
    MIRX_CHECK(side == 56 || side == 28 || side == 14, "conv3x3_terms_pool: side must be 56, 28 or 14");
    MIRX_CHECK(n == 0 || (y_terms && w2 && oscale && out && y_inv && pool_scale && pool_shift && pooled),
               "conv3x3_terms_pool: null buffer");
    if (!out_plane_stride) out_plane_stride = (int64_t)side * side;
    MIRX_CHECK(out_plane_stride >= (int64_t)side * side, "conv3x3_terms_pool: the plane stride is 0 (= side^2) or at least side^2");
    MIRX_CHECK(out_batch_stride >= 32 * out_plane_stride, "conv3x3_terms_pool: output batch stride too small");
    MIRX_CHECK(pooled_batch_stride >= (int64_t)32 * (side / 2) * (side / 2), "conv3x3_terms_pool: pooled batch stride too small");
    MIRX_CHECK(!conv3x3_takes_small(n, side),
               "conv3x3_terms_pool: a launch this small runs on the one-wave-per-block kernel, which has no pooled twin "
               "(mirx_conv3x3_small_launch tells; use mirx_conv3x3_direct_terms_nchw + mirx_bn_relu_avgpool2)");
    MIRX_HIP(launch_conv3x3_d2p(reinterpret_cast<const uint16_t *>(y_terms), reinterpret_cast<const uint16_t *>(w2), oscale, n,
                                side, out, out_batch_stride, y_inv, out_range_or_null, out_plane_stride,
                                reinterpret_cast<hipStream_t>(stream), pool_scale, pool_shift, pooled, pooled_batch_stride));
    return MIRX_OK;
}

int mirx_conv3x3_small_launch(int64_t n, int side) {
    return (side == 56 || side == 28 || side == 14 || side == 7) && n > 0 && conv3x3_takes_small(n, side) ? 1 : 0;
}

int mirx_dense_layer_fused(float *buf, int64_t batch_stride, int64_t plane_stride, int cin, const float *scale1,
                           const float *shift1, const void *w2, const float *oscale, const float *bias, const void *c3w2,
                           const float *c3oscale, int64_t n, int side, float *range_row, float in_ks, float in_kb, float y_ks,
                           float y_kb, void *stream) {
    MIRX_CHECK(n >= 0 && n <= (int64_t)1 << 30, "dense_layer_fused: bad batch");
    MIRX_CHECK(side == 14 || side == 7, "dense_layer_fused: side must be 14 or 7 (the bottleneck of a 196-pixel unit fits the LDS)");
    MIRX_CHECK(cin >= 128 && cin <= 1024 && cin % 32 == 0, "dense_layer_fused: cin must be a multiple of 32 in [128, 1024]");
    if (!plane_stride) plane_stride = (int64_t)side * side;
    MIRX_CHECK(plane_stride == (int64_t)side * side, "dense_layer_fused: channel planes must be packed (plane stride 0 or side^2)");
    MIRX_CHECK(batch_stride >= (int64_t)(cin + 32) * plane_stride && batch_stride % 4 == 0,
               "dense_layer_fused: batch stride smaller than cin + 32 planes, or not a multiple of 4 floats");
    MIRX_CHECK((reinterpret_cast<uintptr_t>(buf) & 15) == 0, "dense_layer_fused: buf must be 16-byte aligned");
    MIRX_CHECK(n == 0 || (buf && scale1 && shift1 && w2 && oscale && bias && c3w2 && c3oscale && range_row),
               "dense_layer_fused: null buffer");
    MIRX_CHECK(in_ks >= 0.f && in_kb >= 0.f && y_ks >= 0.f && y_kb >= 0.f, "dense_layer_fused: bounds are non-negative");
    MIRX_HIP(launch_dense_fused(buf, batch_stride, cin, scale1, shift1, reinterpret_cast<const uint16_t *>(w2), oscale,
                                bias, reinterpret_cast<const uint16_t *>(c3w2), c3oscale, n, side, range_row, in_ks, in_kb, y_ks,
                                y_kb, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_linear_split3(const float *x, int64_t m, int k, const void *w3, const float *bias_or_null, int n, int act,
                       const float *residual_or_null, const float *gamma_or_null, float *y, void *stream) {
    MIRX_CHECK(m >= 0 && k >= 16 && k % 16 == 0 && n >= 1, "linear_split3: k must be a multiple of 16");
    MIRX_CHECK(act >= 0 && act <= 2 && !(act == 2 && residual_or_null), "linear_split3: act is 0 (none), 1 (erf gelu) or 2 (tanh gelu, no residual)");
    MIRX_CHECK(m == 0 || (x && w3 && y), "linear_split3: null buffer");
    MIRX_CHECK(residual_or_null || !gamma_or_null, "linear_split3: gamma scales the residual branch only");
    MIRX_CHECK(x != y, "linear_split3: y may alias the residual, not the input");
    MIRX_HIP(launch_linear_s3(x, m, k, reinterpret_cast<const uint16_t *>(w3), bias_or_null, n, act, residual_or_null,
                              gamma_or_null, y, 0, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_linear_split2h(const float *x, int64_t m, int k, const void *w2, const float *bias_or_null, int n, int act,
                        const float *residual_or_null, const float *gamma_or_null, float x_scale, float out_scale,
                        float *y, void *stream) {
    MIRX_CHECK(m >= 0 && k >= 16 && k % 16 == 0 && n >= 1, "linear_split2h: k must be a multiple of 16");
    MIRX_CHECK(act >= 0 && act <= 2 && !(act == 2 && residual_or_null), "linear_split2h: act is 0 (none), 1 (erf gelu) or 2 (tanh gelu, no residual)");
    MIRX_CHECK(m == 0 || (x && w2 && y), "linear_split2h: null buffer");
    MIRX_CHECK(residual_or_null || !gamma_or_null, "linear_split2h: gamma scales the residual branch only");
    MIRX_CHECK(x != y, "linear_split2h: y may alias the residual, not the input");
    MIRX_CHECK(x_scale > 0.f && out_scale > 0.f, "linear_split2h: scales must be positive powers of two");
    MIRX_HIP(launch_linear_h2(x, m, k, reinterpret_cast<const uint16_t *>(w2), bias_or_null, n, act, residual_or_null,
                              gamma_or_null, x_scale, out_scale, y, 0, nullptr, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_linear_split2h_gelu_grn(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w2,
                                 const float *bias_or_null, int n, float x_scale, float out_scale, float *y, float *partials,
                                 float *gx, void *stream) {
    MIRX_CHECK(n_img >= 0 && tokens_per_image >= 128 && k >= 16 && k % 16 == 0 && n >= 1,
               "linear_split2h_gelu_grn: k must be a multiple of 16, tokens_per_image at least 128");
    MIRX_CHECK(n_img == 0 || (x && w2 && y && partials && gx), "linear_split2h_gelu_grn: null buffer");
    MIRX_CHECK(x_scale > 0.f && out_scale > 0.f, "linear_split2h_gelu_grn: scales must be positive");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    MIRX_HIP(launch_linear_h2(x, n_img * tokens_per_image, k, reinterpret_cast<const uint16_t *>(w2), bias_or_null, n, 1, nullptr,
                              nullptr, x_scale, out_scale, y, tokens_per_image, nullptr, st, false, partials));
    MIRX_HIP(launch_grn_norm_partials(partials, tokens_per_image, n_img, n, gx, st));
    return MIRX_OK;
}

int mirx_linear_split2h_grn_rows(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w2,
                                 const float *bias_or_null, int n, const float *residual_or_null, const float *input_scale,
                                 float x_bound, const float *input_scale_max, float w_inv, float *y, void *stream) {
    MIRX_CHECK(n_img >= 0 && tokens_per_image >= 1 && k >= 16 && k % 16 == 0 && n >= 1,
               "linear_split2h_grn_rows: k must be a multiple of 16");
    MIRX_CHECK(n_img == 0 || (x && w2 && y && input_scale && input_scale_max), "linear_split2h_grn_rows: null buffer");
    MIRX_CHECK(x != y, "linear_split2h_grn_rows: y may alias the residual, not the input");
    MIRX_CHECK(x_bound > 0.f && w_inv > 0.f, "linear_split2h_grn_rows: x_bound and w_inv must be positive");
    MIRX_HIP(launch_linear_h2(x, n_img * tokens_per_image, k, reinterpret_cast<const uint16_t *>(w2), bias_or_null, n, 0,
                              residual_or_null, input_scale, x_bound, w_inv, y, tokens_per_image, input_scale_max,
                              reinterpret_cast<hipStream_t>(stream), true));
    return MIRX_OK;
}

int mirx_linear_split2h_nchw(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w2,
                             const float *bias_or_null, int n, const float *residual_or_null,
                             const float *input_scale_or_null, float x_bound, const float *input_scale_max_or_null,
                             float w_inv, float *y, void *stream) {
    MIRX_CHECK(n_img >= 0 && tokens_per_image >= 1 && k >= 16 && k % 16 == 0 && n >= 1,
               "linear_split2h_nchw: k must be a multiple of 16");
    MIRX_CHECK(n_img == 0 || (x && w2 && y), "linear_split2h_nchw: null buffer");
    MIRX_CHECK(x != y, "linear_split2h_nchw: y may alias the residual, not the input");
    MIRX_CHECK(x_bound > 0.f && w_inv > 0.f, "linear_split2h_nchw: x_bound and w_inv must be positive");
    MIRX_CHECK(!input_scale_or_null == !input_scale_max_or_null,
               "linear_split2h_nchw: an input scale needs the device scalar that bounds it (and only then)");
    if (input_scale_max_or_null) {
        // the kernel derives 2^s from x_bound * input_scale_max[0]
        MIRX_HIP(launch_linear_h2(x, n_img * tokens_per_image, k, reinterpret_cast<const uint16_t *>(w2), bias_or_null, n, 0,
                                  residual_or_null, input_scale_or_null, x_bound, w_inv, y, tokens_per_image,
                                  input_scale_max_or_null, reinterpret_cast<hipStream_t>(stream)));
    } else {
        int e = 0;                                     // x_bound * 2^s in [2^14, 2^15)
        (void)frexpf(x_bound, &e);                     // x_bound = f * 2^e, f in [0.5, 1)
        const float xs = ldexpf(1.f, 15 - e);
        MIRX_CHECK(std::isfinite(x_bound) && std::isfinite(xs) && xs > 0.f, "linear_split2h_nchw: x_bound out of range");
        MIRX_HIP(launch_linear_h2(x, n_img * tokens_per_image, k, reinterpret_cast<const uint16_t *>(w2), bias_or_null, n, 0,
                                  residual_or_null, nullptr, xs, w_inv / xs, y, tokens_per_image, nullptr,
                                  reinterpret_cast<hipStream_t>(stream)));
    }
    return MIRX_OK;
}

int mirx_linear_split3_nchw(const float *x, int64_t n_img, int tokens_per_image, int k, const void *w3,
                            const float *bias_or_null, int n, const float *residual_or_null,
                            const float *input_scale_or_null, float *y, void *stream) {
    MIRX_CHECK(n_img >= 0 && tokens_per_image >= 1 && k >= 16 && k % 16 == 0 && n >= 1,
               "linear_split3_nchw: k must be a multiple of 16");
    MIRX_CHECK(n_img == 0 || (x && w3 && y), "linear_split3_nchw: null buffer");
    MIRX_CHECK(x != y, "linear_split3_nchw: y may alias the residual, not the input");
    MIRX_HIP(launch_linear_s3(x, n_img * tokens_per_image, k, reinterpret_cast<const uint16_t *>(w3), bias_or_null, n, 0,
                              residual_or_null, input_scale_or_null, y, tokens_per_image,
                              reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_grn_norm_nhwc(const float *x, int64_t n, int hw, int c, float *gx, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535 && hw >= 1 && c >= 1 && (n == 0 || (x && gx)), "grn_norm: bad argument");
    MIRX_HIP(launch_grn_norm(x, n, hw, c, gx, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_grn_scale(const float *gx, const float *weight, int64_t n, int c, float eps, float *scale, float *scale_max,
                   void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535 && c >= 1 && eps >= 0.f, "grn_scale: bad argument");
    MIRX_CHECK(n == 0 || (gx && weight && scale && scale_max), "grn_scale: null buffer");
    MIRX_HIP(launch_grn_scale(gx, weight, n, c, eps, scale, scale_max, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}


int mirx_layernorm(const float *x, int64_t m, int c, const float *gamma_or_null, const float *beta_or_null, float eps,
                   float *y, int tokens_per_image, void *stream) {
    MIRX_CHECK(m >= 0 && c >= 4 && c % 4 == 0 && c <= 8192, "layernorm: c must be a multiple of 4, at most 8192");
    MIRX_CHECK(m == 0 || (x && y), "layernorm: null buffer");
    MIRX_CHECK(tokens_per_image >= 0 && (tokens_per_image == 0 || (c <= 512 && x != y)),
               "layernorm: the channels-first form needs c <= 512 and y != x");
    MIRX_CHECK(eps >= 0.f, "layernorm: eps must be non-negative");
    MIRX_HIP(launch_layernorm_rows(x, m, c, gamma_or_null, beta_or_null, eps, y, tokens_per_image,
                                   reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_layernorm_patch2_nhwc(const float *x, int64_t n_img, int h, int w, int c, const float *gamma_or_null,
                               const float *beta_or_null, float eps, float *y, void *stream) {
    MIRX_CHECK(n_img >= 0 && h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "layernorm_patch2_nhwc: h and w must be even");
    MIRX_CHECK(c >= 4 && c % 4 == 0 && c <= 8192, "layernorm_patch2_nhwc: c must be a multiple of 4, at most 8192");
    MIRX_CHECK(n_img == 0 || (x && y), "layernorm_patch2_nhwc: null buffer");
    MIRX_CHECK(x != y && eps >= 0.f, "layernorm_patch2_nhwc: not in place; eps must be non-negative");
    MIRX_HIP(launch_layernorm_rows(x, n_img * h * w, c, gamma_or_null, beta_or_null, eps, y, 0, reinterpret_cast<hipStream_t>(stream),
                                   nullptr, 1.f, w, h));
    return MIRX_OK;
}

int mirx_layernorm_terms(const float *x, int64_t m, int c, const float *gamma_or_null, const float *beta_or_null, float eps,
                         float scale, void *yt, void *stream) {
    MIRX_CHECK(m >= 0 && c >= 4 && c % 4 == 0 && c <= 8160, "layernorm_terms: c must be a multiple of 4, at most 8160");
    MIRX_CHECK(m == 0 || (x && yt), "layernorm_terms: null buffer");
    MIRX_CHECK(eps >= 0.f && scale > 0.f, "layernorm_terms: eps must be non-negative and scale positive");
    MIRX_HIP(launch_layernorm_rows(x, m, c, gamma_or_null, beta_or_null, eps, nullptr, 0, reinterpret_cast<hipStream_t>(stream),
                                   yt, scale));
    return MIRX_OK;
}

int mirx_rows_to_terms(const float *x, int64_t m, int k, int64_t row_stride, float scale, void *xt, void *stream) {
    MIRX_CHECK(m >= 0 && k >= 1 && row_stride >= k && row_stride % 4 == 0, "rows_to_terms: bad geometry (row_stride % 4 == 0)");
    MIRX_CHECK(m == 0 || (x && xt), "rows_to_terms: null buffer");
    MIRX_CHECK((reinterpret_cast<uintptr_t>(x) & 15) == 0, "rows_to_terms: x must be 16-byte aligned");
    MIRX_CHECK(scale > 0.f, "rows_to_terms: scale must be positive");
    MIRX_HIP(launch_rows_to_terms(x, m, k, row_stride, scale, xt, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int64_t mirx_linear_terms_workspace_bytes(int64_t m, int k, int n) { return (int64_t)linear_t2_workspace_bytes(m, k, n); }

int mirx_linear_terms(const void *xt, int64_t m, int k, const void *wt, const float *bias_or_null, int n, int act,
                      const float *residual_or_null, const float *gamma_or_null, float out_scale, float *y_or_null,
                      void *yt_or_null, float yt_scale, void *workspace_or_null, int64_t workspace_bytes, void *stream) {
    MIRX_CHECK(m >= 0 && k >= 1 && n >= 4 && n % 4 == 0, "linear_terms: n must be a multiple of 4");
    MIRX_CHECK(act >= 0 && act <= 2, "linear_terms: act is 0 (none), 1 (GELU) or 2 (tanh GELU)");
    MIRX_CHECK(m == 0 || (xt && wt), "linear_terms: null operand");
    MIRX_CHECK((y_or_null != nullptr) != (yt_or_null != nullptr), "linear_terms: exactly one of y and yt");
    MIRX_CHECK(!(yt_or_null && residual_or_null), "linear_terms: the terms output has no residual form");
    MIRX_CHECK(!(residual_or_null && act), "linear_terms: the residual form has no activation");
    MIRX_CHECK(!gamma_or_null || residual_or_null, "linear_terms: gamma scales the residual form only");
    MIRX_CHECK(!yt_or_null || yt_scale > 0.f, "linear_terms: yt_scale must be positive");
    MIRX_CHECK(workspace_bytes >= 0, "linear_terms: negative workspace size");
    MIRX_HIP(launch_linear_t2(xt, m, k, wt, bias_or_null, n, act, residual_or_null, gamma_or_null, out_scale, y_or_null,
                              yt_or_null, yt_scale, workspace_or_null, workspace_or_null ? (size_t)workspace_bytes : 0,
                              reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_patchify_nchw(const float *x, int64_t n, int c, int h, int w, int patch, const float *ln_gamma_or_null,
                       const float *ln_beta_or_null, float eps, float *out, int row_stride, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535 && c >= 1 && patch >= 1 && h >= patch && w >= patch, "patchify: bad geometry");
    MIRX_CHECK(row_stride >= c * patch * patch, "patchify: row_stride smaller than c * patch * patch");
    MIRX_CHECK((ln_gamma_or_null == nullptr) == (ln_beta_or_null == nullptr), "patchify: gamma and beta go together");
    MIRX_CHECK(n == 0 || (x && out), "patchify: null buffer");
    MIRX_CHECK((size_t)2 * patch * w * sizeof(float) <= 64 * 1024, "patchify: strip too wide for the LayerNorm2d prologue");
    MIRX_HIP(launch_patchify(x, n, c, h, w, patch, ln_gamma_or_null, ln_beta_or_null, eps, out, row_stride,
                             reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_attention_small(const float *q, int64_t q_row_stride, const float *k, const float *v, int64_t kv_row_stride,
                         const uint8_t *key_mask_or_null, int64_t batch, int heads, int head_dim, int n_queries, int n_keys,
                         float scale, float *out, void *stream) {
    MIRX_CHECK(batch >= 0 && heads >= 1 && n_queries >= 0 && n_keys >= 1, "attention_small: bad sizes");
    MIRX_CHECK(head_dim == 16 || head_dim == 32 || head_dim == 64 || head_dim == 72, "attention_small: head_dim 16 / 32 / 64 / 72");
    MIRX_CHECK(q_row_stride >= (int64_t)heads * head_dim && kv_row_stride >= (int64_t)heads * head_dim &&
                   q_row_stride % 4 == 0 && kv_row_stride % 4 == 0,
               "attention_small: row strides must cover heads * head_dim and be multiples of 4");
    MIRX_CHECK(batch == 0 || n_queries == 0 || (q && k && v && out), "attention_small: null buffer");
    MIRX_HIP(launch_attention_small(q, q_row_stride, k, v, kv_row_stride, key_mask_or_null, batch, heads, head_dim,
                                    n_queries, n_keys, scale, out, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_conv3x3_winograd_nchw(const float *x, const float *u, int64_t n, int side, float *out, int64_t out_batch_stride,
                               void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535, "conv3x3: batch must be in [0, 65535]");
    MIRX_CHECK(side == 56 || side == 28 || side == 14 || side == 7, "conv3x3: side must be 56, 28, 14 or 7");
    MIRX_CHECK(n == 0 || (x && u && out), "conv3x3: null buffer");
    MIRX_CHECK(out_batch_stride >= (int64_t)32 * side * side, "conv3x3: output batch stride too small");
    MIRX_HIP(launch_conv3x3_wino(x, u, n, side, out, out_batch_stride, nullptr, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}



int mirx_conv3x3_winograd_split3_nchw(const float *x, const void *u3, int64_t n, int side, float *out,
                                      int64_t out_batch_stride, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535, "conv3x3_split3: batch must be in [0, 65535]");
    MIRX_CHECK(side == 56 || side == 28 || side == 14, "conv3x3_split3: side must be 56, 28 or 14");
    MIRX_CHECK(n == 0 || (x && u3 && out), "conv3x3_split3: null buffer");
    MIRX_CHECK(out_batch_stride >= (int64_t)32 * side * side, "conv3x3_split3: output batch stride too small");
    MIRX_HIP(launch_conv3x3_wino_s3(x, reinterpret_cast<const uint16_t *>(u3), n, side, out, out_batch_stride,
                                    reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_conv3x3_direct_split3_nchw(const float *x, const void *w3, int64_t n, int side, float *out,
                                    int64_t out_batch_stride, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535, "conv3x3_direct: batch must be in [0, 65535]");
    MIRX_CHECK(side == 56 || side == 28 || side == 14, "conv3x3_direct: side must be 56, 28 or 14");
    MIRX_CHECK(n == 0 || (x && w3 && out), "conv3x3_direct: null buffer");
    MIRX_CHECK(out_batch_stride >= (int64_t)32 * side * side, "conv3x3_direct: output batch stride too small");
    MIRX_HIP(launch_conv3x3_d3(x, reinterpret_cast<const uint16_t *>(w3), n, side, out, out_batch_stride,
                               reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_attention_qkv_f32(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim, float scale,
                           float *out, void *stream) {
    MIRX_CHECK(batch >= 0 && n_tokens >= 0 && heads >= 1, "attention: bad sizes");
    MIRX_CHECK(head_dim == 32 || head_dim == 64 || head_dim == 72 || head_dim == 96,
               "attention: head_dim must be 32, 64, 72 or 96");
    MIRX_CHECK(batch == 0 || n_tokens == 0 || (qkv && out), "attention: null buffer");
    MIRX_CHECK(batch <= 65535 && heads <= 65535, "attention: batch and heads must be <= 65535");
    MIRX_HIP(launch_attention(qkv, batch, n_tokens, heads, head_dim, scale, out, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_attention_qkv_f32_split3(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim, float scale,
                                  float *out, void *stream) {
    MIRX_CHECK(batch >= 0 && n_tokens >= 0 && heads >= 1, "attention_split3: bad sizes");
    MIRX_CHECK(head_dim == 32 || head_dim == 64 || head_dim == 72 || head_dim == 96,
               "attention_split3: head_dim must be 32, 64, 72 or 96");
    MIRX_CHECK(batch <= 65535 && heads <= 65535, "attention_split3: batch and heads must be <= 65535");
    MIRX_CHECK(batch == 0 || n_tokens == 0 || (qkv && out), "attention_split3: null buffer");
    MIRX_HIP(launch_attention_s3(qkv, batch, n_tokens, heads, head_dim, scale, out, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_attention_qkv_f32_split2h(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim, float scale,
                                   float qk_bound, float v_bound, float *out, void *stream) {
    MIRX_CHECK(batch >= 0 && n_tokens >= 0 && heads >= 1, "attention_split2h: bad sizes");
    MIRX_CHECK(head_dim == 64 || head_dim == 72 || head_dim == 96 || head_dim == 32,
               "attention_split2h: head_dim must be 32, 64, 72 or 96");
    MIRX_CHECK(batch <= 65535 && heads <= 65535, "attention_split2h: batch and heads must be <= 65535");
    MIRX_CHECK(batch == 0 || n_tokens == 0 || (qkv && out), "attention_split2h: null buffer");
    MIRX_CHECK(qk_bound > 0.f && v_bound > 0.f && scale > 0.f, "attention_split2h: bounds and scale must be positive");
    MIRX_HIP(launch_attention_h2(qkv, batch, n_tokens, heads, head_dim, scale, qk_bound, v_bound, out,
                                 reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_attention_qkv_f32_split2h_terms(const float *qkv, int64_t batch, int n_tokens, int heads, int head_dim, float scale,
                                         float qk_bound, float v_bound, float out_scale, void *out_terms, void *stream) {
    MIRX_CHECK(batch >= 0 && n_tokens >= 0 && heads >= 1, "attention_split2h_terms: bad sizes");
    MIRX_CHECK(head_dim == 64 || head_dim == 72 || head_dim == 96 || head_dim == 32,
               "attention_split2h_terms: head_dim must be 32, 64, 72 or 96");
    MIRX_CHECK(batch <= 65535 && heads <= 65535, "attention_split2h_terms: batch and heads must be <= 65535");
    MIRX_CHECK(batch == 0 || n_tokens == 0 || (qkv && out_terms), "attention_split2h_terms: null buffer");
    MIRX_CHECK(qk_bound > 0.f && v_bound > 0.f && scale > 0.f && out_scale > 0.f,
               "attention_split2h_terms: bounds and scales must be positive");
    MIRX_HIP(launch_attention_h2(qkv, batch, n_tokens, heads, head_dim, scale, qk_bound, v_bound, nullptr,
                                 reinterpret_cast<hipStream_t>(stream), out_terms, out_scale));
    return MIRX_OK;
}

int mirx_rank_metrics(const int64_t *ranks, int64_t nq, int64_t n, int64_t row_stride,
                      const int64_t *gallery_labels, int64_t n_labels, const int64_t *query_labels,
                      const int64_t *query_ids_or_null, int drop_self, int rel_kind, double jaccard_threshold,
                      int ap_kind,
                      const int32_t *kappas, int nk, double *out_ap, int64_t *out_cnt, int64_t *out_nrel,
                      int64_t *out_maxpos, void *stream) {
    MIRX_CHECK(nq >= 0 && n >= 0 && row_stride >= n && n_labels >= 0, "rank_metrics: bad sizes");
    MIRX_CHECK(nq == 0 || (ranks && gallery_labels && query_labels && out_ap && out_nrel && out_maxpos),
               "rank_metrics: null buffer");
    MIRX_CHECK(nk >= 0 && nk <= MIRX_MAX_KAPPAS && (nk == 0 || (kappas && out_cnt)), "rank_metrics: 0 <= nk <= 8");
    MIRX_CHECK((rel_kind == 0 || rel_kind == 1) && (ap_kind == 0 || ap_kind == 1), "rank_metrics: bad kind");
    RankMetricsArgs a{};
    a.ranks = ranks; a.nq = nq; a.n = n; a.row_stride = row_stride;
    a.gallery_labels = gallery_labels; a.n_labels = n_labels; a.query_labels = query_labels;
    a.query_ids = query_ids_or_null; a.drop_self = (drop_self && query_ids_or_null) ? 1 : 0;
    a.jaccard_threshold = jaccard_threshold; a.nk = nk;
    for (int j = 0; j < nk; ++j) {
        MIRX_CHECK(kappas[j] >= 1, "rank_metrics: kappa must be >= 1");
        a.kappas[j] = kappas[j];
    }
    a.out_ap = out_ap; a.out_cnt = out_cnt; a.out_nrel = out_nrel; a.out_maxpos = out_maxpos;
    MIRX_HIP(launch_rank_metrics(a, rel_kind, ap_kind, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_l2_normalize(float *x, int64_t n, int dim, void *stream) {
    MIRX_CHECK(x && n >= 0 && dim >= 1, "l2_normalize: bad argument");
    MIRX_HIP(launch_l2_normalize(x, n, dim, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_bn_relu_gap_l2norm(const float *x, const float *scale, const float *shift, int64_t n, int c,
                            int hw, int normalize, float *y, void *stream) {
    MIRX_CHECK(x && y && n >= 0 && c >= 1 && c <= 16384 && hw >= 1, "bn_relu_gap_l2norm: bad argument");
    MIRX_HIP(launch_bn_relu_gap_l2norm(x, scale, shift, n, c, hw, normalize, y,
                                       reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_bn_relu_nchw(const float *x, int64_t x_batch_stride, const float *scale, const float *shift,
                      int64_t n, int c, int hw, float *y, void *stream) {
    MIRX_CHECK(x && scale && shift && y && n >= 0 && c >= 1 && hw >= 1, "bn_relu_nchw: bad argument");
    MIRX_CHECK(((int64_t)c * hw) % 4 == 0 && x_batch_stride % 4 == 0 && x_batch_stride >= (int64_t)c * hw,
               "bn_relu_nchw: c*hw and the batch stride must be multiples of 4");
    MIRX_HIP(launch_bn_relu_nchw(x, x_batch_stride, scale, shift, n, c, hw, y,
                                 reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_bn_relu_avgpool2(const float *x, int64_t x_batch_stride, const float *scale, const float *shift,
                          int64_t n, int c, int h, int w, float *y, int64_t x_plane_stride, void *stream) {
    MIRX_CHECK(x && scale && shift && y && n >= 0 && c >= 1, "bn_relu_avgpool2: bad argument");
    MIRX_CHECK(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && x_batch_stride % 2 == 0,
               "bn_relu_avgpool2: h, w and the batch stride must be even");
    MIRX_CHECK(x_plane_stride == 0 || (x_plane_stride >= (int64_t)h * w && x_plane_stride % 4 == 0),
               "bn_relu_avgpool2: the plane stride is 0 (= h * w) or a multiple of 4 that is at least h * w");
    MIRX_HIP(launch_bn_relu_avgpool2(x, x_batch_stride, scale, shift, n, c, h, w, y, x_plane_stride,
                                     reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_bn_relu_avgpool2_into(const float *x, int64_t x_batch_stride, const float *scale, const float *shift,
                               int64_t n, int c, int h, int w, float *y, int64_t y_batch_stride, int64_t x_plane_stride,
                               void *stream) {
    MIRX_CHECK(x && scale && shift && y && n >= 0 && c >= 1, "bn_relu_avgpool2_into: bad argument");
    MIRX_CHECK(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0 && x_batch_stride % 2 == 0,
               "bn_relu_avgpool2_into: h, w and the batch stride must be even");
    MIRX_CHECK(x_plane_stride == 0 || (x_plane_stride >= (int64_t)h * w && x_plane_stride % 4 == 0),
               "bn_relu_avgpool2_into: the plane stride is 0 (= h * w) or a multiple of 4 that is at least h * w");
    MIRX_CHECK(y_batch_stride >= (int64_t)c * (h / 2) * (w / 2) && y_batch_stride % 2 == 0,
               "bn_relu_avgpool2_into: the output batch stride must be even and at least c * (h / 2) * (w / 2)");
    MIRX_HIP(launch_bn_relu_avgpool2(x, x_batch_stride, scale, shift, n, c, h, w, y, x_plane_stride,
                                     reinterpret_cast<hipStream_t>(stream), y_batch_stride));
    return MIRX_OK;
}

int mirx_stem_conv7_bn_relu_pool(const float *x, const float *w, const float *scale, const float *shift,
                                 int64_t n, int h, int wd, float *y, void *stream) {
    MIRX_CHECK(x && w && scale && shift && y && n >= 0, "stem: null argument");
    MIRX_CHECK(h >= 8 && wd >= 8 && h % 4 == 0 && wd % 4 == 0, "stem: H and W must be multiples of 4");
    MIRX_HIP(launch_stem(x, w, scale, shift, n, h, wd, y, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_stem_conv7_bn_relu_pool_split3(const float *x, const void *w3, const float *scale, const float *shift,
                                        int64_t n, int h, int wd, float *y, void *stream) {
    MIRX_CHECK(x && w3 && scale && shift && y && n >= 0 && n <= 65535, "stem_split3: null argument or batch > 65535");
    MIRX_CHECK(h >= 8 && wd >= 8 && h % 4 == 0 && wd % 4 == 0, "stem_split3: H and W must be multiples of 4");
    MIRX_HIP(launch_stem_s3(x, reinterpret_cast<const uint16_t *>(w3), scale, shift, n, h, wd, y,
                            (int64_t)64 * (h / 4) * (wd / 4), nullptr, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_range_absmax(const float *x, int64_t per_image, int64_t n, float *range_row, void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535 && per_image >= 0, "range_absmax: batch must be in [0, 65535]");
    MIRX_CHECK(n == 0 || per_image == 0 || (x && range_row), "range_absmax: null buffer");
    MIRX_HIP(launch_range_absmax(x, per_image, n, range_row, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_range_absmax_u8(const uint8_t *x, int64_t hw, int64_t n, const float *mean3, const float *std3, float *range_row,
                         void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535 && hw >= 0, "range_absmax_u8: batch must be in [0, 65535]");
    MIRX_CHECK(n == 0 || hw == 0 || (x && mean3 && std3 && range_row), "range_absmax_u8: null buffer");
    MIRX_HIP(launch_range_absmax_u8(x, hw, n, mean3, std3, range_row, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_stem_conv7_bn_relu_pool_split2h_u8_into(const uint8_t *x, const float *mean3, const float *std3, const void *w2,
                                                 const float *oscale, const float *scale, const float *shift, int64_t n, int h,
                                                 int wd, float *y, int64_t y_batch_stride, const float *in_range,
                                                 float *out_range_or_null, void *stream) {
    MIRX_CHECK(x && mean3 && std3 && w2 && oscale && scale && shift && y && in_range && n >= 0 && n <= 65535,
               "stem_split2h_u8: null argument or batch > 65535");
    MIRX_CHECK(h >= 8 && wd >= 8 && h % 4 == 0 && wd % 4 == 0, "stem_split2h_u8: H and W must be multiples of 4");
    MIRX_CHECK(y_batch_stride >= (int64_t)64 * (h / 4) * (wd / 4), "stem_split2h_u8: output batch stride too small");
    MIRX_HIP(launch_stem_h2_u8(x, mean3, std3, reinterpret_cast<const uint16_t *>(w2), oscale, scale, shift, n, h, wd, y,
                               y_batch_stride, in_range, out_range_or_null, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_stem_conv7_bn_relu_pool_split2h_into(const float *x, const void *w2, const float *oscale, const float *scale,
                                              const float *shift, int64_t n, int h, int wd, float *y, int64_t y_batch_stride,
                                              const float *in_range, float *out_range_or_null, void *stream) {
    MIRX_CHECK(x && w2 && oscale && scale && shift && y && in_range && n >= 0 && n <= 65535,
               "stem_split2h: null argument or batch > 65535");
    MIRX_CHECK(h >= 8 && wd >= 8 && h % 4 == 0 && wd % 4 == 0, "stem_split2h: H and W must be multiples of 4");
    MIRX_CHECK(y_batch_stride >= (int64_t)64 * (h / 4) * (wd / 4), "stem_split2h: output batch stride too small");
    MIRX_HIP(launch_stem_h2(x, reinterpret_cast<const uint16_t *>(w2), oscale, scale, shift, n, h, wd, y, y_batch_stride,
                            in_range, out_range_or_null, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}


int mirx_conv1x1_bn_relu(const float *x, int64_t x_batch_stride, int cin, const float *scale, const float *shift,
                         const float *wt, const float *bias, int64_t n, int hw, int cout, int relu_out, float *y,
                         void *stream) {
    MIRX_CHECK(x && wt && y && n >= 0 && hw >= 1, "conv1x1: null argument");
    MIRX_CHECK((scale == nullptr) == (shift == nullptr), "conv1x1: scale and shift go together");
    MIRX_CHECK(cin >= 32 && cin % 32 == 0 && cout >= 128 && cout % 128 == 0, "conv1x1: cin % 32 and cout % 128 must be 0");
    MIRX_CHECK(x_batch_stride >= (int64_t)cin * hw, "conv1x1: batch stride smaller than cin * hw");
    MIRX_HIP(launch_conv1x1(x, x_batch_stride, cin, scale, shift, wt, bias, n, hw, cout, relu_out, y,
                            reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_dwconv7x7_nhwc(const float *x, const float *w_taps_first, const float *bias, int64_t n, int c, int h, int wd, float *y,
                        void *stream) {
    MIRX_CHECK(n >= 0 && n <= 65535 && c >= 1 && h >= 1 && wd >= 1, "dwconv7x7_nhwc: bad geometry (n <= 65535)");
    MIRX_CHECK((int64_t)h * wd * c <= 0x7fffffff, "dwconv7x7_nhwc: one image must stay below 2^31 elements");
    MIRX_CHECK(n == 0 || (x && w_taps_first && y), "dwconv7x7_nhwc: null buffer");
    MIRX_CHECK(x != y, "dwconv7x7_nhwc: in place is not supported");
    MIRX_HIP(launch_dwconv7_nhwc(x, w_taps_first, bias, n, c, h, wd, y, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

int mirx_dwconv7x7_nchw_to_nhwc(const float *x, const float *w, const float *bias, int64_t n, int c, int h,
                                int wd, float *y, void *stream) {
    MIRX_CHECK(x && w && y && n >= 0 && c >= 1 && h >= 1 && wd >= 1, "dwconv7x7: bad argument");
    MIRX_HIP(launch_dwconv7(x, w, bias, n, c, h, wd, y, reinterpret_cast<hipStream_t>(stream)));
    return MIRX_OK;
}

}  // extern "C"
