// k_finalize.hip -- tier-1 tail: threshold selection from sampled group maxima, and the
// per-query finalize (sort candidates, completeness guard, fp64 re-rank, final top-k).
//
// Why the result is exact (DESIGN.md "Exactness"): the GEMM stores EVERY row whose
// approximate score s~ exceeds tau.  With |s~ - s| <= eps for every row, the true top-k_eff
// rows all have s~ >= kth(s~) - 2*eps, so if tau < kth(s~) - 2*eps and the list did not
// overflow, re-scoring exactly the rows with s~ >= kth(s~) - 2*eps in fp64 and sorting them
// (score desc, id asc) gives the oracle's answer.  Queries failing the guard go to the exact
// scan; nothing is ever answered approximately.
#include "mirx_kernels.h"

#include <math.h>

namespace mirx {

namespace {

constexpr int64_t ID_LAST = INT64_MAX;

__device__ inline uint32_t f32_orderable(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float f32_from_orderable(uint32_t e) {
    return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

template <typename T, typename Before>
__device__ inline void bitonic_lds(T *a, int m, Before before) {
    for (int size = 2; size <= m; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (m >> 1); t += blockDim.x) {
                const int i = 2 * t - (t & (stride - 1));
                const int j = i + stride;
                const bool best_first = (i & size) == 0;
                const T x = a[i], y = a[j];
                if (before(y, x) == best_first) { a[i] = y; a[j] = x; }
            }
        }
    }
    __syncthreads();
}

__device__ inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// tau[q] = rank_j-th largest sampled group maximum of query q; +inf for padding queries.
__global__ __launch_bounds__(256) void k_select_tau(const float *__restrict__ groupmax, int ngroups,
                                                    int64_t nq, int rank_j, float *__restrict__ tau) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *a = reinterpret_cast<float *>(smem);
    const int64_t qi = blockIdx.x;
    if (qi >= nq) {
        if (threadIdx.x == 0) tau[qi] = INFINITY;
        return;
    }
    const int m = pow2_ceil(ngroups);
    for (int i = threadIdx.x; i < m; i += 256) a[i] = i < ngroups ? groupmax[qi * ngroups + i] : -INFINITY;
    bitonic_lds(a, m, [](float x, float y) { return x > y; });
    if (threadIdx.x == 0) {
        const int j = rank_j < ngroups ? rank_j : ngroups;
        tau[qi] = a[j - 1];
    }
}

template <int METRIC>
__global__ __launch_bounds__(256) void k_finalize(FinalizeArgs A) {
    __shared__ unsigned long long keys[CAND_CAP];
    __shared__ Hit hits[CAND_CAP];
    __shared__ int sh_ok, sh_m;
    __shared__ float sh_lo;
    const int slot_q = blockIdx.x;                                  // index into tau / candidate buffers
    const int qi = A.qmap ? A.qmap[slot_q] : slot_q;                // original query number
    const int wave = threadIdx.x >> 6, lane = lane_id();
    // gather the producer regions (+ the shared overflow list) of this query into `keys`
    __shared__ int sh_cnt, sh_raw, sh_ovf;
    if (threadIdx.x == 0) { sh_cnt = 0; sh_raw = 0; sh_ovf = 0; }
    for (int i = threadIdx.x; i < CAND_CAP; i += 256) keys[i] = 0ull;
    __syncthreads();
    auto push = [&](const Cand cd) {
        const int p = atomicAdd(&sh_cnt, 1);
        if (p < CAND_CAP)
            keys[p] = ((unsigned long long)f32_orderable(cd.s) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)cd.row);
    };
    for (int rg = threadIdx.x; rg < A.regions; rg += 256) {
        const int cr = A.region_cnt[(int64_t)slot_q * A.regions + rg];
        if (cr > 0) {
            atomicAdd(&sh_raw, cr);
            const int take = cr < A.slots ? cr : A.slots;
            const Cand *src = A.cand + ((int64_t)slot_q * A.regions + rg) * A.slots;
            for (int i = 0; i < take; ++i) push(src[i]);
        }
    }
    {
        const int oc = A.ovf_cnt[slot_q];
        if (oc > CAND_OVF && threadIdx.x == 0) sh_ovf = 1;         // overflow list itself overflowed
        const int take = oc < CAND_OVF ? oc : CAND_OVF;
        for (int i = threadIdx.x; i < take; i += 256) push(A.ovf[(int64_t)slot_q * CAND_OVF + i]);
    }
    __syncthreads();
    const int c_raw = sh_raw;
    const bool overflow = sh_cnt > CAND_CAP || sh_ovf != 0;
    const int c = sh_cnt > CAND_CAP ? CAND_CAP : sh_cnt;
    const int64_t ex = A.exclude ? A.exclude[qi] : -1;
    const int k_eff = A.k + (ex >= 0 ? 1 : 0);
    const int m2 = pow2_ceil(c > 1 ? c : 1);
    bitonic_lds(keys, m2, [](unsigned long long x, unsigned long long y) { return x > y; });
    if (threadIdx.x == 0) {
        int ok = (!overflow && c >= k_eff) ? 1 : 0, retry = 0;
        float lo = 0.0f;
        if (ok) {
            const float kth = f32_from_orderable((uint32_t)(keys[k_eff - 1] >> 32));
            const float gmax = __uint_as_float(*A.gnorm_max_bits);
            const float qn = A.qnorm[qi];
            // |s~ - s| <= eps: bf16 rounding of both operands (2^-8 + 2^-18), fp32 accumulation
            // of dimp products (dimp * 2^-23, a 2x margin over gamma_n), and for metric 1 the
            // fp32 bias and its add.  Every product below is nudged upward.
            float eps = qn * gmax * (0.00390625f + 3.8147e-6f + (float)A.dimp * 1.1920929e-7f) * 1.00001f;
            if (METRIC == 1) eps += 1.1920929e-7f * (gmax * gmax + qn * gmax) * 1.00001f;
            lo = kth - 2.0f * eps;
            lo -= fabsf(lo) * 2.4e-7f + 1e-37f;            // round toward -inf with margin
            ok = (A.tau[slot_q] < lo) ? 1 : 0;
            if (!ok) {
                atomicAdd((unsigned long long *)&A.stats->incomplete, 1ull);
                if (A.retry_list) {
                    // the k-th best s~ is final (every row above tau was seen), so tau2 just below
                    // `lo` makes the next filter pass complete by construction
                    const int p = atomicAdd(A.fail_count + 1, 1);
                    A.retry_list[p] = qi;
                    A.tau2[qi] = lo - (fabsf(lo) * 2.4e-7f + 1e-37f);
                    retry = 1;
                }
            }
        } else {
            if (overflow) atomicAdd((unsigned long long *)&A.stats->overflowed, 1ull);
            else atomicAdd((unsigned long long *)&A.stats->incomplete, 1ull);
        }
        if (!ok && !retry) {
            const int p = atomicAdd(A.fail_count, 1);
            A.fail_list[p] = qi;
        }
        sh_ok = ok;
        sh_lo = lo;
        sh_m = 0;
    }
    __syncthreads();
    if (!sh_ok) return;
    const float lo = sh_lo;
    {
        int mine = 0;
        for (int i = threadIdx.x; i < c; i += 256)
            mine += f32_from_orderable((uint32_t)(keys[i] >> 32)) >= lo ? 1 : 0;
        if (mine) atomicAdd(&sh_m, mine);
    }
    __syncthreads();
    const int m = sh_m;                                   // sorted, so these are keys[0..m)
    const float *qrow = A.q32p + (int64_t)qi * A.dimp;
    for (int i = wave; i < m; i += 4) {
        const uint32_t row = 0xFFFFFFFFu - (uint32_t)(keys[i] & 0xFFFFFFFFull);
        const double s = lane_tree_score<METRIC>(qrow, A.g32 + (int64_t)row * A.dimp, A.dimp);
        if (lane == 0) {
            const int64_t id = A.ids[row];
            Hit h;
            h.s = (id == ex) ? -INFINITY : s;
            h.id = (id == ex) ? ID_LAST : id;
            hits[i] = h;
        }
    }
    const int mp = pow2_ceil(m > 1 ? m : 1);
    for (int i = m + threadIdx.x; i < mp; i += 256) { hits[i].s = -INFINITY; hits[i].id = ID_LAST; }
    bitonic_lds(hits, mp, [](const Hit &x, const Hit &y) { return hit_before(x.s, x.id, y.s, y.id); });
    for (int r = threadIdx.x; r < A.k; r += 256) {
        Hit h;
        h.s = -INFINITY;
        h.id = ID_LAST;
        if (r < mp) h = hits[r];
        const bool empty = h.id == ID_LAST;
        const int64_t o = (int64_t)qi * A.k + r;
        if (A.out_f64) A.out_f64[o] = empty ? -INFINITY : h.s;
        A.out_ids[o] = empty ? -1 : h.id;
        if (A.out_val) {
            float v = -INFINITY;
            if (!empty) v = METRIC == 0 ? (float)h.s : (float)(-sqrt(fmax(-h.s, 0.0)));
            A.out_val[o] = v;
        }
    }
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long *)&A.stats->tier1_answered, 1ull);
        atomicAdd((unsigned long long *)&A.stats->candidates, (unsigned long long)c_raw);
        atomicAdd((unsigned long long *)&A.stats->reranked, (unsigned long long)m);
    }
}

__global__ __launch_bounds__(256) void k_gather_queries(const uint16_t *__restrict__ q16,
                                                        const float *__restrict__ tau2,
                                                        const int32_t *__restrict__ list, int n, int dimp,
                                                        uint16_t *__restrict__ q16r, float *__restrict__ taur) {
    const int64_t i = blockIdx.x;
    const int chunks = dimp / 8;                                    // 16-byte pieces per row
    uint4 *dst = reinterpret_cast<uint4 *>(q16r + i * dimp);
    if (i < n) {
        const int32_t qi = list[i];
        const uint4 *src = reinterpret_cast<const uint4 *>(q16 + (int64_t)qi * dimp);
        for (int c = threadIdx.x; c < chunks; c += 256) dst[c] = src[c];
        if (threadIdx.x == 0) taur[i] = tau2[qi];
    } else {
        for (int c = threadIdx.x; c < chunks; c += 256) dst[c] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) taur[i] = INFINITY;
    }
}

}  // namespace

hipError_t launch_gather_queries(const uint16_t *q16, const float *tau2, const int32_t *list, int n,
                                 int64_t n_pad, int dimp, uint16_t *q16r, float *taur, hipStream_t st) {
    if (n_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_queries, dim3((unsigned)n_pad), dim3(256), 0, st, q16, tau2, list, n, dimp, q16r,
                       taur);
    return hipGetLastError();
}

hipError_t launch_select_tau(const float *groupmax, int ngroups, int64_t nq, int64_t nq_pad, int rank_j,
                             float *tau, hipStream_t st) {
    if (nq_pad <= 0) return hipSuccess;
    int m = 1;
    while (m < ngroups) m <<= 1;
    if (m > 16384) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_select_tau),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * (int)sizeof(float));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select_tau, dim3((unsigned)nq_pad), dim3(256), (size_t)m * sizeof(float), st,
                       groupmax, ngroups, nq, rank_j, tau);
    return hipGetLastError();
}

hipError_t launch_finalize(const FinalizeArgs &a, hipStream_t st) {
    if (a.nq <= 0) return hipSuccess;
    if (a.metric == MIRX_METRIC_IP)
        hipLaunchKernelGGL(k_finalize<0>, dim3(a.nq), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(k_finalize<1>, dim3(a.nq), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mirx
