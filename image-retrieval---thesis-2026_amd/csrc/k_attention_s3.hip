// k_attention_s3.hip -- the flash attention of k_attention.hip (fp32 in, fp32 out, scores never in HBM) with BOTH
// GEMMs on the bf16 matrix pipe, every fp32 operand carried as three bf16 terms (x = xh + xm + xl exactly; six
// v_mfma_f32_32x32x16_bf16 per product block, dropped cross terms <= 3 * 2^-24 |a b|: fp32-grade):
//   S^T[key, q] = K[key, :] . Q[q, :]         A = K tile (LDS, split while staged), B = Q (split once, registers)
//   O^T[d, q]  += V^T[d, key] . P^T[key, q]   A = V^T tile (LDS, split + transposed while staged),
//                                             B = P, split in registers straight from the S accumulators
// 48 MFMAs of 32 cycles per 32-key tile instead of 64 fp32 MFMAs of 64 cycles: 2.67x less matrix time.
//
// Geometry as k_attention: one workgroup = 128 queries (4 waves x 32) of one (image, head), keys 32 at a time, a
// lane owns one query (online softmax per lane + one exchange with lane ^ 32).  The accumulator layout of the
// first GEMM hands lane (q, h) the keys  kappa(r, h) = 8 (r >> 2) + (r & 3) + 4 h,  r = 0..15; the second GEMM
// contracts the keys in exactly that order -- step s, k-group h, element i  <->  key kappa(8 s + i, h) -- so P
// never moves between lanes, and V^T is stored with its key axis permuted to match (position 16 s + 8 h + i).
// LDS rows: K planes [32 keys][64 ch] bf16 (128 B rows, 16-byte chunk c at c ^ ((key >> 1) & 7)); V^T planes
// [64 d][32 positions] bf16 (64 B rows, chunk c at c ^ ((d >> 2) & 3)): the chunk index changes every 256 bytes.
// k_attention_s3: head_dim 64 (ViT-B / DINOv2); k_attention_s3g<DH>: 32 / 72 / 96.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int DH = 64;                 // head dimension
constexpr int KT = 32;                 // keys per tile
constexpr int K_PLANE = KT * DH * 2;   // bytes of one term of the K tile (4 KiB)
constexpr int V_PLANE = DH * KT * 2;   // bytes of one term of the V^T tile (4 KiB)
constexpr int BUF = 3 * K_PLANE + 3 * V_PLANE;   // 24 KiB

__device__ inline void split2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
    const f32x2 v = {a, b};
    const bf16x2 vh = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(vh, f32x2);
    const bf16x2 vm = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(vm, f32x2);
    const bf16x2 vl = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned, vh);
    m = __builtin_bit_cast(unsigned, vm);
    l = __builtin_bit_cast(unsigned, vl);
}

// 8 fp32 values -> three bf16x8 fragments
__device__ inline void split8(const float (&v)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
    u32x4 ph, pm, pl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned th, tm, tl;
        split2(v[2 * p], v[2 * p + 1], th, tm, tl);
        ph[p] = th; pm[p] = tm; pl[p] = tl;
    }
    h = __builtin_bit_cast(bf16x8, ph);
    m = __builtin_bit_cast(bf16x8, pm);
    l = __builtin_bit_cast(bf16x8, pl);
}

#define MIRX_MFMA6(C, AH, AM, AL, BH, BM, BL)                                       \
    {                                                                               \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BM, C, 0, 0, 0);            \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, C, 0, 0, 0);            \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, C, 0, 0, 0);            \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BH, C, 0, 0, 0);            \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BM, C, 0, 0, 0);            \
        C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, C, 0, 0, 0);            \
    }

__global__ __launch_bounds__(256, 3) void k_attention_s3(const float *__restrict__ qkv, int n, int heads,
                                                         float scale_log2e, float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) char sm[2 * BUF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, nq = lane & 31;
    const int head = blockIdx.y;
    const int64_t img = blockIdx.z;
    const int64_t tok = 3 * (int64_t)heads * DH;                          // floats per token in qkv
    const float *base = qkv + img * n * tok + head * DH;                   // q of token t: base + t*tok; k: + heads*DH; v: + 2*heads*DH
    const int q_idx = blockIdx.x * 128 + wave * 32 + nq;
    const int q_ld = q_idx < n ? q_idx : n - 1;

    // this lane's query: channels 16 ks + 8 half + i, pre-multiplied by scale * log2(e), three terms each
    bf16x8 qh[4], qm[4], ql[4];
    {
        const float *qp = base + q_ld * tok + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(qp + 16 * ks), b = *reinterpret_cast<const f32x4 *>(qp + 16 * ks + 4);
            const float v[8] = {a[0] * scale_log2e, a[1] * scale_log2e, a[2] * scale_log2e, a[3] * scale_log2e,
                                b[0] * scale_log2e, b[1] * scale_log2e, b[2] * scale_log2e, b[3] * scale_log2e};
            split8(v, qh[ks], qm[ks], ql[ks]);
        }
    }

    // ---- staging assignments ------------------------------------------------------------------------------------
    // K: thread -> key t >> 3, channels 8 (t & 7) .. + 7 (two float4)        -> one 16-byte chunk per term
    // V: thread -> keys 2 (t >> 4), 2 (t >> 4) + 1, channels 4 (t & 15) .. + 3  -> four bf16 pairs per term
    const int kk = threadIdx.x >> 3, kc = threadIdx.x & 7;
    const int k_lds = kk * 128 + ((kc ^ ((kk >> 1) & 7)) << 4);                   // + term * K_PLANE
    const int vm_ = threadIdx.x >> 4, vc = threadIdx.x & 15;
    const int vkey = 2 * vm_;
    // position of key k on the permuted axis: 16 (k >> 4) + 8 ((k >> 2) & 1) + (k & 3) + 4 ((k >> 3) & 1)
    const int vpos = 16 * (vkey >> 4) + 8 * ((vkey >> 2) & 1) + (vkey & 3) + 4 * ((vkey >> 3) & 1);   // even
    int v_lds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = 4 * vc + j;
        v_lds[j] = 3 * K_PLANE + d * 64 + ((((vpos >> 3) ^ ((d >> 2) & 3))) << 4) + (vpos & 7) * 2;   // + term * V_PLANE
    }
    f32x4 rk[2], rv[2];
    auto load_tile = [&](int kt) {
        int key = kt * KT + kk;
        if (key >= n) key = n - 1;                                        // masked later
        const float *kp = base + key * tok + heads * DH + 8 * kc;
        rk[0] = *reinterpret_cast<const f32x4 *>(kp);
        rk[1] = *reinterpret_cast<const f32x4 *>(kp + 4);
        int k0 = kt * KT + vkey, k1 = k0 + 1;
        if (k0 >= n) k0 = n - 1;
        if (k1 >= n) k1 = n - 1;                                          // P is 0 there
        rv[0] = *reinterpret_cast<const f32x4 *>(base + k0 * tok + 2 * heads * DH + 4 * vc);
        rv[1] = *reinterpret_cast<const f32x4 *>(base + k1 * tok + 2 * heads * DH + 4 * vc);
    };
    auto store_tile = [&](int buf) {
        char *sb = sm + buf * BUF;
        u32x4 ph, pm, pl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned th, tm, tl;
            split2(rk[p >> 1][2 * (p & 1)], rk[p >> 1][2 * (p & 1) + 1], th, tm, tl);
            ph[p] = th; pm[p] = tm; pl[p] = tl;
        }
        *reinterpret_cast<u32x4 *>(sb + k_lds) = ph;
        *reinterpret_cast<u32x4 *>(sb + k_lds + K_PLANE) = pm;
        *reinterpret_cast<u32x4 *>(sb + k_lds + 2 * K_PLANE) = pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned th, tm, tl;
            split2(rv[0][j], rv[1][j], th, tm, tl);                        // keys 2m, 2m + 1 of channel 4 vc + j
            *reinterpret_cast<unsigned *>(sb + v_lds[j]) = th;
            *reinterpret_cast<unsigned *>(sb + v_lds[j] + V_PLANE) = tm;
            *reinterpret_cast<unsigned *>(sb + v_lds[j] + 2 * V_PLANE) = tl;
        }
    };

    // fragment addresses: K row nq (key), chunk 2 ks + half; V^T row 32 t + nq (d), chunk 2 s + half
    int fk[4], fv[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fk[ks] = nq * 128 + (((2 * ks + half) ^ ((nq >> 1) & 7)) << 4);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int d = 32 * t + nq;
            fv[t][s] = 3 * K_PLANE + d * 64 + (((2 * s + half) ^ ((d >> 2) & 3)) << 4);
        }

    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int ntiles = (n + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int cur = kt & 1;
        __syncthreads();                                   // tile kt visible; buffer cur ^ 1 free
        load_tile(kt + 1 < ntiles ? kt + 1 : kt);          // branch-free: the last tile re-loads itself
        __builtin_amdgcn_sched_barrier(0);
        const char *sb = sm + cur * BUF;

        // ---- S^T = K Q^T ----------------------------------------------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fk[ks]);
            const bf16x8 am = *reinterpret_cast<const bf16x8 *>(sb + fk[ks] + K_PLANE);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fk[ks] + 2 * K_PLANE);
            MIRX_MFMA6(sacc, ah, am, al, qh[ks], qm[ks], ql[ks])
        }

        // ---- online softmax over this lane's 16 keys (base 2) -----------------------------------------------------
        const int key0 = kt * KT + 4 * half;
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + 8 * (r >> 2) + (r & 3);
            if (key >= n) sacc[r] = -INFINITY;
            mt = fmaxf(mt, sacc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));            // the other 16 keys of the same query
        const float m_new = fmaxf(m_run, mt);              // finite: every tile holds at least one valid key
        const float alpha = exp2f(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = exp2f(sacc[r] - m_new);
            psum += sacc[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // ---- O^T += V^T P^T: step s contracts the keys kappa(8 s + i, half) = this lane's sacc[8 s + i] -------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float pv[8] = {sacc[8 * s], sacc[8 * s + 1], sacc[8 * s + 2], sacc[8 * s + 3],
                                 sacc[8 * s + 4], sacc[8 * s + 5], sacc[8 * s + 6], sacc[8 * s + 7]};
            bf16x8 bh, bm, bl;
            split8(pv, bh, bm, bl);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s]);
                const bf16x8 am = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s] + V_PLANE);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s] + 2 * V_PLANE);
                MIRX_MFMA6(o[t], ah, am, al, bh, bm, bl)
            }
        }
        store_tile(cur ^ 1);
    }

    // ---- normalise and store: register r of o[t] is channel 32 t + 8 (r >> 2) + (r & 3) + 4 half ------------------
    l_run += __shfl_xor(l_run, 32, 64);
    if (q_idx < n) {
        const float inv = 1.0f / l_run;
        float *op = out + ((img * n + q_idx) * heads + head) * DH + 4 * half;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[t][4 * g + j] * inv;
                *reinterpret_cast<f32x4 *>(op + 32 * t + 8 * g) = v;
            }
    }
}

// ---- any head_dim that is a multiple of 8 (72: the SigLIP-So400m tower) --------------------------------------------
// Same algorithm; what changes against the head_dim-64 kernel above: the K rows hold KS = ceil(DH / 16) MFMA steps
// (channels beyond DH are zero in Q and K), their 16-byte chunks are rotated by key >> 2 instead of XORed (the
// chunk count is not a power of two); the output has NT = ceil(DH / 32) tiles, the lanes of a partly filled last
// tile re-read row DH - 1 of V^T and their results are dropped; staging walks item lists (K: key x chunk, V: key
// pair x 4 channels) because they no longer divide by 256 threads.
template <int DH>
__global__ __launch_bounds__(256, 2) void k_attention_s3g(const float *__restrict__ qkv, int n, int heads,
                                                          float scale_log2e, float *__restrict__ out) {
    constexpr int KS = (DH + 15) / 16, KCH = 2 * KS, KROW = KCH * 16;      // K row: KCH chunks of 16 B
    constexpr int NT = (DH + 31) / 32;
    constexpr int KPL = KT * KROW, VPL = DH * 64, GBUF = 3 * (KPL + VPL);
    extern __shared__ __attribute__((aligned(16))) char smg[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, nq = lane & 31;
    const int head = blockIdx.y;
    const int64_t img = blockIdx.z;
    const int64_t tok = 3 * (int64_t)heads * DH;
    const float *base = qkv + img * n * tok + head * DH;
    const int q_idx = blockIdx.x * 128 + wave * 32 + nq;
    const int q_ld = q_idx < n ? q_idx : n - 1;

    bf16x8 qh[KS], qm[KS], ql[KS];
    {
        const float *qp = base + q_ld * tok;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = 16 * ks + 8 * half;                             // DH % 8 == 0: a chunk is all in or all out
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (c0 < DH) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(qp + c0), b = *reinterpret_cast<const f32x4 *>(qp + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = a[j] * scale_log2e; v[4 + j] = b[j] * scale_log2e; }
            }
            split8(v, qh[ks], qm[ks], ql[ks]);
        }
    }

    // ---- staging item lists ----------------------------------------------------------------------------------------
    constexpr int NKI = KT * KCH, IPK = (NKI + 255) / 256;                // K items: (key, chunk of 8 channels)
    constexpr int NVI = (KT / 2) * (DH / 4), IPV = (NVI + 255) / 256;     // V items: (key pair, 4 channels)
    f32x4 rk[IPK][2], rv[IPV][2];
    auto k_item = [&](int i, int &key, int &c, bool &live) {
        int it = threadIdx.x + 256 * i;
        live = it < NKI;
        if (!live) it = NKI - 1;
        key = it / KCH;
        c = it % KCH;
    };
    auto v_item = [&](int i, int &m, int &vc, bool &live) {
        int it = threadIdx.x + 256 * i;
        live = it < NVI;
        if (!live) it = NVI - 1;
        m = it / (DH / 4);
        vc = it % (DH / 4);
    };
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < IPK; ++i) {
            int key, c; bool live;
            k_item(i, key, c, live);
            int g = kt * KT + key;
            if (g >= n) g = n - 1;
            const int cc = 8 * c < DH ? 8 * c : DH - 8;                    // pad chunk: read something valid, store zeros
            const float *kp = base + g * tok + heads * DH + cc;
            rk[i][0] = *reinterpret_cast<const f32x4 *>(kp);
            rk[i][1] = *reinterpret_cast<const f32x4 *>(kp + 4);
        }
#pragma unroll
        for (int i = 0; i < IPV; ++i) {
            int m, vc; bool live;
            v_item(i, m, vc, live);
            int k0 = kt * KT + 2 * m, k1 = k0 + 1;
            if (k0 >= n) k0 = n - 1;
            if (k1 >= n) k1 = n - 1;
            rv[i][0] = *reinterpret_cast<const f32x4 *>(base + k0 * tok + 2 * heads * DH + 4 * vc);
            rv[i][1] = *reinterpret_cast<const f32x4 *>(base + k1 * tok + 2 * heads * DH + 4 * vc);
        }
    };
    auto store_tile = [&](int buf) {
        char *sb = smg + buf * GBUF;
#pragma unroll
        for (int i = 0; i < IPK; ++i) {
            int key, c; bool live;
            k_item(i, key, c, live);
            const bool real = 8 * c < DH;
            u32x4 ph, pm, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned th, tm, tl;
                split2(real ? rk[i][p >> 1][2 * (p & 1)] : 0.f, real ? rk[i][p >> 1][2 * (p & 1) + 1] : 0.f, th, tm, tl);
                ph[p] = th; pm[p] = tm; pl[p] = tl;
            }
            if (live) {
                const int pos = (c + (key >> 2)) % KCH;
                char *d = sb + key * KROW + pos * 16;
                *reinterpret_cast<u32x4 *>(d) = ph;
                *reinterpret_cast<u32x4 *>(d + KPL) = pm;
                *reinterpret_cast<u32x4 *>(d + 2 * KPL) = pl;
            }
        }
#pragma unroll
        for (int i = 0; i < IPV; ++i) {
            int m, vc; bool live;
            v_item(i, m, vc, live);
            const int vkey = 2 * m;
            const int vpos = 16 * (vkey >> 4) + 8 * ((vkey >> 2) & 1) + (vkey & 3) + 4 * ((vkey >> 3) & 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned th, tm, tl;
                split2(rv[i][0][j], rv[i][1][j], th, tm, tl);
                const int d = 4 * vc + j;
                char *dst = sb + 3 * KPL + d * 64 + (((vpos >> 3) ^ ((d >> 2) & 3)) << 4) + (vpos & 7) * 2;
                if (live) {
                    *reinterpret_cast<unsigned *>(dst) = th;
                    *reinterpret_cast<unsigned *>(dst + VPL) = tm;
                    *reinterpret_cast<unsigned *>(dst + 2 * VPL) = tl;
                }
            }
        }
    };

    int fk[KS], fv[NT][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int pos = (2 * ks + half + (nq >> 2)) % KCH;
        fk[ks] = nq * KROW + pos * 16;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int d = 32 * t + nq < DH ? 32 * t + nq : DH - 1;
            fv[t][s] = 3 * KPL + d * 64 + (((2 * s + half) ^ ((d >> 2) & 3)) << 4);
        }

    f32x16 o[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int ntiles = (n + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int cur = kt & 1;
        __syncthreads();
        load_tile(kt + 1 < ntiles ? kt + 1 : kt);
        __builtin_amdgcn_sched_barrier(0);
        const char *sb = smg + cur * GBUF;

        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fk[ks]);
            const bf16x8 am = *reinterpret_cast<const bf16x8 *>(sb + fk[ks] + KPL);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fk[ks] + 2 * KPL);
            MIRX_MFMA6(sacc, ah, am, al, qh[ks], qm[ks], ql[ks])
        }

        const int key0 = kt * KT + 4 * half;
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + 8 * (r >> 2) + (r & 3);
            if (key >= n) sacc[r] = -INFINITY;
            mt = fmaxf(mt, sacc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2f(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = exp2f(sacc[r] - m_new);
            psum += sacc[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float pv[8] = {sacc[8 * s], sacc[8 * s + 1], sacc[8 * s + 2], sacc[8 * s + 3],
                                 sacc[8 * s + 4], sacc[8 * s + 5], sacc[8 * s + 6], sacc[8 * s + 7]};
            bf16x8 bh, bm, bl;
            split8(pv, bh, bm, bl);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s]);
                const bf16x8 am = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s] + VPL);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s] + 2 * VPL);
                MIRX_MFMA6(o[t], ah, am, al, bh, bm, bl)
            }
        }
        store_tile(cur ^ 1);
    }

    l_run += __shfl_xor(l_run, 32, 64);
    if (q_idx < n) {
        const float inv = 1.0f / l_run;
        float *op = out + ((img * n + q_idx) * heads + head) * DH + 4 * half;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * t + 8 * g + 4 * half >= DH) continue;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[t][4 * g + j] * inv;
                *reinterpret_cast<f32x4 *>(op + 32 * t + 8 * g) = v;
            }
    }
}

template <int DH>
hipError_t launch_s3g(const float *qkv, int64_t batch, int n, int heads, float scale, float *out, hipStream_t st) {
    constexpr int KS = (DH + 15) / 16;
    const size_t lds = (size_t)2 * 3 * (KT * KS * 32 + DH * 64);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attention_s3g<DH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((n + 127) / 128), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(k_attention_s3g<DH>, grid, dim3(256), lds, st, qkv, n, heads, scale * 1.4426950408889634f, out);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_s3(const float *qkv, int64_t batch, int n, int heads, int head_dim, float scale, float *out,
                               hipStream_t st) {
    if (batch <= 0 || n <= 0) return hipSuccess;
    if (heads <= 0 || heads > 65535 || batch > 65535) return hipErrorInvalidValue;
    if (head_dim == 72) return launch_s3g<72>(qkv, batch, n, heads, scale, out, st);
    if (head_dim == 96) return launch_s3g<96>(qkv, batch, n, heads, scale, out, st);
    if (head_dim == 32) return launch_s3g<32>(qkv, batch, n, heads, scale, out, st);
    if (head_dim != DH) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 127) / 128), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(k_attention_s3, grid, dim3(256), 0, st, qkv, n, heads, scale * 1.4426950408889634f, out);
    return hipGetLastError();
}

}  // namespace mirx
