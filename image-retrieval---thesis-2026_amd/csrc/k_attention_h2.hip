// k_attention_h2.hip -- k_attention_s3 (flash attention, fp32 in / fp32 out, head_dim 64) with every operand carried as
// TWO fp16 terms and THREE v_mfma_f32_32x32x16_f16 per product block instead of three bf16 terms and six MFMAs (see
// k_linear_h2.hip for the arithmetic and the measured error).  fp16's range is the caller's contract: `qk_bound` >=
// max |q|, |k| and `v_bound` >= max |v| over the packed projection (mirx.model derives them from the LayerNorm and
// the projection's row norms); the launcher turns them into exact power-of-two scales: Q is staged as
// q * (scale log2 e) * qs, K as k * ks, V as v * vs, the probabilities (in [0, 1]) as p * 1024, and the
// accumulators are multiplied back by 1 / (qs ks) before the softmax and 1 / (1024 vs) at the end.
// Layout, key permutation and LDS swizzles are k_attention_s3's; an LDS buffer is 16 KiB instead of 24.
// k_attention_h2: head_dim 64 (ViT-B / DINOv2); k_attention_h2g<DH>: 32 / 72 / 96.
#include "mirx_kernels.h"

namespace mirx {

namespace {

// 2^x as the bare v_exp_f32.  exp2f() wraps the instruction in a compare, two selects, an add and a multiply so that results below
// 2^-126 come out as denormals; a softmax weight that small changes neither the running sum (>= 1) nor its bf16 terms, and the
// wrapper was 4 of every 6 VALU instructions of the softmax.
__device__ inline float exp2_raw(float x) { return __builtin_amdgcn_exp2f(x); }

// max of a value over the two 32-lane halves of the wave, in every lane: v_permlane32_swap (gfx950) hands each half the other's
// value inside the vector unit -- the ds_bpermute of __shfl_xor(.., 32) was an LDS round trip on the critical path of every tile
__device__ inline float max_over_halves(float v) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
    const unsigned b = __float_as_uint(v);
    const u32x2_ r = __builtin_amdgcn_permlane32_swap(b, b, false, false);    // r[0] = the low half's value, r[1] = the high half's
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;   // (fp16 here)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;

// Diagnostic builds (results wrong, timing only; -DMIRX_DIAG): bits of MIRX_ATT_EXP in k_attention_h2 -- 1 no softmax arithmetic,
// 2 no tile staging after the first tile, 4 no P V MFMAs, 8 no Q K MFMAs, 16 no workgroup barrier in the K loop
#ifndef MIRX_ATT_EXP
#define MIRX_ATT_EXP 0
#endif
constexpr int DH = 64;                 // head dimension
constexpr int KT = 32;                 // keys per tile
constexpr int K_PLANE = KT * DH * 2;   // bytes of one term of the K tile (4 KiB)
constexpr int V_PLANE = DH * KT * 2;   // bytes of one term of the V^T tile (4 KiB)
constexpr int BUF = 2 * K_PLANE + 2 * V_PLANE;   // 16 KiB

__device__ inline void split2(float a, float b, unsigned &h, unsigned &l) {
    split2h_pair(a, b, h, l);
}

// 8 fp32 values -> two fp16x8 fragments
__device__ inline void split8(const float (&v)[8], bf16x8 &h, bf16x8 &l) {
    u32x4 ph, pl;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned th, tl;
        split2(v[2 * p], v[2 * p + 1], th, tl);
        ph[p] = th; pl[p] = tl;
    }
    h = __builtin_bit_cast(bf16x8, ph);
    l = __builtin_bit_cast(bf16x8, pl);
}

#define MIRX_MFMA3(C, AH, AL, BH, BL)                                               \
    {                                                                               \
        C = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, C, 0, 0, 0);             \
        C = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, C, 0, 0, 0);             \
        C = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, C, 0, 0, 0);             \
    }

// XCD-aware order: workgroups are dealt to the 8 XCDs round-robin in linear order (x fastest), so the query tiles of one
// (image, head) -- which all stream the same K and V -- would sit behind eight different L2s.  Re-dealt, XCD x takes the
// (image, head) pairs x, x + 8, .. query tile by query tile: K / V of a pair cross the fabric once.  (Any bijection is correct.)
__device__ inline void attention_block(int &qt, int &head, int64_t &img) {
    qt = blockIdx.x;
    head = blockIdx.y;
    img = blockIdx.z;
#ifndef MIRX_ATT_PLAIN_ORDER
    const unsigned nqt = gridDim.x, npair = gridDim.y * gridDim.z, full = npair & ~7u;
    const unsigned lin = blockIdx.x + nqt * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin < nqt * full) {
        const unsigned j = lin >> 3;
        const unsigned pair = (j / nqt) * 8 + (lin & 7);
        qt = (int)(j % nqt);
        head = (int)(pair % gridDim.y);
        img = pair / gridDim.y;
    }
#endif
}

// Four consecutive channels `ch ..` (ch % 4 == 0) of token row `row` written as "terms rows" (k_linear_t2.hip: per 32 features one
// 128-byte line, fp16 high terms | fp16 low terms of scale * value): the input format of the DMA-fed Linear that follows.
__device__ inline void store_terms4(char *out_t, int64_t row, int c, int ch, const f32x4 &v, float scale) {
    unsigned h0, l0, h1, l1;
    split2h_pair(v[0] * scale, v[1] * scale, h0, l0);
    split2h_pair(v[2] * scale, v[3] * scale, h1, l1);
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
    const u32x2_ hi = {h0, h1}, lo = {l0, l1};
    char *dst = out_t + row * ((int64_t)((c + 31) / 32 * 32) * 4) + (ch >> 5) * 128 + (ch & 31) * 2;
    *reinterpret_cast<u32x2_ *>(dst) = hi;
    *reinterpret_cast<u32x2_ *>(dst + 64) = lo;
}

__global__ __launch_bounds__(256, 3) void k_attention_h2(const float *__restrict__ qkv, int n, int heads,
                                                         float q_mul, float k_mul, float v_mul, float s_inv, float o_inv,
                                                         float *__restrict__ out, char *__restrict__ out_t, float t_scale) {
    __shared__ __attribute__((aligned(16))) char sm[2 * BUF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, nq = lane & 31;
    int qt, head;
    int64_t img;
    attention_block(qt, head, img);
    const int64_t tok = 3 * (int64_t)heads * DH;                          // floats per token in qkv
    const float *base = qkv + img * n * tok + head * DH;                   // q of token t: base + t*tok; k: + heads*DH; v: + 2*heads*DH
    const int q_idx = qt * 128 + wave * 32 + nq;
    const int q_ld = q_idx < n ? q_idx : n - 1;

    // this lane's query: channels 16 ks + 8 half + i, pre-multiplied by scale * log2(e), three terms each
    bf16x8 qh[4], ql[4];
    {
        const float *qp = base + q_ld * tok + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(qp + 16 * ks), b = *reinterpret_cast<const f32x4 *>(qp + 16 * ks + 4);
            const float v[8] = {a[0] * q_mul, a[1] * q_mul, a[2] * q_mul, a[3] * q_mul,
                                b[0] * q_mul, b[1] * q_mul, b[2] * q_mul, b[3] * q_mul};
            split8(v, qh[ks], ql[ks]);
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
        v_lds[j] = 2 * K_PLANE + d * 64 + ((((vpos >> 3) ^ ((d >> 2) & 3))) << 4) + (vpos & 7) * 2;   // + term * V_PLANE
    }
    // K / V rows come through a buffer descriptor over THIS image's n token rows: a lane's byte offset inside a tile is fixed
    // for the whole launch and the tile's offset is one scalar, so the loop holds no address arithmetic (it was ~40 of its ~250
    // vector instructions: clamps, 32-bit multiplies, 64-bit adds); rows beyond n are out of range and read as zero -- their
    // scores are set to -inf in the tail tile and their probabilities are exactly 0.
    const __amdgpu_buffer_rsrc_t kvrs = __builtin_amdgcn_make_buffer_rsrc((void *)(qkv + img * n * tok), 0,
                                                                          (unsigned)((int64_t)n * tok * 4), 0x00020000);
    const unsigned vo_k = (unsigned)((kk * tok + (heads + head) * DH + 8 * kc) * 4);
    const unsigned vo_v = (unsigned)((vkey * tok + (2 * heads + head) * DH + 4 * vc) * 4);
    const unsigned row_b = (unsigned)(tok * 4), tile_b = (unsigned)(KT * tok * 4);
    f32x4 rk[2], rv[2];
    auto load_tile = [&](int kt) {
        const unsigned so = (unsigned)kt * tile_b;                        // wave-uniform
        rk[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_k, so, 0));
        rk[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_k + 16u, so, 0));
        rv[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_v, so, 0));
        rv[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_v + row_b, so, 0));
    };
    auto store_tile = [&](int buf) {
        char *sb = sm + buf * BUF;
        u32x4 ph, pl;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned th, tl;
            split2(rk[p >> 1][2 * (p & 1)] * k_mul, rk[p >> 1][2 * (p & 1) + 1] * k_mul, th, tl);
            ph[p] = th; pl[p] = tl;
        }
        *reinterpret_cast<u32x4 *>(sb + k_lds) = ph;
        *reinterpret_cast<u32x4 *>(sb + k_lds + K_PLANE) = pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned th, tl;
            split2(rv[0][j] * v_mul, rv[1][j] * v_mul, th, tl);            // keys 2m, 2m + 1 of channel 4 vc + j
            *reinterpret_cast<unsigned *>(sb + v_lds[j]) = th;
            *reinterpret_cast<unsigned *>(sb + v_lds[j] + V_PLANE) = tl;
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
            fv[t][s] = 2 * K_PLANE + d * 64 + (((2 * s + half) ^ ((d >> 2) & 3)) << 4);
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
        if (!(MIRX_ATT_EXP & 16)) __syncthreads();         // tile kt visible; buffer cur ^ 1 free
        if (!(MIRX_ATT_EXP & 2)) load_tile(kt + 1 < ntiles ? kt + 1 : kt);          // branch-free: the last tile re-loads itself
        __builtin_amdgcn_sched_barrier(0);
        const char *sb = sm + cur * BUF;

        // ---- S^T = K Q^T ----------------------------------------------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fk[ks]);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fk[ks] + K_PLANE);
            if (MIRX_ATT_EXP & 8) { sacc[ks] += (float)ah[0] + (float)al[1] + (float)qh[ks][2]; continue; }
            MIRX_MFMA3(sacc, ah, al, qh[ks], ql[ks])
        }
        // (the scores stay unscaled: s_inv, a power of two, goes into the exponent's multiply-add below -- the same bits)

        // ---- online softmax over this lane's 16 keys (base 2) -----------------------------------------------------
        const int key0 = kt * KT + 4 * half;
        const bool tail_tile = (kt + 1) * KT > n;          // wave-uniform: only the last tile can hold keys beyond n
        float mt = -INFINITY;
        if (tail_tile) {
            // a real branch (the empty asm keeps the compiler from turning it into 16 compares, 16 selects and 16 scalar ORs
            // executed for EVERY tile: a fifth of the loop's vector instructions)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + 8 * (r >> 2) + (r & 3);
                if (key >= n) sacc[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
        mt = max_over_halves(mt) * s_inv;    // the other 16 keys of the same query; max commutes with the positive scale
        const float m_new = fmaxf(m_run, mt);              // finite: every tile holds at least one valid key
        const float alpha = exp2_raw(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!(MIRX_ATT_EXP & 1)) sacc[r] = exp2_raw(fmaf(sacc[r], s_inv, -m_new));
            psum += sacc[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!(MIRX_ATT_EXP & 1) && !__all(alpha == 1.0f)) {                       // wave-uniform: once the running maxima have settled there is nothing to rescale
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }

        // ---- O^T += V^T P^T: step s contracts the keys kappa(8 s + i, half) = this lane's sacc[8 s + i] -------------
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float pv[8] = {sacc[8 * s] * 1024.f, sacc[8 * s + 1] * 1024.f, sacc[8 * s + 2] * 1024.f, sacc[8 * s + 3] * 1024.f,
                                 sacc[8 * s + 4] * 1024.f, sacc[8 * s + 5] * 1024.f, sacc[8 * s + 6] * 1024.f, sacc[8 * s + 7] * 1024.f};
            bf16x8 bh, bl;
            split8(pv, bh, bl);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s]);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s] + V_PLANE);
                if (MIRX_ATT_EXP & 4) { o[t][s] += (float)ah[0] + (float)al[1] + (float)bh[2] + (float)bl[3]; continue; }
                MIRX_MFMA3(o[t], ah, al, bh, bl)
            }
        }
        if (!(MIRX_ATT_EXP & 2)) store_tile(cur ^ 1);
    }

    // ---- normalise and store: register r of o[t] is channel 32 t + 8 (r >> 2) + (r & 3) + 4 half ------------------
    l_run += __shfl_xor(l_run, 32, 64);
    if (q_idx < n) {
        const float inv = o_inv / l_run;
        float *op = out + ((img * n + q_idx) * heads + head) * DH + 4 * half;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[t][4 * g + j] * inv;
                if (out_t) store_terms4(out_t, (img * n + q_idx), heads * DH, head * DH + 4 * half + 32 * t + 8 * g, v, t_scale);
                else *reinterpret_cast<f32x4 *>(op + 32 * t + 8 * g) = v;
            }
    }
}

// ---- any head_dim that is a multiple of 8 (72: the SigLIP-So400m tower) --------------------------------------------
// k_attention_s3g's generalisation (K rows of KS = ceil(DH / 16) MFMA steps with zero channels beyond DH, 16-byte chunks
// rotated by key >> 2, NT = ceil(DH / 32) output tiles, staging by item lists) on the two-fp16-term arithmetic above.
template <int DH>
__global__ __launch_bounds__(256, 2) void k_attention_h2g(const float *__restrict__ qkv, int n, int heads, float q_mul,
                                                          float k_mul, float v_mul, float s_inv, float o_inv,
                                                          float *__restrict__ out, char *__restrict__ out_t, float t_scale) {
    constexpr int KS = (DH + 15) / 16, KCH = 2 * KS, KROW = KCH * 16;      // K row: KCH chunks of 16 B
    constexpr int NT = (DH + 31) / 32;
    constexpr int KPL = KT * KROW, VPL = DH * 64, GBUF = 2 * (KPL + VPL);
    extern __shared__ __attribute__((aligned(16))) char smg[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, nq = lane & 31;
    int qt, head;
    int64_t img;
    attention_block(qt, head, img);
    const int64_t tok = 3 * (int64_t)heads * DH;
    const float *base = qkv + img * n * tok + head * DH;
    const int q_idx = qt * 128 + wave * 32 + nq;
    const int q_ld = q_idx < n ? q_idx : n - 1;

    bf16x8 qh[KS], ql[KS];
    {
        const float *qp = base + q_ld * tok;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = 16 * ks + 8 * half;                             // DH % 8 == 0: a chunk is all in or all out
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (c0 < DH) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(qp + c0), b = *reinterpret_cast<const f32x4 *>(qp + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = a[j] * q_mul; v[4 + j] = b[j] * q_mul; }
            }
            split8(v, qh[ks], ql[ks]);
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
    // (buffer loads with one scalar tile offset: see k_attention_h2)
    const __amdgpu_buffer_rsrc_t kvrs = __builtin_amdgcn_make_buffer_rsrc((void *)(qkv + img * n * tok), 0,
                                                                          (unsigned)((int64_t)n * tok * 4), 0x00020000);
    const unsigned row_b = (unsigned)(tok * 4), tile_b = (unsigned)(KT * tok * 4);
    unsigned vo_k[IPK], vo_v[IPV];
#pragma unroll
    for (int i = 0; i < IPK; ++i) {
        int key, c; bool live;
        k_item(i, key, c, live);
        const int cc = 8 * c < DH ? 8 * c : DH - 8;                        // pad chunk: read something valid, store zeros
        vo_k[i] = (unsigned)((key * tok + (heads + head) * DH + cc) * 4);
    }
#pragma unroll
    for (int i = 0; i < IPV; ++i) {
        int m, vc; bool live;
        v_item(i, m, vc, live);
        vo_v[i] = (unsigned)((2 * m * tok + (2 * heads + head) * DH + 4 * vc) * 4);
    }
    auto load_tile = [&](int kt) {
        const unsigned so = (unsigned)kt * tile_b;                        // wave-uniform
#pragma unroll
        for (int i = 0; i < IPK; ++i) {
            rk[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_k[i], so, 0));
            rk[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_k[i] + 16u, so, 0));
        }
#pragma unroll
        for (int i = 0; i < IPV; ++i) {
            rv[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_v[i], so, 0));
            rv[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kvrs, vo_v[i] + row_b, so, 0));
        }
    };
    auto store_tile = [&](int buf) {
        char *sb = smg + buf * GBUF;
#pragma unroll
        for (int i = 0; i < IPK; ++i) {
            int key, c; bool live;
            k_item(i, key, c, live);
            const float km = 8 * c < DH ? k_mul : 0.f;                     // pad chunk: zeros
            u32x4 ph, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned th, tl;
                split2(rk[i][p >> 1][2 * (p & 1)] * km, rk[i][p >> 1][2 * (p & 1) + 1] * km, th, tl);
                ph[p] = th; pl[p] = tl;
            }
            if (live) {
                const int pos = (c + (key >> 2)) % KCH;
                char *d = sb + key * KROW + pos * 16;
                *reinterpret_cast<u32x4 *>(d) = ph;
                *reinterpret_cast<u32x4 *>(d + KPL) = pl;
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
                unsigned th, tl;
                split2(rv[i][0][j] * v_mul, rv[i][1][j] * v_mul, th, tl);
                const int d = 4 * vc + j;
                char *dst = sb + 2 * KPL + d * 64 + (((vpos >> 3) ^ ((d >> 2) & 3)) << 4) + (vpos & 7) * 2;
                if (live) {
                    *reinterpret_cast<unsigned *>(dst) = th;
                    *reinterpret_cast<unsigned *>(dst + VPL) = tl;
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
            fv[t][s] = 2 * KPL + d * 64 + (((2 * s + half) ^ ((d >> 2) & 3)) << 4);
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
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fk[ks] + KPL);
            MIRX_MFMA3(sacc, ah, al, qh[ks], ql[ks])
        }
        // (the scores stay unscaled: s_inv, a power of two, goes into the exponent's multiply-add below -- the same bits)

        const int key0 = kt * KT + 4 * half;
        const bool tail_tile = (kt + 1) * KT > n;          // wave-uniform: only the last tile can hold keys beyond n
        float mt = -INFINITY;
        if (tail_tile) {
            // a real branch (the empty asm keeps the compiler from turning it into 16 compares, 16 selects and 16 scalar ORs
            // executed for EVERY tile: a fifth of the loop's vector instructions)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + 8 * (r >> 2) + (r & 3);
                if (key >= n) sacc[r] = -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[r]);
        mt = max_over_halves(mt) * s_inv;
        const float m_new = fmaxf(m_run, mt);
        const float alpha = exp2_raw(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sacc[r] = exp2_raw(fmaf(sacc[r], s_inv, -m_new));
            psum += sacc[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }

#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = sacc[8 * s + j] * 1024.f;
            bf16x8 bh, bl;
            split8(pv, bh, bl);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s]);
                const bf16x8 al = *reinterpret_cast<const bf16x8 *>(sb + fv[t][s] + VPL);
                MIRX_MFMA3(o[t], ah, al, bh, bl)
            }
        }
        store_tile(cur ^ 1);
    }

    l_run += __shfl_xor(l_run, 32, 64);
    if (q_idx < n) {
        const float inv = o_inv / l_run;
        float *op = out + ((img * n + q_idx) * heads + head) * DH + 4 * half;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (32 * t + 8 * g + 4 * half >= DH) continue;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[t][4 * g + j] * inv;
                if (out_t) store_terms4(out_t, (img * n + q_idx), heads * DH, head * DH + 4 * half + 32 * t + 8 * g, v, t_scale);
                else *reinterpret_cast<f32x4 *>(op + 32 * t + 8 * g) = v;
            }
    }
}

template <int DH>
hipError_t launch_h2g(const float *qkv, int64_t batch, int n, int heads, float q_mul, float k_mul, float v_mul, float s_inv,
                      float o_inv, float *out, char *out_t, float t_scale, hipStream_t st) {
    constexpr int KS = (DH + 15) / 16;
    const size_t lds = (size_t)2 * 2 * (KT * KS * 32 + DH * 64);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_attention_h2g<DH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)((n + 127) / 128), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(k_attention_h2g<DH>, grid, dim3(256), lds, st, qkv, n, heads, q_mul, k_mul, v_mul, s_inv, o_inv, out, out_t, t_scale);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_h2(const float *qkv, int64_t batch, int n, int heads, int head_dim, float scale,
                               float qk_bound, float v_bound, float *out, hipStream_t st, void *out_terms, float terms_scale) {
    if (batch <= 0 || n <= 0) return hipSuccess;
    if (heads <= 0 || heads > 65535 || batch > 65535) return hipErrorInvalidValue;
    if (head_dim != DH && head_dim != 72 && head_dim != 96 && head_dim != 32) return hipErrorInvalidValue;
    if (!(qk_bound > 0.f) || !(v_bound > 0.f) || !(scale > 0.f)) return hipErrorInvalidValue;
    if ((out != nullptr) == (out_terms != nullptr) || (heads * head_dim) % 4) return hipErrorInvalidValue;
    if ((int64_t)n * 3 * heads * head_dim * 4 >= ((int64_t)1 << 31)) return hipErrorInvalidValue;   // an image's qkv rows: 32-bit buffer offsets
    const float sl = scale * 1.4426950408889634f;
    // powers of two that bring each operand's bound to at most 2^14 (fp16 max 65504)
    const float qs = exp2f(floorf(log2f(16384.0f / (qk_bound * sl))));
    const float ks = exp2f(floorf(log2f(16384.0f / qk_bound)));
    const float vs = exp2f(floorf(log2f(16384.0f / v_bound)));
    const float s_inv = 1.0f / (qs * ks), o_inv = 1.0f / (1024.0f * vs);
    if (head_dim == 72) return launch_h2g<72>(qkv, batch, n, heads, sl * qs, ks, vs, s_inv, o_inv, out, reinterpret_cast<char *>(out_terms), terms_scale, st);
    if (head_dim == 96) return launch_h2g<96>(qkv, batch, n, heads, sl * qs, ks, vs, s_inv, o_inv, out, reinterpret_cast<char *>(out_terms), terms_scale, st);
    if (head_dim == 32) return launch_h2g<32>(qkv, batch, n, heads, sl * qs, ks, vs, s_inv, o_inv, out, reinterpret_cast<char *>(out_terms), terms_scale, st);
    const dim3 grid((unsigned)((n + 127) / 128), (unsigned)heads, (unsigned)batch);
    hipLaunchKernelGGL(k_attention_h2, grid, dim3(256), 0, st, qkv, n, heads, sl * qs, ks, vs, s_inv, o_inv, out, reinterpret_cast<char *>(out_terms), terms_scale);
    return hipGetLastError();
}

}  // namespace mirx
