// k_attention.hip -- fp32 multi-head self-attention for the ViT backbones (DINOv2 ViT-B/14: 12 heads of
// 64 over 1370 tokens at 518x518), on v_mfma_f32_32x32x2_f32.
//
// Replaces softmax(q k^T / sqrt(d)) v of the timm / DINOv2 attention block (reference
// model.py:459-463 `vit_base_patch14_dinov2`, nih_multilabel_retrieval.py:175-221) in one pass: scores
// and probabilities never reach HBM (flash-attention recurrence), fp32 throughout.
//
// Input is the packed projection `qkv` [B, N, 3, H, 64] exactly as the block's Linear produces it and
// the output is [B, N, H, 64] = [B, N, C], so neither the head permute nor the transpose back exist.
//
// One workgroup = 128 queries (4 waves x 32) of one (image, head); it walks the keys 32 at a time.
// Both GEMMs are computed TRANSPOSED so that a lane owns ONE query and no cross-lane traffic is needed
// between them:
//   S^T[key, q] = K[key, :] . Q[q, :]        A = K tile (LDS), B = Q (32 registers, pre-scaled)
//       -> lane (q, half) holds the 16 keys k(r) = 8 (r >> 2) + (r & 3) + 4 half of its query:
//          max / exp / sum are per-lane loops plus ONE exchange with lane ^ 32;
//   O^T[d, q]  += V^T[d, key] . P^T[key, q]  A = V tile (LDS), B = P straight from the registers above:
//       the contraction visits the keys in the order k(r), half -- V rows are fetched in that order.
// K is staged de-interleaved by channel parity (the A operand of step s is channel 2s + half), rows
// pitched at 4 mod 8 floats (36 for head_dim 64 and 72) so the ds_read_b128 of 16 different keys hit 16
// different bank groups.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int KT = 32;                 // keys per tile

// DH = head dimension, a multiple of 8 up to 96 (64: DINOv2 / ViT-B; 72: the SigLIP-So400m tower of MedSigLIP).
// A head wider than a multiple of 32 pays a partly idle third output tile (72: 84 MFMAs per key tile for 72
// channels' worth instead of 64 for 64).
template <int DH>
__global__ __launch_bounds__(256) void k_attention(const float *__restrict__ qkv, int n, int heads, float scale_log2e,
                                                   float *__restrict__ out) {
    constexpr int HP = DH / 2;                           // channels per parity plane
    constexpr int KP = (HP % 8 == 4) ? HP : HP + 4;      // plane row pitch = 4 mod 8 floats: 16 keys' ds_read_b128 hit 16 bank groups
    constexpr int VP = DH;                               // pitch of a V row
    constexpr int NT = (DH + 31) / 32;                   // output tiles of 32 channels
    constexpr int C4 = DH / 4;                           // float4 chunks of a key row
    constexpr int CPT = (C4 + 7) / 8;                    // chunks per staging thread
    __shared__ __attribute__((aligned(16))) float s_k[2][2][KT][KP];      // [buffer][parity][key][channel / 2]
    __shared__ __attribute__((aligned(16))) float s_v[2][KT][VP];         // [buffer][key][channel]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, nq = lane & 31;
    const int head = blockIdx.y;
    const int64_t img = blockIdx.z;
    const int64_t tok = 3 * (int64_t)heads * DH;                          // floats per token in qkv
    const float *base = qkv + img * n * tok + head * DH;                   // q of token t: base + t*tok; k: + heads*DH; v: + 2*heads*DH
    const int q_idx = blockIdx.x * 128 + wave * 32 + nq;
    const int q_ld = q_idx < n ? q_idx : n - 1;

    // this lane's query, channels 2s + half, pre-multiplied by scale * log2(e): softmax runs on exp2
    float qf[HP];
    {
        const float *qp = base + q_ld * tok + half;
#pragma unroll
        for (int s = 0; s < HP; ++s) qf[s] = qp[2 * s] * scale_log2e;
    }

    // staging: thread -> key t / 8, float4 chunks (t & 7), (t & 7) + 8, ... of K and of V (8 threads = 128 B)
    const int st_key = threadIdx.x >> 3, st_c = threadIdx.x & 7;
    f32x4 rk[CPT], rv[CPT];
    auto load_tile = [&](int kt) {
        int key = kt * KT + st_key;
        if (key >= n) key = n - 1;                                        // masked later
        const float *kp = base + key * tok + heads * DH;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = st_c + 8 * j;
            if (c < C4) {
                rk[j] = *reinterpret_cast<const f32x4 *>(kp + 4 * c);
                rv[j] = *reinterpret_cast<const f32x4 *>(kp + heads * DH + 4 * c);
            }
        }
    };
    auto store_tile = [&](int buf) {
        // float4 chunk c covers channels 4 c .. 4 c + 3 -> parity planes, index channel / 2
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = st_c + 8 * j;
            if (c < C4) {
                *reinterpret_cast<float2 *>(&s_k[buf][0][st_key][2 * c]) = make_float2(rk[j][0], rk[j][2]);
                *reinterpret_cast<float2 *>(&s_k[buf][1][st_key][2 * c]) = make_float2(rk[j][1], rk[j][3]);
                *reinterpret_cast<f32x4 *>(&s_v[buf][st_key][4 * c]) = rv[j];
            }
        }
    };

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
        __syncthreads();                                   // tile kt visible; buffer cur ^ 1 free
        if (kt + 1 < ntiles) load_tile(kt + 1);

        // ---- S^T = K Q^T -------------------------------------------------------------------------
        float kf[HP];
#pragma unroll
        for (int c = 0; c < HP / 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(&s_k[cur][half][nq][4 * c]);
            kf[4 * c] = v[0]; kf[4 * c + 1] = v[1]; kf[4 * c + 2] = v[2]; kf[4 * c + 3] = v[3];
        }
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < HP; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], sacc, 0, 0, 0);

        // ---- online softmax over this lane's 16 keys (base 2) --------------------------------------
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
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;

        // ---- O^T += V^T P^T: step r contracts keys k(r) (half 0) and k(r) + 4 (half 1) ------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float *vrow = &s_v[cur][8 * (r >> 2) + (r & 3) + 4 * half][0];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // a partly filled last tile: the lanes beyond DH re-read channel DH - 1; their output rows are dropped
                const int ch = (32 * t + 31 < DH) ? 32 * t + nq : (32 * t + nq < DH ? 32 * t + nq : DH - 1);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[ch], sacc[r], o[t], 0, 0, 0);
            }
        }
        if (kt + 1 < ntiles) store_tile(cur ^ 1);
    }

    // ---- normalise and store: register r of o[t] is channel 32 t + 8 (r >> 2) + (r & 3) + 4 half ------
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

}  // namespace

hipError_t launch_attention(const float *qkv, int64_t batch, int n, int heads, int head_dim, float scale, float *out,
                            hipStream_t st) {
    if (batch <= 0 || n <= 0) return hipSuccess;
    if (heads <= 0 || heads > 65535 || batch > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n + 127) / 128), (unsigned)heads, (unsigned)batch);
    const float sl = scale * 1.4426950408889634f;
    switch (head_dim) {
        case 32: hipLaunchKernelGGL(k_attention<32>, grid, dim3(256), 0, st, qkv, n, heads, sl, out); break;
        case 64: hipLaunchKernelGGL(k_attention<64>, grid, dim3(256), 0, st, qkv, n, heads, sl, out); break;
        case 72: hipLaunchKernelGGL(k_attention<72>, grid, dim3(256), 0, st, qkv, n, heads, sl, out); break;
        case 96: hipLaunchKernelGGL(k_attention<96>, grid, dim3(256), 0, st, qkv, n, heads, sl, out); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mirx
