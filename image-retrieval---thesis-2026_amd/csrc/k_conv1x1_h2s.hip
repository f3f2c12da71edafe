// k_conv1x1_h2s.hip -- the 1x1 convolution of k_conv1x1_h2.hip for SMALL launches (the reference's own batch sizes: one
// image per query in milvus/milvus_retrieval.py:53-66, 32 in ingest_embeddings.py:465-469, 64 in test.py:1513).
//
// k_conv1x1_h2 gives a workgroup 128 output channels x 128 / 256 pixels and walks the input channels in a serial loop of
// 16-channel stages (barrier, LDS round trip, ~0.6 us each).  A 14 x 14 image is two such workgroups, a 7 x 7 one a quarter
// of one: below a few hundred images a launch is a handful of workgroups, each a chain of up to 64 dependent stages,
// and 250 of the chip's CUs idle (B = 1: 25 us per layer, 2.4 ms per forward).
//
// This kernel is the same arithmetic cut the other way: ONE WAVE = one 32 x 32 accumulator tile (32 output channels x 32
// pixels), no LDS, no barrier.  A lane loads the 8 input values of its pixel and channel half straight into the B
// fragment (BN + ReLU + the fp16 split applied in registers, as the staging threads of the big kernel do before their
// LDS store) and its 16 bytes of each weight term straight from L2 into the A fragment; loads run NSTAGE stages ahead in
// a register ring (counted vmcnt by the compiler: every access is a plain load in program order).  4 x more workgroups
// than tiles of 128 channels and 4-8 x more than tiles of 128 / 256 pixels; the four waves that share a pixel tile
// re-read its activations (L2 hits; at these sizes bytes are not what a layer costs).
//
// Bit-identical to k_conv1x1_h2 by construction: an output is accumulated by the same v_mfma_f32_32x32x16_f16 sequence
// (stages in order; per stage lo_w hi_x, hi_w lo_x, hi_w hi_x; the same 8 channels per lane half at the same fragment
// positions), the operands are formed by the same instructions (split2h_pair on act(x) * 2^s), and the epilogue applies
// the same power-of-two scales and the same fmaf.  tests/test_model_gpu.py checks rows of a small batch against the same
// images inside a 2048-image batch (which takes the big kernel) bit for bit.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int KC = 16;             // channels per stage
constexpr int CMB = 128;           // output channels per weight block of w2 ([cout / 128][cin / 16][2][128][16])
constexpr int MAX_CIN = 1024;      // the BN vectors of the prologue live in LDS (DenseNet: cin <= 1024)

// NCB = 32-channel blocks per wave.  1: the most waves per launch (one image: every wave is a chain of cin / 16 dependent
// stages, and the layer's latency is that chain).  2: the activation loads -- 8 of the 10 load instructions of a stage -- feed
// twice the MFMAs (mid-size launches, where the waves queue on the load path rather than wait for one chain).
template <bool PROLOGUE, bool RELU_OUT, bool YTERMS, int NCB>
__global__ __launch_bounds__(64) void k_conv1x1_h2s(const float *__restrict__ x, int64_t xbs, int cin,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    const uint16_t *__restrict__ w2, const float *__restrict__ oscale,
                                                    const float *__restrict__ bias, int64_t n, int hw, int cout,
                                                    float *__restrict__ y, int64_t ybs, const float *__restrict__ in_amax,
                                                    float in_ks, float in_kb, unsigned *__restrict__ out_amax, float y_ks,
                                                    float y_kb, float *__restrict__ y_inv_out, int64_t xps, int64_t yps) {
    constexpr int NSTAGE = NCB == 1 ? 6 : 4;               // stages of loads in flight (24 / 40 registers per stage)
    __shared__ __attribute__((aligned(16))) float s_bn[PROLOGUE ? 2 * MAX_CIN : 4];     // [scale | shift][cin]
    const int lane = threadIdx.x & 63;
    const int kg = lane >> 5;                               // which 8 of a stage's 16 channels this lane stages
    const int64_t total = n * (int64_t)hw;
    const int nk = cin / KC;
    const int cb0 = blockIdx.y * NCB;                       // first 32-channel block of the output this wave owns

    // ---- this lane's pixel: loads, scale and epilogue all belong to it ----------------------------------
    const unsigned pp_raw = blockIdx.x * 32u + (lane & 31);
    const bool live = pp_raw < (unsigned)total;
    const unsigned pp = live ? pp_raw : (unsigned)total - 1u;       // a dead lane computes a valid pixel and stores nothing
    const unsigned pimg = pp / (unsigned)hw;
    const unsigned off = pp - pimg * (unsigned)hw;
    const float *xsrc = x + (int64_t)pimg * xbs + off + (int64_t)(8 * kg) * xps;
    // weights: row (lane & 31) of each 32-channel block, chunk kg of the stage, both terms
    const uint16_t *wsrc[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
        const int co0 = (cb0 + j) * 32;
        wsrc[j] = w2 + ((int64_t)(co0 / CMB) * nk * 2 * CMB + (co0 % CMB) + (lane & 31)) * KC + 8 * kg;
    }
    const int64_t w_stage = 2 * CMB * KC, w_term = CMB * KC;

    float xr[NSTAGE][8];
    f16x8 wr[NSTAGE][NCB][2];
    auto issue = [&](int kt, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[s][j] = xsrc[((int64_t)kt * KC + j) * xps];
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            wr[s][j][0] = *reinterpret_cast<const f16x8 *>(wsrc[j] + (int64_t)kt * w_stage);
            wr[s][j][1] = *reinterpret_cast<const f16x8 *>(wsrc[j] + (int64_t)kt * w_stage + w_term);
        }
    };
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s) issue(s < nk ? s : nk - 1, s);

    // the prologue's BN vectors: once into LDS (a ring slot of their own per stage cost 16 registers per stage in flight)
    if (PROLOGUE) {
        for (int i = lane * 4; i < cin; i += 256) {
            *reinterpret_cast<f32x4 *>(s_bn + i) = *reinterpret_cast<const f32x4 *>(scale + i);
            *reinterpret_cast<f32x4 *>(s_bn + MAX_CIN + i) = *reinterpret_cast<const f32x4 *>(shift + i);
        }
    }
    // the range of this pixel's image -> its power-of-two staging scale (read behind the first loads)
    const float xb = fmaf(in_ks, in_amax ? in_amax[pimg] : 0.f, in_kb);
    float x_scale, x_inv;
    range_scales(xb, x_scale, x_inv);

    f32x16 acc[NCB];
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    f32x4 bnc[2][2], bnh[2][2];                           // scale / shift of the stage in hand and of the next one
    auto read_bn = [&](int kt, int q) __attribute__((always_inline)) {
        if (PROLOGUE) {
            const int kk = kt < nk ? kt : nk - 1;
            bnc[q][0] = *reinterpret_cast<const f32x4 *>(s_bn + kk * KC + 8 * kg);
            bnc[q][1] = *reinterpret_cast<const f32x4 *>(s_bn + kk * KC + 8 * kg + 4);
            bnh[q][0] = *reinterpret_cast<const f32x4 *>(s_bn + MAX_CIN + kk * KC + 8 * kg);
            bnh[q][1] = *reinterpret_cast<const f32x4 *>(s_bn + MAX_CIN + kk * KC + 8 * kg + 4);
        }
    };
    static_assert(NSTAGE % 2 == 0, "the BN double buffer alternates with the stage parity");
    read_bn(0, 0);

    for (int kt0 = 0; kt0 < nk; kt0 += NSTAGE) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {
            const int kt = kt0 + s;
            read_bn(kt + 1, (s + 1) & 1);
            // the two fp16 terms of act(x) * 2^s, pairs of channels (round to nearest even)
            u32x4 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0 = xr[s][2 * j], v1 = xr[s][2 * j + 1];
                if (PROLOGUE) {
                    v0 = fmaxf(fmaf(v0, bnc[s & 1][j >> 1][(2 * j) & 3], bnh[s & 1][j >> 1][(2 * j) & 3]), 0.f);
                    v1 = fmaxf(fmaf(v1, bnc[s & 1][j >> 1][(2 * j + 1) & 3], bnh[s & 1][j >> 1][(2 * j + 1) & 3]), 0.f);
                }
                unsigned th, tl;
                split2h_pair(v0 * x_scale, v1 * x_scale, th, tl);
                ph[j] = th;
                pl[j] = tl;
            }
            const f16x8 bh = __builtin_bit_cast(f16x8, ph), bl = __builtin_bit_cast(f16x8, pl);
            f16x8 ah[NCB], al[NCB];
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                ah[j] = wr[s][j][0];
                al[j] = wr[s][j][1];
            }
            // refill the slot: stage kt + NSTAGE (clamped: a load is never behind a branch)
            issue(kt + NSTAGE < nk ? kt + NSTAGE : nk - 1, s);
            if (kt < nk) {
#pragma unroll
                for (int j = 0; j < NCB; ++j) {
                    // smallest terms first, as k_conv1x1_h2
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[j], bh, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bl, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[j], bh, acc[j], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: register r of block j = channel 32 (cb0 + j) + (r & 3) + 8 (r >> 2) + 4 kg, pixel = this lane's ----------
    if (YTERMS) {
        // y as the 3x3 conv wants it: [image][group of 16 channels][term][pixel][16] fp16; group 2 cb + kg holds exactly the 16
        // channels this lane owns, in register order: a pixel's record of a (group, term) is the 32 bytes of ONE lane
        float ys_, y_inv;
        range_scales(fmaf(y_ks, xb, y_kb), ys_, y_inv);
        if (live && off == 0 && cb0 == 0 && kg == 0) y_inv_out[pimg] = y_inv;     // one writer per image
#pragma unroll
        for (int jb = 0; jb < NCB; ++jb) {
            const int co0 = (cb0 + jb) * 32;
            u32x4 h[2], l[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int r = 2 * j + e;
                    const int ch = co0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    float t = fmaf(acc[jb][r], oscale[ch] * x_inv, bias ? bias[ch] : 0.f);
                    t = __int_as_float(max(__float_as_int(t), 0));      // ReLU on the bits (k_conv1x1_h2)
                    v[e] = t * ys_;
                }
                unsigned hh, ll;
                split2h_pair(v[0], v[1], hh, ll);
                h[j >> 2][j & 3] = hh;
                l[j >> 2][j & 3] = ll;
            }
            if (live) {
                uint16_t *yt = reinterpret_cast<uint16_t *>(y);
                const int g = 2 * (cb0 + jb) + kg;
                uint16_t *dst = yt + ((int64_t)pimg * 16 * hw + off + (int64_t)(2 * g) * hw) * 16;
                *reinterpret_cast<u32x4 *>(dst) = h[0];
                *reinterpret_cast<u32x4 *>(dst + 8) = h[1];
                *reinterpret_cast<u32x4 *>(dst + (int64_t)hw * 16) = l[0];
                *reinterpret_cast<u32x4 *>(dst + (int64_t)hw * 16 + 8) = l[1];
            }
        }
        return;
    }
    float vmax = 0.f;
    if (live) {
        float *yo = y + (int64_t)pimg * ybs + off;
#pragma unroll
        for (int jb = 0; jb < NCB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (cb0 + jb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                float v = fmaf(acc[jb][r], oscale[ch] * x_inv, bias ? bias[ch] : 0.f);
                if (RELU_OUT) v = v < 0.f ? 0.f : v;          // keeps a NaN (fmaxf would turn it into 0)
                vmax = range_max(vmax, v);
                yo[(int64_t)ch * yps] = v;
            }
    }
    if (out_amax) range_publish_lanes(out_amax, (int)pimg, vmax, lane);
}

}  // namespace

// called by launch_conv1x1_h2 (k_conv1x1_h2.hip) for launches that would leave most CUs without a workgroup
hipError_t launch_conv1x1_h2_small(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                                   const uint16_t *w2, const float *oscale, const float *bias, int64_t n, int hw, int cout,
                                   int relu_out, float *y, int64_t ybs, const float *in_amax, float in_ks, float in_kb,
                                   float *out_amax, float y_ks, float y_kb, float *y_inv_out, int64_t xps, int64_t yps,
                                   hipStream_t st) {
    if (cin > MAX_CIN && scale) return hipErrorInvalidValue;      // (launch_conv1x1_h2 keeps such a layer on the tiled kernel)
    const int64_t px = n * (int64_t)hw;
    const int64_t waves1 = ((px + 31) / 32) * (cout / 32);
    // two channel blocks per wave once one block per wave gives every CU several waves anyway
    const int ncb = (waves1 >= 1024 && cout % 64 == 0) ? 2 : 1;
    const dim3 grid((unsigned)((px + 31) / 32), (unsigned)(cout / (32 * ncb)));
    unsigned *oa = reinterpret_cast<unsigned *>(out_amax);
#define MIRX_H2S_(P, R, T, N)                                                                                       \
    hipLaunchKernelGGL((k_conv1x1_h2s<P, R, T, N>), grid, dim3(64), 0, st, x, xbs, cin, scale, shift, w2, oscale, bias, n, \
                       hw, cout, y, ybs, in_amax, in_ks, in_kb, oa, y_ks, y_kb, y_inv_out, xps, yps)
#define MIRX_H2S(P, R, T) \
    {                     \
        if (ncb == 2) MIRX_H2S_(P, R, T, 2); else MIRX_H2S_(P, R, T, 1); \
    }
    if (y_inv_out) {
        MIRX_H2S(true, true, true)
    } else if (scale) {
        if (relu_out) MIRX_H2S(true, true, false) else MIRX_H2S(true, false, false)
    } else {
        if (relu_out) MIRX_H2S(false, true, false) else MIRX_H2S(false, false, false)
    }
#undef MIRX_H2S
#undef MIRX_H2S_
    return hipGetLastError();
}

}  // namespace mirx
