// k_conv3x3_d2q.hip -- k_conv3x3_d2p (DenseNet dense-layer 3x3 conv, 128 -> 32 channels, the bottleneck handed over as two
// fp16 terms, staging by LDS DMA alone) re-tiled for v_mfma_f32_16x16x32_f16.
//
// Why: the kernel is POWER-bound, not pipe- or HBM-bound.  A diagnostic build without any DMA behind stage 0 ran as fast as
// the real one, and a one-wave clock probe (tools/clock_probe.py --h2) reads 1.2 GHz under it (2.4 idle): the matrix pipe is
// ~87 % busy at half the nominal clock.  At equal cycles per FLOP the 16x16x32 shape sustains 1.13-1.17x the FLOP/s of
// 32x32x16 under the power limit (tools/mfma_power.hip: 1.97-2.02 vs 1.72-1.76 PFLOP/s fp16, whole chip, random data).
//
// RESULT (same box, 1024 images): parity identical; the clock under this kernel is 1.60 GHz instead of 1.29 -- and the layer
// takes 0.747 ms instead of 0.727 (28: 0.229 / 0.222, 14: 0.073 / 0.065).  Busy cycles x clock is the same for both: what the
// power limit fixes is the rate of delivered fp16 FLOP *with their operand traffic* (54-56 KiB of ds_read_b128 per wave and
// stage either way), and the cheaper instruction shape buys nothing once every operand comes from LDS.  k_conv3x3_d2p stays
// the default; this file is the A/B arm (mirx_conv3x3_direct_terms_nchw_mfma16).
//
//   D[pixel, oc] += X[pixel, k] * Wt[k, oc]      A rows = 16 output pixels, B columns = 16 output channels, K = 32
//
// K = 32 with 16-channel stages (two stages of a 32-channel K step would double the LDS footprint and halve the occupancy):
//   * taps (0,1), (2,3), (4,5), (6,7): K = [tap t, 16 c | tap t + 1, 16 c] -- lanes 0..31 of an operand read tap t, lanes
//     32..63 tap t + 1 (a per-lane constant in the address); three MFMAs per pair: xh wl + xl wh + xh wh;
//   * tap 8: K = [hi term | lo term] of the pixels against [wh | wh] and [wl | wl]: two MFMAs = xh wh + xl wh + xh wl + xl wl
//     (the last product is the one the three-MFMA form drops as 2^-22; here it rides along in the K half that would idle).
//   14 MFMAs of 16 cycles per (16 x 16 tile, stage) instead of the ideal 13.5; 56 instead of 54 ds_read_b128 per wave-stage.
// LDS images are NOT swizzled: a ds_read_b128 lane group holds rows {0-3, 12-15} of K chunk c and rows {4-11} of chunk
// c + 1, i.e. rows r and r + 8 always read opposite 16-byte halves of their 32-byte rows -- conflict-free on linear rows.
// The accumulator of a lane is 4 consecutive pixels of one channel: one 16-byte store per tile.
#include <cstdint>

#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int CIN = 128, COUT = 32;
constexpr int KC = 16;                        // channels per stage
constexpr int NST = CIN / KC;                 // 8 stages
constexpr int NT = 4;                         // pixel tiles per wave (tile = wave + 4 t)

// W = map side (56 / 28 / 14); R = output rows per strip (4 / 8 / 14)
template <int W, int R>
__global__ __launch_bounds__(256, 2) void k_conv3x3_d2q(const uint16_t *__restrict__ yt, const uint16_t *__restrict__ w3,
                                                        const float *__restrict__ oscale, float *__restrict__ out,
                                                        int64_t out_bs, const float *__restrict__ in_inv,
                                                        unsigned *__restrict__ out_range, int64_t out_ps) {
    constexpr int PW = W + 2, PR = R + 2;     // padded strip
    constexpr int NPIX = PR * PW;             // padded pixels of a stage
    constexpr int NP = (NPIX + 31) / 32;      // 1-KiB DMA pieces per term plane
    constexpr int PLANE = NP * 32 * 32;       // bytes of one term of one stage (32 B per pixel, rounded up to whole pieces)
    constexpr int STAGE = 2 * PLANE;
    constexpr int WSTAGE = 9 * 2 * COUT * KC * 2;   // bytes of one stage of weights (18 KiB): [tap][term][32 oc][16 c]
    constexpr int W_LDS0 = 2 * STAGE;               // weight buffers behind the two activation buffers
    constexpr int NOUT = R * W;               // output pixels of a full strip
    constexpr int NTILE = (NOUT + 15) / 16;   // 14 / 14 / 13
    static_assert(NTILE <= 4 * NT, "four pixel tiles per wave");
    static_assert(W % 4 == 0 || R == W, "a lane's 4 pixels stay inside one image row, or the strip is the whole image");
    constexpr int PPW = (2 * NP + 3) / 4;     // activation pieces per wave and stage (both terms)
    extern __shared__ __attribute__((aligned(16))) char sm[];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row16 = lane & 15, chunk = lane >> 4, hi = chunk >> 1, half = chunk & 1;
    // XCD-aware order: XCD x walks images x, x + 8, .. strip by strip, so the halo rows of a strip are in the L2 its
    // neighbour just filled (see k_conv3x3_d2p)
    int strip = blockIdx.x;
    int64_t img = blockIdx.y;
    {
        const unsigned nstrip = gridDim.x, lin = blockIdx.x + nstrip * blockIdx.y, full = gridDim.y & ~7u;
        if (lin < nstrip * full) {
            const unsigned j = lin >> 3;
            strip = (int)(j % nstrip);
            img = (int64_t)(j / nstrip) * 8 + (lin & 7);
        }
    }
    const int oy0 = strip * R;                                // first output row of the strip
    constexpr unsigned IMG_BYTES = 16u * W * W * 32u;        // 8 groups x 2 terms x W*W pixels x 32 B
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(yt + img * (int64_t)(IMG_BYTES / 2)), 0,
                                                                           IMG_BYTES, 0x00020000);
    // this wave's activation pieces: q = wave + 4 i over the 2 NP pieces of a stage (term = q / NP, piece = q % NP): lane l
    // -> padded pixel 32 piece + l / 2, 16-byte half l & 1; the source is the pixel's place inside the image or an offset
    // beyond num_records for the padding ring (out-of-range buffer loads return zero)
    unsigned a_src[PPW];
    int a_dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave + 4 * i;
        const int term = q / NP, piece = q % NP;
        const int pix = piece * 32 + (lane >> 1);
        const int pr = pix / PW, pc = pix % PW;
        const int iy = oy0 - 1 + pr, ix = pc - 1;
        const bool inside = q < 2 * NP && pix < NPIX && iy >= 0 && iy < W && ix >= 0 && ix < W;
        a_src[i] = inside ? (unsigned)((term * W * W + iy * W + ix) * 32 + (lane & 1) * 16) : 0xfffffff0u;
        a_dst[i] = q < 2 * NP ? term * PLANE + piece * 1024 : -1;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w3, 0, NST * WSTAGE, 0x00020000);
    auto dma_stage = [&](int st, int buf) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int piece = wave + 4 * i;               // 18 weight pieces: waves 0, 1 take five, waves 2, 3 four
            if (piece < 18)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(sm + W_LDS0 + buf * WSTAGE + piece * 1024), 16, lane * 16,
                                                         st * WSTAGE + piece * 1024, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (a_dst[i] >= 0) {                               // wave-uniform
                const unsigned v = a_src[i] == 0xfffffff0u ? a_src[i] : a_src[i] + (unsigned)st * (2u * W * W * 32u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, LDS_PTR(sm + buf * STAGE + a_dst[i]), 16, v, 0, 0, 0);
            }
    };

    // ---- operand addresses ----------------------------------------------------------------------------------------------
    // pixels (A): row16 = pixel of the tile, chunk -> (tap of the pair, 8-channel half); byte offset inside a term plane of the
    // pixel's tap (0, 0) corner, + the tap's offset (per lane: taps t / t + 1 for the two lane halves)
    bool live[NT];
    int px_off[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = wave + 4 * t;
        live[t] = tile < NTILE;                                // wave-uniform
        int p = tile * 16 + row16;
        if (p >= NOUT) p = NOUT - 1;                           // idle rows shadow a valid pixel (never stored)
        px_off[t] = ((p / W) * PW + (p % W)) * 32 + half * 16;
    }
    int tap_off[5];                                            // pairs 0..3: tap 2 j + hi; 4: tap 8 with the term by lane half
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int tap = 2 * j + hi;
        tap_off[j] = ((tap / 3) * PW + tap % 3) * 32;
    }
    tap_off[4] = (2 * PW + 2) * 32 + hi * PLANE;
    // weights (B): row16 = output channel of the tile, chunk -> (tap of the pair, 8-channel half); [tap][term][32 oc][16 c]
    const int w_off = hi * 2048 + row16 * 32 + half * 16;     // + j * 4096 + term * 1024 + oc tile * 512
    const int w8_off = 16 * 1024 + row16 * 32 + half * 16;    // tap 8: both K halves read the same term plane

    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int o = 0; o < 2; ++o) acc[t][o] = f32x4{0.f, 0.f, 0.f, 0.f};

    dma_stage(0, 0);
    for (int st = 0; st < NST; ++st) {
        const int cur = st & 1;
        // stage st landed (this wave's DMA: vmcnt(0); every wave's: the barrier); buffers cur ^ 1 free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < NST) dma_stage(st + 1, cur ^ 1);          // wave-uniform
        __builtin_amdgcn_sched_barrier(0);
        const char *sb = sm + cur * STAGE;
        const char *wb = sm + W_LDS0 + cur * WSTAGE;
        f16x8 xf[2][NT][2], wf[2][2][2];                       // [set][pixel tile][term], [set][oc tile][term]
        auto read_step = [&](int j, int set) {
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
                    wf[set][o][tm] = *reinterpret_cast<const f16x8 *>(wb + (j < 4 ? w_off + j * 4096 : w8_off) + tm * 1024 + o * 512);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (live[t]) {
                    xf[set][t][0] = *reinterpret_cast<const f16x8 *>(sb + px_off[t] + tap_off[j]);
                    if (j < 4) xf[set][t][1] = *reinterpret_cast<const f16x8 *>(sb + px_off[t] + tap_off[j] + PLANE);
                }
        };
        read_step(0, 0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int set = j & 1;
            if (j + 1 < 5) read_step(j + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (live[t]) {
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        f32x4 c = acc[t][o];
                        if (j < 4) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[set][t][0], wf[set][o][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[set][t][1], wf[set][o][0], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[set][t][0], wf[set][o][0], c, 0, 0, 0);
                        } else {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[set][t][0], wf[set][o][1], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[set][t][0], wf[set][o][0], c, 0, 0, 0);
                        }
                        acc[t][o] = c;
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- outputs: a lane holds pixels 4 chunk .. 4 chunk + 3 of its tile for channel 16 o + row16 ---------------------------
    const float x_inv = in_inv[0];
    float *oi = out + img * out_bs + (int64_t)oy0 * W;
    float vmax = 0.f;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int oc = 16 * o + row16;
        const float osc = oscale[oc] * x_inv;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int p = (wave + 4 * t) * 16 + 4 * chunk;
            if (live[t] && p < NOUT && oy0 + p / W < W) {
                f32x4 v = acc[t][o];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] *= osc;
                    vmax = range_max(vmax, v[r]);
                }
                *reinterpret_cast<f32x4 *>(oi + (int64_t)oc * out_ps + p) = v;
            }
        }
    }
    if (out_range) range_publish(out_range, vmax, lane);
}

template <int W, int R>
hipError_t launch_d2q(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, float *out, int64_t out_bs,
                      const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st) {
    constexpr int NP = ((R + 2) * (W + 2) + 31) / 32;
    const size_t lds = (size_t)2 * 2 * NP * 1024 + 2 * 9 * 2 * 32 * 16 * 2;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_d2q<W, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_conv3x3_d2q<W, R>), dim3((W + R - 1) / R, (unsigned)n), dim3(256), lds, st, yt, w2, oscale, out,
                       out_bs, in_inv, reinterpret_cast<unsigned *>(out_range), out_ps);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_conv3x3_d2q(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, int side, float *out,
                              int64_t out_bs, const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535 || !in_inv || !oscale) return hipErrorInvalidValue;
    if (!out_ps) out_ps = (int64_t)side * side;
    if (out_ps < (int64_t)side * side || (out_ps & 3)) return hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(out) & 15) || (out_bs & 3)) return hipErrorInvalidValue;   // 16-byte stores
    if (side == 56) return launch_d2q<56, 4>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
    if (side == 28) return launch_d2q<28, 8>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
    if (side == 14) return launch_d2q<14, 14>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
    return hipErrorInvalidValue;
}

}  // namespace mirx
