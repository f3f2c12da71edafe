// k_conv3x3_d2h.hip -- the DenseNet dense layer's 3x3 convolution (128 -> 32 channels, stride 1, pad 1, no bias) as a
// direct implicit GEMM with every fp32 operand carried as TWO fp16 terms (x = xh + xl to 22 bits) and THREE
// v_mfma_f32_32x32x16_f16 per product block (wl xh + wh xl + wh xh; the dropped wl xl is 2^-22 of the product).
//
//   D[oc, pixel] += W[oc, (tap, c)] * X[(tap, c), pixel]     K = 9 taps x 128 channels, one MFMA step = one tap x 16 c
//
// k_conv3x3_d2p reads the bottleneck ALREADY SPLIT into its fp16 terms by the 1x1 conv that produced it (scaled per image
// by 2^t, 2^-t in in_inv[image]), reads weights that were scaled per OUTPUT channel by the caller (largest |w| of the channel
// in [2^13, 2^14): mirx.model._conv3x3_weights_split2h), multiplies the accumulator by oscale[oc] * 2^-t and folds the largest
// |output| of image b into out_range[b] (ranges are per image: mirx_common.h).
//
// One workgroup (4 waves) = a strip of R rows x W columns = 224 (196 for the 14 x 14 map) output pixels of one image
// = 7 column blocks of 32 pixels (waves 0..2 take two, wave 3 one) x all 32 output channels.
//   * X: 16-channel stages of the padded strip ((R + 2) x (W + 2) pixels), pixel-major -- [term][padded pixel][16 channels]
//     fp16, the two 16-byte halves of a pixel at slot h ^ ((pixel >> 3) & 1) -- so the B fragment of a lane (its pixel
//     shifted by the tap, 8 channels) is ONE ds_read_b128 per term.  Double-buffered, staged by LDS DMA alone.
//   * W: pre-split and pre-ordered by the caller ([8 stages][9 taps][2 terms][32 oc][16 c] fp16).  The 18 KiB of a
//     stage go global -> LDS by DMA ONCE per workgroup (18 one-KiB pieces = one (tap, term) plane each, double-buffered,
//     issued right after the stage barrier) and every wave reads its A fragments from there.
// (The round-2 variant with an fp32 bottleneck split inside this kernel, k_conv3x3_d2h, and the 16x16x32 re-tiling
// k_conv3x3_d2q were A/B arms that lost to this kernel; they live in the history of this file)
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int CIN = 128, COUT = 32;
constexpr int KC = 16;                        // channels per stage = one MFMA K
constexpr int NST = CIN / KC;                 // 8 stages

// ---------------------------------------------------------------------------------------------------------------------
// k_conv3x3_d2p: yt = [image][group of 16 channels][term][pixel][16] fp16 (k_conv1x1_h2<.., YTERMS>), scaled by 2^t, 2^-t in
// in_inv[image].  Staging is then pure LDS DMA -- no register prefetch, no split, no LDS stores: piece j of a term plane =
// padded pixels 32 j .. 32 j + 31 (lane l -> pixel 32 j + l / 2, 16-byte slot l & 1, which must hold channel chunk
// (l & 1) ^ ((pixel >> 3) & 1)); a lane's source offset is its pixel's position inside the image, or an offset beyond the
// buffer's num_records for the padding ring (out-of-range buffer loads return zero: the halo needs no branch and no
// pre-zeroed LDS).  Weights by DMA as in k_conv3x3_d2h.  The tap loop is k_conv3x3_d2h's.
// IPW > 1 (7 x 7 maps): a workgroup takes IPW whole images (R = W), each with its own zero ring, so that its 7 column blocks
// are as full as on the larger maps (4 x 49 = 196 pixels)
// NW = waves per workgroup.  NW = 8 (-DMIRX_D2P_WAVES=8, the A/B arm): one workgroup takes twice the pixels (a strip of twice
// the rows, or twice the images) against ONE staged copy of the weights -- a stage of weights (18 KiB) is almost as many bytes
// as a 4-row strip's activations (22 KiB), so the 4-wave version pulls 80 KiB into the CU per 8 rows of a 56 x 56 map where
// this one pulls 56.  Parity-identical and SLOWER: 0.95 vs 0.75 ms per 56 x 56 layer, 44.3 vs 45.1 k img/s on the forward --
// with a single workgroup per CU all eight waves meet at every stage barrier and nothing else feeds the matrix pipe, which
// is what bounds this kernel (unlike the 1x1 conv, where sharing the weight stage between two pixel tiles gained 2.5 %).
template <int W, int R, int IPW = 1, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void k_conv3x3_d2p(const uint16_t *__restrict__ yt, const uint16_t *__restrict__ w3,
                                                        const float *__restrict__ oscale, float *__restrict__ out,
                                                        int64_t out_bs, const float *__restrict__ in_inv,
                                                        unsigned *__restrict__ out_range, int64_t out_ps, int64_t n_img,
                                                        const float *__restrict__ pool_sc, const float *__restrict__ pool_sh,
                                                        float *__restrict__ pool_out, int64_t pool_bs) {
    constexpr int PW = W + 2, PR = R + 2;     // padded strip
    static_assert(IPW == 1 || R == W, "several images per workgroup: whole images only");
    constexpr int NPIX = IPW * PR * PW;       // padded pixels of a stage
    constexpr int NP = (NPIX + 31) / 32;      // 1-KiB DMA pieces per term plane
    constexpr int PLANE = NP * 32 * 32;       // bytes of one term of one stage (32 B per pixel, rounded up to whole pieces)
    constexpr int STAGE = 2 * PLANE;
    constexpr int WSTAGE = 9 * 2 * COUT * KC * 2;   // bytes of one stage of weights (18 KiB): 18 pieces of 1 KiB
    constexpr int W_LDS0 = 2 * STAGE;               // weight buffers behind the two activation buffers
    constexpr int NOUT = IPW * R * W;         // output pixels of a full strip
    constexpr int NBLK = (NOUT + 31) / 32;    // 7
    static_assert(NBLK <= 2 * NW, "two column blocks per wave");
    constexpr int PPW = (2 * NP + NW - 1) / NW;   // activation pieces per wave and stage (both terms)
    constexpr int WPW = (18 + NW - 1) / NW;       // weight pieces per wave and stage
    extern __shared__ __attribute__((aligned(16))) char sm[];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = lane >> 5, n = lane & 31;
    // Workgroups are dealt to the 8 XCDs round-robin in linear order, and each XCD has its own L2: with (strip, image) =
    // blockIdx the neighbouring strips of an image -- which share two halo rows each -- would sit on different XCDs and
    // every halo row would cross the fabric twice (1.5x the map at R = 4).  Re-deal so that XCD x walks images x, x + 8, ..
    // strip by strip: the halo of a strip is then in the L2 its neighbour just filled (+3 %).  (Any bijection is correct.)
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
    img *= IPW;                                               // first image of this workgroup
    const int n_here = (int)(n_img - img < IPW ? n_img - img : IPW);   // images that exist (the batch's tail)
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(yt + img * (int64_t)(IMG_BYTES / 2)), 0,
                                                                           (unsigned)n_here * IMG_BYTES, 0x00020000);
    // this wave's activation pieces: q = wave + NW i over the 2 NP pieces of a stage (term = q / NP, piece = q % NP); the
    // source offset of this lane inside a term plane, or "out of range" for the zero ring
    unsigned a_src[PPW];
    int a_dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave + NW * i;
        const int term = q / NP, piece = q % NP;
        const int pix = piece * 32 + (lane >> 1);
        const int ji = pix / (PR * PW), prem = pix % (PR * PW);    // image of the workgroup, padded pixel inside it
        const int pr = prem / PW, pc = prem % PW;
        const int iy = oy0 - 1 + pr, ix = pc - 1;
        const bool inside = q < 2 * NP && pix < NPIX && iy >= 0 && iy < W && ix >= 0 && ix < W;
        const int chunk = (lane & 1) ^ ((pix >> 3) & 1);
        // images beyond the batch lie beyond num_records: zero like the padding ring
        a_src[i] = inside ? (unsigned)(ji * IMG_BYTES + (term * W * W + iy * W + ix) * 32 + chunk * 16) : 0xfffffff0u;
        a_dst[i] = q < 2 * NP ? term * PLANE + piece * 1024 : -1;
    }
    auto dma_a = [&](int st, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            if (a_dst[i] >= 0) {                               // wave-uniform
                const unsigned v = a_src[i] == 0xfffffff0u ? a_src[i] : a_src[i] + (unsigned)st * (2u * W * W * 32u);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(yrsrc, LDS_PTR(sm + buf * STAGE + a_dst[i]), 16, v, 0, 0, 0);
            }
    };

    // ---- this wave's column blocks and this lane's pixels --------------------------------------------------------
    int pbase[2];                                             // padded index of the pixel's tap (0, 0) corner
    bool live_blk[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int blk = wave + NW * t;
        live_blk[t] = blk < NBLK;                              // wave-uniform
        int p = blk * 32 + n;
        if (p >= NOUT) p = NOUT - 1;                           // idle lanes shadow a valid pixel (never stored)
        const int pj = p / (R * W), pq = p % (R * W);          // image of the workgroup (0 unless IPW > 1), pixel inside it
        pbase[t] = pj * (PR * PW) + (pq / W) * PW + (pq % W);
    }
    const int a_off = n * 32 + ((half ^ ((n >> 3) & 1)) << 4);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w3, 0, NST * WSTAGE, 0x00020000);
    const int w_voff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    auto dma_w = [&](int st, int buf) {
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int piece = wave + NW * i;              // 18 pieces dealt round-robin to the waves
            if (piece < 18)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(sm + W_LDS0 + buf * WSTAGE + piece * 1024), 16, w_voff,
                                                         st * WSTAGE + piece * 1024, 0, 0);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#define MIRX_D2P_READA(DST, TAP)                                                                   \
    {                                                                                              \
        DST[0] = *reinterpret_cast<const f16x8 *>(wb + (2 * (TAP)) * 1024 + a_off);                \
        DST[1] = *reinterpret_cast<const f16x8 *>(wb + (2 * (TAP) + 1) * 1024 + a_off);            \
    }
#define MIRX_D2P_READB(DST, TAP, T)                                                                \
    {                                                                                              \
        const int pix_ = pbase[T] + ((TAP) / 3) * PW + (TAP) % 3;                                  \
        const char *pb_ = sb + pix_ * 32 + ((half ^ ((pix_ >> 3) & 1)) << 4);                      \
        DST[0] = *reinterpret_cast<const f16x8 *>(pb_);                                            \
        DST[1] = *reinterpret_cast<const f16x8 *>(pb_ + PLANE);                                    \
    }
#define MIRX_D2P_MFMA(T, A, B)                                                                     \
    {                                                                                              \
        f32x16 c_ = acc[T];                                                                        \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], B[0], c_, 0, 0, 0);                      \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[1], c_, 0, 0, 0);                      \
        c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], c_, 0, 0, 0);                      \
        acc[T] = c_;                                                                               \
    }
    const bool two = live_blk[1];                          // wave-uniform: waves 0..2 own two column blocks
    dma_w(0, 0);
    dma_a(0, 0);
    for (int st = 0; st < NST; ++st) {
        const int cur = st & 1;
        // stage st landed (this wave's DMA: vmcnt(0); every wave's: the barrier); buffers cur ^ 1 free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 1 < NST) {                                // wave-uniform
            dma_w(st + 1, cur ^ 1);
            dma_a(st + 1, cur ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const char *sb = sm + cur * STAGE;
        const char *wb = sm + W_LDS0 + cur * WSTAGE;
        f16x8 a0[2], a1[2], b0[2], b1[2];
        MIRX_D2P_READA(a0, 0)
        MIRX_D2P_READB(b0, 0, 0)
        if (two) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                MIRX_D2P_READB(b1, tap, 1)
                if (tap + 1 < 9) {
                    if (tap & 1) { MIRX_D2P_READA(a0, tap + 1) } else { MIRX_D2P_READA(a1, tap + 1) }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tap & 1) { MIRX_D2P_MFMA(0, a1, b0) } else { MIRX_D2P_MFMA(0, a0, b0) }
                if (tap + 1 < 9) MIRX_D2P_READB(b0, tap + 1, 0)
                __builtin_amdgcn_sched_barrier(0);
                if (tap & 1) { MIRX_D2P_MFMA(1, a1, b1) } else { MIRX_D2P_MFMA(1, a0, b1) }
            }
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                if (tap + 1 < 9) {
                    MIRX_D2P_READB(b1, tap + 1, 0)
                    if (tap & 1) { MIRX_D2P_READA(a0, tap + 1) } else { MIRX_D2P_READA(a1, tap + 1) }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (tap & 1) { MIRX_D2P_MFMA(0, a1, b0) } else { MIRX_D2P_MFMA(0, a0, b0) }
#pragma unroll
                for (int q = 0; q < 2; ++q) b0[q] = b1[q];
            }
        }
    }
#undef MIRX_D2P_READA
#undef MIRX_D2P_READB
#undef MIRX_D2P_MFMA

    // ---- outputs straight from the accumulators: register r = channel 8 (r >> 2) + (r & 3) + 4 half, lane = pixel ----
    float *oi = out + img * out_bs + (int64_t)oy0 * W;        // IPW > 1: image j of the workgroup at + j * out_bs
    float osc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) osc[r] = oscale[8 * (r >> 2) + (r & 3) + 4 * half];
    // POOLED TWIN (pool_out != nullptr; dense blocks 1-3, whose transition starts with norm + relu + avgpool 2x2): the strip
    // holds whole row pairs, so this workgroup also writes avgpool2(relu(bn_t(v))) of its 32 new channels into the transition's
    // pooled input -- the same fp32 operations in the same order as mirx_bn_relu_avgpool2 on the stored values, bit-identical
    // -- and the pooling pass over the whole block (every channel read once more from HBM) shrinks to the block's first
    // channels.  The activated values cross lanes through the LDS the K loop has released.
    constexpr int ACT_PITCH = NOUT + 1;                       // odd: the lanes of a pooled read fall on different banks
    float *s_act = reinterpret_cast<float *>(sm);             // [32 channels][ACT_PITCH]
    const bool pooled_twin = IPW == 1 && pool_out != nullptr; // wave-uniform
    float psc[16], psh[16];
    if (pooled_twin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            psc[r] = pool_sc[8 * (r >> 2) + (r & 3) + 4 * half];
            psh[r] = pool_sh[8 * (r >> 2) + (r & 3) + 4 * half];
        }
        __syncthreads();                                      // every wave has read its last fragments: the stage buffers are free
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = (wave + NW * t) * 32 + n;
        const int pj = p / (R * W), pq = p % (R * W);
        const bool live = live_blk[t] && p < NOUT && pj < n_here && oy0 + pq / W < W;
        const int pimg = (int)img + (pj < n_here ? pj : 0);            // a lane that carries nothing: a valid image, vmax = 0
        float vmax = 0.f;
        if (live) {
            const float x_inv = in_inv[pimg];                          // 2^-t of this pixel's image (oscale * 2^-t is exact)
            float *op = oi + (int64_t)pj * out_bs + pq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = 8 * (r >> 2) + (r & 3) + 4 * half;
                const float v = acc[t][r] * (osc[r] * x_inv);
                vmax = range_max(vmax, v);
                op[(int64_t)oc * out_ps] = v;
                if (pooled_twin) s_act[oc * ACT_PITCH + p] = fmaxf(fmaf(v, psc[r], psh[r]), 0.0f);
            }
        }
        if (out_range && live_blk[t]) {                                // wave-uniform
            if (IPW == 1) range_publish(out_range, (int)img, vmax, lane);
            else range_publish_lanes(out_range, pimg, vmax, lane);
        }
    }
    if constexpr (IPW == 1) {
        if (pooled_twin) {
            __syncthreads();
            constexpr int W2 = W / 2, R2 = R / 2, NP2 = R2 * W2;      // pooled pixels of the strip, per channel
            static_assert(R % 2 == 0 && W % 2 == 0, "whole row pairs");
            const int rows_here = W - oy0 < R ? W - oy0 : R;          // the map's last strip may be short (an even number of rows)
            float *po = pool_out + img * pool_bs + (int64_t)(oy0 / 2) * W2;
            for (int i = threadIdx.x; i < COUT * NP2; i += 64 * NW) {
                const int oc = i / NP2, q = i % NP2, py = q / W2, px = q % W2;
                if (2 * py < rows_here) {
                    const float *a = s_act + oc * ACT_PITCH + (2 * py) * W + 2 * px;
                    po[(int64_t)oc * (W2 * W2) + q] = (a[0] + a[1] + a[W] + a[W + 1]) * 0.25f;
                }
            }
        }
    }
}

template <int W, int R, int IPW = 1, int NW = 4>
hipError_t launch_d2p(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, float *out, int64_t out_bs,
                      const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st, const float *pool_sc = nullptr,
                      const float *pool_sh = nullptr, float *pool_out = nullptr, int64_t pool_bs = 0) {
    constexpr int NP = (IPW * (R + 2) * (W + 2) + 31) / 32;
    const size_t lds = (size_t)2 * 2 * NP * 1024 + 2 * 9 * 2 * 32 * 16 * 2;
    static_assert((size_t)2 * 2 * NP * 1024 + 2 * 9 * 2 * 32 * 16 * 2 <= 160 * 1024, "LDS of one CU");
    static unsigned long long attr_devs = 0;          // per instantiation
    if (first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_d2p<W, R, IPW, NW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_conv3x3_d2p<W, R, IPW, NW>), dim3((W + R - 1) / R, (unsigned)((n + IPW - 1) / IPW)), dim3(64 * NW), lds,
                       st, yt, w2, oscale, out, out_bs, in_inv, reinterpret_cast<unsigned *>(out_range), out_ps, n, pool_sc, pool_sh,
                       pool_out, pool_bs);
    return hipGetLastError();
}

}  // namespace

// does a launch of n images take the one-wave-per-block kernel (k_conv3x3_d2s.hip)?  (the pooled twin exists in the strip
// kernel only: the caller asks before it plans a block)
bool conv3x3_takes_small(int64_t n, int side) {
    const int64_t wgs = side == 56 ? 14 * n : side == 28 ? 4 * n : side == 14 ? n : (n + 3) / 4;
    return wgs < conv3x3_small_max_wg();
}

hipError_t launch_conv3x3_d2p(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, int side, float *out,
                              int64_t out_bs, const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st,
                              const float *pool_sc, const float *pool_sh, float *pool_out, int64_t pool_bs) {
    if (n <= 0) return hipSuccess;
    if (n > 65535 || !in_inv || !oscale) return hipErrorInvalidValue;
    if (!out_ps) out_ps = (int64_t)side * side;
    if (out_ps < (int64_t)side * side) return hipErrorInvalidValue;
    if (pool_out && (!pool_sc || !pool_sh || side == 7 || pool_bs < (int64_t)COUT * (side / 2) * (side / 2)))
        return hipErrorInvalidValue;
    // small launches (the reference's own batch sizes): one wave per block of 32 output pixels, no LDS (k_conv3x3_d2s.hip)
    if (conv3x3_takes_small(n, side)) {
        if (pool_out) return hipErrorInvalidValue;            // the pooled twin lives in the strip kernel (conv3x3_takes_small)
        return launch_conv3x3_d2s(yt, w2, oscale, n, side, out, out_bs, in_inv, out_range, out_ps, st);
    }
#ifndef MIRX_D2P_WAVES
#define MIRX_D2P_WAVES 4          // 8: one 8-wave workgroup per CU on twice the pixels (the A/B arm, measured slower)
#endif
#if MIRX_D2P_WAVES == 8
    if (pool_out) return hipErrorInvalidValue;
    if (side == 56) return launch_d2p<56, 8, 1, 8>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
    if (side == 28) return launch_d2p<28, 14, 1, 8>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
    if (side == 14) return launch_d2p<14, 14, 2, 8>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
    if (side == 7) return launch_d2p<7, 7, 8, 8>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
#else
    if (side == 56) return launch_d2p<56, 4>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st, pool_sc, pool_sh, pool_out, pool_bs);
    if (side == 28) return launch_d2p<28, 8>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st, pool_sc, pool_sh, pool_out, pool_bs);
    if (side == 14) return launch_d2p<14, 14>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st, pool_sc, pool_sh, pool_out, pool_bs);
    if (side == 7) return launch_d2p<7, 7, 4>(yt, w2, oscale, n, out, out_bs, in_inv, out_range, out_ps, st);
#endif
    return hipErrorInvalidValue;
}

}  // namespace mirx
