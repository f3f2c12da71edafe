// k_stem_h2.hip -- the DenseNet stem of k_stem_s3.hip (conv 7x7 / 2 + BN + ReLU + max-pool 3x3 / 2, one kernel, only the
// pooled map reaches HBM) with both fp32 operands of the implicit GEMM carried as TWO fp16 terms (three
// v_mfma_f32_32x32x16_f16 per product block: wl xh + wh xl + wh xh, the dropped wl xl is 2^-22 of the product) instead
// of three bf16 terms and six MFMAs: 66 MFMAs of 32 cycles per wave instead of 132.
//
// fp16 needs the range of the input: `in_range[b]` = the largest |pixel| of image b (mirx_range_absmax, one pass over
// the images; ranges are per image, mirx_common.h); the patch is multiplied by 2^s (bound * 2^s in [2^14, 2^15)) while it is
// staged, the weights arrive scaled per output channel (mirx.model._stem_weights_split2h) and the accumulator is
// multiplied by oscale[oc] / 2^s before norm0.  Tiling, patch layout and pooling are k_stem_s3's; it writes image b at
// y + b * y_bs (the channel prefix of dense block 1's buffer) and folds the largest pooled value into `out_range[b]`.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

constexpr int PTH = 8, PTW = 7;              // pooled tile
constexpr int CTH = 2 * PTH + 1;             // 17 conv rows
constexpr int CTW = 2 * PTW + 1;             // 15 conv cols
constexpr int NPX = CTH * CTW;               // 255 conv pixels = 8 MFMA columns (one idle lane)
constexpr int ITH = 2 * (CTH - 1) + 7;       // 39 input rows
constexpr int ITW = 2 * (CTW - 1) + 7;       // 35 input cols
constexpr int PH = 24;                       // plane pitch: 18 columns used (+ 3 read ahead)
constexpr int PLANE = 3 * ITH * PH;          // floats per parity plane
constexpr int OCB = 32;                      // output channels per MFMA row block
constexpr int NOB = 2;                       // row blocks per workgroup: all 64 channels (a split B fragment feeds both)
constexpr int NSTEP = 11;                    // MFMA steps: 22 (c, ky) rows, row 21 = zero weights
constexpr int CONV_PITCH = 260;              // 255 pixels + pad, 260 = 4 (mod 32) banks per channel
constexpr int S_IN = 2 * PLANE;
constexpr int S_CONV = NOB * OCB * CONV_PITCH;
constexpr int S_ALL = S_IN > S_CONV ? S_IN : S_CONV;
constexpr int W_BYTES = NOB * NSTEP * 2 * OCB * 16 * 2;      // all weights of the stem: 44 pieces of 1 KiB
constexpr int W_OFF = S_IN * 4;                              // behind the patch (the conv tile reuses both after the K loop)
constexpr int PAR_OFF = (W_OFF + W_BYTES > S_ALL * 4 ? W_OFF + W_BYTES : S_ALL * 4);   // epilogue constants behind everything
constexpr int LDS_BYTES = PAR_OFF + 4 * NOB * OCB * 4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

__device__ inline void split2(float a, float b, unsigned &h, unsigned &l) {
    const f32x2 v = {a, b};
    const f16x2 vh = __builtin_convertvector(v, f16x2);
    const f32x2 r1 = v - __builtin_convertvector(vh, f32x2);
    const f16x2 vl = __builtin_convertvector(r1, f16x2);
    h = __builtin_bit_cast(unsigned, vh);
    l = __builtin_bit_cast(unsigned, vl);
}

// TIN = float: normalised fp32 images.  TIN = uint8_t: raw 8-bit images [B, 3, H, W]; the reference's ToTensor + Normalize
// (test.py:1309-1332: x = u / 255, then (x - mean[c]) / std[c], fp32, correctly rounded) is applied while the patch is staged,
// through a 3 x 256 table built once per workgroup with exactly those operations -- the staged values, hence the embeddings,
// are bit-identical to feeding the normalised fp32 tensor, at a quarter of the input bytes (PCIe and HBM).
template <typename TIN>
__global__ __launch_bounds__(256, 2) void k_stem_h2(const TIN *__restrict__ x, const uint16_t *__restrict__ w3,
                                                    const float *__restrict__ oscale, const float *__restrict__ scale,
                                                    const float *__restrict__ shift, int h, int wd, float *__restrict__ y,
                                                    int64_t y_bs, const float *__restrict__ in_range,
                                                    unsigned *__restrict__ out_range, const float *__restrict__ mean,
                                                    const float *__restrict__ stdv) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr bool U8 = sizeof(TIN) == 1;
    __shared__ float s_lut[U8 ? 768 : 1];
    if (U8) {
        for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = ((float)(i & 255) / 255.0f - mean[i >> 8]) / stdv[i >> 8];
        __syncthreads();
    }
    float *s_in = sm;                    // [2 parity][3][ITH][PH]
    float *s_conv = sm;                  // [OCB][CONV_PITCH], after the K loop
    const int ph = h / 4, pw = wd / 4, ch = h / 2, cw = wd / 2;
    const int tiles_x = (pw + PTW - 1) / PTW;
    // XCD-aware order (as in k_conv3x3_d2p): workgroups are dealt to the 8 XCDs round-robin in linear order, so with (tile,
    // image) = blockIdx the neighbouring tiles of an image -- whose input patches overlap by a quarter in each direction and
    // whose 28-byte output row segments share 128-byte lines -- would sit behind eight different L2s.  Re-dealt, XCD x walks
    // images x, x + 8, .. tile by tile: the overlap is an L2 hit and the partial output lines merge in one L2.
    int tile = blockIdx.x;
    int64_t img = blockIdx.y;
#ifndef MIRX_STEM_PLAIN_ORDER
    {
        const unsigned ntile = gridDim.x, lin = blockIdx.x + ntile * blockIdx.y, full = gridDim.y & ~7u;
        if (lin < ntile * full) {
            const unsigned j = lin >> 3;
            tile = (int)(j % ntile);
            img = (int64_t)(j / ntile) * 8 + (lin & 7);
        }
    }
#endif
    const int tile_y = tile / tiles_x, tile_x = tile % tiles_x;
    const int oc0 = 0;
    const int py0 = tile_y * PTH, px0 = tile_x * PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;     // first conv row/col of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;     // first input row/col of the patch
    const TIN *xi = x + img * 3 * (int64_t)h * wd;
    float x_scale, x_inv;
    range_scales(in_range[img], x_scale, x_inv);
    // per-channel epilogue constants (accumulator scale, norm0 scale and shift) once per workgroup: the epilogue reads them
    // from LDS instead of three cached global loads per accumulator register
    float *s_par = sm + PAR_OFF / 4;     // [64][4]: oscale * 2^-s, scale, shift
    if (threadIdx.x < NOB * OCB) {
        s_par[4 * threadIdx.x] = oscale[threadIdx.x] * x_inv;
        s_par[4 * threadIdx.x + 1] = scale[threadIdx.x];
        s_par[4 * threadIdx.x + 2] = shift[threadIdx.x];
    }

    // ---- the 44 KiB of weights go global -> LDS by DMA once per workgroup (11 one-KiB pieces per wave, issued before the
    // patch loads; the region behind the patch is free until the conv tile is written after the K loop).  Every wave needs
    // every weight: read per wave from L2 (the first version) that is 176 KiB per workgroup through a vector L1 the set does
    // not fit in -- ten times the bytes of the input patch.  Piece = (channel block, step, term) = [32 oc][16 k] fp16; lane l
    // -> LDS (row l / 2, slot l & 1) <- source chunk (l & 1) ^ ((row >> 3) & 1): conflict-free ds_read_b128 fragments.
    char *s_wt = reinterpret_cast<char *>(sm) + W_OFF;
    {
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = threadIdx.x & 63;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w3, 0, W_BYTES, 0x00020000);
        const int voff = (ln >> 1) * 32 + (((ln & 1) ^ ((ln >> 4) & 1)) << 4);
#pragma unroll
        for (int i = 0; i < W_BYTES / 1024 / 4; ++i) {
            const int piece = wv + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(s_wt + piece * 1024), 16, voff, piece * 1024, 0, 0);
        }
    }

    // ---- patch staging: all global loads of a thread are issued before its first LDS store ---------------------
    // (channel, row, column of a 36-wide row: column 35 = kx 7 of the last pixel is written as zero); coalesced along the row,
    // written to the plane of its column parity.  Thread -> (row slot tid / 36, column q = tid % 36) once; pass t covers patch rows 7 t .. 7 t + 6 of the 3 x 39 (channel,
    // row) pairs: the column, its bounds check and its parity plane are per-thread constants and the address advances by a
    // constant per pass (the first version recomputed (c, r, q) from a flat index with two divisions per element).
    constexpr int ROWW = 36, RPP = 7;                            // 7 x 36 = 252 of the 256 threads carry an element
    constexpr int N_IN = (3 * ITH + RPP - 1) / RPP;              // 17 passes
    const int s_row = threadIdx.x / ROWW, s_q = threadIdx.x % ROWW;
    const int xx = ix0 + s_q;
    const bool col_ok = s_row < RPP && s_q < ITW && xx >= 0 && xx < wd;
    const bool lane_live = s_row < RPP;
    float vin[N_IN];
#pragma unroll
    for (int t = 0; t < N_IN; ++t) {
        const int row = RPP * t + s_row;                         // (channel, patch row) pair
        const int c = (row >= ITH) + (row >= 2 * ITH), r = row - c * ITH;
        const int yy = iy0 + r;
        vin[t] = 0.0f;
        if (col_ok && row < 3 * ITH && yy >= 0 && yy < h) {
            if constexpr (U8) vin[t] = s_lut[c * 256 + xi[((int64_t)c * h + yy) * wd + xx]];
            else vin[t] = xi[((int64_t)c * h + yy) * wd + xx];
        }
    }
    unsigned *s_w = reinterpret_cast<unsigned *>(s_in);
    // The patch is split into its two fp16 terms HERE, once per input value, and stored as one 32-bit word (hi | lo << 16)
    // in the slot the fp32 value used to take: a B fragment is then 8 word reads and 8 v_perm_b32 -- a value is part of ~11
    // fragments, and splitting it in every one of them (~30 VALU instructions per fragment) bounded the K loop, not the MFMAs.
    const int s_dst = (s_q & 1) * PLANE + s_row * PH + (s_q >> 1);
#pragma unroll
    for (int t = 0; t < N_IN; ++t) {
        const float v = vin[t] * x_scale;
        const _Float16 vh = (_Float16)v;
        const _Float16 vl = (_Float16)(v - (float)vh);
        const unsigned word = (unsigned)__builtin_bit_cast(unsigned short, vh) | ((unsigned)__builtin_bit_cast(unsigned short, vl) << 16);
        if (lane_live && RPP * t + s_row < 3 * ITH) s_w[s_dst + RPP * t * PH] = word;   // row (c ITH + r) = 7 t + slot
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's weight pieces have landed (the barrier: everyone's)
    __syncthreads();

    // ---- implicit GEMM: wave -> pixel blocks 2 wave, 2 wave + 1 (32 pixels each) -----------------------------------
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, n = lane & 31;
    int xbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int p = (2 * wave + t) * 32 + n;
        if (p >= NPX) p = NPX - 1;                       // the one idle lane reads a valid pixel
        const int r = p / CTW, q = p % CTW;
        xbase[t] = 2 * r * PH + q;                       // + row offset of (c, ky) + j (+ PLANE for odd kx)
    }
    // A fragments: LDS image of w3[oc block][step][term][oc = n][16 k], this lane's 16 bytes at k = 8 half.  Both channel blocks run
    // in this workgroup: the B fragment of a (pixel block, step) is split ONCE (the ~40 VALU instructions of the split,
    // not the three MFMAs of one block, bounded the one-block version) and feeds 2 x 3 MFMAs.
    const char *wp = s_wt + n * 32 + ((half ^ ((n >> 3) & 1)) << 4);
    constexpr int WBLK = NSTEP * 2 * 1024;               // bytes of one channel block's weights
    f32x16 acc[2][NOB];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int b = 0; b < NOB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.0f;
#ifndef MIRX_STEM_EXP
#define MIRX_STEM_EXP 0            // diagnostic builds (wrong results, timing only): 1 no K loop, 2 no conv-tile epilogue, 4 no pooling
#endif
#pragma unroll
    for (int s = 0; s < ((MIRX_STEM_EXP & 1) ? 0 : NSTEP); ++s) {
        f16x8 ah[NOB], al[NOB];
#pragma unroll
        for (int b = 0; b < NOB; ++b) {
            ah[b] = *reinterpret_cast<const f16x8 *>(wp + b * WBLK + (s * 2 + 0) * 1024);
            al[b] = *reinterpret_cast<const f16x8 *>(wp + b * WBLK + (s * 2 + 1) * 1024);
        }
        // this k-group's (c, ky) row: 2 s + half, row 21 (zero weights) re-reads row 20
        const int rho0 = 2 * s, rho1 = 2 * s + 1 < 21 ? 2 * s + 1 : 20;
        const int ro0 = ((rho0 / 7) * ITH + rho0 % 7) * PH, ro1 = ((rho1 / 7) * ITH + rho1 % 7) * PH;
        const int ro = half ? ro1 : ro0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned *pe = s_w + xbase[t] + ro;    // even plane: kx = 0, 2, 4, 6
            const unsigned *po = pe + PLANE;             // odd plane:  kx = 1, 3, 5, (7: zero weight)
            u32x4 bh, bl;
            // word = hi | lo << 16: the hi halves of two words -> one fp16 pair of xh, the lo halves -> one pair of xl
            bh[0] = __builtin_amdgcn_perm(pe[1], pe[0], 0x05040100u); bl[0] = __builtin_amdgcn_perm(pe[1], pe[0], 0x07060302u);
            bh[1] = __builtin_amdgcn_perm(pe[3], pe[2], 0x05040100u); bl[1] = __builtin_amdgcn_perm(pe[3], pe[2], 0x07060302u);
            bh[2] = __builtin_amdgcn_perm(po[1], po[0], 0x05040100u); bl[2] = __builtin_amdgcn_perm(po[1], po[0], 0x07060302u);
            bh[3] = __builtin_amdgcn_perm(po[3], po[2], 0x05040100u); bl[3] = __builtin_amdgcn_perm(po[3], po[2], 0x07060302u);
            const f16x8 xh = __builtin_bit_cast(f16x8, bh), xl = __builtin_bit_cast(f16x8, bl);
#pragma unroll
            for (int b = 0; b < NOB; ++b) {
                f32x16 c = acc[t][b];
                // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], xh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], xl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], xh, c, 0, 0, 0);
                acc[t][b] = c;
            }
            __builtin_amdgcn_sched_barrier(0);           // keeps the reads of the next fragment from being hoisted over 11 steps
        }
    }

    // ---- BN + ReLU, conv tile to LDS (register r: channel 8 (r >> 2) + (r & 3) + 4 half, pixel n) ---
    __syncthreads();                                     // every wave is done with the patch
    if (!(MIRX_STEM_EXP & 2)) {
        int pp[2];
        bool live[2], inside[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pp[t] = (2 * wave + t) * 32 + n;
            live[t] = pp[t] < NPX;
            const int r0 = pp[t] / CTW, q = pp[t] % CTW;
            const int cy = cy0 + r0, cx = cx0 + q;
            // outside the conv map = pool padding; 0 never wins over a relu output
            inside[t] = cy >= 0 && cy < ch && cx >= 0 && cx < cw;
        }
#pragma unroll
        for (int b = 0; b < NOB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = b * OCB + 8 * (r >> 2) + (r & 3) + 4 * half;
                const f32x4 pr = *reinterpret_cast<const f32x4 *>(s_par + 4 * oc);      // once for both pixel blocks
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float v = fmaxf(fmaf(acc[t][b][r] * pr[0], pr[1], pr[2]), 0.0f);
                    if (live[t]) s_conv[oc * CONV_PITCH + pp[t]] = inside[t] ? v : 0.0f;
                }
            }
    }
    __syncthreads();

    // ---- max-pool 3x3 / 2: only the pooled map goes to HBM ----------------------------------------
    float *yi = y + img * y_bs + oc0 * (int64_t)ph * pw;   // y_bs: batch stride (the dense block's buffer)
    float vmax = 0.f;
    for (int i = threadIdx.x; i < ((MIRX_STEM_EXP & 4) ? 0 : NOB * OCB * PTH * 8); i += 256) {
        const int oc = i / (PTH * 8), r = (i / 8) % PTH, q = i % 8;
        const int py = py0 + r, px = px0 + q;
        if (q < PTW && py < ph && px < pw) {
            const float *cbase = s_conv + oc * CONV_PITCH + (2 * r) * CTW + 2 * q;
            float m = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, cbase[dy * CTW + dx]);
            if (x_inv != x_inv) m = x_inv;               // a non-finite input range: NaN out (relu and max above drop NaNs)
            yi[((int64_t)oc * ph + py) * pw + px] = m;
            vmax = range_max(vmax, m);
        }
    }
    if (out_range) range_publish(out_range, (int)img, vmax, threadIdx.x & 63);
}

template <typename TIN>
static hipError_t launch_stem_h2_t(const TIN *x, const uint16_t *w2, const float *oscale, const float *scale, const float *shift,
                                   int64_t n, int h, int wd, float *y, int64_t y_bs, const float *in_range, float *out_range,
                                   const float *mean, const float *stdv, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535 || !in_range || !oscale) return hipErrorInvalidValue;
    const int ph = h / 4, pw = wd / 4;
    const int tiles = ((ph + PTH - 1) / PTH) * ((pw + PTW - 1) / PTW);
    const size_t lds = LDS_BYTES;
    static unsigned long long attr_devs = 0;           // per instantiation
    if (first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stem_h2<TIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_stem_h2<TIN>, dim3((unsigned)tiles, (unsigned)n, 1), dim3(256), lds, st, x, w2, oscale, scale, shift,
                       h, wd, y, y_bs, in_range, reinterpret_cast<unsigned *>(out_range), mean, stdv);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_stem_h2(const float *x, const uint16_t *w2, const float *oscale, const float *scale, const float *shift,
                          int64_t n, int h, int wd, float *y, int64_t y_bs, const float *in_range, float *out_range,
                          hipStream_t st) {
    return launch_stem_h2_t<float>(x, w2, oscale, scale, shift, n, h, wd, y, y_bs, in_range, out_range, nullptr, nullptr, st);
}

hipError_t launch_stem_h2_u8(const uint8_t *x, const float *mean, const float *stdv, const uint16_t *w2, const float *oscale,
                             const float *scale, const float *shift, int64_t n, int h, int wd, float *y, int64_t y_bs,
                             const float *in_range, float *out_range, hipStream_t st) {
    if (!mean || !stdv) return hipErrorInvalidValue;
    return launch_stem_h2_t<uint8_t>(x, w2, oscale, scale, shift, n, h, wd, y, y_bs, in_range, out_range, mean, stdv, st);
}

// ---- largest |value| of every image (`per` contiguous fp32 each) into its range: grid (chunks, images) -----------------
namespace {
__global__ __launch_bounds__(256) void k_range_absmax(const float *__restrict__ x, int64_t per, unsigned *__restrict__ row) {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    const float *xi = x + (int64_t)blockIdx.y * per;
    float m = 0.f;
    if ((reinterpret_cast<uintptr_t>(xi) & 15) == 0) {
        const int64_t nv = per >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
            const f32x4 v = reinterpret_cast<const f32x4 *>(xi)[i];
            m = range_max(range_max(range_max(range_max(m, v[0]), v[1]), v[2]), v[3]);
        }
        if (blockIdx.x == 0 && threadIdx.x < (per & 3)) m = range_max(m, xi[(nv << 2) + threadIdx.x]);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) m = range_max(m, xi[i]);
    }
    range_publish(row, (int)blockIdx.y, m, threadIdx.x & 63);
}
}  // namespace

namespace {
// the same for raw 8-bit images [n, 3, hw]: the largest |(u / 255 - mean[c]) / std[c]| of every image (the normalisation is
// monotone per channel, but the table keeps this one pass over the bytes trivially consistent with the stem's)
__global__ __launch_bounds__(256) void k_range_absmax_u8(const uint8_t *__restrict__ x, int64_t hw, const float *__restrict__ mean,
                                                         const float *__restrict__ stdv, unsigned *__restrict__ row) {
    __shared__ float s_lut[768];
    for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = ((float)(i & 255) / 255.0f - mean[i >> 8]) / stdv[i >> 8];
    __syncthreads();
    const uint8_t *xi = x + (int64_t)blockIdx.y * 3 * hw;
    float m = 0.f;
    for (int c = 0; c < 3; ++c)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
            m = range_max(m, s_lut[c * 256 + xi[c * hw + i]]);
    range_publish(row, (int)blockIdx.y, m, threadIdx.x & 63);
}
}  // namespace

hipError_t launch_range_absmax_u8(const uint8_t *x, int64_t hw, int64_t n, const float *mean, const float *stdv, float *row,
                                  hipStream_t st) {
    if (n <= 0 || hw <= 0) return hipSuccess;
    if (n > 65535 || !mean || !stdv) return hipErrorInvalidValue;
    const int64_t want = (hw + 255) / 256;
    const unsigned chunks = (unsigned)(want > 16 ? 16 : want);
    hipLaunchKernelGGL(k_range_absmax_u8, dim3(chunks, (unsigned)n), dim3(256), 0, st, x, hw, mean, stdv, reinterpret_cast<unsigned *>(row));
    return hipGetLastError();
}

hipError_t launch_range_absmax(const float *x, int64_t per, int64_t n, float *row, hipStream_t st) {
    if (n <= 0 || per <= 0) return hipSuccess;
    if (n > 65535) return hipErrorInvalidValue;
    // ~16 16-byte loads per thread (64 chunks of one load each took 2.7x the time of the old whole-batch pass)
    const int64_t want = (per / 4 + 16 * 256 - 1) / (16 * 256);
    const unsigned chunks = (unsigned)(want < 1 ? 1 : (want > 16 ? 16 : want));
    hipLaunchKernelGGL(k_range_absmax, dim3(chunks, (unsigned)n), dim3(256), 0, st, x, per, reinterpret_cast<unsigned *>(row));
    return hipGetLastError();
}

}  // namespace mirx
