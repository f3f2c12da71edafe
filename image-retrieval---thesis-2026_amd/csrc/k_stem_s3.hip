// k_stem_s3.hip -- the DenseNet stem of k_embed.hip (conv 7x7 / 2 + BN + ReLU + max-pool 3x3 / 2, one kernel, only
// the pooled map reaches HBM) with the implicit GEMM on the bf16 matrix pipe, both fp32 operands carried as three
// bf16 terms (six v_mfma_f32_32x32x16_bf16 per product block, fp32 accumulation; dropped cross terms
// <= 3 * 2^-24 |w x|, the rounding class of one fp32 product).
//
// Same tiling as k_stem: one workgroup (4 waves) = an 8 x 7 tile of pooled pixels x 32 output channels = a
// 17 x 15 tile of conv pixels (8 MFMA column blocks, two per wave); the 39 x 35 input patch sits in LDS
// de-interleaved into an even- and an odd-column plane.  The K axis is (c, ky) x 8 (kx padded 7 -> 8): one MFMA
// step = TWO (c, ky) rows (k-group h of the wave = row 2 s + h), 11 steps (22 rows, the last one zero weights):
//   * B: a lane reads the 8 input values of its pixel and row -- 4 from the even plane (kx = 0, 2, 4, 6), 4 from
//     the odd plane (kx = 1, 3, 5, 7) -- and splits them into three bf16 terms in registers;
//   * A: the weights arrive pre-split and pre-ordered ([2 channel blocks][11 steps][3 terms][32 oc][16 k] bf16,
//     mirx.model._stem_weights_split3) and are read straight from global memory (33 KiB, L2-resident): 3 x 16 B per
//     lane and step, no LDS.
// 132 MFMAs of 32 cycles per wave instead of 168 of 64: 2.5x less matrix time; LDS drops to the patch / conv tile
// (33 KiB).
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int PTH = 8, PTW = 7;              // pooled tile
constexpr int CTH = 2 * PTH + 1;             // 17 conv rows
constexpr int CTW = 2 * PTW + 1;             // 15 conv cols
constexpr int NPX = CTH * CTW;               // 255 conv pixels = 8 MFMA columns (one idle lane)
constexpr int ITH = 2 * (CTH - 1) + 7;       // 39 input rows
constexpr int ITW = 2 * (CTW - 1) + 7;       // 35 input cols
constexpr int PH = 24;                       // plane pitch: 18 columns used (+ 3 read ahead)
constexpr int PLANE = 3 * ITH * PH;          // floats per parity plane
constexpr int OCB = 32;                      // output channels per workgroup
constexpr int NSTEP = 11;                    // MFMA steps: 22 (c, ky) rows, row 21 = zero weights
constexpr int CONV_PITCH = 260;              // 255 pixels + pad, 260 = 4 (mod 32) banks per channel
constexpr int S_IN = 2 * PLANE;
constexpr int S_CONV = OCB * CONV_PITCH;
constexpr int S_ALL = S_IN > S_CONV ? S_IN : S_CONV;

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

__global__ __launch_bounds__(256, 3) void k_stem_s3(const float *__restrict__ x, const uint16_t *__restrict__ w3,
                                                    const float *__restrict__ scale, const float *__restrict__ shift,
                                                    int h, int wd, float *__restrict__ y, int64_t y_bs,
                                                    unsigned *__restrict__ out_range) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_in = sm;                    // [2 parity][3][ITH][PH]
    float *s_conv = sm;                  // [OCB][CONV_PITCH], after the K loop
    const int ph = h / 4, pw = wd / 4, ch = h / 2, cw = wd / 2;
    const int tiles_x = (pw + PTW - 1) / PTW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
    const int64_t img = blockIdx.y;
    const int oc0 = blockIdx.z * OCB;
    const int py0 = tile_y * PTH, px0 = tile_x * PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;     // first conv row/col of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;     // first input row/col of the patch
    const float *xi = x + img * 3 * (int64_t)h * wd;

    // ---- patch staging: all global loads of a thread are issued before its first LDS store ---------------------
    // element i -> (channel, row, column of a 36-wide row: columns 35 = kx 7 of the last pixel is written as zero);
    // coalesced along the row, written to the plane of its column parity
    constexpr int ROWW = 36;
    constexpr int N_IN = (3 * ITH * ROWW + 255) / 256;           // 17 per thread
    float vin[N_IN];
#pragma unroll
    for (int t = 0; t < N_IN; ++t) {
        const int i = threadIdx.x + 256 * t;
        const int c = i / (ITH * ROWW), r = (i / ROWW) % ITH, q = i % ROWW;
        const int yy = iy0 + r, xx = ix0 + q;
        vin[t] = 0.0f;
        if (i < 3 * ITH * ROWW && q < ITW && yy >= 0 && yy < h && xx >= 0 && xx < wd)
            vin[t] = xi[((int64_t)c * h + yy) * wd + xx];
    }
#pragma unroll
    for (int t = 0; t < N_IN; ++t) {
        const int i = threadIdx.x + 256 * t;
        const int c = i / (ITH * ROWW), r = (i / ROWW) % ITH, q = i % ROWW;
        if (i < 3 * ITH * ROWW) s_in[(q & 1) * PLANE + (c * ITH + r) * PH + (q >> 1)] = vin[t];
    }
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
    // A fragments: w3[oc block][step][term][oc = n][16 k], this lane's 16 bytes at k = 8 half
    const uint16_t *wp = w3 + ((int64_t)blockIdx.z * NSTEP * 3 * OCB + n) * 16 + 8 * half;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(wp + (s * 3 + 0) * OCB * 16);
        const bf16x8 am = *reinterpret_cast<const bf16x8 *>(wp + (s * 3 + 1) * OCB * 16);
        const bf16x8 al = *reinterpret_cast<const bf16x8 *>(wp + (s * 3 + 2) * OCB * 16);
        // this k-group's (c, ky) row: 2 s + half, row 21 (zero weights) re-reads row 20
        const int rho0 = 2 * s, rho1 = 2 * s + 1 < 21 ? 2 * s + 1 : 20;
        const int ro0 = ((rho0 / 7) * ITH + rho0 % 7) * PH, ro1 = ((rho1 / 7) * ITH + rho1 % 7) * PH;
        const int ro = half ? ro1 : ro0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float *pe = s_in + xbase[t] + ro;      // even plane: kx = 0, 2, 4, 6
            const float *po = pe + PLANE;                // odd plane:  kx = 1, 3, 5, (7: zero weight)
            u32x4 bh, bm, bl;
            unsigned th, tm, tl;
            split2(pe[0], pe[1], th, tm, tl); bh[0] = th; bm[0] = tm; bl[0] = tl;
            split2(pe[2], pe[3], th, tm, tl); bh[1] = th; bm[1] = tm; bl[1] = tl;
            split2(po[0], po[1], th, tm, tl); bh[2] = th; bm[2] = tm; bl[2] = tl;
            split2(po[2], po[3], th, tm, tl); bh[3] = th; bm[3] = tm; bl[3] = tl;
            const bf16x8 xh = __builtin_bit_cast(bf16x8, bh), xm = __builtin_bit_cast(bf16x8, bm),
                         xl = __builtin_bit_cast(bf16x8, bl);
            f32x16 c = acc[t];
            // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh, c, 0, 0, 0);
            acc[t] = c;
        }
    }

    // ---- BN + ReLU, conv tile to LDS (register r: channel 8 (r >> 2) + (r & 3) + 4 half, pixel n) ---
    __syncthreads();                                     // every wave is done with the patch
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = (2 * wave + t) * 32 + n;
        if (p < NPX) {
            const int r0 = p / CTW, q = p % CTW;
            const int cy = cy0 + r0, cx = cx0 + q;
            // outside the conv map = pool padding; 0 never wins over a relu output
            const bool inside = cy >= 0 && cy < ch && cx >= 0 && cx < cw;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = 8 * (r >> 2) + (r & 3) + 4 * half;
                const float v = fmaxf(fmaf(acc[t][r], scale[oc0 + oc], shift[oc0 + oc]), 0.0f);
                s_conv[oc * CONV_PITCH + p] = inside ? v : 0.0f;
            }
        }
    }
    __syncthreads();

    // ---- max-pool 3x3 / 2: only the pooled map goes to HBM ----------------------------------------
    float *yi = y + img * y_bs + oc0 * (int64_t)ph * pw;   // y_bs: batch stride (the dense block's buffer)
    float vmax = 0.f;
    for (int i = threadIdx.x; i < OCB * PTH * 8; i += 256) {
        const int oc = i / (PTH * 8), r = (i / 8) % PTH, q = i % 8;
        const int py = py0 + r, px = px0 + q;
        if (q < PTW && py < ph && px < pw) {
            const float *cbase = s_conv + oc * CONV_PITCH + (2 * r) * CTW + 2 * q;
            float m = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, cbase[dy * CTW + dx]);
            yi[((int64_t)oc * ph + py) * pw + px] = m;
            vmax = range_max(vmax, m);
        }
    }
    (void)vmax; (void)out_range;        // range publishing lives on the two-fp16-term path only (per image, mirx_common.h)
}

}  // namespace

hipError_t launch_stem_s3(const float *x, const uint16_t *w3, const float *scale, const float *shift, int64_t n, int h,
                          int wd, float *y, int64_t y_bs, float *out_range, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535) return hipErrorInvalidValue;
    const int ph = h / 4, pw = wd / 4;
    const int tiles = ((ph + PTH - 1) / PTH) * ((pw + PTW - 1) / PTW);
    const size_t lds = (size_t)S_ALL * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stem_s3),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stem_s3, dim3((unsigned)tiles, (unsigned)n, 64 / OCB), dim3(256), lds, st, x, w3, scale, shift, h,
                       wd, y, y_bs, reinterpret_cast<unsigned *>(out_range));
    return hipGetLastError();
}

}  // namespace mirx
