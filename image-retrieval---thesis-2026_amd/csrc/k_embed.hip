// k_embed.hip -- DenseNet stem: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BatchNorm + ReLU
// + max-pool 3x3 / stride 2 / pad 1, fused so the 112x112x64 conv map never reaches HBM.
//
// Replaces features.conv0 / norm0 / relu0 / pool0 of the torchvision DenseNet-121 that
// model.py:53 instantiates.  NCHW fp32 in and out (the reference's layout and dtype).
//
// One workgroup (4 waves) = an 8 x 7 tile of pooled pixels x 32 output channels of one image, i.e. a
// 17 x 15 tile of conv pixels (255 = 8 MFMA columns of 32) computed as an implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains):
//     D[oc, pixel] += W[oc, k] * X[k, pixel],   k = (c, ky, kx) with kx padded 7 -> 8 (zero weight),
// so that the two K values of one MFMA step are always (kx even, kx odd) of the same (c, ky).
// LDS (44 KiB, three workgroups per CU; the conv tile reuses the patch + weight space after the K loop):
//   * the 39 x 35 input patch, zero padded, de-interleaved into an even-column and an odd-column plane:
//     the lanes of wave half h read plane h at [c][2r + ky][q + j] -- consecutive pixels hit consecutive
//     banks (the raw stride-2 walk of a stride-2 convolution would be a 2-way conflict);
//   * weights re-laid [k][32];
//   * the 32 x 255 conv tile after BN + ReLU, from which the pool phase writes only the pooled map.
// Every LDS address inside the K loop is one per-lane base plus an immediate.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int PTH = 8, PTW = 7;              // pooled tile
constexpr int CTH = 2 * PTH + 1;             // 17 conv rows
constexpr int CTW = 2 * PTW + 1;             // 15 conv cols
constexpr int NPX = CTH * CTW;               // 255 conv pixels = 8 MFMA columns (one idle lane)
constexpr int ITH = 2 * (CTH - 1) + 7;       // 39 input rows
constexpr int ITW = 2 * (CTW - 1) + 7;       // 35 input cols
constexpr int PH = 24;                       // plane pitch: 18 columns used; 2 * PH = 16 (mod 32) banks
constexpr int PLANE = 3 * ITH * PH;          // floats per parity plane
constexpr int OCB = 32;                      // output channels per workgroup
constexpr int KX8 = 8;                       // kx padded to 8
constexpr int NK = 3 * 7 * KX8;              // 168
constexpr int CONV_PITCH = 260;              // 255 pixels + pad, 260 = 4 (mod 32) banks per channel
constexpr int S_IN = 2 * PLANE;
constexpr int S_W = NK * OCB;
constexpr int S_CONV = OCB * CONV_PITCH;

__global__ __launch_bounds__(256) void k_stem(const float *__restrict__ x, const float *__restrict__ w,
                                              const float *__restrict__ scale,
                                              const float *__restrict__ shift, int h, int wd,
                                              float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_in = sm;                    // [2 parity][3][ITH][PH]
    float *s_w = s_in + S_IN;            // [NK][OCB]
    float *s_conv = sm;                  // [OCB][CONV_PITCH], after the K loop
    static_assert(S_CONV <= S_IN + S_W, "the conv tile must fit in the staging space it reuses");
    const int ph = h / 4, pw = wd / 4, ch = h / 2, cw = wd / 2;
    const int tiles_x = (pw + PTW - 1) / PTW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
    const int64_t img = blockIdx.y;
    const int oc0 = blockIdx.z * OCB;
    const int py0 = tile_y * PTH, px0 = tile_x * PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;     // first conv row/col of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;     // first input row/col of the patch
    const float *xi = x + img * 3 * (int64_t)h * wd;

    // Staging.  All global loads of a thread are issued before the first LDS store, so the workgroup
    // pays one memory latency, not one per element.
    // input patch: element i -> (channel, row, column of a 36-wide row); coalesced along the row,
    // written to the plane of its column parity
    constexpr int N_IN = (3 * ITH * 36 + 255) / 256;            // 17 per thread
    constexpr int N_W = (OCB * 147 + 255) / 256;                // 19 per thread
    float vin[N_IN], vw[N_W];
#pragma unroll
    for (int t = 0; t < N_IN; ++t) {
        const int i = threadIdx.x + 256 * t;
        const int c = i / (ITH * 36), r = (i / 36) % ITH, q = i % 36;
        const int yy = iy0 + r, xx = ix0 + q;
        vin[t] = 0.0f;
        if (i < 3 * ITH * 36 && q < ITW && yy >= 0 && yy < h && xx >= 0 && xx < wd)
            vin[t] = xi[((int64_t)c * h + yy) * wd + xx];
    }
    // weights of channels oc0 .. oc0 + 31 are one contiguous run of [32][3][7][7]
#pragma unroll
    for (int t = 0; t < N_W; ++t) {
        const int i = threadIdx.x + 256 * t;
        vw[t] = i < OCB * 147 ? w[oc0 * 147 + i] : 0.0f;
    }
#pragma unroll
    for (int t = 0; t < N_IN; ++t) {
        const int i = threadIdx.x + 256 * t;
        const int c = i / (ITH * 36), r = (i / 36) % ITH, q = i % 36;
        if (i < 3 * ITH * 36) s_in[(q & 1) * PLANE + (c * ITH + r) * PH + (q >> 1)] = vin[t];
    }
    // -> s_w[(c*7 + ky)*8 + kx][oc]; the kx = 7 rows are zero
#pragma unroll
    for (int t = 0; t < N_W; ++t) {
        const int i = threadIdx.x + 256 * t;
        const int oc = i / 147, rem = i % 147;
        if (i < OCB * 147) s_w[((rem / 7) * KX8 + rem % 7) * OCB + oc] = vw[t];
    }
    for (int i = threadIdx.x; i < 21 * OCB; i += 256) s_w[((i / OCB) * KX8 + 7) * OCB + (i % OCB)] = 0.0f;
    __syncthreads();

    // ---- implicit GEMM: wave -> pixel columns 2*wave, 2*wave + 1 (32 pixels each) ------------------
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, n = lane & 31;
    int xbase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int p = (2 * wave + t) * 32 + n;
        if (p >= NPX) p = NPX - 1;                       // the one idle lane reads a valid pixel
        const int r = p / CTW, q = p % CTW;
        xbase[t] = half * PLANE + 2 * r * PH + q;
    }
    const int wbase = half * OCB + n;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = s_w[wbase + ((c * 7 + ky) * KX8 + 2 * j) * OCB];
                const int xo = (c * ITH + ky) * PH + j;
                const float b0 = s_in[xbase[0] + xo];
                const float b1 = s_in[xbase[1] + xo];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            }

    // ---- BN + ReLU, conv tile to LDS (register r: channel 8 (r >> 2) + (r & 3) + 4 half, pixel n) ---
    __syncthreads();                                     // every wave is done with the patch and the weights
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
    float *yi = y + (img * 64 + oc0) * (int64_t)ph * pw;
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
        }
    }
}

}  // namespace

hipError_t launch_stem(const float *x, const float *w, const float *scale, const float *shift, int64_t n,
                       int h, int wd, float *y, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int ph = h / 4, pw = wd / 4;
    const int tiles = ((ph + PTH - 1) / PTH) * ((pw + PTW - 1) / PTW);
    const size_t lds = (size_t)(S_IN + S_W) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stem),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stem, dim3((unsigned)tiles, (unsigned)n, 64 / OCB), dim3(256), lds, st, x, w, scale, shift,
                       h, wd, y);
    return hipGetLastError();
}

}  // namespace mirx
