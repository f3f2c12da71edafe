// k_embed.hip -- DenseNet stem: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BatchNorm + ReLU
// + max-pool 3x3 / stride 2 / pad 1, fused so the 112x112x64 conv map never reaches HBM.
//
// Replaces features.conv0 / norm0 / relu0 / pool0 of the torchvision DenseNet-121 that
// model.py:53 instantiates.  NCHW fp32 in and out (the reference's layout and dtype).
//
// One workgroup = a 4 x 14 tile of pooled pixels x 32 output channels of one image:
//   LDS (70 KiB, two workgroups per CU): input patch 3 x 23 x 64 (zero padded), weights re-laid
//   [147][32], conv tile [32][9*29].  Conv phase: each thread owns 3 adjacent conv pixels x 16
//   channels (48 accumulators); per (input channel, kernel row) it reads 11 inputs and 7 x 16
//   weights from LDS for 336 FMAs.  Pool phase reads the conv tile back and writes only the pooled
//   map (NCHW, 56-byte rows).
#include "mirx_kernels.h"

namespace mirx {

namespace {

// Tile geometry: one workgroup = 4 x 14 pooled pixels x 32 output channels of one image.
constexpr int PTH = 4, PTW = 14;             // pooled tile
constexpr int CTH = 2 * PTH + 1;             // 9 conv rows
constexpr int CTW = 2 * PTW + 1;             // 29 conv cols
constexpr int ITH = 2 * (CTH - 1) + 7;       // 23 input rows
constexpr int IT_PITCH = 64;                 // 63 input cols, padded
constexpr int NTAP = 3 * 7 * 7;              // 147
constexpr int OCB = 32;                      // output channels per workgroup
constexpr int PXG = 3;                       // conv pixels per thread (along x)
constexpr int NG = (CTW + PXG - 1) / PXG;    // 10 pixel groups per conv row
constexpr int CONV_PITCH = CTH * CTW + 3;    // 264 floats per channel
constexpr int S_IN = 3 * ITH * IT_PITCH + 8; // + slack: the last pixel group reads one float past a row
constexpr int S_W = NTAP * OCB;
constexpr int S_CONV = OCB * CONV_PITCH;

// conv phase: thread = (16-channel half of the 32, conv row, group of 3 conv pixels): per (c, ky) it
// reads 11 inputs (float2 x 6) and 7 x 16 weights (float4 x 28) for 336 FMAs.
__global__ __launch_bounds__(256) void k_stem(const float *__restrict__ x, const float *__restrict__ w,
                                              const float *__restrict__ scale,
                                              const float *__restrict__ shift, int h, int wd,
                                              float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_in = sm;                    // [3][ITH][IT_PITCH]
    float *s_w = s_in + S_IN;            // [NTAP][OCB]
    float *s_conv = s_w + S_W;           // [OCB][CONV_PITCH]
    const int ph = h / 4, pw = wd / 4, ch = h / 2, cw = wd / 2;
    const int tiles_x = (pw + PTW - 1) / PTW;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
    const int64_t img = blockIdx.y;
    const int oc0 = blockIdx.z * OCB;
    const int py0 = tile_y * PTH, px0 = tile_x * PTW;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;     // first conv row/col of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;     // first input row/col of the patch
    const float *xi = x + img * 3 * (int64_t)h * wd;

    for (int i = threadIdx.x; i < 3 * ITH * IT_PITCH; i += 256) {
        const int c = i / (ITH * IT_PITCH), r = (i / IT_PITCH) % ITH, q = i % IT_PITCH;
        const int yy = iy0 + r, xx = ix0 + q;
        float v = 0.0f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < wd) v = xi[((int64_t)c * h + yy) * wd + xx];
        s_in[i] = v;
    }
    if (threadIdx.x < 8) s_in[3 * ITH * IT_PITCH + threadIdx.x] = 0.0f;
    for (int i = threadIdx.x; i < NTAP * OCB; i += 256) {
        const int oc = i / NTAP, tap = i % NTAP;         // w is [64][3][7][7]
        s_w[tap * OCB + oc] = w[(oc0 + oc) * NTAP + tap];
    }
    __syncthreads();

    if (threadIdx.x < 2 * CTH * NG) {
        const int chh = threadIdx.x / (CTH * NG);        // channels 16*chh .. 16*chh+15 of this block
        const int pu = threadIdx.x % (CTH * NG);
        const int r = pu / NG, g = pu % NG;
        float acc[PXG][16];
#pragma unroll
        for (int p = 0; p < PXG; ++p)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[p][j] = 0.0f;
        for (int c = 0; c < 3; ++c)
            for (int ky = 0; ky < 7; ++ky) {
                const float *irow = s_in + (c * ITH + 2 * r + ky) * IT_PITCH + 2 * PXG * g;
                float in[12];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float2 t = *reinterpret_cast<const float2 *>(irow + 2 * q);
                    in[2 * q] = t.x;
                    in[2 * q + 1] = t.y;
                }
                const float *wrow = s_w + ((c * 7 + ky) * 7) * OCB + 16 * chh;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    float wv[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 t = *reinterpret_cast<const float4 *>(wrow + kx * OCB + 4 * q);
                        wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
                    }
#pragma unroll
                    for (int p = 0; p < PXG; ++p)
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[p][j] = fmaf(in[2 * p + kx], wv[j], acc[p][j]);
                }
            }
        const int cy = cy0 + r;
#pragma unroll
        for (int p = 0; p < PXG; ++p) {
            const int q = PXG * g + p, cx = cx0 + q;
            if (q < CTW) {
                // outside the conv map = pool padding; 0 never wins over a relu output
                const bool inside = cy >= 0 && cy < ch && cx >= 0 && cx < cw;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int oc = 16 * chh + j;
                    const float v = fmaxf(fmaf(acc[p][j], scale[oc0 + oc], shift[oc0 + oc]), 0.0f);
                    s_conv[oc * CONV_PITCH + r * CTW + q] = inside ? v : 0.0f;
                }
            }
        }
    }
    __syncthreads();

    float *yi = y + (img * 64 + oc0) * (int64_t)ph * pw;
    for (int i = threadIdx.x; i < OCB * PTH * 16; i += 256) {
        const int oc = i / (PTH * 16), r = (i / 16) % PTH, q = i % 16;
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
    const size_t lds = (size_t)(S_IN + S_W + S_CONV) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stem),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stem, dim3((unsigned)tiles, (unsigned)n, 64 / OCB), dim3(256), lds, st, x, w, scale, shift,
                       h, wd, y);
    return hipGetLastError();
}

}  // namespace mirx
