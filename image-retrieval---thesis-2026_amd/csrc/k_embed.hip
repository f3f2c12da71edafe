// k_embed.hip -- DenseNet stem: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BatchNorm + ReLU
// + max-pool 3x3 / stride 2 / pad 1, fused so the 112x112x64 conv map never reaches HBM.
//
// Replaces features.conv0 / norm0 / relu0 / pool0 of the torchvision DenseNet-121 that
// model.py:53 instantiates.  NCHW fp32 in and out (the reference's layout and dtype).
//
// One workgroup = one 8x8 tile of pooled pixels x all 64 channels of one image:
//   LDS: input patch 3 x 39 x 39 (zero padded), weights re-laid [147][64], conv tile [64][17*17].
//   conv phase: thread = (8-channel group, pixel lane); each tap costs one broadcast input
//   read and two 16-B weight reads for 8 FMAs.
#include "mirx_kernels.h"

namespace mirx {

namespace {

constexpr int PT = 8;                   // pooled tile edge
constexpr int CT = 2 * PT + 1;          // conv tile edge (17)
constexpr int IT = 2 * (CT - 1) + 7;    // input patch edge (39)
constexpr int IT_PAD = IT + 1;          // row pitch 40
constexpr int NTAP = 3 * 7 * 7;         // 147
constexpr int CONV_PITCH = CT * CT + 3; // 292: de-phase channel rows

__global__ __launch_bounds__(256) void k_stem(const float *__restrict__ x, const float *__restrict__ w,
                                              const float *__restrict__ scale,
                                              const float *__restrict__ shift, int h, int wd,
                                              float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_in = sm;                                   // [3][IT][IT_PAD]
    float *s_w = s_in + 3 * IT * IT_PAD;                // [NTAP][64]   (offset 4680 floats, 16-B aligned)
    float *s_conv = s_w + NTAP * 64;                    // [64][CONV_PITCH]
    const int ph = h / 4, pw = wd / 4, ch = h / 2, cw = wd / 2;
    const int tiles_x = (pw + PT - 1) / PT;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x % tiles_x;
    const int64_t img = blockIdx.y;
    const int py0 = tile_y * PT, px0 = tile_x * PT;
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;     // first conv row/col of the tile
    const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;     // first input row/col of the patch
    const float *xi = x + img * 3 * (int64_t)h * wd;

    for (int i = threadIdx.x; i < 3 * IT * IT; i += 256) {
        const int c = i / (IT * IT), r = (i / IT) % IT, q = i % IT;
        const int yy = iy0 + r, xx = ix0 + q;
        float v = 0.0f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < wd) v = xi[((int64_t)c * h + yy) * wd + xx];
        s_in[(c * IT + r) * IT_PAD + q] = v;
    }
    for (int i = threadIdx.x; i < NTAP * 64; i += 256) {
        const int oc = i / NTAP, tap = i % NTAP;         // w is [64][3][7][7]
        s_w[tap * 64 + oc] = w[i];
    }
    __syncthreads();

    const int cg = threadIdx.x >> 5;                     // channels 8*cg .. 8*cg+7
    const int pl = threadIdx.x & 31;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[8 * cg + j]; sh[j] = shift[8 * cg + j]; }
    for (int p = pl; p < CT * CT; p += 32) {
        const int r = p / CT, q = p % CT;
        const int cy = cy0 + r, cx = cx0 + q;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        const bool inside = cy >= 0 && cy < ch && cx >= 0 && cx < cw;
        if (inside) {
            for (int c = 0; c < 3; ++c)
                for (int ky = 0; ky < 7; ++ky) {
                    const float *irow = s_in + (c * IT + 2 * r + ky) * IT_PAD + 2 * q;
                    const float *wrow = s_w + ((c * 7 + ky) * 7) * 64 + 8 * cg;
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) {
                        const float v = irow[kx];
                        const float4 w0 = *reinterpret_cast<const float4 *>(wrow + kx * 64);
                        const float4 w1 = *reinterpret_cast<const float4 *>(wrow + kx * 64 + 4);
                        acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]);
                        acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                        acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]);
                        acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)   // outside the conv map = pool padding; 0 never wins over relu output
            s_conv[(8 * cg + j) * CONV_PITCH + p] = inside ? fmaxf(fmaf(acc[j], sc[j], sh[j]), 0.0f) : 0.0f;
    }
    __syncthreads();

    float *yi = y + img * 64 * (int64_t)ph * pw;
    for (int i = threadIdx.x; i < 64 * PT * PT; i += 256) {
        const int oc = i / (PT * PT), r = (i / PT) % PT, q = i % PT;
        const int py = py0 + r, px = px0 + q;
        if (py < ph && px < pw) {
            const float *cbase = s_conv + oc * CONV_PITCH + (2 * r) * CT + 2 * q;
            float m = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, cbase[dy * CT + dx]);
            yi[((int64_t)oc * ph + py) * pw + px] = m;
        }
    }
}

}  // namespace

hipError_t launch_stem(const float *x, const float *w, const float *scale, const float *shift, int64_t n,
                       int h, int wd, float *y, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int ph = h / 4, pw = wd / 4;
    const int tiles = ((ph + PT - 1) / PT) * ((pw + PT - 1) / PT);
    const size_t lds = (size_t)(3 * IT * IT_PAD + NTAP * 64 + 64 * CONV_PITCH) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_stem),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stem, dim3((unsigned)tiles, (unsigned)n), dim3(256), lds, st, x, w, scale, shift, h,
                       wd, y);
    return hipGetLastError();
}

}  // namespace mirx
