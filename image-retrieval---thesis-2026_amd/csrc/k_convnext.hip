// k_convnext.hip -- ConvNeXt(V2) block front end: depthwise 7x7 convolution (pad 3, bias) read from
// NCHW and written NHWC, i.e. conv_dw + the permute(0, 2, 3, 1) that timm's ConvNeXtBlock does
// before its LayerNorm / MLP (reference model.py:96-100 -> timm convnextv2_base blocks).
//
// HBM-bound: 49 MACs per output against 8 bytes of traffic.  One workgroup = 16 x 16 output
// pixels x 32 channels of one image: the 22 x 22 input patches of the 32 channels sit in LDS
// (channel pitch 485 floats: odd, so the 32 lanes of a half-wave -- one per channel -- never
// collide on a bank); a thread owns one channel and two output rows, keeps that channel's 49
// weights in registers, and for every kernel row reads 22 inputs for 7 x 16 FMAs.  The NHWC store
// puts the 32 channels of a pixel in one 128-byte segment.
#include "mirx_kernels.h"

namespace mirx {

namespace {

constexpr int DT = 16;                 // output tile edge
constexpr int DP = DT + 6;             // input patch edge (22)
constexpr int DCH = 32;                // channels per workgroup
constexpr int DPITCH = DP * DP + 1;    // 485

__global__ __launch_bounds__(256) void k_dwconv7(const float *__restrict__ x, const float *__restrict__ w,
                                                 const float *__restrict__ bias, int c, int h, int wd,
                                                 float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [DCH][DPITCH]
    const int tiles_x = (wd + DT - 1) / DT;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int c0 = blockIdx.y * DCH;
    const int64_t img = blockIdx.z;
    const int y0 = ty * DT, x0 = tx * DT;
    const float *xi = x + (img * c + c0) * (int64_t)h * wd;
    for (int i = threadIdx.x; i < DCH * DP * DP; i += 256) {
        const int ch = i / (DP * DP), r = (i / DP) % DP, q = i % DP;
        const int yy = y0 + r - 3, xx = x0 + q - 3;
        float v = 0.0f;
        if (c0 + ch < c && yy >= 0 && yy < h && xx >= 0 && xx < wd) v = xi[((int64_t)ch * h + yy) * wd + xx];
        sm[ch * DPITCH + r * DP + q] = v;
    }
    __syncthreads();
    const int ch = threadIdx.x & 31, rg = threadIdx.x >> 5;        // channel, pair of output rows
    if (c0 + ch >= c) return;
    float wk[49];
#pragma unroll
    for (int i = 0; i < 49; ++i) wk[i] = w[(int64_t)(c0 + ch) * 49 + i];
    const float b = bias ? bias[c0 + ch] : 0.0f;
    const float *pch = sm + ch * DPITCH;
    float *yo = y + img * (int64_t)h * wd * c + c0 + ch;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = rg * 2 + rr;
        if (y0 + r >= h) break;
        float acc[DT];
#pragma unroll
        for (int q = 0; q < DT; ++q) acc[q] = b;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            float in[DP];
#pragma unroll
            for (int q = 0; q < DP; ++q) in[q] = pch[(r + ky) * DP + q];
#pragma unroll
            for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                for (int q = 0; q < DT; ++q) acc[q] = fmaf(in[q + kx], wk[ky * 7 + kx], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < DT; ++q)
            if (x0 + q < wd) yo[((int64_t)(y0 + r) * wd + x0 + q) * c] = acc[q];
    }
}

}  // namespace

hipError_t launch_dwconv7(const float *x, const float *w, const float *bias, int64_t n, int c, int h, int wd,
                          float *y, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int tiles = ((h + DT - 1) / DT) * ((wd + DT - 1) / DT);
    const size_t lds = (size_t)DCH * DPITCH * sizeof(float);
    if (n > 65535 || (c + DCH - 1) / DCH > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dwconv7, dim3((unsigned)tiles, (unsigned)((c + DCH - 1) / DCH), (unsigned)n), dim3(256), lds,
                       st, x, w, bias, c, h, wd, y);
    return hipGetLastError();
}

}  // namespace mirx
