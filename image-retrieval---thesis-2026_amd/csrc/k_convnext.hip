// k_convnext.hip -- ConvNeXt(V2) block front end: depthwise 7x7 convolution (pad 3, bias) read from
// NCHW and written NHWC, i.e. conv_dw + the permute(0, 2, 3, 1) that timm's ConvNeXtBlock does
// before its LayerNorm / MLP (reference model.py:96-100 -> timm convnextv2_base blocks).
//
// 49 MACs per output against 8 bytes of traffic.  One workgroup = 16 x 16 (or 12 x 12) output
// pixels x 32 channels of one image: the 22 x 22 (18 x 18) input patches of the 32 channels sit in LDS
// (channel pitch 485 / 325 floats: odd, so the 32 lanes of a half-wave -- one per channel -- never
// collide on a bank); a thread owns one channel and two (three) output rows, keeps that channel's 49
// weights in registers, and for every kernel row reads 22 inputs for 7 x 16 FMAs.  The NHWC store
// puts the 32 channels of a pixel in one 128-byte segment.
#include <cstdio>

#include "mirx_kernels.h"

namespace mirx {

namespace {

constexpr int DCH = 32;                // channels per workgroup

// DT = output tile edge, RPT = output rows per thread: 16 / 2 (256 threads, 22 x 22 patches, 62 KiB) for maps whose
// side is a multiple of 16 (96, 48 in ConvNeXtV2-base @384), 12 / 3 (128 threads, 18 x 18 patches, 42 KiB) for the
// 24- and 12-wide maps of stages 3 and 4, where a 16-wide tile would leave 44 % of its outputs outside the map.
template <int DT, int RPT>
__global__ __launch_bounds__(32 * (DT / RPT), 2) void k_dwconv7(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, int c, int h, int wd,
                                                              float *__restrict__ y) {
    constexpr int DP = DT + 6;             // input patch edge
    constexpr int DPITCH = DP * DP + 1;    // odd: the 32 lanes of a half-wave (one per channel) never share a bank
    constexpr int NTH = 32 * (DT / RPT);   // threads
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [DCH][DPITCH]
    const int tiles_x = (wd + DT - 1) / DT;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int c0 = blockIdx.y * DCH;
    const int64_t img = blockIdx.z;
    const int y0 = ty * DT, x0 = tx * DT;
    const float *xi = x + (img * c + c0) * (int64_t)h * wd;
    // Staging.  A wavefront step covers RW whole patch rows (lane -> row lane / DP, column lane % DP: one division per
    // thread, outside the loop), the workgroup RSTEP rows; a thread's (channel, patch row) advances by RSTEP rows per step
    // with at most one wrap.  (The first version numbered the patch elements linearly and took channel / row / column of
    // every element by division: ~35 VALU per loaded value -- in-kernel stamps showed the staging phase at 40 k cycles
    // against 20 k for the 49-tap arithmetic, VALU-bound on index math.)  Loads go in batches of NB before the matching LDS
    // stores (addresses clamped, values zeroed when stored) so that a batch costs one memory round trip.
    constexpr int RW = 64 / DP, RSTEP = RW * (NTH / 64);
    constexpr int NSTEP = (DCH * DP + RSTEP - 1) / RSTEP;        // 88 steps (16 / 2), 96 (12 / 3)
#ifndef MIRX_DW_NB
#define MIRX_DW_NB 16
#endif
    constexpr int NB = MIRX_DW_NB;
#ifdef MIRX_DW_STAMPS
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int rsub = lane / DP, q = lane - rsub * DP;
        const bool lane_on = rsub < RW;
        const int row0 = wave * RW + (lane_on ? rsub : 0);
        int chp = row0 / DP, r = row0 - chp * DP;                // this thread's patch row: channel chp, row r
        const int xx = x0 + q - 3;
        const bool x_in = xx >= 0 && xx < wd;
        const int xc = xx < 0 ? 0 : (xx >= wd ? wd - 1 : xx);
        const int ch_max = (c - c0 < DCH ? c - c0 : DCH) - 1;    // last valid channel of this workgroup
#pragma unroll 1
        for (int t0 = 0; t0 < NSTEP; t0 += NB) {
            float vin[NB];
            int chs = chp, rs = r;
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int chc = chp > ch_max ? ch_max : chp;
                int yy = y0 + r - 3;
                yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
                vin[t] = xi[(chc * h + yy) * wd + xc];
                r += RSTEP;
                if (r >= DP) {
                    r -= DP;
                    ++chp;
                }
            }
#pragma unroll
            for (int t = 0; t < NB; ++t) {
                const int yy = y0 + rs - 3;
                const bool in = chs <= ch_max && yy >= 0 && yy < h && x_in;
                if (lane_on && chs < DCH && t0 + t < NSTEP) sm[chs * DPITCH + rs * DP + q] = in ? vin[t] : 0.0f;
                rs += RSTEP;
                if (rs >= DP) {
                    rs -= DP;
                    ++chs;
                }
            }
        }
    }
    __syncthreads();
#ifdef MIRX_DW_STAMPS
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
    const int ch = threadIdx.x & 31, rg = threadIdx.x >> 5;        // channel, group of RPT output rows
    if (c0 + ch >= c) return;
    float wk[49];
#pragma unroll
    for (int i = 0; i < 49; ++i) wk[i] = w[(int64_t)(c0 + ch) * 49 + i];
#ifdef MIRX_DW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2_ = __builtin_amdgcn_s_memtime();
#endif
    const float b = bias ? bias[c0 + ch] : 0.0f;
    const float *pch = sm + ch * DPITCH;
    float *yo = y + img * (int64_t)h * wd * c + c0 + ch;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int r = rg * RPT + rr;
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
#ifdef MIRX_DW_STAMPS
    const unsigned long long t3_ = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 3 && (blockIdx.z & 15) == 5)
        printf("dwconv<%d,%d> img %d: staging %llu, weights %llu, compute+store %llu cycles\n", DT, RPT, (int)blockIdx.z, t1_ - t0_,
               t2_ - t1_, t3_ - t2_);
#endif
}

// ---- depthwise 7x7 on a channels-last map (NHWC in, NHWC out): no LDS, no staging phase ---------------------------------
// With channels last, a wavefront's lanes are 64 consecutive channels of one pixel: every load and store is one 256-byte
// run, and a thread (one channel) can walk a strip of DR = 3 output rows along x with a 7-column window of its DR + 6 input
// rows in registers: a step loads ONE new column (DR + 6 values), multiplies the whole window (DR x 49 FMAs) and stores DR
// outputs.  The window rotates by renaming (the x loop is unrolled by 7), an input value is fetched (DR + 6) / DR times
// (from L2 after the first), weights arrive as [49][c] (transposed once on the host side of the model) in 49 registers.
// grid: (ceil(h / DR) strips, ceil(c / 64), n); block: 64.
#ifndef MIRX_DW_DR
#define MIRX_DW_DR 3
#endif
constexpr int DR = MIRX_DW_DR;      // output rows per thread (ConvNeXtV2 @384, B = 64, same box: 2 / 3 / 4 / 6 / 8 rows -> 1 960 / 2 024 / 2 010 / 1 897 / 1 848 img/s)

__global__ __launch_bounds__(64) void k_dwconv7_nhwc(const float *__restrict__ x, const float *__restrict__ wt,
                                                     const float *__restrict__ bias, int c, int h, int wd,
                                                     float *__restrict__ y) {
    const int ch = blockIdx.y * 64 + threadIdx.x;
    if (ch >= c) return;
    const int y0 = blockIdx.x * DR;
    const int64_t img = blockIdx.z;
    const float *xi = x + img * (int64_t)h * wd * c + ch;
    float *yo = y + img * (int64_t)h * wd * c + ch;
    float wk[49];
#pragma unroll
    for (int i = 0; i < 49; ++i) wk[i] = wt[(int64_t)i * c + ch];
    const float b = bias ? bias[ch] : 0.0f;
    // row offsets (in elements) and validity of the DR + 6 input rows of this strip
    int roff[DR + 6];
    bool rok[DR + 6];
#pragma unroll
    for (int r = 0; r < DR + 6; ++r) {
        const int yy = y0 + r - 3;
        rok[r] = yy >= 0 && yy < h;
        roff[r] = (yy < 0 ? 0 : (yy >= h ? h - 1 : yy)) * wd * c;
    }
    // win[k][r]: input column (x_out - 3 + k) of row r, for the output column being computed
    float win[7][DR + 6];
    auto load_col = [&](int xx, float (&col)[DR + 6]) {
        const bool xok = xx >= 0 && xx < wd;
        const int xc = (xx < 0 ? 0 : (xx >= wd ? wd - 1 : xx)) * c;
#pragma unroll
        for (int r = 0; r < DR + 6; ++r) {
            const float v = xi[roff[r] + xc];
            col[r] = (xok && rok[r]) ? v : 0.0f;
        }
    };
#pragma unroll
    for (int k = 0; k < 6; ++k) load_col(k - 3, win[k]);          // columns -3 .. 2 of output column 0
    // one output column: the newest window column (input x_out + 3) arrives in slot S, the window's columns in x order are
    // slots S + 1, .., S + 7 (mod 7)
#define MIRX_DW_STEP(S)                                                                          \
    if (xo < wd) {                                                                               \
        load_col(xo + 3, win[((S) + 6) % 7]);                                                    \
        float acc[DR];                                                                           \
        _Pragma("unroll") for (int r = 0; r < DR; ++r) acc[r] = b;                               \
        _Pragma("unroll") for (int ky = 0; ky < 7; ++ky)                                         \
            _Pragma("unroll") for (int kx = 0; kx < 7; ++kx)                                     \
                _Pragma("unroll") for (int r = 0; r < DR; ++r)                                   \
                    acc[r] = fmaf(win[((S) + kx) % 7][r + ky], wk[ky * 7 + kx], acc[r]);         \
        _Pragma("unroll") for (int r = 0; r < DR; ++r)                                           \
            if (y0 + r < h) yo[((int64_t)(y0 + r) * wd + xo) * c] = acc[r];                      \
        ++xo;                                                                                    \
    }
    // at step S the window slots in x order start at slot S: slot (S + k) % 7 holds input column xo - 3 + k; the new column
    // xo + 3 goes to slot (S + 6) % 7, which held column xo - 4 (no longer needed)
    int xo = 0;
#pragma unroll 1
    while (xo < wd) {
        MIRX_DW_STEP(0) MIRX_DW_STEP(1) MIRX_DW_STEP(2) MIRX_DW_STEP(3) MIRX_DW_STEP(4) MIRX_DW_STEP(5) MIRX_DW_STEP(6)
    }
#undef MIRX_DW_STEP
}

// Global response normalisation (timm GlobalResponseNorm, channels last), split in two HBM passes instead of
// PyTorch's eight: (1) gx[b][c] = || x[b, :, :, c] ||_2, (2) x = x * scale[b][c] + shift[c] in place, with
// scale = 1 + weight * gx / (mean_c gx + eps) computed by the caller on the tiny [n][c] tensor.
// Norm: workgroup = (image, 64 channels); lanes walk channels (256-byte rows), the 4 waves take every 4th
// position; fixed summation order -> bit-reproducible.
__global__ __launch_bounds__(256) void k_grn_norm(const float *__restrict__ x, int hw, int c, float *__restrict__ gx) {
    __shared__ float part[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * 64 + lane;
    const int64_t img = blockIdx.y;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < c) {
        const float *p = x + img * hw * (int64_t)c + c0;
        int pos = wave;
        for (; pos + 12 < hw; pos += 16) {               // four independent chains per lane
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = p[(int64_t)(pos + 4 * u) * c];
                acc[u] = fmaf(v, v, acc[u]);
            }
        }
        for (; pos < hw; pos += 4) {
            const float v = p[(int64_t)pos * c];
            acc[0] = fmaf(v, v, acc[0]);
        }
    }
    part[wave][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (wave == 0 && c0 < c) gx[img * c + c0] = sqrtf((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}

// GRN scale vector: scale[b, c] = 1 + weight[c] * gx[b, c] / (mean_c gx[b, :] + eps), and the largest |scale| of the whole
// batch into smax[0] (the two-fp16-term Linear that folds the scale into its staging reads it as its device-side bound):
// unsigned atomic max on the float's bits (|scale| >= 0; NaN / inf sort above everything), smax zeroed by the caller.
// One workgroup per image; replaces five ATen launches per ConvNeXt block.
__global__ __launch_bounds__(256) void k_grn_scale(const float *__restrict__ gx, const float *__restrict__ weight, int c,
                                                   float eps, float *__restrict__ scale, unsigned *__restrict__ smax) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float *g = gx + (int64_t)blockIdx.x * c;
    float acc = 0.f;
    for (int i = threadIdx.x; i < c; i += 256) acc += g[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    const float inv = 1.f / (((red[0] + red[1]) + (red[2] + red[3])) / (float)c + eps);
    float vmax = 0.f;
    for (int i = threadIdx.x; i < c; i += 256) {
        const float v = fmaf(weight[i], g[i] * inv, 1.f);
        scale[(int64_t)blockIdx.x * c + i] = v;
        vmax = range_max(vmax, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = range_max(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) atomicMax(smax, __float_as_uint(vmax));
}

}  // namespace

hipError_t launch_dwconv7_nhwc(const float *x, const float *wt, const float *bias, int64_t n, int c, int h, int wd, float *y,
                               hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535 || (c + 63) / 64 > 65535 || (int64_t)h * wd * c > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dwconv7_nhwc, dim3((unsigned)((h + DR - 1) / DR), (unsigned)((c + 63) / 64), (unsigned)n), dim3(64), 0, st,
                       x, wt, bias, c, h, wd, y);
    return hipGetLastError();
}

hipError_t launch_dwconv7(const float *x, const float *w, const float *bias, int64_t n, int c, int h, int wd,
                          float *y, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535 || (c + DCH - 1) / DCH > 65535) return hipErrorInvalidValue;
    // tile edge: 12 when it wastes fewer outputs than 16 (24- and 12-wide maps), else 16
    auto waste = [&](int dt) { return (int64_t)((h + dt - 1) / dt * dt) * ((wd + dt - 1) / dt * dt); };
    if (waste(12) < waste(16)) {
        constexpr int DT = 12, DP = DT + 6;
        const int tiles = ((h + DT - 1) / DT) * ((wd + DT - 1) / DT);
        const size_t lds = (size_t)DCH * (DP * DP + 1) * sizeof(float);
        hipLaunchKernelGGL((k_dwconv7<12, 3>), dim3((unsigned)tiles, (unsigned)((c + DCH - 1) / DCH), (unsigned)n),
                           dim3(128), lds, st, x, w, bias, c, h, wd, y);
    } else {
        constexpr int DT = 16, DP = DT + 6;
        const int tiles = ((h + DT - 1) / DT) * ((wd + DT - 1) / DT);
        const size_t lds = (size_t)DCH * (DP * DP + 1) * sizeof(float);
        hipLaunchKernelGGL((k_dwconv7<16, 2>), dim3((unsigned)tiles, (unsigned)((c + DCH - 1) / DCH), (unsigned)n),
                           dim3(256), lds, st, x, w, bias, c, h, wd, y);
    }
    return hipGetLastError();
}

hipError_t launch_grn_scale(const float *gx, const float *weight, int64_t n, int c, float eps, float *scale, float *smax,
                            hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535 || c < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_grn_scale, dim3((unsigned)n), dim3(256), 0, st, gx, weight, c, eps, scale, reinterpret_cast<unsigned *>(smax));
    return hipGetLastError();
}

hipError_t launch_grn_norm(const float *x, int64_t n, int hw, int c, float *gx, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_grn_norm, dim3((unsigned)((c + 63) / 64), (unsigned)n), dim3(256), 0, st, x, hw, c, gx);
    return hipGetLastError();
}

}  // namespace mirx
