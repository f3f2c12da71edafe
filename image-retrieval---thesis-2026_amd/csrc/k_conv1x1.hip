// k_conv1x1.hip -- DenseNet dense-layer front half as ONE fp32 MFMA kernel:
//     y = act_out( W * act_in(x) + bias )        (1x1 convolution = GEMM over channels)
// with act_in(x)[k] = relu(x[k] * scale[k] + shift[k]) applied while the activation tile is staged
// (norm1 + relu1 of torchvision's _DenseLayer) and act_out = relu (norm2's shift is the bias, its
// scale is folded into W by the caller) -- so the concatenated feature buffer is read ONCE per layer
// instead of three times (BN pass, ReLU pass, conv read).  Replaces norm1 -> relu1 -> conv1 -> norm2
// -> relu2 of model.py:53's densenet121 (and, with the prologue off, the transition 1x1 conv).
//
// x is the channel-prefix view of the block buffer: image b, channel k, pixel p at
// x[b * x_batch_stride + k * hw + p].  y is packed NCHW [n, cout, hw].
//
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD): workgroup tile = 128 output
// channels x 64 pixels, 4 waves as 2 (channels) x 2 (pixels), K staged 16 channels at a time through
// a double-buffered LDS image (A = W^T [k][128], B = act [k][64], 16 channels per stage); both operand reads are one
// ds_read_b32 per lane with consecutive lanes on consecutive words (conflict-free).
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int CM = 128;   // output channels per workgroup

// NREP = 32-pixel tiles per wave: workgroup tile = 128 channels x (64 * NREP) pixels
template <bool VEC4, bool PROLOGUE, bool RELU_OUT, int NREP, int KC>
__global__ __launch_bounds__(256) void k_conv1x1(const float *__restrict__ x, int64_t xbs, int cin,
                                                 const float *__restrict__ scale,
                                                 const float *__restrict__ shift,
                                                 const float *__restrict__ wt, const float *__restrict__ bias,
                                                 int64_t n, int hw, int cout, float *__restrict__ y) {
    constexpr int CP = 64 * NREP;
    __shared__ __attribute__((aligned(16))) float sA[2][KC][CM];
    __shared__ __attribute__((aligned(16))) float sB[2][KC][CP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t total = n * (int64_t)hw;
    const int64_t p0 = (int64_t)blockIdx.x * CP;
    const int co0 = blockIdx.y * CM;

    // staging assignments
    const int a_k = threadIdx.x >> 5, a_m4 = threadIdx.x & 31;            // A: rows a_k + 8 i, 4 channels at 4 a_m4
    constexpr int PG = CP / 4;                                            // float4 pixel groups per row
    const int b_k = threadIdx.x / PG, b_p4 = threadIdx.x % PG;            // B (VEC4): rows b_k + (256/PG) i, 4 pixels at 4 b_p4
    int64_t b_src = -1;                                                   // element offset of this thread's pixel (group)
    if (VEC4) {
        const int64_t pp = p0 + 4 * b_p4;
        if (pp < total) b_src = (pp / hw) * xbs + (pp % hw);
    } else {
        const int64_t pp = p0 + (threadIdx.x % CP);                       // scalar path: one pixel per thread
        if (pp < total) b_src = (pp / hw) * xbs + (pp % hw);
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < KC / 8; ++i) {
            const int k = a_k + 8 * i;
            const float4 v = *reinterpret_cast<const float4 *>(wt + (int64_t)(k0 + k) * cout + co0 + 4 * a_m4);
            *reinterpret_cast<float4 *>(&sA[buf][k][4 * a_m4]) = v;
        }
        if (VEC4) {
            constexpr int BIT = (KC * PG + 255) / 256;                 // float4 groups per thread
#pragma unroll
            for (int i = 0; i < BIT; ++i) {
                const int k = b_k + (256 / PG) * i;
                if (KC * PG < 256 && k >= KC) break;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b_src >= 0) {
                    v = *reinterpret_cast<const float4 *>(x + b_src + (int64_t)(k0 + k) * hw);
                    if (PROLOGUE) {
                        const float sc = scale[k0 + k], sh = shift[k0 + k];
                        v.x = fmaxf(fmaf(v.x, sc, sh), 0.f); v.y = fmaxf(fmaf(v.y, sc, sh), 0.f);
                        v.z = fmaxf(fmaf(v.z, sc, sh), 0.f); v.w = fmaxf(fmaf(v.w, sc, sh), 0.f);
                    }
                }
                *reinterpret_cast<float4 *>(&sB[buf][k][4 * b_p4]) = v;
            }
        } else {
            // hw not a multiple of 4 (7x7 maps): scalar gather, 8 elements per thread
#pragma unroll
            for (int i = 0; i < KC * CP / 256; ++i) {
                const int k = threadIdx.x / CP + (256 / CP) * i, p = threadIdx.x % CP;
                float v = 0.f;
                if (b_src >= 0) {
                    v = x[b_src + (int64_t)(k0 + k) * hw];
                    if (PROLOGUE) v = fmaxf(fmaf(v, scale[k0 + k], shift[k0 + k]), 0.f);
                }
                sB[buf][k][p] = v;
            }
        }
    };

    f32x16 acc[2][NREP];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NREP; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nk = cin / KC;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __syncthreads();                                   // stage kt visible; buffer cur^1 free
        if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * KC);
        const int kh = lane >> 5, nn = wn * 32 * NREP + (lane & 31), m0 = wm * 64 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < KC / 2; ++kk) {
            float bv[NREP];
#pragma unroll
            for (int ni = 0; ni < NREP; ++ni) bv[ni] = sB[cur][2 * kk + kh][nn + 32 * ni];
            const float a0 = sA[cur][2 * kk + kh][m0];
            const float a1 = sA[cur][2 * kk + kh][m0 + 32];
#pragma unroll
            for (int ni = 0; ni < NREP; ++ni) {
                acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[ni], acc[0][ni], 0, 0, 0);
                acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[ni], acc[1][ni], 0, 0, 0);
            }
        }
    }

    // epilogue: register r of tile (mi, ni) = channel co0 + 64 wm + 32 mi + (r&3) + 8 (r>>2) + 4 (lane>>5),
    // pixel p0 + 32 NREP wn + 32 ni + (lane & 31)
#pragma unroll
    for (int ni = 0; ni < NREP; ++ni) {
        const int64_t pp = p0 + wn * 32 * NREP + 32 * ni + (lane & 31);
        if (pp >= total) continue;
        const int64_t bimg = pp / hw, off = pp % hw;
        float *yo = y + bimg * (int64_t)cout * hw + off;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = co0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[mi][ni][r] + (bias ? bias[ch] : 0.f);
                if (RELU_OUT) v = fmaxf(v, 0.f);
                yo[(int64_t)ch * hw] = v;
            }
    }
}

template <bool VEC4, int NREP, int KC>
hipError_t launch_v(const float *x, int64_t xbs, int cin, const float *scale, const float *shift, const float *wt,
                    const float *bias, int64_t n, int hw, int cout, int relu_out, float *y, hipStream_t st) {
    constexpr int CP = 64 * NREP;
    const dim3 grid((unsigned)((n * (int64_t)hw + CP - 1) / CP), (unsigned)(cout / CM));
#define MIRX_C1(P, R)                                                                                       \
    hipLaunchKernelGGL((k_conv1x1<VEC4, P, R, NREP, KC>), grid, dim3(256), 0, st, x, xbs, cin, scale, shift, wt, bias, n, \
                       hw, cout, y)
    if (scale) {
        if (relu_out) MIRX_C1(true, true); else MIRX_C1(true, false);
    } else {
        if (relu_out) MIRX_C1(false, true); else MIRX_C1(false, false);
    }
#undef MIRX_C1
    return hipGetLastError();
}

}  // namespace

hipError_t launch_conv1x1(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                          const float *wt, const float *bias, int64_t n, int hw, int cout, int relu_out, float *y,
                          hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (cin % 32 || cout % CM) return hipErrorInvalidValue;
    // NREP = 1 (64-pixel tiles, 48 KiB LDS, 3 workgroups per CU) measured faster than NREP = 2
    // (128-pixel tiles, 2 per CU) on every DenseNet-121 layer shape: 13.6k vs 13.3k img/s end to end
    // Tile choice measured on DenseNet-121 (B = 256, end-to-end img/s): 64 pixels x 16 channels per
    // stage (24 KiB LDS, 6 workgroups per CU) 14.07k; 64 x 32: 13.62k; 64 x 8: 13.39k; 128 x 16: 13.47k;
    // 128 x 32: 13.24k -- occupancy beats weight-tile reuse here.
    if ((hw & 3) == 0 && (xbs & 3) == 0)
        return launch_v<true, 1, 16>(x, xbs, cin, scale, shift, wt, bias, n, hw, cout, relu_out, y, st);
    return launch_v<false, 1, 16>(x, xbs, cin, scale, shift, wt, bias, n, hw, cout, relu_out, y, st);
}

}  // namespace mirx
