// k_linear_h2.hip -- k_linear_s3 (token-major Linear, fp32 in / fp32 out) with every fp32 operand carried as TWO fp16
// terms (x = xh + xl exactly to 22 bits) and THREE v_mfma_f32_32x32x16_f16 per product block (xl wh + xh wl + xh wh;
// the dropped xl wl is 2^-22 of the product) instead of three bf16 terms and six MFMAs: half the matrix work and 4
// instead of 6 LDS bytes per element at the same measured error (both are dominated by the fp32 accumulation).
// 301-330 TFLOP/s fp32-equivalent on the DINOv2 shapes vs 171-199 for k_linear_s3 (the chip holds 1.7 instead of
// 1.36 GHz under it).
//
// fp16 has 5 exponent bits, so the CALLER supplies range information: the weights arrive multiplied by a power of two
// (mirx.model._linear_h2_weights: largest |w| in [2^13, 2^14)), x is multiplied by the power of two `x_scale` while it
// is staged, and the accumulator by `out_scale` = 1 / (x_scale * w_scale) before bias / activation -- all exact.
// (Two token tiles per workgroup against one staged copy of the weights -- the change that gave the DenseNet 1x1 conv 6 % --
// was built here too and measured 7-10 % SLOWER on all three token-major backbones: 212 VGPRs, two workgroups per CU instead
// of three, and this kernel lives on the matrix pipe, not on the bytes it pulls in.)
// Contract: |x * x_scale| <= 65504 for every element (mirx.model uses it where a bound is provable: the inputs that
// come out of a LayerNorm).  Everything else (tiling, DMA'd weight stages, paired x loads, XCD-contiguous tile order,
// epilogues) is k_linear_s3's.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;   /* fp16 in this experiment */
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
// A stage is published by a barrier only after THIS wave's LDS DMA has landed: the compiler's own s_waitcnt
// before s_barrier covers the registers it knows about, not the asynchronous buffer_load ... lds writes.
#define STAGE_BARRIER()                                      \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        __syncthreads();                                     \
    } while (0)

// Diagnostic builds (results wrong; attribute the kernel's time): -DMIRX_LH2_EXP=1 no fp16 split (raw bits staged),
// 2 no x loads inside the K loop, 3 plain-store epilogue (no activation / residual), 4 no MFMAs
#ifndef MIRX_LH2_EXP
#define MIRX_LH2_EXP 0
#endif

constexpr int TM = 128;            // tokens per workgroup
constexpr int TN = 128;            // outputs per workgroup
constexpr int KC = 16;             // features per stage
constexpr int PLANE = 128 * KC * 2;            // bytes of one term of one operand stage (4 KiB)
constexpr int STAGE = 4 * PLANE;               // x terms [0, 2), w terms [2, 4): 16 KiB

// SQ (row-major outputs only): the workgroup also returns the column sums of v^2 over its token rows, split by image (a tile
// of 128 rows spans at most two images when tokens_per_image >= 128): sq_out[tile_m][0 / 1][n] -- the partial sums of
// ConvNeXtV2's global response norm ||hid[b, :, c]||_2, so that no separate pass has to read the 4C-wide hidden map again.
// Fixed summation order (registers by row, the two lane halves, the two wave rows): bit-reproducible.
template <int ACT, bool RES, bool NCHW, bool GRN, bool SQ = false>
__global__ __launch_bounds__(256, 3) void k_linear_h2(const float *__restrict__ x, int64_t m, int k,
                                                      const uint16_t *__restrict__ w3,
                                                      const float *__restrict__ bias, int n, const float *res,
                                                      const float *__restrict__ gamma, float *y, int ntn,
                                                      int64_t total_tiles, int64_t per_xcd, int tpi, float x_scale,
                                                      float out_scale, const float *__restrict__ xb_dev,
                                                      float *__restrict__ sq_out = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    // xb_dev: the input's bound lives (partly) on the device -- |x * gamma| <= x_scale * xb_dev[0] with `x_scale` the host's
    // bound on |x| and xb_dev[0] the largest |gamma| (ConvNeXt GRN scale, computed per forward); the kernel derives the
    // power-of-two scales itself and `out_scale` arrives as 1 / w_scale.  A non-finite bound turns every output into NaN.
    if (xb_dev) {
        float xs, xi;
        range_scales(x_scale * xb_dev[0], xs, xi);
        x_scale = xs;
        out_scale *= xi;
    }
    const int64_t tile = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= total_tiles) return;
    const int tn = (int)(tile % ntn);
    const int64_t m0 = (tile / ntn) * TM;
    const int n0 = tn * TN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = k / KC;

    // ---- staging: x: thread -> token row (t >> 1), 8 features at 8 (t & 1); w: three 16-byte chunks ----------
    const int x_row = threadIdx.x >> 1, x_half = threadIdx.x & 1;
    int64_t xr = m0 + x_row;
    if (xr >= m) xr = m - 1;                                   // ragged last tile: read a valid row, never stored
    const float *xsrc = x + xr * k + 8 * x_half;
    const int x_lds = x_row * 32 + ((x_half ^ ((x_row >> 3) & 1)) << 4);               // + term * PLANE
    // w: the 12 KiB stage image goes global -> LDS by DMA (buffer_load ... lds: lane l of a wave writes 16 B at
    // piece base + 16 l), three 1-KiB pieces per wave.  Piece p, lane l is LDS (term p / 4, row 32 (p & 3) +
    // l / 2, slot l & 1), which holds source chunk (l & 1) ^ ((row >> 3) & 1) -- the same for every piece.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(w3 + ((int64_t)tn * nk) * (2 * TN * KC)), 0, nk * (2 * TN * KC * 2), 0x00020000);
    const int w_voff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);

    f32x4 rx[2], rx2[2];
    // GRN (ConvNeXt block tail): x is multiplied by the per-(image, feature) scale `gamma` = [m / tpi][k] while it is
    // staged -- the GRN apply pass over the 4C-wide hidden map disappears (its shift is folded into the bias by the
    // caller: W (x s + b) = W (x s) + W b)
    f32x4 sx[2], sx2[2];
    const float *gsp = GRN ? gamma + (xr / tpi) * k + 8 * x_half : nullptr;
    auto dma_w = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(sm + buf * STAGE + 2 * PLANE + piece * 1024), 16,
                                                     w_voff, kt * (2 * TN * KC * 2) + piece * 1024, 0, 0);
        }
    };
    auto load_x = [&](int kt, f32x4 (&r)[2], f32x4 (&sc)[2]) {
        if (MIRX_LH2_EXP == 2 && kt > 1) return;
        r[0] = *reinterpret_cast<const f32x4 *>(xsrc + kt * KC);
        r[1] = *reinterpret_cast<const f32x4 *>(xsrc + kt * KC + 4);
        if (GRN) {
            sc[0] = *reinterpret_cast<const f32x4 *>(gsp + kt * KC);
            sc[1] = *reinterpret_cast<const f32x4 *>(gsp + kt * KC + 4);
        }
    };
    auto store_x = [&](int buf, const f32x4 (&r)[2], const f32x4 (&sc)[2]) {
        char *sb = sm + buf * STAGE;
        u32x4 ph, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 v = {r[j >> 1][2 * (j & 1)], r[j >> 1][2 * (j & 1) + 1]};
            if (GRN) {
                v[0] *= sc[j >> 1][2 * (j & 1)];
                v[1] *= sc[j >> 1][2 * (j & 1) + 1];
            }
            unsigned th, tl;
            if (MIRX_LH2_EXP == 1) {
                th = __float_as_uint(v[0]);
                tl = __float_as_uint(v[1]);
            } else {
                split2h_pair(v[0] * x_scale, v[1] * x_scale, th, tl);
            }
            ph[j] = th;
            pl[j] = tl;
        }
        *reinterpret_cast<u32x4 *>(sb + x_lds) = ph;
        *reinterpret_cast<u32x4 *>(sb + x_lds + PLANE) = pl;
    };

    // ---- fragment addressing: lane -> row (lane & 31), K chunk (lane >> 5) -------------------------------
    const int kg = lane >> 5;
    int fx[2], fw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rx_ = wm * 64 + t * 32 + (lane & 31);
        fx[t] = rx_ * 32 + ((kg ^ ((rx_ >> 3) & 1)) << 4);
        const int rw_ = wn * 64 + t * 32 + (lane & 31);
        fw[t] = 2 * PLANE + rw_ * 32 + ((kg ^ ((rw_ >> 3) & 1)) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto mfma_stage = [&](int cur) {
        const char *sb = sm + cur * STAGE;
        bf16x8 a[2][2], b[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                a[t][p] = *reinterpret_cast<const bf16x8 *>(sb + fx[t] + p * PLANE);
                b[t][p] = *reinterpret_cast<const bf16x8 *>(sb + fw[t] + p * PLANE);
            }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                f32x16 c = acc[mi][ni];
                if (MIRX_LH2_EXP == 4) {
                    c[0] += (float)a[mi][0][0] + (float)a[mi][1][1] + (float)b[ni][0][2] + (float)b[ni][1][3];
                    acc[mi][ni] = c;
                    continue;
                }
                // smallest terms first; NCHW: outputs on the MFMA rows, tokens on the lanes
#define MIRX_L3_MFMA(TA, TB)                                                                               \
    c = NCHW ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b[ni][TB], a[mi][TA], c, 0, 0, 0)                    \
             : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][TA], b[ni][TB], c, 0, 0, 0);
                MIRX_L3_MFMA(1, 0)
                MIRX_L3_MFMA(0, 1)
                MIRX_L3_MFMA(0, 0)
#undef MIRX_L3_MFMA
                acc[mi][ni] = c;
            }
    };

    // Stages go in pairs: the x loads of an even and the following odd stage touch the SAME 128-byte lines
    // (16 features = 64 B per token row per stage), so both are issued together while the line is in the
    // vector L1; issued a stage apart, the second half comes from L2 again.
    dma_w(0, 0);
    load_x(0, rx, sx);
    load_x(nk > 1 ? 1 : 0, rx2, sx2);
    store_x(0, rx, sx);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        // ---- even stage kt in buffer 0; registers: rx2 = x of stage kt + 1 -------------------------------
        STAGE_BARRIER();
        dma_w(kt + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_stage(0);
        store_x(1, rx2, sx2);
        // ---- odd stage kt + 1 in buffer 1; loads the x of BOTH stages of the next pair -------------------
        STAGE_BARRIER();
        const int k2 = kt + 2 < nk ? kt + 2 : nk - 1, k3 = kt + 3 < nk ? kt + 3 : nk - 1;   // branch-free tail
        dma_w(k2, 0);
        load_x(k2, rx, sx);
        load_x(k3, rx2, sx2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_stage(1);
        store_x(0, rx, sx);
    }
    if (kt < nk) {                                         // odd stage count: the last stage sits in buffer 0
        STAGE_BARRIER();
        mfma_stage(0);
    }

    if (NCHW) {
        // register r of tile (mi, ni) = output n0 + 64 wn + 32 ni + (r&3) + 8 (r>>2) + 4 (lane>>5), token
        // m0 + 64 wm + 32 mi + (lane & 31): a half-wave stores 32 consecutive pixels of one output plane
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int64_t tok = m0 + wm * 64 + 32 * mi + (lane & 31);
            if (tok >= m) continue;
            const int64_t img = tok / tpi;
            const int64_t base = img * n * tpi + (tok - img * tpi);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = n0 + wn * 64 + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (col >= n) continue;                     // zero-padded weight rows of the last output tile
                    float v = acc[mi][ni][r] * out_scale + (bias ? bias[col] : 0.f);
                    if (ACT == 1) v = gelu_erf(v);
                    if (ACT == 2) v = gelu_tanh(v);
                    const int64_t idx = base + (int64_t)col * tpi;
                    if (RES) v = res[idx] + v;
                    y[idx] = v;
                }
        }
        return;
    }
    // epilogue: register r of tile (mi, ni) = token m0 + 64 wm + 32 mi + (r&3) + 8 (r>>2) + 4 (lane>>5),
    // output n0 + 64 wn + 32 ni + (lane & 31): a half-wave stores 128 contiguous bytes of one token row
    float sq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};                                  // SQ: [ni][first / second image of the tile]
    const int64_t second = SQ ? (m0 / tpi + 1) * (int64_t)tpi : 0;              // first row of the tile's second image
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + 32 * ni + (lane & 31);
        if (col >= n) continue;                                 // zero-padded weight rows of the last output tile
        const float bv = bias ? bias[col] : 0.f;
        const float gv = (RES && gamma && !GRN) ? gamma[col] : 1.f;       // (GRN: `gamma` is the input scale, not a LayerScale)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= m) continue;
                float v = acc[mi][ni][r] * out_scale + bv;
                if (MIRX_LH2_EXP != 3) {
                    if (ACT == 1) v = gelu_erf(v);
                    if (ACT == 2) v = gelu_tanh(v);
                    if (RES) v = res[row * n + col] + gv * v;
                }
                y[row * n + col] = v;
                if (SQ) {
                    if (row < second) sq[ni][0] = fmaf(v, v, sq[ni][0]);
                    else sq[ni][1] = fmaf(v, v, sq[ni][1]);
                }
            }
    }
    if (SQ) {
        __syncthreads();                                        // every wave is done with the stage buffers
        float *ssq = reinterpret_cast<float *>(sm);             // [wave row wm][image 0 / 1][128 columns]
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const float tot = sq[ni][im] + __shfl_xor(sq[ni][im], 32, 64);
                if (lane < 32) ssq[(wm * 2 + im) * TN + wn * 64 + ni * 32 + lane] = tot;
            }
        __syncthreads();
        const int im = threadIdx.x >> 7, cl = threadIdx.x & 127;
        if (n0 + cl < n) sq_out[((m0 / TM) * 2 + im) * (int64_t)n + n0 + cl] = ssq[im * TN + cl] + ssq[(2 + im) * TN + cl];
    }
}

}  // namespace

namespace {
// gx[img][col] = sqrt( sum over the 128-row tiles that touch image img of its partial ): k_linear_h2<.., SQ>'s sq_out summed in tile
// order (deterministic).  thread -> (image, column).
__global__ __launch_bounds__(256) void k_grn_norm_partials(const float *__restrict__ part, int tpi, int64_t n_img, int c,
                                                           float *__restrict__ gx) {
    const int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (it >= n_img * c) return;
    const int64_t img = it / c;
    const int col = (int)(it - img * c);
    const int64_t t0 = img * tpi / TM, t1 = ((img + 1) * tpi - 1) / TM;
    float acc = 0.f;
    for (int64_t t = t0; t <= t1; ++t) {
        const int sel = (t * TM) / tpi == img ? 0 : 1;
        acc += part[(t * 2 + sel) * (int64_t)c + col];
    }
    gx[it] = sqrtf(acc);
}
}  // namespace

hipError_t launch_grn_norm_partials(const float *part, int tpi, int64_t n_img, int c, float *gx, hipStream_t st) {
    if (n_img <= 0) return hipSuccess;
    if (tpi < TM || c < 1) return hipErrorInvalidValue;
    const int64_t blocks = (n_img * c + 255) / 256;
    if (blocks > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_grn_norm_partials, dim3((unsigned)blocks), dim3(256), 0, st, part, tpi, n_img, c, gx);
    return hipGetLastError();
}

hipError_t launch_linear_h2(const float *x, int64_t m, int k, const uint16_t *w2, const float *bias, int n, int act,
                            const float *res, const float *gamma, float x_scale, float out_scale, float *y,
                            int tokens_per_image, const float *xb_dev, hipStream_t st, bool rows_out, float *sq_out) {
    if (m <= 0) return hipSuccess;
    if (k % KC || n < 1 || act < 0 || act > 2) return hipErrorInvalidValue;
    const int ntn = (n + TN - 1) / TN;                 // w2 holds ntn * 128 rows, zero beyond n
    const int64_t total = ((m + TM - 1) / TM) * ntn;
    const int64_t per_xcd = (total + 7) / 8;
    if (per_xcd * 8 > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(per_xcd * 8));
    const size_t lds = 2 * (size_t)STAGE;
#define MIRX_H2(A, R, C, G)                                                                                \
    {                                                                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_h2<A, R, C, G>),        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL((k_linear_h2<A, R, C, G>), grid, dim3(256), lds, st, x, m, k, w2, bias, n, res, gamma, y, ntn, \
                           total, per_xcd, tokens_per_image, x_scale, out_scale, xb_dev);                  \
    }
    if (sq_out) {                                     // ConvNeXt fc1 + GELU with the GRN partial sums (row-major)
        if (act != 1 || res || gamma || xb_dev || tokens_per_image < TM) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_h2<1, false, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_linear_h2<1, false, false, false, true>), grid, dim3(256), lds, st, x, m, k, w2, bias, n, res, gamma, y,
                           ntn, total, per_xcd, tokens_per_image, x_scale, out_scale, xb_dev, sq_out);
    } else if (tokens_per_image > 0 && rows_out) {    // ConvNeXt block tail on a channels-last stream: GRN input scale, row-major output
        if (act || !gamma) return hipErrorInvalidValue;
        if (res) MIRX_H2(0, true, false, true) else MIRX_H2(0, false, false, true)
    } else if (tokens_per_image > 0) {                // ConvNeXt block tail / downsample: `gamma` = GRN input scale [images][k] or null
        if (act) return hipErrorInvalidValue;
        if (gamma) {
            if (res) MIRX_H2(0, true, true, true) else MIRX_H2(0, false, true, true)
        } else {
            if (res) MIRX_H2(0, true, true, false) else MIRX_H2(0, false, true, false)
        }
    } else if (res) {
        if (act == 2 || xb_dev) return hipErrorInvalidValue;
        if (act) MIRX_H2(1, true, false, false) else MIRX_H2(0, true, false, false)
    } else {
        if (xb_dev) return hipErrorInvalidValue;
        if (act == 2) MIRX_H2(2, false, false, false) else if (act) MIRX_H2(1, false, false, false) else MIRX_H2(0, false, false, false)
    }
#undef MIRX_H2
    return hipGetLastError();
}

}  // namespace mirx
