// k_linear_s3.hip -- fp32-grade Linear layer  y = epi(x W^T + b)  for the token-major models (DINOv2 /
// MedSigLIP blocks, ConvNeXtV2 point-wise MLPs), on the bf16 matrix pipe with every fp32 operand carried as
// THREE bf16 terms (the scheme of k_conv1x1_s3.hip: 6 v_mfma_f32_32x32x16_bf16 per product block, dropped
// cross terms <= 3 * 2^-24 |w x|, i.e. the rounding class of one fp32 product).
//
//   x   [m][k] fp32, k contiguous (tokens x features)        -> MFMA row operand, split while it is staged
//   w3  [ceil(n / 128)][k / 16][3 terms][128][16] bf16, rows >= n zero -> MFMA column operand (mirx.model._split3_weights)
//   y   [m][n] fp32:  v = acc + bias[n];  ACT == 1: v = gelu(v) (erf form);
//                     RES: v = res[m][n] + gamma[n] * v  (LayerScale + residual; y may alias res)
//   NCHW variant (ConvNeXt block tail): token t = image t / tpi, pixel t % tpi; y and res are [image][n][tpi].
//   The MFMA operand roles are swapped (rows = outputs, lanes = tokens) so that a half-wave still stores 128
//   contiguous bytes.
//
// Workgroup tile 128 tokens x 128 outputs, 4 waves (2 x 2, 64 x 64 each), 16 features per stage,
// double-buffered LDS (48 KiB, 3 workgroups per CU).  The weights go global -> LDS by DMA (no registers); the
// activations are register-prefetched one stage ahead, two stages per load group so that each 128-byte line of
// a token row is fetched while it is still in the vector L1.  The output tile index runs
// with n fastest and is laid out per XCD (workgroup id % 8 = XCD on gfx950), so the n-tiles that share a token
// tile are resident on the SAME L2 at the same time: x is read from HBM once per token tile.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
// A stage is published by a barrier only after THIS wave's LDS DMA has landed: the compiler's own s_waitcnt
// before s_barrier covers the registers it knows about, not the asynchronous buffer_load ... lds writes.
#define STAGE_BARRIER()                                      \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        __syncthreads();                                     \
    } while (0)

constexpr int TM = 128;            // tokens per workgroup
constexpr int TN = 128;            // outputs per workgroup
constexpr int KC = 16;             // features per stage
constexpr int PLANE = 128 * KC * 2;            // bytes of one term of one operand stage (4 KiB)
constexpr int STAGE = 6 * PLANE;               // x terms [0, 3), w terms [3, 6): 24 KiB

template <int ACT, bool RES, bool NCHW, bool GRN>
__global__ __launch_bounds__(256, 3) void k_linear_s3(const float *__restrict__ x, int64_t m, int k,
                                                      const uint16_t *__restrict__ w3,
                                                      const float *__restrict__ bias, int n, const float *res,
                                                      const float *__restrict__ gamma, float *y, int ntn,
                                                      int64_t total_tiles, int64_t per_xcd, int tpi) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
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
        (void *)(w3 + ((int64_t)tn * nk) * (3 * TN * KC)), 0, nk * (3 * TN * KC * 2), 0x00020000);
    const int w_voff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);

    f32x4 rx[2], rx2[2];
    // GRN (ConvNeXt block tail): x is multiplied by the per-(image, feature) scale `gamma` = [m / tpi][k] while it is
    // staged -- the GRN apply pass over the 4C-wide hidden map disappears (its shift is folded into the bias by the
    // caller: W (x s + b) = W (x s) + W b)
    f32x4 sx[2], sx2[2];
    const float *gsp = GRN ? gamma + (xr / tpi) * k + 8 * x_half : nullptr;
    auto dma_w = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(sm + buf * STAGE + 3 * PLANE + piece * 1024), 16,
                                                     w_voff, kt * (3 * TN * KC * 2) + piece * 1024, 0, 0);
        }
    };
    auto load_x = [&](int kt, f32x4 (&r)[2], f32x4 (&sc)[2]) {
        r[0] = *reinterpret_cast<const f32x4 *>(xsrc + kt * KC);
        r[1] = *reinterpret_cast<const f32x4 *>(xsrc + kt * KC + 4);
        if (GRN) {
            sc[0] = *reinterpret_cast<const f32x4 *>(gsp + kt * KC);
            sc[1] = *reinterpret_cast<const f32x4 *>(gsp + kt * KC + 4);
        }
    };
    auto store_x = [&](int buf, const f32x4 (&r)[2], const f32x4 (&sc)[2]) {
        char *sb = sm + buf * STAGE;
        u32x4 ph, pm, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 v = {r[j >> 1][2 * (j & 1)], r[j >> 1][2 * (j & 1) + 1]};
            if (GRN) {
                v[0] *= sc[j >> 1][2 * (j & 1)];
                v[1] *= sc[j >> 1][2 * (j & 1) + 1];
            }
            const bf16x2 h = __builtin_convertvector(v, bf16x2);
            const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
            const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
            const f32x2 r2 = r1 - __builtin_convertvector(mm, f32x2);
            const bf16x2 l = __builtin_convertvector(r2, bf16x2);
            ph[j] = __builtin_bit_cast(unsigned, h);
            pm[j] = __builtin_bit_cast(unsigned, mm);
            pl[j] = __builtin_bit_cast(unsigned, l);
        }
        *reinterpret_cast<u32x4 *>(sb + x_lds) = ph;
        *reinterpret_cast<u32x4 *>(sb + x_lds + PLANE) = pm;
        *reinterpret_cast<u32x4 *>(sb + x_lds + 2 * PLANE) = pl;
    };

    // ---- fragment addressing: lane -> row (lane & 31), K chunk (lane >> 5) -------------------------------
    const int kg = lane >> 5;
    int fx[2], fw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int rx_ = wm * 64 + t * 32 + (lane & 31);
        fx[t] = rx_ * 32 + ((kg ^ ((rx_ >> 3) & 1)) << 4);
        const int rw_ = wn * 64 + t * 32 + (lane & 31);
        fw[t] = 3 * PLANE + rw_ * 32 + ((kg ^ ((rw_ >> 3) & 1)) << 4);
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
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[t][p] = *reinterpret_cast<const bf16x8 *>(sb + fx[t] + p * PLANE);
                b[t][p] = *reinterpret_cast<const bf16x8 *>(sb + fw[t] + p * PLANE);
            }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                f32x16 c = acc[mi][ni];
                // smallest terms first; NCHW: outputs on the MFMA rows, tokens on the lanes
#define MIRX_L3_MFMA(TA, TB)                                                                               \
    c = NCHW ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni][TB], a[mi][TA], c, 0, 0, 0)                   \
             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][TA], b[ni][TB], c, 0, 0, 0);
                MIRX_L3_MFMA(1, 1)
                MIRX_L3_MFMA(2, 0)
                MIRX_L3_MFMA(0, 2)
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
                    float v = acc[mi][ni][r] + (bias ? bias[col] : 0.f);
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
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + 32 * ni + (lane & 31);
        if (col >= n) continue;                                 // zero-padded weight rows of the last output tile
        const float bv = bias ? bias[col] : 0.f;
        const float gv = (RES && gamma) ? gamma[col] : 1.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= m) continue;
                float v = acc[mi][ni][r] + bv;
                if (ACT == 1) v = gelu_erf(v);
                    if (ACT == 2) v = gelu_tanh(v);
                if (RES) v = res[row * n + col] + gv * v;
                y[row * n + col] = v;
            }
    }
}

}  // namespace

hipError_t launch_linear_s3(const float *x, int64_t m, int k, const uint16_t *w3, const float *bias, int n, int act,
                            const float *res, const float *gamma, float *y, int tokens_per_image, hipStream_t st) {
    if (m <= 0) return hipSuccess;
    if (k % KC || n < 1 || act < 0 || act > 2 || tokens_per_image < 0) return hipErrorInvalidValue;
    const int ntn = (n + TN - 1) / TN;                 // w3 holds ntn * 128 rows, zero beyond n
    const int64_t total = ((m + TM - 1) / TM) * ntn;
    const int64_t per_xcd = (total + 7) / 8;
    if (per_xcd * 8 > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(per_xcd * 8));
    const size_t lds = 2 * (size_t)STAGE;
#define MIRX_L3(A, R, C, G)                                                                                \
    {                                                                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_s3<A, R, C, G>),        \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL((k_linear_s3<A, R, C, G>), grid, dim3(256), lds, st, x, m, k, w3, bias, n, res, gamma, y, \
                           ntn, total, per_xcd, tokens_per_image);                                         \
    }
    if (tokens_per_image > 0) {                       // ConvNeXt block tail: `gamma` = GRN input scale [images][k] or null
        if (act) return hipErrorInvalidValue;
        if (gamma) {
            if (res) MIRX_L3(0, true, true, true) else MIRX_L3(0, false, true, true)
        } else {
            if (res) MIRX_L3(0, true, true, false) else MIRX_L3(0, false, true, false)
        }
    } else if (res) {
        if (act == 2) return hipErrorInvalidValue;
        if (act) MIRX_L3(1, true, false, false) else MIRX_L3(0, true, false, false)
    } else {
        if (act == 2) MIRX_L3(2, false, false, false) else if (act) MIRX_L3(1, false, false, false) else MIRX_L3(0, false, false, false)
    }
#undef MIRX_L3
    return hipGetLastError();
}

}  // namespace mirx
