// k_conv1x1_s3.hip -- the dense-layer 1x1 convolution of k_conv1x1.hip for the MFMA-bound layers
// (many input channels, small maps), with each fp32 operand carried as THREE bf16 terms so that the
// products run on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate):
//     x = xh + xm + xl,  xh = bf16(x), xm = bf16(x - xh), xl = bf16(x - xh - xm)      (exact: 3 x 8 = 24 bits)
//     w * x ~= wh xh + wh xm + wm xh + wh xl + wl xh + wm xm                           (6 MFMAs, fp32 accumulate)
// The dropped cross terms are <= 3 * 2^-24 |w x|: the rounding class of one fp32 product, so the result is
// fp32-grade (tests: 2e-6 of the fp32 kernel).  6 bf16 MFMAs of 32 cycles replace 8 fp32 MFMAs of 64
// cycles per 16 channels: 2.67x less matrix-pipe time; the layer then runs against its HBM stream.
//
// Same contract as mirx_conv1x1_bn_relu (y = act_out(W * act_in(x) + bias), NCHW), except that the
// weights arrive pre-split: w3 = [cout / 128][cin / 16][3 terms][128 out][16 in] bf16 (mirx.model
// prepares it once per layer).  Workgroup tile 128 output channels x 128 pixels, 4 waves (2 x 2), one
// 16-channel K step per stage, double-buffered LDS (48 KiB, 3 workgroups per CU); weights by LDS DMA,
// activations register-prefetched.
// Activations are split while they are staged: a thread owns one pixel and 8 consecutive channels (8
// coalesced loads, BN + ReLU with wave-uniform scalars, three 16-byte LDS stores).  LDS rows are 32 B
// (16 bf16); chunk c of row r sits at c ^ ((r >> 3) & 1): conflict-free ds_read_b128 for the hardware's
// 16-lane groups.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int CM = 128;            // output channels per workgroup
constexpr int CP = 128;            // pixels per workgroup
constexpr int KC = 16;             // channels per stage
constexpr int PLANE_A = CM * KC * 2;   // bytes of one term of the weight stage (4 KiB)
constexpr int PLANE_B = CP * KC * 2;
constexpr int STAGE = 3 * PLANE_A + 3 * PLANE_B;   // 24 KiB

template <bool PROLOGUE, bool RELU_OUT>
__global__ __launch_bounds__(256, 3) void k_conv1x1_s3(const float *__restrict__ x, int64_t xbs, int cin,
                                                       const float *__restrict__ scale,
                                                       const float *__restrict__ shift,
                                                       const uint16_t *__restrict__ w3,
                                                       const float *__restrict__ bias, int64_t n, int hw, int cout,
                                                       float *__restrict__ y, int64_t ybs) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    __shared__ float sBias[CM];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t total = n * (int64_t)hw;
    const int64_t p0 = (int64_t)blockIdx.x * CP;
    const int co0 = blockIdx.y * CM;
    const int nk = cin / KC;

    // ---- staging assignments ------------------------------------------------------------------------
    // B: thread -> pixel (t & 127), channel group kg = t >> 7 (wave-uniform): channels 8 kg .. 8 kg + 7
    const int b_px = threadIdx.x & 127;
    const int b_kg = wave >> 1;
    int64_t b_off = 0;
    {
        const int64_t pp = p0 + b_px;
        if (pp < total) b_off = (pp / hw) * xbs + (pp % hw);
    }
    const float *xsrc = x + b_off + (int64_t)(8 * b_kg) * hw;
    const int b_lds = 3 * PLANE_A + b_px * 32 + ((b_kg ^ ((b_px >> 3) & 1)) << 4);     // + term * PLANE_B
    // A: the 12 KiB weight stage goes global -> LDS by DMA (buffer_load ... lds: lane l of a wave writes 16 B at
    // piece base + 16 l), three 1-KiB pieces per wave.  Piece p, lane l is LDS (term p / 4, row 32 (p & 3) +
    // l / 2, slot l & 1), which holds source chunk (l & 1) ^ ((row >> 3) & 1) -- the same for every piece.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(w3 + ((int64_t)blockIdx.y * nk) * (3 * CM * KC)), 0, nk * (3 * CM * KC * 2), 0x00020000);
    const int w_voff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    auto dma_w = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int piece = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(sm + buf * STAGE + piece * 1024), 16, w_voff,
                                                     kt * (3 * CM * KC * 2) + piece * 1024, 0, 0);
        }
    };

    float rb[8], rsc[8], rsh[8];
    auto load = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rb[j] = xsrc[((int64_t)kt * KC + j) * hw];
        if (PROLOGUE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                rsc[j] = scale[kt * KC + 8 * b_kg + j];
                rsh[j] = shift[kt * KC + 8 * b_kg + j];
            }
        }
    };
    auto store = [&](int buf) {
        char *sb = sm + buf * STAGE;
        // three bf16 terms of each value, two values at a time: fptrunc <2 x float> -> <2 x bfloat> is one
        // v_cvt_pk_bf16_f32 (round to nearest even) on gfx950
        u32x4 ph, pm, pl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x2 v = {rb[2 * j], rb[2 * j + 1]};
            if (PROLOGUE) {
                v[0] = fmaxf(fmaf(v[0], rsc[2 * j], rsh[2 * j]), 0.f);
                v[1] = fmaxf(fmaf(v[1], rsc[2 * j + 1], rsh[2 * j + 1]), 0.f);
            }
            const bf16x2 h = __builtin_convertvector(v, bf16x2);
            const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
            const bf16x2 m = __builtin_convertvector(r1, bf16x2);
            const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
            const bf16x2 l = __builtin_convertvector(r2, bf16x2);
            ph[j] = __builtin_bit_cast(unsigned, h);
            pm[j] = __builtin_bit_cast(unsigned, m);
            pl[j] = __builtin_bit_cast(unsigned, l);
        }
        *reinterpret_cast<u32x4 *>(sb + b_lds) = ph;
        *reinterpret_cast<u32x4 *>(sb + b_lds + PLANE_B) = pm;
        *reinterpret_cast<u32x4 *>(sb + b_lds + 2 * PLANE_B) = pl;
    };

    // ---- fragment addressing: lane -> row (lane & 31), K chunk (lane >> 5) -------------------------------
    const int kg = lane >> 5;
    int fa[2], fb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra_ = wm * 64 + t * 32 + (lane & 31);
        fa[t] = ra_ * 32 + ((kg ^ ((ra_ >> 3) & 1)) << 4);
        const int rb_ = wn * 64 + t * 32 + (lane & 31);
        fb[t] = 3 * PLANE_A + rb_ * 32 + ((kg ^ ((rb_ >> 3) & 1)) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if (threadIdx.x < CM) sBias[threadIdx.x] = bias ? bias[co0 + threadIdx.x] : 0.f;
    dma_w(0, 0);
    load(0);
    store(0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        // stage kt visible (this wave's weight DMA has landed: vmcnt(0) -- the compiler's own wait before
        // s_barrier does not cover the asynchronous LDS writes); buffer cur ^ 1 free
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int nxt = kt + 1 < nk ? kt + 1 : kt;         // branch-free: the last stage re-loads itself
        dma_w(nxt, cur ^ 1);
        load(nxt);
        __builtin_amdgcn_sched_barrier(0);
        const char *sb = sm + cur * STAGE;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[t][p] = *reinterpret_cast<const bf16x8 *>(sb + fa[t] + p * PLANE_A);
                b[t][p] = *reinterpret_cast<const bf16x8 *>(sb + fb[t] + p * PLANE_B);
            }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                f32x16 c = acc[mi][ni];
                // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][2], b[ni][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][1], b[ni][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][0], b[ni][0], c, 0, 0, 0);
                acc[mi][ni] = c;
            }
        store(cur ^ 1);
    }

    // epilogue: register r of tile (mi, ni) = channel co0 + 64 wm + 32 mi + (r&3) + 8 (r>>2) + 4 (lane>>5),
    // pixel p0 + 64 wn + 32 ni + (lane & 31)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t pp = p0 + wn * 64 + 32 * ni + (lane & 31);
        if (pp >= total) continue;
        const int64_t bimg = pp / hw, off = pp % hw;
        float *yo = y + bimg * ybs + off;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = co0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[mi][ni][r] + sBias[ch - co0];
                if (RELU_OUT) v = fmaxf(v, 0.f);
                yo[(int64_t)ch * hw] = v;
            }
    }
}

}  // namespace

hipError_t launch_conv1x1_s3(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                             const uint16_t *w3, const float *bias, int64_t n, int hw, int cout, int relu_out,
                             float *y, int64_t ybs, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (cin % KC || cout % CM) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((n * (int64_t)hw + CP - 1) / CP), (unsigned)(cout / CM));
    const size_t lds = 2 * (size_t)STAGE;
#define MIRX_S3(P, R)                                                                                      \
    {                                                                                                      \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv1x1_s3<P, R>),             \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
        if (e != hipSuccess) return e;                                                                     \
        hipLaunchKernelGGL((k_conv1x1_s3<P, R>), grid, dim3(256), lds, st, x, xbs, cin, scale, shift, w3, bias, n, hw, \
                           cout, y, ybs);                                                                       \
    }
    if (scale) {
        if (relu_out) MIRX_S3(true, true) else MIRX_S3(true, false)
    } else {
        if (relu_out) MIRX_S3(false, true) else MIRX_S3(false, false)
    }
#undef MIRX_S3
    return hipGetLastError();
}

}  // namespace mirx
