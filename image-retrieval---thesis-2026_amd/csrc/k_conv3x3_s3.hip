// k_conv3x3_s3.hip -- the Winograd F(2x2, 3x3) convolution of k_conv3x3.hip with the 16 channel GEMMs on the
// bf16 matrix pipe, every fp32 operand carried as THREE bf16 terms (the scheme of k_conv1x1_s3.hip):
//     M_ij[oc, tile] = sum_c U_ij[oc, c] V_ij[c, tile],   U = Uh + Um + Ul,  V = Vh + Vm + Vl  (exact, 3 x 8 bits)
//     U V ~= Um Vm + Ul Vh + Uh Vl + Um Vh + Uh Vm + Uh Vh          (6 v_mfma_f32_32x32x16_bf16, fp32 accumulation)
// The dropped cross terms are <= 3 * 2^-24 |U V|: the rounding class of one fp32 product.  Per 16 channels and
// Winograd component a wave issues 6 MFMAs of 32 cycles instead of 8 fp32 MFMAs of 64: 2.67x less matrix time,
// which the fp32 kernel is bound by (61 % of its cycles).
//
// Same geometry as k_conv3x3_wino: one workgroup = a strip of up to 28 tiles of one image (28 of 32 MFMA
// columns), wave i owns Winograd row i and its four chains j = 0..3, the output transform and the direct write
// into the dense block's buffer are unchanged.  What changes is the operand path:
//   * V: a lane now owns 8 CONSECUTIVE channels of its tile (the K chunk of its half-wave): it forms T_i and the
//     four V_ij of each channel from the raw input rows in LDS (fp32, as before), then splits the 32 values into
//     three bf16 terms in registers (v_cvt_pk_bf16_f32) -- ~200 VALU instructions per 16-channel stage.
//   * U: pre-transformed AND pre-split by the caller ([stage][ij][term][oc][16 channels] bf16); each lane reads
//     its 16-byte A fragments straight from global memory (L2-resident: 393 KiB per layer shared by every
//     workgroup) at the top of the stage; no LDS, no staging registers.
// 2 workgroups per CU (<= 256 VGPRs): LDS holds only the raw input rows (16 channels per stage, double-buffered,
// register-prefetched two stages ahead).
//
// Measured (1024 images, -DMIRX_W3_CYCLES breakdown per stage and wave: barrier 150, issuing the 12 U + 7 input
// loads 1220, LDS reads + transform 960, split + 24 MFMAs 1310, LDS stores 530 ticks): with only 32 output
// channels a V value feeds 192 MACs, so once the matrix time shrinks 2.67x the kernel is bound by operand
// DELIVERY -- 12 KiB of U through the vector L1 and 16 KiB of input rows through LDS per wave and stage for 768
// cycles of MFMA.  Net: 28x28 maps 0.325 vs 0.364 ms, 14x14 0.094 vs 0.102 ms, 56x56 1.42 vs 1.375 ms against the
// fp32-MFMA kernel; the model uses this kernel on the 28 / 14 maps only.  A wave tile of two column blocks (U
// fragments reused for 64 tiles) is what would lift the bound.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int CIN = 128, COUT = 32;
constexpr int KC = 16;                        // channels per stage = one MFMA K
constexpr int NST = CIN / KC;                 // 8 stages

// two fp32 values -> the packed bf16 pairs of their three terms
__device__ inline void split_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
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

// W = map side (56 / 28 / 14); R = tile rows per strip (1 / 2 / 4)
template <int W, int R>
__global__ __launch_bounds__(256, 2) void k_conv3x3_wino_s3(const float *__restrict__ x, const uint16_t *__restrict__ u3,
                                                            float *__restrict__ out, int64_t out_bs) {
    constexpr int TW = W / 2;                 // tiles per row
    constexpr int ROWS = 2 * R + 2;           // input rows of a strip
    constexpr int PITCH = W + 4;              // input row pitch in LDS: col -1 at index 1, even, >= W + 3
    constexpr int PLANE = ROWS * PITCH;       // one channel
    constexpr int IN_STAGE = KC * PLANE;
    constexpr int XB = 2 * COUT * 32;         // floats one wave writes to the exchange buffer
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_in = sm;                         // [2][KC][ROWS][PITCH]
    float *s_x = sm;                          // [4 waves][2][32 oc][32 tiles], after the K loop

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = lane >> 5, n = lane & 31;
    const int strip = blockIdx.x;
    const int64_t img = blockIdx.y;
    const int oy0 = strip * 2 * R;                           // first output row of the strip
    const float *xi = x + img * CIN * (int64_t)(W * W);

    // ---- input staging: KC * ROWS rows of W floats as float2 pairs; thread t takes items t, t + 256, ... ----
    // TWO register sets: the loads of stage st + 2 are issued while stage st computes and stage st + 1 waits in
    // the other set (one stage of compute is shorter than the HBM latency; with a single set every stage ended
    // in a ~2 us wait).
    constexpr int IN_F2 = KC * ROWS * (W / 2);
    constexpr int IN_PER = (IN_F2 + 255) / 256;
    float2 rin_a[IN_PER], rin_b[IN_PER];
    // Loads are unconditional (rows outside the map read a clamped row and are zeroed when they are stored): a
    // conditional load makes hipcc wait for every outstanding load before it re-initialises the register.
    auto load = [&](int st, float2 (&rin)[IN_PER]) {
#pragma unroll
        for (int i = 0; i < IN_PER; ++i) {
            int it = threadIdx.x + 256 * i;
            if (it >= IN_F2) it = IN_F2 - 1;
            const int c = it / (ROWS * (W / 2)), r = (it / (W / 2)) % ROWS, q = it % (W / 2);
            int iy = oy0 - 1 + r;
            iy = iy < 0 ? 0 : (iy >= W ? W - 1 : iy);
            rin[i] = *reinterpret_cast<const float2 *>(xi + ((int64_t)(st * KC + c) * W + iy) * W + 2 * q);
        }
    };
    auto store = [&](int buf, const float2 (&rin)[IN_PER]) {
#pragma unroll
        for (int i = 0; i < IN_PER; ++i) {
            const int it = threadIdx.x + 256 * i;
            const int c = it / (ROWS * (W / 2)), r = (it / (W / 2)) % ROWS, q = it % (W / 2);
            const int iy = oy0 - 1 + r;
            const bool inside = iy >= 0 && iy < W;
            if (it < IN_F2) {
                float *d = s_in + buf * IN_STAGE + (c * ROWS + r) * PITCH + 1 + 2 * q;     // col x at index x + 1
                d[0] = inside ? rin[i].x : 0.f;
                d[1] = inside ? rin[i].y : 0.f;
            }
        }
    };
    // the halo columns (x = -1 and x = W) are zero in both buffers for the whole kernel
    for (int i = threadIdx.x; i < 2 * KC * ROWS; i += 256) {
        float *row = s_in + (i / (KC * ROWS)) * IN_STAGE + (i % (KC * ROWS)) * PITCH;
        row[0] = 0.f;
        row[W + 1] = 0.f;
        row[W + 2] = 0.f;
        row[W + 3] = 0.f;
    }

    // ---- this lane's tile, this wave's pair of input rows, this lane's U fragments -------------------------
    const int tile = n < R * TW ? n : R * TW - 1;            // idle lanes shadow the last tile (never stored)
    const int tr = tile / TW, tc = tile % TW;
    // T_i = d[ra] + sb * d[rb]:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const int in_a = 8 * half * PLANE + (2 * tr + ra) * PITCH + 2 * tc;      // + q * PLANE; cols 2tc-1 .. 2tc+2 at +0..+3
    const int in_b = 8 * half * PLANE + (2 * tr + rb) * PITCH + 2 * tc;
    // U3[stage][ij = 4 wave + j][term][oc = n][16]: this lane's 16 bytes at + 8 half
    const uint16_t *up = u3 + ((int64_t)(4 * wave) * 3 * COUT + n) * KC + 8 * half;
    constexpr int U_IJ = 3 * COUT * KC;                      // bf16 per Winograd component of one stage
    constexpr int U_ST = 16 * U_IJ;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // registers: rin_a / rin_b alternate; at the top of stage st, LDS holds stage st and one set holds stage st + 1
    load(0, rin_a);
    store(0, rin_a);
    load(1, rin_b);
#ifdef MIRX_W3_CYCLES
    unsigned long long cyc[5] = {0, 0, 0, 0, 0};
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#define W3_STAMP(K) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cyc[K] += t_ - t_prev; t_prev = t_; }
#else
#define W3_STAMP(K)
#endif
    auto stage = [&](int st, float2 (&rnext)[IN_PER], const float2 (&rstore)[IN_PER]) {
        const int cur = st & 1;
#ifdef MIRX_W3_CYCLES
        unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();                                   // stage st visible; buffer cur ^ 1 free
        W3_STAMP(0)
        // A fragments of this stage straight from global memory (L2): 12 x 16 B per lane
        bf16x8 ua[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                ua[j][t] = *reinterpret_cast<const bf16x8 *>(up + (int64_t)st * U_ST + j * U_IJ + t * COUT * KC);
        load(st + 2 < NST ? st + 2 : NST - 1, rnext);      // branch-free: the tail re-loads the last stage
        __builtin_amdgcn_sched_barrier(0);
        W3_STAMP(1)

        // B fragments: 8 channels of this lane's tile -> T_i, the four V_ij, three bf16 terms each
        const float *si = s_in + cur * IN_STAGE;
        float v[4][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 a0 = *reinterpret_cast<const float2 *>(si + q * PLANE + in_a);
            const float2 a1 = *reinterpret_cast<const float2 *>(si + q * PLANE + in_a + 2);
            const float2 b0 = *reinterpret_cast<const float2 *>(si + q * PLANE + in_b);
            const float2 b1 = *reinterpret_cast<const float2 *>(si + q * PLANE + in_b + 2);
            const float t0 = fmaf(sb, b0.x, a0.x), t1 = fmaf(sb, b0.y, a0.y);
            const float t2 = fmaf(sb, b1.x, a1.x), t3 = fmaf(sb, b1.y, a1.y);
            v[0][q] = t0 - t2;
            v[1][q] = t1 + t2;
            v[2][q] = t2 - t1;
            v[3][q] = t1 - t3;
        }
#ifdef MIRX_W3_CYCLES
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W3_STAMP(2)
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 ph, pm, pl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned th, tm, tl;
                split_pair(v[j][2 * p], v[j][2 * p + 1], th, tm, tl);
                ph[p] = th; pm[p] = tm; pl[p] = tl;
            }
            const bf16x8 bh = __builtin_bit_cast(bf16x8, ph), bm = __builtin_bit_cast(bf16x8, pm),
                         bl = __builtin_bit_cast(bf16x8, pl);
            f32x16 c = acc[j];
            // smallest terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j][1], bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j][2], bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j][0], bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j][1], bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j][0], bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua[j][0], bh, c, 0, 0, 0);
            acc[j] = c;
        }
#ifdef MIRX_W3_CYCLES
        __builtin_amdgcn_sched_barrier(0);
        W3_STAMP(3)
        __builtin_amdgcn_sched_barrier(0);
#endif
        store(cur ^ 1, rstore);                            // stage st + 1 (loaded one stage ago)
        W3_STAMP(4)
    };
    static_assert(NST % 2 == 0, "the stage loop runs in pairs");
    for (int st = 0; st < NST; st += 2) {
        stage(st, rin_a, rin_b);                           // fetch st + 2 into a, publish b = st + 1
        stage(st + 1, rin_b, rin_a);
    }

#ifdef MIRX_W3_CYCLES
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    // ---- output transform (as k_conv3x3_wino) ------------------------------------------------------------------
    __syncthreads();                                       // every wave is done with the staging buffers
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = 8 * (r >> 2) + (r & 3) + 4 * half;
        const float p0 = acc[0][r] + acc[1][r] + acc[2][r];
        const float p1 = acc[1][r] - acc[2][r] - acc[3][r];
        s_x[wave * XB + oc * 32 + n] = p0;
        s_x[wave * XB + COUT * 32 + oc * 32 + n] = p1;
    }
    __syncthreads();
    // item = (oc, tile): 32 x 28 items, 256 threads; lanes walk tiles -> coalesced float2 stores
    float *oi = out + img * out_bs;
    for (int it = threadIdx.x; it < COUT * 32; it += 256) {
        const int oc = it >> 5, t = it & 31;
        if (t < R * TW && oy0 + 2 * (t / TW) < W) {
            const int otr = t / TW, otc = t % TW;
            float p[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                p[i][0] = s_x[i * XB + oc * 32 + t];
                p[i][1] = s_x[i * XB + COUT * 32 + oc * 32 + t];
            }
            float *o = oi + ((int64_t)oc * W + oy0 + 2 * otr) * W + 2 * otc;
            *reinterpret_cast<float2 *>(o) = make_float2(p[0][0] + p[1][0] + p[2][0], p[0][1] + p[1][1] + p[2][1]);
            *reinterpret_cast<float2 *>(o + W) = make_float2(p[1][0] - p[2][0] - p[3][0], p[1][1] - p[2][1] - p[3][1]);
        }
    }
#ifdef MIRX_W3_CYCLES
    if (blockIdx.x == 5 && blockIdx.y == 300 && lane == 0)
        printf("W=%d wave %d: barrier %llu  issue-loads %llu  lds+transform %llu  split+mfma %llu  store %llu | prologue %llu  loop %llu  epilogue %llu (100 MHz ticks x cycles?)\n",
               W, wave, cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], (t_loop - t_start) - (cyc[0] + cyc[1] + cyc[2] + cyc[3] + cyc[4]),
               cyc[0] + cyc[1] + cyc[2] + cyc[3] + cyc[4], __builtin_amdgcn_s_memtime() - t_loop);
#endif
}

template <int W, int R>
hipError_t launch_ws3(const float *x, const uint16_t *u3, int64_t n, float *out, int64_t out_bs, hipStream_t st) {
    constexpr int ROWS = 2 * R + 2, PITCH = W + 4;
    const size_t stage = (size_t)(2 * KC * ROWS * PITCH) * sizeof(float);
    const size_t xch = (size_t)4 * 2 * COUT * 32 * sizeof(float);
    const size_t lds = stage > xch ? stage : xch;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_wino_s3<W, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_conv3x3_wino_s3<W, R>), dim3((W / 2 + R - 1) / R, (unsigned)n), dim3(256), lds, st, x, u3, out,
                       out_bs);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_conv3x3_wino_s3(const float *x, const uint16_t *u3, int64_t n, int side, float *out, int64_t out_bs,
                                  hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (n > 65535) return hipErrorInvalidValue;
    if (side == 56) return launch_ws3<56, 1>(x, u3, n, out, out_bs, st);
    if (side == 28) return launch_ws3<28, 2>(x, u3, n, out, out_bs, st);
    if (side == 14) return launch_ws3<14, 4>(x, u3, n, out, out_bs, st);
    return hipErrorInvalidValue;
}

}  // namespace mirx
