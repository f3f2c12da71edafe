// k_conv3x3.hip -- the 3x3 convolution of a DenseNet dense layer (128 -> 32 channels, stride 1, pad 1,
// no bias) as Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32, writing its 32 channels straight into
// the block buffer (the separate copy of the new channels disappears).
//
// Replaces conv2 of torchvision's _DenseLayer (model.py:53 `densenet121`) on the 56x56, 28x28 and 14x14
// maps (k_conv3x3_wino) and on the 7x7 maps of the last block (k_conv3x3_wino7): all 58 dense layers.
//
//   Y = A^T [ sum_c (G g G^T)_c  .  (B^T d_c B) ] A        per 2x2 output tile, 16 products instead of 36
//
// The sum over the 128 input channels of each of the 16 Winograd components is a GEMM
//   M_xi[oc, tile] = sum_c U_xi[oc, c] V_xi[c, tile],   xi = (i, j) in 4 x 4.
// One workgroup = one strip of up to 28 tiles of one image (one tile row of a 56-wide map, two of a
// 28-wide one, four of a 14-wide one) = 28 MFMA columns (4 idle).  Wave i owns Winograd row i: it forms T_i = row i of B^T d from TWO
// input rows, the four V_ij = T_i B with 8 adds per (tile, channel), and runs four MFMA chains
// (j = 0..3) -- the transformed input never exists in memory.  U (transformed weights, prepared once
// by the caller) and the raw input rows are staged 8 channels at a time through LDS (register
// prefetched, 47 KiB, three workgroups per CU).  The output transform first folds j inside the wave
// (P_i0 = M_i0 + M_i1 + M_i2, P_i1 = M_i1 - M_i2 - M_i3), exchanges the P's through LDS and finishes
// with Y_0p = P_0p + P_1p + P_2p, Y_1p = P_1p - P_2p - P_3p.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int CIN = 128, COUT = 32;
constexpr int KC = 8;                         // channels per stage
constexpr int NCH = CIN / KC;                 // 16 stages
constexpr int U_STAGE = 16 * KC * COUT;       // floats of transformed weights per stage (16 KiB)

// W = map side (56 / 28 / 14); R = tile rows per strip (1 / 2 / 4): 28 tiles per strip; the last strip of
// a 14-wide map holds three tile rows
template <int W, int R>
__global__ __launch_bounds__(256, 3) void k_conv3x3_wino(const float *__restrict__ x, const float *__restrict__ u,
                                                      float *__restrict__ out, int64_t out_bs,
                                                      unsigned *__restrict__ out_range) {
    constexpr int TW = W / 2;                 // tiles per row
    constexpr int ROWS = 2 * R + 2;           // input rows of a strip
    constexpr int PITCH = W + 4;              // input row pitch in LDS: col -1 at index 1, even, >= W + 3
    constexpr int IN_STAGE = KC * ROWS * PITCH;
    constexpr int XB = 2 * COUT * 32;         // floats one wave writes to the exchange buffer
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_u = sm;                          // [2][16][KC][32]
    float *s_in = sm + 2 * U_STAGE;           // [2][KC][ROWS][PITCH]
    float *s_x = sm;                          // [4 waves][2][32 oc][32 tiles], after the K loop

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, n = lane & 31;
    const int strip = blockIdx.x;
    const int64_t img = blockIdx.y;
    const int oy0 = strip * 2 * R;                           // first output row of the strip
    const float *xi = x + img * CIN * (int64_t)(W * W);

    // ---- staging assignments ------------------------------------------------------------------------
    // weights: stage block is 4096 contiguous floats -> 4 float4 per thread
    // input: KC * ROWS rows of W floats as float2 pairs; thread t takes items t, t + 256, ...
    constexpr int IN_F2 = KC * ROWS * (W / 2);
    constexpr int IN_PER = (IN_F2 + 255) / 256;
    f32x4 ru[4];
    float2 rin[IN_PER];
    auto load = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ru[i] = *reinterpret_cast<const f32x4 *>(u + (int64_t)ch * U_STAGE + 4 * (threadIdx.x + 256 * i));
#pragma unroll
        for (int i = 0; i < IN_PER; ++i) {
            const int it = threadIdx.x + 256 * i;
            const int c = it / (ROWS * (W / 2)), r = (it / (W / 2)) % ROWS, q = it % (W / 2);
            const int iy = oy0 - 1 + r;
            rin[i] = make_float2(0.f, 0.f);
            if (it < IN_F2 && iy >= 0 && iy < W)
                rin[i] = *reinterpret_cast<const float2 *>(xi + ((int64_t)(ch * KC + c) * W + iy) * W + 2 * q);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(s_u + buf * U_STAGE + 4 * (threadIdx.x + 256 * i)) = ru[i];
#pragma unroll
        for (int i = 0; i < IN_PER; ++i) {
            const int it = threadIdx.x + 256 * i;
            const int c = it / (ROWS * (W / 2)), r = (it / (W / 2)) % ROWS, q = it % (W / 2);
            if (it < IN_F2) {
                float *d = s_in + buf * IN_STAGE + (c * ROWS + r) * PITCH + 1 + 2 * q;     // col x at index x + 1
                d[0] = rin[i].x; d[1] = rin[i].y;
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

    // ---- this lane's tile and this wave's pair of input rows --------------------------------------------
    const int tile = n < R * TW ? n : R * TW - 1;            // idle lanes shadow the last tile (never stored)
    const int tr = tile / TW, tc = tile % TW;
    // T_i = s_a * d[ra] + s_b * d[rb]:  i=0: d0 - d2,  i=1: d1 + d2,  i=2: d2 - d1,  i=3: d1 - d3
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const int in_a = (2 * tr + ra) * PITCH + 2 * tc;         // + channel * ROWS * PITCH; cols 2tc-1 .. 2tc+2 at +0..+3
    const int in_b = (2 * tr + rb) * PITCH + 2 * tc;
    const int u_off = wave * 4 * KC * COUT + n;              // + (j * KC + c) * 32

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    load(0);
    store(0);
    for (int ch = 0; ch < NCH; ++ch) {
        const int cur = ch & 1;
        __syncthreads();                                   // stage ch visible; buffer cur ^ 1 free
        const int nxt = ch + 1 < NCH ? ch + 1 : ch;         // branch-free: the last stage re-loads itself
        load(nxt);
        __builtin_amdgcn_sched_barrier(0);
        const float *su = s_u + cur * U_STAGE + u_off;
        const float *si = s_in + cur * IN_STAGE;
        // operands of step s (channels 2s + half): two input rows x 4 columns, four weights.  Step s + 1 is
        // read while the MFMAs of step s run (hipcc otherwise waits for every read right before its use).
        float2 da[2][2], db[2][2];
        float uw[2][4];
#define MIRX_W_READ(S, SET)                                                                    \
    {                                                                                          \
        const int c_ = 2 * (S) + half;                                                         \
        da[SET][0] = *reinterpret_cast<const float2 *>(si + c_ * ROWS * PITCH + in_a);         \
        da[SET][1] = *reinterpret_cast<const float2 *>(si + c_ * ROWS * PITCH + in_a + 2);     \
        db[SET][0] = *reinterpret_cast<const float2 *>(si + c_ * ROWS * PITCH + in_b);         \
        db[SET][1] = *reinterpret_cast<const float2 *>(si + c_ * ROWS * PITCH + in_b + 2);     \
        uw[SET][0] = su[(0 * KC + c_) * COUT];                                                 \
        uw[SET][1] = su[(1 * KC + c_) * COUT];                                                 \
        uw[SET][2] = su[(2 * KC + c_) * COUT];                                                 \
        uw[SET][3] = su[(3 * KC + c_) * COUT];                                                 \
    }
#define MIRX_W_MFMA(SET)                                                                       \
    {                                                                                          \
        const float t0 = fmaf(sb, db[SET][0].x, da[SET][0].x), t1 = fmaf(sb, db[SET][0].y, da[SET][0].y); \
        const float t2 = fmaf(sb, db[SET][1].x, da[SET][1].x), t3 = fmaf(sb, db[SET][1].y, da[SET][1].y); \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(uw[SET][0], t0 - t2, acc[0], 0, 0, 0);   \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(uw[SET][1], t1 + t2, acc[1], 0, 0, 0);   \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(uw[SET][2], t2 - t1, acc[2], 0, 0, 0);   \
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(uw[SET][3], t1 - t3, acc[3], 0, 0, 0);   \
    }
        MIRX_W_READ(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        MIRX_W_READ(1, 1)
        MIRX_W_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        MIRX_W_READ(2, 0)
        MIRX_W_MFMA(1)
        __builtin_amdgcn_sched_barrier(0);
        MIRX_W_READ(3, 1)
        MIRX_W_MFMA(0)
        __builtin_amdgcn_sched_barrier(0);
        MIRX_W_MFMA(1)
#undef MIRX_W_READ
#undef MIRX_W_MFMA
        store(cur ^ 1);
    }

    // ---- output transform ---------------------------------------------------------------------------------
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
    float vmax = 0.f;                                       // largest |output| (range slots, mirx_common.h)
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
            const float2 v0 = make_float2(p[0][0] + p[1][0] + p[2][0], p[0][1] + p[1][1] + p[2][1]);
            const float2 v1 = make_float2(p[1][0] - p[2][0] - p[3][0], p[1][1] - p[2][1] - p[3][1]);
            *reinterpret_cast<float2 *>(o) = v0;
            *reinterpret_cast<float2 *>(o + W) = v1;
            vmax = range_max(range_max(range_max(range_max(vmax, v0.x), v0.y), v1.x), v1.y);
        }
    }
    (void)vmax; (void)out_range;        // range publishing lives on the two-fp16-term path only (per image, mirx_common.h)
}

// The 7x7 maps (last dense block): a map is 4 x 4 tiles (the 8th row / column of outputs is computed and
// dropped), so one workgroup takes TWO images = 32 tiles = all 32 MFMA columns.  Same algorithm and weight
// layout as k_conv3x3_wino; the input is staged cell by cell (rows of 7 floats have no vector alignment)
// into zero-initialised 10 x 12 planes whose border cells are never written.
__global__ __launch_bounds__(256, 3) void k_conv3x3_wino7(const float *__restrict__ x, const float *__restrict__ u,
                                                          float *__restrict__ out, int64_t out_bs, int64_t n_img,
                                                          unsigned *__restrict__ out_range) {
    constexpr int W = 7, ROWS = 10, PITCH = 12;
    constexpr int PLANE = ROWS * PITCH;                  // one channel of one image
    constexpr int IN_STAGE = 2 * KC * PLANE;
    constexpr int XB = 2 * COUT * 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *s_u = sm;                                     // [2][16][KC][32]
    float *s_in = sm + 2 * U_STAGE;                      // [2][2 images][KC][ROWS][PITCH]
    float *s_x = sm;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, n = lane & 31;
    const int64_t img0 = (int64_t)blockIdx.x * 2;

    // staging: weights 4 float4 per thread; input cells (image, channel, y, x): 2*8*49 = 784 -> 4 per thread
    constexpr int CELLS = 2 * KC * W * W;
    constexpr int IN_PER = (CELLS + 255) / 256;
    f32x4 ru[4];
    float rin[IN_PER];
    auto load = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ru[i] = *reinterpret_cast<const f32x4 *>(u + (int64_t)ch * U_STAGE + 4 * (threadIdx.x + 256 * i));
#pragma unroll
        for (int i = 0; i < IN_PER; ++i) {
            const int it = threadIdx.x + 256 * i;
            const int im = it / (KC * W * W), c = (it / (W * W)) % KC, cell = it % (W * W);
            rin[i] = 0.f;
            if (it < CELLS && img0 + im < n_img)
                rin[i] = x[((img0 + im) * CIN + ch * KC + c) * (int64_t)(W * W) + cell];
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(s_u + buf * U_STAGE + 4 * (threadIdx.x + 256 * i)) = ru[i];
#pragma unroll
        for (int i = 0; i < IN_PER; ++i) {
            const int it = threadIdx.x + 256 * i;
            const int im = it / (KC * W * W), c = (it / (W * W)) % KC, cell = it % (W * W);
            if (it < CELLS)
                s_in[buf * IN_STAGE + (im * KC + c) * PLANE + (cell / W + 1) * PITCH + cell % W + 1] = rin[i];
        }
    };
    for (int i = threadIdx.x; i < 2 * IN_STAGE; i += 256) s_in[i] = 0.f;     // borders stay zero for good
    __syncthreads();

    const int im = n >> 4, tile = n & 15;
    const int tr = tile >> 2, tc = tile & 3;
    const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rb = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sb = wave == 1 ? 1.f : -1.f;
    const int in_a = im * KC * PLANE + (2 * tr + ra) * PITCH + 2 * tc;
    const int in_b = im * KC * PLANE + (2 * tr + rb) * PITCH + 2 * tc;
    const int u_off = wave * 4 * KC * COUT + n;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    load(0);
    store(0);
    for (int ch = 0; ch < NCH; ++ch) {
        const int cur = ch & 1;
        __syncthreads();
        const int nxt = ch + 1 < NCH ? ch + 1 : ch;
        load(nxt);
        __builtin_amdgcn_sched_barrier(0);
        const float *su = s_u + cur * U_STAGE + u_off;
        const float *si = s_in + cur * IN_STAGE;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const int c = 2 * s + half;
            const float2 a0 = *reinterpret_cast<const float2 *>(si + c * PLANE + in_a);
            const float2 a1 = *reinterpret_cast<const float2 *>(si + c * PLANE + in_a + 2);
            const float2 b0 = *reinterpret_cast<const float2 *>(si + c * PLANE + in_b);
            const float2 b1 = *reinterpret_cast<const float2 *>(si + c * PLANE + in_b + 2);
            const float t0 = fmaf(sb, b0.x, a0.x), t1 = fmaf(sb, b0.y, a0.y);
            const float t2 = fmaf(sb, b1.x, a1.x), t3 = fmaf(sb, b1.y, a1.y);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(su[(0 * KC + c) * COUT], t0 - t2, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(su[(1 * KC + c) * COUT], t1 + t2, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(su[(2 * KC + c) * COUT], t2 - t1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(su[(3 * KC + c) * COUT], t1 - t3, acc[3], 0, 0, 0);
        }
        store(cur ^ 1);
    }

    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = 8 * (r >> 2) + (r & 3) + 4 * half;
        s_x[wave * XB + oc * 32 + n] = acc[0][r] + acc[1][r] + acc[2][r];
        s_x[wave * XB + COUT * 32 + oc * 32 + n] = acc[1][r] - acc[2][r] - acc[3][r];
    }
    __syncthreads();
    float vmax = 0.f;
    for (int it = threadIdx.x; it < COUT * 32; it += 256) {
        const int oc = it >> 5, t = it & 31;
        const int oim = t >> 4, otr = (t & 15) >> 2, otc = t & 3;
        if (img0 + oim >= n_img) continue;
        float p[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i][0] = s_x[i * XB + oc * 32 + t];
            p[i][1] = s_x[i * XB + COUT * 32 + oc * 32 + t];
        }
        float *o = out + (img0 + oim) * out_bs + ((int64_t)oc * W + 2 * otr) * W + 2 * otc;
        const bool col1 = 2 * otc + 1 < W, row1 = 2 * otr + 1 < W;
        const float v00 = p[0][0] + p[1][0] + p[2][0], v01 = p[0][1] + p[1][1] + p[2][1];
        const float v10 = p[1][0] - p[2][0] - p[3][0], v11 = p[1][1] - p[2][1] - p[3][1];
        o[0] = v00;
        vmax = range_max(vmax, v00);
        if (col1) { o[1] = v01; vmax = range_max(vmax, v01); }
        if (row1) {
            o[W] = v10;
            vmax = range_max(vmax, v10);
            if (col1) { o[W + 1] = v11; vmax = range_max(vmax, v11); }
        }
    }
    (void)vmax; (void)out_range;        // range publishing lives on the two-fp16-term path only (per image, mirx_common.h)
}

template <int W, int R>
hipError_t launch_w(const float *x, const float *u, int64_t n, float *out, int64_t out_bs, unsigned *out_range,
                    hipStream_t st) {
    constexpr int ROWS = 2 * R + 2, PITCH = W + 4;
    const size_t stage = (size_t)(2 * U_STAGE + 2 * KC * ROWS * PITCH) * sizeof(float);
    const size_t xch = (size_t)4 * 2 * COUT * 32 * sizeof(float);
    const size_t lds = stage > xch ? stage : xch;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_wino<W, R>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_conv3x3_wino<W, R>), dim3((W / 2 + R - 1) / R, (unsigned)n), dim3(256), lds, st, x, u, out,
                       out_bs, out_range);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_conv3x3_wino(const float *x, const float *u, int64_t n, int side, float *out, int64_t out_bs,
                               float *out_range_f, hipStream_t st) {
    unsigned *out_range = reinterpret_cast<unsigned *>(out_range_f);
    if (n <= 0) return hipSuccess;
    if (n > 65535) return hipErrorInvalidValue;
    if (side == 56) return launch_w<56, 1>(x, u, n, out, out_bs, out_range, st);
    if (side == 28) return launch_w<28, 2>(x, u, n, out, out_bs, out_range, st);
    if (side == 14) return launch_w<14, 4>(x, u, n, out, out_bs, out_range, st);
    if (side == 7) {
        const size_t lds = (size_t)(2 * U_STAGE + 2 * 2 * KC * 10 * 12) * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_wino7),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv3x3_wino7, dim3((unsigned)((n + 1) / 2)), dim3(256), lds, st, x, u, out, out_bs, n, out_range);
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace mirx
