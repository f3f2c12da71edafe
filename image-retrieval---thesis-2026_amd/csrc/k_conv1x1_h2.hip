// k_conv1x1_h2.hip -- the dense-layer 1x1 convolution of k_conv1x1_s3.hip with every fp32 operand carried as TWO fp16
// terms (x = xh + xl to 22 bits) and THREE v_mfma_f32_32x32x16_f16 per product block (xl wh + xh wl + xh wh; the
// dropped xl wl is 2^-22 of the product) instead of three bf16 terms and six MFMAs -- the scheme of k_linear_h2.hip.
// Half the matrix-pipe time and 4 instead of 6 LDS bytes per element; the layer then runs against its HBM stream.
//
// fp16 has 5 exponent bits, so the kernel needs the RANGE of its input.  DenseNet activations have no a-priori bound,
// so the range travels with the data, PER IMAGE (mirx_common.h): every kernel that writes into a dense block's buffer
// also folds the largest |value| it wrote for image b into the buffer's range row[b] (unsigned atomic max -- the bit
// patterns of non-negative floats order like the floats).  This kernel reads row[b] of its input buffer, and
//     bound_b = in_ks * row[b] + in_kb      (in_ks = max |BN scale|, in_kb = max |BN shift| of the prologue; 1, 0 without)
// is an upper bound of every value of image b it stages; x_scale_b = the power of two with bound_b * x_scale_b in [2^14,
// 2^15).  A pixel tile may straddle images: the scale belongs to the staging thread's pixel and, in the epilogue, to the
// lane's pixel, so an image's result never depends on its batch mates.  A value
// below 2^-18 of the bound keeps an ABSOLUTE error of 2^-40 of the bound (fp16 subnormal low term), everything else 22
// bits.  The weights arrive pre-split with one power-of-two scale PER OUTPUT CHANNEL (largest |w| of the row in
// [2^13, 2^14)); `oscale[o]` = 1 / that scale is applied to the accumulator together with 1 / x_scale -- all exact.
// The epilogue folds the largest |output| of image b into `out_amax[b]` the same way.
//
// Contract otherwise as mirx_conv1x1_bn_relu_split3: y = act_out(W * act_in(x) + bias), NCHW, tile 128 output
// channels x 128 (small launches) or 2 x 128 pixels, 16-channel stages, double-buffered LDS (32 / 48 KiB), weights by LDS
// DMA, activations register-prefetched.  w2 = [cout / 128][cin / 16][2 terms][128 out][16 in] fp16.
#include <atomic>

#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// Diagnostic build -DMIRX_C1H2_STAMPS: s_memtime at the phase boundaries of every stage, wave 0 of each workgroup, into
// g_c1_stamps (read back by tools/bench_conv1x1.py --stamps through mirx_debug_c1_stamps); never in the shipped library.
#ifdef MIRX_C1H2_STAMPS
__device__ unsigned long long g_c1_stamps[8192 * 8];      // [7] = (store total << 32) | (its VALU part)
#define C1_STAMP(ACC)                                                 \
    {                                                                 \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ACC += now_ - st_t;                                           \
        st_t = now_;                                                  \
    }
#else
#define C1_STAMP(ACC)
#endif

constexpr int CM = 128;            // output channels per workgroup
constexpr int CP = 128;            // pixels per workgroup
constexpr int KC = 16;             // channels per stage
constexpr int PLANE_A = CM * KC * 2;   // bytes of one term of the weight stage (4 KiB)
constexpr int PLANE_B = CP * KC * 2;
#ifndef MIRX_C1H2_MIN_WG
#define MIRX_C1H2_MIN_WG 256        // two-tile workgroups only when the launch has at least this many of them (one per CU; 512: -0.3 %)
#endif
#ifndef MIRX_C1H2_NPT
#define MIRX_C1H2_NPT 2             // pixel tiles per workgroup on large launches (1: the A/B arm)
#endif

// NPT = pixel tiles (of 128) per workgroup.  NPT = 2: both tiles run against ONE staged copy of the weights -- the weight
// stage (8 KiB per 16 channels) is as many bytes as a pixel tile's activations, so one tile per workgroup pulls twice the
// layer's bytes into the CU; two tiles halve the L2 -> LDS weight stream and the weight-fragment reads per MFMA, at 128
// accumulator registers per lane (two workgroups per CU instead of three).
// TABLED: the images a workgroup's pixels can span fit the 8-row oscale table (hw >= 37 with two pixel tiles) -- decided by the
// launcher, so that the epilogue holds ONE form of the multiply (a run-time choice made the compiler emit both for every value)
// Lanes 2 j and 2 j + 1 exchange half of what they hold: a = (even lane) its own p, (odd lane) the even neighbour's q;
// b = (even lane) the odd neighbour's p, (odd lane) its own q.  One v_cndmask_b32_dpp per word (D = vcc ? src1 : dpp(src0));
// written out because the compiler turns the same thing in C into a select, a v_mov_b32_dpp and two more selects per word.
// The s_nop covers the two wait states a DPP read needs behind the VALU write of its source (the hazard recogniser does not
// look into inline assembly).
__device__ __forceinline__ void pair_exchange(const u32x4 &p, const u32x4 &q, u32x4 &a, u32x4 &b) {
    unsigned a0, a1, a2, a3, b0, b1, b2, b3;
    asm volatile("s_nop 1\n"
                 "s_mov_b64 vcc, %[me]\n"
                 "v_cndmask_b32_dpp %[a0], %[q0], %[p0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %[a1], %[q1], %[p1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %[a2], %[q2], %[p2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %[a3], %[q3], %[p3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_mov_b64 vcc, %[mo]\n"
                 "v_cndmask_b32_dpp %[b0], %[p0], %[q0], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %[b1], %[p1], %[q1], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %[b2], %[p2], %[q2], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "v_cndmask_b32_dpp %[b3], %[p3], %[q3], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 : [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1),
                   [b2] "=&v"(b2), [b3] "=&v"(b3)
                 : [p0] "v"(p[0]), [p1] "v"(p[1]), [p2] "v"(p[2]), [p3] "v"(p[3]), [q0] "v"(q[0]), [q1] "v"(q[1]),
                   [q2] "v"(q[2]), [q3] "v"(q[3]), [me] "s"(0x5555555555555555ull), [mo] "s"(0xaaaaaaaaaaaaaaaaull)
                 : "vcc");
    a = u32x4{a0, a1, a2, a3};
    b = u32x4{b0, b1, b2, b3};
}

// (An eight-wave arm -- ONE workgroup per CU, waves 0-3 and 4-7 each on two pixel tiles against one staged copy of the weights,
// half the L2 -> LDS weight stream per pixel -- was built in round 3, parity-green and SLOWER (89.8 / 90.1 vs 88.5 / 88.5 ms
// per forward): eight waves meeting at one barrier lose more than the shared operand saves.  Removed in round 4)
template <bool PROLOGUE, bool RELU_OUT, bool YTERMS, int NPT, bool TABLED>
__global__ __launch_bounds__(256, NPT == 2 ? 2 : 3) void k_conv1x1_h2(const float *__restrict__ x, int64_t xbs, int cin,
                                                       const float *__restrict__ scale,
                                                       const float *__restrict__ shift,
                                                       const uint16_t *__restrict__ w2,
                                                       const float *__restrict__ oscale,
                                                       const float *__restrict__ bias, int64_t n, int hw, int cout,
                                                       float *__restrict__ y, int64_t ybs,
                                                       const float *__restrict__ in_amax, float in_ks, float in_kb,
                                                       unsigned *__restrict__ out_amax, float y_ks, float y_kb,
                                                       float *__restrict__ y_inv_out, int64_t xps, int64_t yps) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    constexpr int KIMG = 8;                        // images a workgroup's pixels may span with a table row each (else: multiply)
    __shared__ float sBias[CM], sOsc[KIMG][CM];
    // YTERMS with a table: the image's output scale 2^t is folded into BOTH constants of the epilogue's multiply-add (a power of
    // two: fma(a, s 2^t, b 2^t) = 2^t fma(a, s, b) exactly), which takes one multiply per value out of the epilogue
    constexpr bool YFOLD = YTERMS && TABLED;
    __shared__ float sBiasY[YFOLD ? KIMG : 1][CM];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tid = threadIdx.x;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t total = n * (int64_t)hw;
    constexpr int STAGE_N = 2 * PLANE_A + NPT * 2 * PLANE_B;       // bytes of one LDS stage
    const int64_t p0 = (int64_t)blockIdx.x * (CP * NPT);
    const int co0 = blockIdx.y * CM;
    const int nk = cin / KC;

    // ---- the range of an image -> its x_scale (a power of two) ---------------------------------------------------------
    // YTERMS: the output is written already split into its two fp16 terms, scaled by 2^t with
    //     |y| <= y_ks * x_bound + y_kb      (y_ks = max_o sum_c |W[o, c]|, y_kb = max |bias|: a provable bound known BEFORE the
    // kernel runs, unlike the true maximum) * 2^t in [2^14, 2^15); 2^-t goes to y_inv_out[image] for the consumer.
    auto image_bound = [&](int64_t img) { return fmaf(in_ks, in_amax ? in_amax[img] : 0.f, in_kb); };

    // ---- staging assignments ------------------------------------------------------------------------
    // B: thread -> pixel (t & 127), channel group kg = t >> 7 (wave-uniform): channels 8 kg .. 8 kg + 7
    const int b_px = tid & 127;
    const int b_kg = wave >> 1;
    const int64_t in_hw = xps;                                        // channel-plane stride of x (>= hw)
    const float *xsrc[NPT];
    float x_scale[NPT];                                               // of this thread's pixel's image (set behind the first loads)
    unsigned st_img[NPT];
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        // pixel indices fit 32 bits (checked by the launcher): one 32-bit division per pixel instead of 64-bit ones
        const unsigned pp = (unsigned)(p0 + u * CP + b_px);
        const bool live = pp < (unsigned)total;
        const unsigned pimg = (live ? pp : (unsigned)total - 1u) / (unsigned)hw;     // a dead pixel: a valid image, data of image 0
        const int64_t b_off = live ? (int64_t)pimg * xbs + (pp - pimg * (unsigned)hw) : 0;
        st_img[u] = pimg;
        xsrc[u] = x + b_off + (int64_t)(8 * b_kg) * in_hw;
    }
    // YTERMS: within every block of 32 pixels, pixel p is staged into LDS row ((p & 15) << 1) | (p >> 4), so that accumulator
    // column l (which reads row l) is pixel 16 (l & 1) + (l >> 1): the lanes 2 j and 2 j + 1 hold the pixels j and 16 + j, and
    // after a swap between those two neighbours every store instruction of the epilogue writes whole 32-byte pixel records, 512
    // contiguous bytes per half-wave (with the pixels in lane order each instruction wrote 16 bytes of every 32: -5 % on the
    // whole forward).  The ds_write below then lands two rows of a bank group per 16 lanes.
#if defined(MIRX_C1H2_EXP_SKIP) && (MIRX_C1H2_EXP_SKIP & 32)         // diagnostic (pixels land in the wrong records): the rows in pixel order
    const int b_row = b_px;
#else
    const int b_row = YTERMS ? ((b_px & ~31) | ((b_px & 15) << 1) | ((b_px >> 4) & 1)) : b_px;
#endif
    const int b_lds = 2 * PLANE_A + b_row * 32 + ((b_kg ^ ((b_row >> 3) & 1)) << 4);   // + tile * 2 PLANE_B + term * PLANE_B
    const int ep_px = YTERMS ? 16 * (lane & 1) + ((lane & 31) >> 1) : (lane & 31);     // this lane's pixel within its 32-block
    // A: the 8 KiB weight stage goes global -> LDS by DMA (buffer_load ... lds: lane l of a wave writes 16 B at
    // piece base + 16 l), two 1-KiB pieces per wave.  Piece p, lane l is LDS (term p / 4, row 32 (p & 3) +
    // l / 2, slot l & 1), which holds source chunk (l & 1) ^ ((row >> 3) & 1) -- the same for every piece.
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(w2 + ((int64_t)blockIdx.y * nk) * (2 * CM * KC)), 0, nk * (2 * CM * KC * 2), 0x00020000);
    const int w_voff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    [[maybe_unused]] auto dma_w1 = [&](int kt, int buf, int i) {
        const int piece = wave + 4 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, LDS_PTR(sm + buf * STAGE_N + piece * 1024), 16, w_voff,
                                                 kt * (2 * CM * KC * 2) + piece * 1024, 0, 0);
    };
    [[maybe_unused]] auto dma_w = [&](int kt, int buf) {
        dma_w1(kt, buf, 0);
        dma_w1(kt, buf, 1);
    };

    // (An arm with the weights through registers -- plain loads + ds_write_b128, so that hipcc's counted s_waitcnt keeps two
    // stages of loads in flight across the barrier instead of the vmcnt(0) the DMA needs -- measured 2-4 % faster on isolated
    // 28 / 14 layers, 6 % slower on the 7 maps and 1.2 % SLOWER on the whole forward: the layers are not latency-bound.  A
    // build with one MFMA per product instead of three (MIRX_C1H2_EXP_ONE_MFMA) gains 5-7 %: not matrix-bound either.)
    // TWO register sets: the activation loads of stage kt + 2 are issued while stage kt computes and stage kt + 1
    // waits in the other set.  With one set a workgroup has 8 KiB of HBM reads in flight (32 KiB per CU at four
    // workgroups): by Little's law that caps the layer near 4 TB/s, which is where the one-set kernel sat.
    constexpr int NR = 8;                               // raw values per thread and stage
    float ra[NPT][NR], rb[NPT][NR], sca[8], sha[8], scb[8], shb[8];
    auto load = [&](int kt, float (&r)[NPT][NR], float (&rsc)[8], float (&rsh)[8]) {
        // (`nt` loads here, to keep the weights L2-resident, measured 5 % SLOWER: 13.35 -> 14.0 ms per 1024 images)
#pragma unroll
        for (int u = 0; u < NPT; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // (agent-scope `sc1` loads, which pass the vector L1 by so that it holds only weight pieces, measured slower: DESIGN 12)
                r[u][j] = xsrc[u][((int64_t)kt * KC + j) * in_hw];
            }
        if (PROLOGUE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                rsc[j] = scale[kt * KC + 8 * b_kg + j];
                rsh[j] = shift[kt * KC + 8 * b_kg + j];
            }
        }
    };
#ifdef MIRX_C1H2_STAMPS
    unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_wait = 0, st_bar = 0, st_issue = 0, st_mfma = 0, st_store = 0, st_valu = 0;
    const unsigned long long st_begin = st_t, rt_begin = __builtin_amdgcn_s_memrealtime();
#endif
    auto store = [&](int buf, const float (&r)[NPT][NR], const float (&rsc)[8], const float (&rsh)[8]) {
        char *sb = sm + buf * STAGE_N;
#pragma unroll
        for (int u = 0; u < NPT; ++u) {
            // two fp16 terms of each (scaled) value, two values at a time (round to nearest even)
            u32x4 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x2 v;
                v[0] = r[u][2 * j];
                v[1] = r[u][2 * j + 1];
                if (PROLOGUE) {
                    v[0] = fmaxf(fmaf(v[0], rsc[2 * j], rsh[2 * j]), 0.f);
                    v[1] = fmaxf(fmaf(v[1], rsc[2 * j + 1], rsh[2 * j + 1]), 0.f);
                }
                unsigned th, tl;
#if defined(MIRX_C1H2_EXP_SPLIT) && MIRX_C1H2_EXP_SPLIT == 0        // diagnostic (wrong results): no BN / ReLU / split arithmetic at all
                th = __float_as_uint(r[u][2 * j]);
                tl = __float_as_uint(r[u][2 * j + 1]);
#elif defined(MIRX_C1H2_EXP_SPLIT) && MIRX_C1H2_EXP_SPLIT == 1      // diagnostic (wrong results): BN / ReLU / scale kept, the fp16 split dropped
                th = __float_as_uint(v[0] * x_scale[u]);
                tl = __float_as_uint(v[1] * x_scale[u]);
#else
                split2h_pair(v[0] * x_scale[u], v[1] * x_scale[u], th, tl);
#endif
                ph[j] = th;
                pl[j] = tl;
            }
#ifdef MIRX_C1H2_STAMPS
            if (u == NPT - 1) {                              // (diagnostic: everything of `store` but the last two LDS writes)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                C1_STAMP(st_valu)
            }
#endif
            *reinterpret_cast<u32x4 *>(sb + b_lds + u * 2 * PLANE_B) = ph;
            *reinterpret_cast<u32x4 *>(sb + b_lds + u * 2 * PLANE_B + PLANE_B) = pl;
        }
    };

    // ---- fragment addressing: lane -> row (lane & 31), K chunk (lane >> 5) -------------------------------
    const int kg = lane >> 5;
    constexpr int NN = 2 * NPT;                        // accumulator tiles along the pixels: ni = 2 * pixel tile + t
    int fa[2], fb[2];                                  // fb: + (ni >> 1) * 2 PLANE_B for the pixel tile
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra_ = wm * 64 + t * 32 + (lane & 31);
        fa[t] = ra_ * 32 + ((kg ^ ((ra_ >> 3) & 1)) << 4);
        const int rb_ = wn * 64 + t * 32 + (lane & 31);
        fb[t] = 2 * PLANE_A + rb_ * 32 + ((kg ^ ((rb_ >> 3) & 1)) << 4);
    }

    const unsigned img_first = (unsigned)(p0 < total ? p0 : total - 1) / (unsigned)hw;
    const unsigned p_last = (unsigned)(p0 + CP * NPT - 1 < total ? p0 + CP * NPT - 1 : total - 1);
    constexpr bool tabled = TABLED;

    f32x16 acc[2][NN];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // one stage: LDS buffer `cur` holds stage kt; `rnext` receives stage kt + 2; `rstore` holds stage kt + 1
    auto stage = [&](int kt, int cur, float (&rnext)[NPT][NR], float (&scn)[8], float (&shn)[8], const float (&rstore)[NPT][NR],
                     const float (&scs)[8], const float (&shs)[8]) {
        // stage kt visible: this wave's weight DMA of stage kt has landed.  vmcnt(0), NOT a counted wait: the two DMA
        // pieces are older than the 8 activation loads issued behind them, and `vmcnt(8)` was tried to keep those loads
        // in flight across the barrier -- it produced state-dependent results (embeddings off by 2e-5 once the caches
        // were warm: the LDS-DMA pieces were still landing when the count had already dropped to 8), i.e. LDS-DMA and
        // loads to registers must not be assumed to retire in one common order.  The loads of stage kt + 1 therefore
        // complete here too; they were issued a whole stage earlier (two register sets), which is what matters.
        C1_STAMP(st_store)
        // (A counted wait that keeps the activation loads in flight across the barrier -- safe once a sched_barrier fences the
        // DMA pieces ahead of the loads, tools/dma_order_probe.hip -- bought 2.7 % on the layers back to back and nothing on
        // the forward: DESIGN 12; the arm was removed in round 4.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C1_STAMP(st_wait)
        __syncthreads();
        C1_STAMP(st_bar)
#if defined(MIRX_C1H2_EXP_SKIP) && (MIRX_C1H2_EXP_SKIP & 1)          // diagnostic (wrong results): no weight DMA inside the K loop
        if (kt < 0)
#endif
        dma_w(kt + 1 < nk ? kt + 1 : kt, cur ^ 1);         // branch-free tails: re-load the last stage
        __builtin_amdgcn_sched_barrier(0);                 // (the DMA pieces stay older than the loads behind them)
        load(kt + 2 < nk ? kt + 2 : nk - 1, rnext, scn, shn);
        __builtin_amdgcn_sched_barrier(0);
        C1_STAMP(st_issue)
        const char *sb = sm + cur * STAGE_N;
        f16x8 a[2][2], b[NN][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[t][p] = *reinterpret_cast<const f16x8 *>(sb + fa[t] + p * PLANE_A);
#pragma unroll
        for (int ni = 0; ni < NN; ++ni)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                b[ni][p] = *reinterpret_cast<const f16x8 *>(sb + fb[ni & 1] + (ni >> 1) * 2 * PLANE_B + p * PLANE_B);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < NN; ++ni) {
                f32x16 c = acc[mi][ni];
                // smallest terms first
#ifndef MIRX_C1H2_EXP_ONE_MFMA          // diagnostic build (wrong results, timing only): one MFMA per product block
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][1], b[ni][0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], b[ni][1], c, 0, 0, 0);
#endif
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], b[ni][0], c, 0, 0, 0);
                acc[mi][ni] = c;
            }
        }
        C1_STAMP(st_mfma)
        store(cur ^ 1, rstore, scs, shs);                  // stage kt + 1 (loaded one stage ago)
    };
    dma_w(0, 0);
    load(0, ra, sca, sha);
    load(nk > 1 ? 1 : 0, rb, scb, shb);
    if (tid < CM) {
        const float bias_v = bias ? bias[co0 + tid] : 0.f;
        sBias[tid] = bias_v;
        const float osc = oscale[co0 + tid];
        sOsc[0][tid] = osc;
        if (tabled) {
            const unsigned nimg = p_last / (unsigned)hw - img_first + 1;
            for (unsigned k = 0; k < nimg; ++k) {
                float xs_, xi_;
                const float xb_ = image_bound(img_first + k);
                range_scales(xb_, xs_, xi_);
                float ys_ = 1.f, yi_;
                if (YFOLD) range_scales(fmaf(y_ks, xb_, y_kb), ys_, yi_);
                sOsc[k][tid] = osc * xi_ * ys_;                         // powers of two: exact
                if (YFOLD) sBiasY[k][tid] = bias_v * ys_;
            }
        }
    }
    // the ranges are read only NOW, behind the first two stages of loads: a range load in front of them held every
    // workgroup's first loads back by one memory round trip (+3 % on the layer)
#pragma unroll
    for (int u = 0; u < NPT; ++u) {
        float inv_;
        range_scales(image_bound(st_img[u]), x_scale[u], inv_);
    }
    // its image's 2^-s times oscale[channel] comes from a small LDS table, one row per image the workgroup's pixels span
    // (built below) -- one LDS read per output value as before ranges were per image, instead of an extra multiply per value
    float ep_xinv[NN], ep_ys[NN];
    int ep_k[NN];
#pragma unroll
    for (int ni = 0; ni < NN; ++ni) {
        unsigned pp = (unsigned)(p0 + (ni >> 1) * CP + wn * 64 + 32 * (ni & 1) + ep_px);
        if (pp >= (unsigned)total) pp = (unsigned)total - 1u;
        const unsigned pimg = pp / (unsigned)hw;
        const float xb = image_bound(pimg);
        float xs_, yi_;
        range_scales(xb, xs_, ep_xinv[ni]);
        ep_k[ni] = tabled ? (int)(pimg - img_first) : 0;
        if (tabled) ep_xinv[ni] = 1.f;                                  // folded into the table row
        ep_ys[ni] = 1.f;
        if (YTERMS) range_scales(fmaf(y_ks, xb, y_kb), ep_ys[ni], yi_);
    }

    store(0, ra, sca, sha);
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        stage(kt, 0, ra, sca, sha, rb, scb, shb);
        stage(kt + 1, 1, rb, scb, shb, ra, sca, sha);
    }
    if (kt < nk) stage(kt, 0, ra, sca, sha, rb, scb, shb);
#ifdef MIRX_C1H2_STAMPS
    if (threadIdx.x == 0) {
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();
        unsigned long long *o = g_c1_stamps + ((blockIdx.x + blockIdx.y * gridDim.x) & 8191) * 8;
        o[0] = now_ - st_begin;
        o[1] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        o[2] = nk;
        o[3] = st_wait;
        o[4] = st_bar;
        o[5] = st_issue;
        o[6] = st_mfma;
        o[7] = (st_store << 32) | (st_valu & 0xffffffffull);          // st_store excludes the part already counted in st_valu
    }
#endif

#if defined(MIRX_C1H2_EXP_SKIP) && (MIRX_C1H2_EXP_SKIP & 2)          // diagnostic (wrong results): nothing is written
    if (in_ks > -1e30f) return;
#endif
    // epilogue: register r of tile (mi, ni) = channel co0 + 64 wm + 32 mi + (r&3) + 8 (r>>2) + 4 (lane>>5),
    // pixel p0 + 128 (ni >> 1) + 64 wn + 32 (ni & 1) + (lane & 31)
    // per accumulator tile column: this lane's pixel, its image and that image's scales (oscale and 2^-s are powers of two,
    // their product is exact)
    if (YTERMS) {
        // y as the 3x3 conv wants it: [image][group g of 16 channels][term][pixel][16] fp16, where group g = 4 wm + 2 mi +
        // (lane >> 5) holds exactly the 16 channels this lane owns in accumulator tile mi (the consumer's weights are
        // permuted to the same channel order: mirx.model.YTERMS_CHANNEL_ORDER): a pixel's record of a (group, term) is the 32
        // bytes of ONE lane.  The lanes 2 j and 2 j + 1 (pixels j and 16 + j of the 32-block, see b_row) swap halves, so that a
        // store instruction writes whole records: the pair writes pixel j in instruction A and pixel 16 + j in instruction B, a
        // half-wave 512 contiguous bytes in each
        uint16_t *yt = reinterpret_cast<uint16_t *>(y);
        // the neighbour's value (lanes 2 j <-> 2 j + 1)
        auto nb = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); };   // quad_perm [1,0,3,2]
        const bool even = !(lane & 1);
#pragma unroll
        for (int ni = 0; ni < NN; ++ni) {
            const unsigned pp = (unsigned)(p0 + (ni >> 1) * CP + wn * 64 + 32 * (ni & 1) + ep_px);
            const bool live = pp < (unsigned)total;
            const unsigned ubimg = (live ? pp : 0u) / (unsigned)hw;
            const int64_t bimg = ubimg;
            const unsigned off = pp - ubimg * (unsigned)hw;
            if (live && off == 0 && wm == 0 && lane < 32) {                    // one writer per image: the lane of its pixel 0
                float ys_, y_inv;
                range_scales(fmaf(y_ks, image_bound(bimg), y_kb), ys_, y_inv);
                y_inv_out[bimg] = y_inv;
            }
            // this pixel's record of group 0, term 0, in records of 32 bytes (fits 32 bits: the launcher checks n hw < 2^27)
            // instruction A writes the even lane's pixel, instruction B the odd lane's: both lanes of a pair need both
            const unsigned q_own = live ? ubimg * 16u * (unsigned)hw + off : 0xffffffffu;
            const unsigned q_nb = nb(q_own);
            const unsigned qa = even ? q_own : q_nb, qb = even ? q_nb : q_own;
            [[maybe_unused]] const float x_inv = ep_xinv[ni], y_scale = ep_ys[ni];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int g = 4 * wm + 2 * mi + (lane >> 5);
                u32x4 h0, h1, l0, l1;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x2 v;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * j + e;
                        const int ch = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#if defined(MIRX_C1H2_EXP_SKIP) && (MIRX_C1H2_EXP_SKIP & 8)          // diagnostic: the epilogue's stores without its arithmetic
                        v[e] = acc[mi][ni][r];
                        (void)ch;
#else
                        float t = fmaf(acc[mi][ni][r], TABLED ? sOsc[ep_k[ni]][ch] : sOsc[0][ch] * x_inv,
                                       YFOLD ? sBiasY[ep_k[ni]][ch] : sBias[ch]);
                        // ReLU on the bits: a negative float is a negative integer (one v_max_i32 instead of compare + select
                        // and their VCC wait states; -0 becomes +0, a NaN keeps its bits unless its sign bit is set)
                        t = __int_as_float(max(__float_as_int(t), 0));
                        v[e] = YFOLD ? t : t * y_scale;
#endif
                    }
                    unsigned hh, ll;
#if defined(MIRX_C1H2_EXP_SKIP) && (MIRX_C1H2_EXP_SKIP & 8)
                    hh = __float_as_uint(v[0]);
                    ll = __float_as_uint(v[1]);
#else
                    split2h_pair(v[0], v[1], hh, ll);
#endif
                    if (j < 4) { h0[j] = hh; l0[j] = ll; }
                    else { h1[j - 4] = hh; l1[j - 4] = ll; }
                }
#if defined(MIRX_C1H2_EXP_SKIP) && (MIRX_C1H2_EXP_SKIP & 4)          // diagnostic: (a quarter of) the epilogue's arithmetic without its stores
                if ((h0[0] ^ h1[1] ^ l0[2] ^ l1[3]) != 0x5a5a1234u) continue;
#endif
                // even lane: channels 0-7 of its own pixel (A) and of its neighbour's (B); odd lane: channels 8-15 of both
                u32x4 ah, al, bh, bl;
                pair_exchange(h0, h1, ah, bh);
                pair_exchange(l0, l1, al, bl);
                const int64_t gofs = (int64_t)(2 * g) * hw;                    // records from group 0 to group g
                const int half = (lane & 1) * 8;
                // (non-temporal stores for these records, which no workgroup reads back, measured equal: DESIGN 12)
#define C1_ST(P, V) *reinterpret_cast<u32x4 *>(P) = V
                if (qa != 0xffffffffu) {
                    uint16_t *dst = yt + ((int64_t)qa + gofs) * 16 + half;
                    C1_ST(dst, ah);
                    C1_ST(dst + (int64_t)hw * 16, al);
                }
                if (qb != 0xffffffffu) {
                    uint16_t *dst = yt + ((int64_t)qb + gofs) * 16 + half;
                    C1_ST(dst, bh);
                    C1_ST(dst + (int64_t)hw * 16, bl);
                }
#undef C1_ST
            }
        }
        return;
    }
#pragma unroll
    for (int ni = 0; ni < NN; ++ni) {
        unsigned pp = (unsigned)(p0 + (ni >> 1) * CP + wn * 64 + 32 * (ni & 1) + (lane & 31));
        const bool live = pp < (unsigned)total;
        if (!live) pp = (unsigned)total - 1u;                 // carries nothing: a valid image index and vmax = 0
        const unsigned ubimg = pp / (unsigned)hw;
        const int64_t bimg = ubimg, off = pp - ubimg * (unsigned)hw;
        const float x_inv = ep_xinv[ni];
        float vmax = 0.f;
        float *yo = y + bimg * ybs + off;
        if (live) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = co0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = fmaf(acc[mi][ni][r], TABLED ? sOsc[ep_k[ni]][ch - co0] : sOsc[0][ch - co0] * x_inv, sBias[ch - co0]);
                    if (RELU_OUT) v = v < 0.f ? 0.f : v;          // keeps a NaN (fmaxf would turn it into 0)
                    vmax = range_max(vmax, v);
                    yo[(int64_t)ch * yps] = v;
                }
        }
        if (out_amax) range_publish_lanes(out_amax, (int)bimg, vmax, lane);
    }
}

}  // namespace

#ifdef MIRX_C1H2_STAMPS
extern "C" int mirx_debug_c1_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c1_stamps), sizeof(g_c1_stamps));
}
#endif

// launches of fewer than this many 128 x 128 workgroups take the one-wave-per-tile kernel (mirx_set_tuning
// MIRX_TUNE_CONV1X1_SMALL_MAX_WG; 0 switches the small kernel off -- the two kernels agree bit for bit, so this is speed only)
static std::atomic<int> g_small_max_wg{128};
void set_conv1x1_small_max_wg(int v) { g_small_max_wg.store(v < 0 ? 0 : v); }
static int conv1x1_small_max_wg() { return g_small_max_wg.load(std::memory_order_relaxed); }

hipError_t launch_conv1x1_h2(const float *x, int64_t xbs, int cin, const float *scale, const float *shift,
                             const uint16_t *w2, const float *oscale, const float *bias, int64_t n, int hw, int cout,
                             int relu_out, float *y, int64_t ybs, const float *in_amax, float in_ks, float in_kb,
                             float *out_amax, float y_ks, float y_kb, float *y_inv_out, int64_t xps, int64_t yps,
                             hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (cin % KC || cout % CM || !oscale) return hipErrorInvalidValue;
    const bool yterms = y_inv_out != nullptr;
    if (yterms && (cout != CM || !relu_out || !scale)) return hipErrorInvalidValue;
    if (!xps) xps = hw;                                // compact channel planes
    if (!yps) yps = hw;
    if (xps < hw || yps < hw) return hipErrorInvalidValue;
    // two pixel tiles per workgroup (one staged copy of the weights for both) when the launch still fills the chip
    const int64_t px = n * (int64_t)hw;
    if (px >= ((int64_t)1 << 31) - 2 * CP) return hipErrorInvalidValue;      // the kernel indexes pixels with 32 bits
    if (yterms && px >= ((int64_t)1 << 27)) return hipErrorInvalidValue;     // ... and the 32-byte records of y (16 per pixel) too
    // small launches (the reference's own batch sizes): one wave per 32 x 32 tile, no LDS (k_conv1x1_h2s.hip) -- when this
    // kernel's 128 x 128 tiles would leave most CUs without a workgroup
    if (px * (cout / CM) < (int64_t)CP * conv1x1_small_max_wg() && (cin <= 1024 || !scale))
        return launch_conv1x1_h2_small(x, xbs, cin, scale, shift, w2, oscale, bias, n, hw, cout, relu_out, y, ybs, in_amax,
                                       in_ks, in_kb, out_amax, y_ks, y_kb, y_inv_out, xps, yps, st);
    const int npt = px >= (int64_t)2 * CP * MIRX_C1H2_MIN_WG ? MIRX_C1H2_NPT : 1;
    const dim3 grid((unsigned)((px + CP * npt - 1) / (CP * npt)), (unsigned)(cout / CM));
    const size_t lds = 2 * (size_t)(2 * PLANE_A + npt * 2 * PLANE_B);
    unsigned *oa = reinterpret_cast<unsigned *>(out_amax);
    // the oscale table has 8 rows: a workgroup's CP * npt pixels span at most (CP * npt - 1) / hw + 2 images
    const bool tabled = (CP * npt - 1) / hw + 2 <= 8;
#define MIRX_H2K(P, R, T, N)                                                                               \
    {                                                                                                      \
        if (tabled) MIRX_H2KT(P, R, T, N, true) else MIRX_H2KT(P, R, T, N, false)                          \
    }
#define MIRX_H2KT(P, R, T, N, TB)                                                                          \
    {                                                                                                      \
        static unsigned long long attr_devs = 0;      /* per instantiation: the attribute call costs a host microsecond per launch */ \
        if (first_use_on_device(attr_devs)) {                                                                                   \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv1x1_h2<P, R, T, N, TB>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (2 * PLANE_A + N * 2 * PLANE_B)); \
            if (e != hipSuccess) return e;                                                                 \
        }                                                                                                  \
        hipLaunchKernelGGL((k_conv1x1_h2<P, R, T, N, TB>), grid, dim3(256), lds, st, x, xbs, cin, scale, shift, w2, oscale, bias, \
                           n, hw, cout, y, ybs, in_amax, in_ks, in_kb, oa, y_ks, y_kb, y_inv_out, xps, yps); \
    }
#define MIRX_H2C(P, R, T)                                                                                  \
    {                                                                                                      \
        if (npt == 2) MIRX_H2K(P, R, T, 2) else MIRX_H2K(P, R, T, 1)                                        \
    }
    if (yterms) {
        MIRX_H2C(true, true, true)
    } else if (scale) {
        if (relu_out) MIRX_H2C(true, true, false) else MIRX_H2C(true, false, false)
    } else {
        if (relu_out) MIRX_H2C(false, true, false) else MIRX_H2C(false, false, false)
    }
#undef MIRX_H2K
#undef MIRX_H2KT
#undef MIRX_H2C
    return hipGetLastError();
}

}  // namespace mirx
