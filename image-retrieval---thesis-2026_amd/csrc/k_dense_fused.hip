// k_dense_fused.hip -- a whole DenseNet dense layer of the SMALL maps (14 x 14 and 7 x 7: dense blocks 3 and 4, 40 of the 58
// layers) in ONE kernel:
//
//     norm1 + relu1 -> conv1 1x1 (cin -> 128, norm2 folded) -> relu2 -> conv2 3x3 (128 -> 32, pad 1) -> 32 new channels
//
// The 128-channel bottleneck of ONE image at 14 x 14 is 196 pixels x 128 channels x (2 fp16 terms) = 98 KiB: it fits the
// CU's LDS.  So a workgroup owns one "unit" = 196 pixels (one 14 x 14 image, or four 7 x 7 images), runs the 1x1 conv of
// k_conv1x1_h2 over the unit's channel prefix (the only HBM stream of the layer: cin x 196 x 4 B per unit), leaves the
// bottleneck in LDS already split into the fp16 terms and in the layout the 3x3 conv's B fragments want, and runs the tap
// loop of k_conv3x3_d2p straight from there.  Against the two-kernel layer this removes the bottleneck's round trip through
// HBM (2 x 100 KB per image and layer: 28 % of block 3's bytes), the 3x3 conv's DMA staging (its matrix pipe was 0.29 /
// 0.15 busy on these maps, waiting for that DMA), one launch per layer, and the image-straddling pixel tiles.
// The MFMA sequences (stage order, term order, tap order) are those of the two kernels, so the 32 output channels are
// BIT-IDENTICAL to mirx_conv1x1_bn_relu_split2h_terms + mirx_conv3x3_direct_terms_nchw (tests/test_model_gpu.py).
//
// Geometry: 512 threads = 8 waves, one workgroup per CU.  196 pixels = 7 MFMA column blocks of 32 (28 idle columns).
//   1x1 phase, wave-specialised.  Waves 0..3 CONSUME: wave w = the 32 bottleneck channels of row block w x all 7 column
//     blocks = 21 MFMAs per 16-channel stage, back to back; its A fragments (the weights of its 32 channels: 16 bytes per lane
//     and term) come STRAIGHT from global memory into a register ring -- every weight is used by exactly one wave, so LDS
//     would only add a round trip; its B fragments come from LDS, those of the next column group (and, across the stage
//     barrier, of the next stage) read while the current group multiplies; the three dependent MFMAs of a block alternate with
//     its group mates'.  Waves 4..7 PRODUCE: thread = (pixel pair, 8 channels): eight 8-byte activation loads per stage
//     through a register ring of D = 4 stages, norm1 + relu1 (scale / shift from LDS, fetched a stage ahead), split into fp16
//     terms, into a ring of THREE LDS stage buffers (pixel-major 32-byte rows, the swizzle of k_conv1x1_h2).  A SIMD then hosts
//     one matrix wave and one memory / VALU wave -- the pairing that overlaps.  History of this loop, cycles per stage for 672
//     cycles of MFMA: all waves alike, stage-wait-multiply in lockstep 1 920; wave roles with a load behind a branch 2 600
//     (hipcc's waits are per program point: a load on one side of a branch makes every later wait assume the worst, ring
//     depth 1); straight-line 1 950 with the producers the pole (1 430 of it norm1 + split + LDS, 540 the 18 loads).
//     Everything comes through registers, so every wait is a compiler-counted s_waitcnt on plain loads: no LDS-DMA in the mix,
//     whose completion is not ordered with register loads and forces vmcnt(0), i.e. a single stage in flight.
//   persistent: a workgroup walks units blockIdx.x, + gridDim.x, ..; both rings keep running across the unit boundary, so
//     the next unit's first stages load while this unit's 3x3 phase computes.
//   3x3 phase (all waves alike): waves 0..6 take one column block each x all 32 output channels; 8 stages (one 16-channel
//     group of the bottleneck each) x 9 taps x 3 MFMAs.  A lane's B fragment = its pixel shifted by the tap = ONE ds_read_b128
//     per term at a per-(lane, tap) offset computed once; a tap that leaves the image reads an all-zero pixel slot (no padded
//     copy: the padded image would not fit).  The 18 KiB of weights per stage go global -> registers -> LDS, double-buffered
//     in the space the 1x1 phase staged in; stages 0 and 1 are requested before the 1x1 epilogue runs.  The 32 x 196 outputs
//     leave through LDS as 16-byte stores of whole channel rows (16 four-byte stores per lane took 8 000 cycles to issue).
// Ranges are per image (mirx_common.h): the unit's images read their own row entries and fold their own output maxima in.
// Channel planes must be packed (plane stride = side^2: the stage offsets are instruction immediates).
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

constexpr int CM = 128;                        // bottleneck channels
constexpr int KC = 16;                         // channels per stage = one MFMA K
constexpr int NBLK = 7;                        // column blocks of 32 pixels
constexpr int NLIVE = 196;                     // pixels of a unit
constexpr int PLANE_A = CM * KC * 2;           // one term of a 1x1 weight stage in global memory: 4 KiB
constexpr int PLANE_B = NBLK * 32 * 32;        // one term of an activation stage: 7 KiB (32 B per pixel)
constexpr int STAGE1 = 2 * PLANE_B;            // 14 KiB
constexpr int NBUF = 3;                        // LDS stage buffers of the 1x1 phase
constexpr int YPIX = 200;                      // 196 pixels + the all-zero slot(s) behind them
constexpr int ZERO_PIX = NLIVE;
constexpr int YPLANE = YPIX * 32;              // one term of one 16-channel group of the bottleneck
constexpr int YGROUP = 2 * YPLANE;
constexpr int Y_BYTES = 8 * YGROUP;            // 100 KiB
constexpr int W3STAGE = 9 * 2 * 32 * KC * 2;   // one stage of 3x3 weights: 18 KiB
constexpr int STG_BYTES = NBUF * STAGE1;       // staging space: three 1x1 stages, then two 3x3 weight stages, then the output tile
static_assert(2 * W3STAGE <= STG_BYTES, "3x3 weight buffers live in the 1x1 staging space");
constexpr int OPITCH = 200;                    // floats per output channel row of the store tile (196 + pad)
static_assert(32 * OPITCH * 4 <= STG_BYTES, "the output tile lives in the staging space");
constexpr int BN_OFF = Y_BYTES + STG_BYTES;    // norm1 scale | shift, fp32 [2][cin] behind the staging space
constexpr int MAX_CIN = 1024;
constexpr int BNX_OFF = BN_OFF + 2 * MAX_CIN * 4;      // the same table times 2^s of the unit's image (14 x 14: one image per unit)
constexpr int LDS_BYTES = BNX_OFF + 2 * MAX_CIN * 4;  // 158 KiB
constexpr int D = 4;                           // consumer ring depth (A fragments, from L2)
constexpr int DP = 8;                          // producer ring depth: a stage is stored 6 steps after its loads were issued
constexpr int NST3 = CM / KC;                  // 8 stages of the 3x3 conv

#ifdef MIRX_DF_STAMPS          // diagnostic build: per-workgroup cycle sums of the phases (tools/df_stamps.py); never in the product
__device__ unsigned long long g_df_stamps[256 * 8];
__device__ unsigned long long g_df_roles[256 * 8];      // consumer busy, wait; producer busy, wait, store part (K loop, summed)
#define MIRX_DF_T(i) if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); stamp[i] += t_ - tprev; tprev = t_; }
// busy / barrier-wait split of one role's stage loop: lane 0 of wave 0 (consumer) and of wave 4 (producer)
#define MIRX_DF_SYNC() { const unsigned long long b0_ = __builtin_amdgcn_s_memtime(); role[0] += b0_ - rprev; __syncthreads(); rprev = __builtin_amdgcn_s_memtime(); role[1] += rprev - b0_; }
#define MIRX_DF_MID() { role2 += __builtin_amdgcn_s_memtime() - rprev; }
#else
#define MIRX_DF_T(i)
#define MIRX_DF_SYNC() __syncthreads();
#define MIRX_DF_MID()
#endif

#define MIRX_DF_RB(SB, NB, DST)                                                                    \
    {                                                                                              \
        DST[0] = *reinterpret_cast<const f16x8 *>(SB + fb0 + (NB) * 1024);                         \
        DST[1] = *reinterpret_cast<const f16x8 *>(SB + fb0 + (NB) * 1024 + PLANE_B);               \
    }
#ifndef MIRX_DF_EXP
#define MIRX_DF_EXP 0          // diagnostic builds (timing only, wrong results): 1 consumers skip the MFMAs, 2 producers skip norm1 + split
#endif
#define MIRX_DF_M3(I0, I1, BX, BY)                                                                 \
    acc[I0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], BX[0], acc[I0], 0, 0, 0);               \
    acc[I1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], BY[0], acc[I1], 0, 0, 0);               \
    acc[I0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], BX[1], acc[I0], 0, 0, 0);               \
    acc[I1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], BY[1], acc[I1], 0, 0, 0);               \
    acc[I0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], BX[0], acc[I0], 0, 0, 0);               \
    acc[I1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], BY[0], acc[I1], 0, 0, 0);

// W = map side: 14 (one image per unit) or 7 (four images per unit).  CONSUMER: the role of this wave in the 1x1 phase.  The
// two roles are two instantiations of the whole body, entered through one wave-uniform branch of the kernel: on a common
// path the compiler keeps the producers' ring AND the consumers' accumulators alive through each other's loops (435
// spilled registers in the first attempt); on disjoint paths each role pays for its own.
template <int W, bool CONSUMER>
__device__ __forceinline__ void dense_fused_body(char *sm, float *sBias, float *sOsc, float *sC3, float *__restrict__ buf,
                                                 int64_t bs, int cin, const float *__restrict__ scale,
                                                 const float *__restrict__ shift, const uint16_t *__restrict__ w2,
                                                 const float *__restrict__ oscale, const float *__restrict__ bias,
                                                 const uint16_t *__restrict__ w3, const float *__restrict__ c3osc,
                                                 int64_t n_img, unsigned *__restrict__ range_row, float in_ks, float in_kb,
                                                 float y_ks, float y_kb) {
    constexpr int HW = W * W;
    constexpr int PS4 = HW * 4;                // bytes of a channel plane (packed)
    constexpr int IPW = NLIVE / HW;            // images per unit: 1 or 4
    static_assert(IPW * HW == NLIVE, "a unit is 196 pixels");
    constexpr int NPAIR = (HW + 1) / 2;        // pixel pairs of an image (7 x 7: the last pair has one pixel)
    constexpr int NITEM = IPW * NPAIR;         // producer items per 8-channel half: 98 / 100
    constexpr bool consumer = CONSUMER;        // waves 0..3
    char *ylds = sm;
    char *stg = sm + Y_BYTES;
    float *s_bn = reinterpret_cast<float *>(sm + BN_OFF);               // [cin] scale, [cin] shift
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int half = lane >> 5, n = lane & 31;
    const int nk = cin / KC;
    const int64_t units = (n_img + IPW - 1) / IPW;

    if (tid < CM) {
        sBias[tid] = bias[tid];
        sOsc[tid] = oscale[tid];
    }
    if (tid < 32) sC3[tid] = c3osc[tid];
    for (int i = tid; i < cin; i += 512) {
        s_bn[i] = scale[i];
        s_bn[cin + i] = shift[i];
    }
    // the all-zero pixel slots of every (group, term) plane: written once, never overwritten
    for (int i = tid; i < 16 * (YPIX - NLIVE) * 2; i += 512) {
        const int plane = i / ((YPIX - NLIVE) * 2), r = i % ((YPIX - NLIVE) * 2);
        *reinterpret_cast<u32x4 *>(ylds + plane * YPLANE + NLIVE * 32 + r * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();                                                    // the tables above are read before the first stage barrier

    // ---- producer role: item = (8-channel half ph, image pj of the unit, pixel pair pq) -----------------------------------
    const int pt = tid & 255;
    const bool p_live = pt < 2 * NITEM;
    const int ph = p_live ? pt / NITEM : 0;
    const int pj = p_live ? (pt % NITEM) / NPAIR : 0, pq = p_live ? (pt % NITEM) % NPAIR : 0;
    const int px0 = pj * HW + 2 * pq;                                   // first pixel of the pair inside the unit
    const bool px1_ok = 2 * pq + 1 < HW;                                // 7 x 7: the 25th pair is a single pixel
    // LDS rows of the two pixels: 16-byte chunk `ph` of row px at slot ph ^ ((px >> 3) & 1)
    const int b_lds0 = px0 * 32 + ((ph ^ ((px0 >> 3) & 1)) << 4);
    const int b_lds1 = (px0 + 1) * 32 + ((ph ^ (((px0 + 1) >> 3) & 1)) << 4);
    const int x_voff = p_live ? (int)(((int64_t)pj * bs + 2 * pq + 8 * ph * HW) * 4) : 0;       // idle threads load (and drop) valid bytes

    // ---- consumer role: lane -> row n of row block `wave`, K chunk `half`; column block nb at fb0 + 1024 nb -----------------
    const int fb0 = n * 32 + ((half ^ ((n >> 3) & 1)) << 4);
    const int a_voff = ((wave & 3) * 32 + n) * 32 + half * 16;          // this lane's 16 bytes inside a term plane of w2

    // ---- 3x3 phase roles ------------------------------------------------------------------------------------------
    // wave w < 7 -> column block w; lane -> pixel p3; per tap the byte offset of the B fragment inside a term plane
    const int p3 = wave * 32 + n;
    const bool live3 = wave < NBLK && p3 < NLIVE;
    const int j3 = live3 ? p3 / HW : 0, q3 = live3 ? p3 % HW : 0;
    int toff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int rr = q3 / W + t / 3 - 1, cc = q3 % W + t % 3 - 1;
        const bool ok = live3 && rr >= 0 && rr < W && cc >= 0 && cc < W;
        const int pix = ok ? j3 * HW + rr * W + cc : ZERO_PIX;
        toff[t] = pix * 32 + ((half ^ ((pix >> 3) & 1)) << 4);
    }
    const int a3_off = n * 32 + ((half ^ ((n >> 3) & 1)) << 4);        // A fragment inside a (tap, term) piece
    // 3x3 weights: chunk q = tid + 512 i (q < 1152) -> piece q >> 6 = wave + 8 i, row lane >> 1, chunk lane & 1
    const int w3_lds = wave * 1024 + (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);

    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void *)w2, 0, nk * (2 * PLANE_A), 0x00020000);
    const __amdgpu_buffer_rsrc_t w3rs = __builtin_amdgcn_make_buffer_rsrc((void *)w3, 0, NST3 * W3STAGE, 0x00020000);
    // a unit's resource covers exactly its images, so pixels of images beyond the batch read as zero by themselves
    auto unit_base = [&](int64_t uu) { return buf + uu * IPW * bs; };
    auto unit_bytes = [&](int64_t uu) {
        const int64_t here = n_img - uu * IPW < IPW ? n_img - uu * IPW : IPW;
        return (int)(here * bs * 4);
    };
    float *base_cur = nullptr, *base_next = nullptr;
    int bytes_cur = 0, bytes_next = 0;

    // ---- the rings.  Stage t of the current unit, or -- past its last stage -- stage t - nk of the next one.  NO branch around
    // a load: the waits hipcc inserts are per program point (see the header) ------------------------------------------------
    u32x2 ra[DP][8];                                                    // producer: 8 channels x 2 pixels per stage
    u32x4 wa[D][2];                                                     // consumer: the two terms of its A fragment per stage
    auto load_x = [&](int t, u32x2 (&r)[8]) {
        const bool wrap = t >= nk;
        const int st = wrap ? t - nk : t;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(wrap ? base_next : base_cur), 0,
                                                                            wrap ? bytes_next : bytes_cur, 0x00020000);
        const int so = st * (KC * PS4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // the channel offset as an immediate where it fits the instruction's 12 bits, else folded into the scalar offset
            if (j * PS4 < 4096) r[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, x_voff + j * PS4, so, 0);
            else r[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, x_voff + (j * PS4 - 4 * PS4), so + 4 * PS4, 0);
        }
    };
    auto load_a = [&](int t, u32x4 (&w)[2]) {
        const int st = t >= nk ? t - nk : t;
        w[0] = __builtin_amdgcn_raw_buffer_load_b128(w2rs, a_voff, st * (2 * PLANE_A), 0);
        w[1] = __builtin_amdgcn_raw_buffer_load_b128(w2rs, a_voff + PLANE_A, st * (2 * PLANE_A), 0);
    };

    // producer: a stage from ring slot r into LDS buffer offset `bo`; bnc = norm1 of the stage's 8 channels (fetched a stage ahead)
    float x_scale = 0.f;      // 2^s of this thread's image, set per unit; 0 for an idle thread or an absent image
    f32x4 bnc[4];             // scale[0..3], scale[4..7], shift[0..3], shift[4..7] of the NEXT stage to be stored
    // norm1 constants come from the table ALREADY MULTIPLIED by the unit's 2^s where a unit is one image (14 x 14):
    // relu(r sc + sh) 2^s == relu(r (sc 2^s) + sh 2^s) bit for bit after the fp16 split (a power of two commutes with the fma's
    // rounding; the fp32 subnormals where it does not lie 2^-100 below what an fp16 term can hold), one multiply less per value
    const float *s_bnp = IPW == 1 ? reinterpret_cast<const float *>(sm + BNX_OFF) : s_bn;
    auto fetch_bn = [&](int st) {
        const int c0 = (st < nk ? st : 0) * KC + 8 * ph;
        bnc[0] = *reinterpret_cast<const f32x4 *>(s_bnp + c0);
        bnc[1] = *reinterpret_cast<const f32x4 *>(s_bnp + c0 + 4);
        bnc[2] = *reinterpret_cast<const f32x4 *>(s_bnp + cin + c0);
        bnc[3] = *reinterpret_cast<const f32x4 *>(s_bnp + cin + c0 + 4);
    };
    // the two fp16 terms of a pair of values: hi = RNE(v) (one v_cvt_pk_f16_f32), v - hi by v_fma_mix_f32, which reads the fp16
    // half in place (the compiler's form converts hi back with two more instructions and subtracts with a packed op; packed fp32
    // ops and SDWA conversions are what made this path 15 cycles per instruction beside the consumers' MFMAs)
    auto split_pair = [&](float v0, float v1, unsigned &hi, unsigned &lo) {
        const f32x2 vv = {v0, v1};
        hi = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, f16x2));
        float r0, r1;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v1));
        const f32x2 rr = {r0, r1};
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2));
    };
    auto store_x = [&](int bo, const u32x2 (&r)[8]) {
        if ((MIRX_DF_EXP & 2) && p_live) {
            char *sb = stg + bo;
            *reinterpret_cast<u32x4 *>(sb + b_lds0) = u32x4{r[0][0], r[1][0], r[2][0], r[3][0]};
            *reinterpret_cast<u32x4 *>(sb + b_lds0 + PLANE_B) = u32x4{r[4][0], r[5][0], r[6][0], r[7][0]};
            *reinterpret_cast<u32x4 *>(sb + b_lds1) = u32x4{r[0][1], r[1][1], r[2][1], r[3][1]};
            *reinterpret_cast<u32x4 *>(sb + b_lds1 + PLANE_B) = u32x4{r[4][1], r[5][1], r[6][1], r[7][1]};
            return;
        }
        if (p_live) {
            u32x4 h0, l0, h1, l1;                                       // pixel 0 / pixel 1 of the pair: 8 channels = 16 bytes per term
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[2][2];                                          // [pixel][channel 2 j, 2 j + 1]
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int c = 2 * j + e;
                    const float sc = bnc[c >> 2][c & 3], sh = bnc[2 + (c >> 2)][c & 3];
                    v[0][e] = fmaxf(fmaf(__uint_as_float(r[c][0]), sc, sh), 0.f);
                    v[1][e] = fmaxf(fmaf(__uint_as_float(r[c][1]), sc, sh), 0.f);
                    if (IPW != 1) {
                        v[0][e] *= x_scale;
                        v[1][e] *= x_scale;
                    }
                }
                unsigned th, tl;
                split_pair(v[0][0], v[0][1], th, tl);
                h0[j] = th; l0[j] = tl;
                split_pair(v[1][0], v[1][1], th, tl);
                h1[j] = th; l1[j] = tl;
            }
            char *sb = stg + bo;
            *reinterpret_cast<u32x4 *>(sb + b_lds0) = h0;
            *reinterpret_cast<u32x4 *>(sb + b_lds0 + PLANE_B) = l0;
            if (px1_ok) {
                *reinterpret_cast<u32x4 *>(sb + b_lds1) = h1;
                *reinterpret_cast<u32x4 *>(sb + b_lds1 + PLANE_B) = l1;
            }
        }
    };

    // consumer: one stage = 21 MFMAs.  Column groups (0, 1), (2, 3), (4, 5, 6): the fragments of the next group are read while
    // this group multiplies -- those of the next stage's first group during the last group -- and the three dependent MFMAs of a
    // block (smallest terms first: the order of k_conv1x1_h2) alternate with its group mates'.
    f32x16 acc[NBLK];
    f16x8 b0[2], b1[2];                                                 // column blocks 0, 1 of the stage about to be multiplied
    auto compute1 = [&](int bo, int bo_next, bool has_next, const u32x4 (&w)[2]) {
        const char *sb = stg + bo;
        const f16x8 a[2] = {__builtin_bit_cast(f16x8, w[0]), __builtin_bit_cast(f16x8, w[1])};
        f16x8 b2[2], b3[2], b4[2], b5[2], b6[2];
        MIRX_DF_RB(sb, 2, b2) MIRX_DF_RB(sb, 3, b3)
        __builtin_amdgcn_sched_barrier(0);
        if (MIRX_DF_EXP & 1) {
            MIRX_DF_RB(sb, 4, b4) MIRX_DF_RB(sb, 5, b5) MIRX_DF_RB(sb, 6, b6)
            if (has_next) { const char *sn = stg + bo_next; MIRX_DF_RB(sn, 0, b0) MIRX_DF_RB(sn, 1, b1) }
            acc[0][0] += (float)(a[0][0] + a[1][0] + b2[0][0] + b3[0][0] + b4[0][0] + b5[0][0] + b6[0][0] + b2[1][0] + b3[1][0] + b4[1][0] + b5[1][0] + b6[1][0]);
            return;
        }
        MIRX_DF_M3(0, 1, b0, b1)
        MIRX_DF_RB(sb, 4, b4) MIRX_DF_RB(sb, 5, b5) MIRX_DF_RB(sb, 6, b6)
        __builtin_amdgcn_sched_barrier(0);
        MIRX_DF_M3(2, 3, b2, b3)
        if (has_next) {                                                 // wave-uniform; LDS reads only (no vmcnt at stake)
            const char *sn = stg + bo_next;
            MIRX_DF_RB(sn, 0, b0) MIRX_DF_RB(sn, 1, b1)
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b4[0], acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b5[0], acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b6[0], acc[6], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b4[1], acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b5[1], acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b6[1], acc[6], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b4[0], acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b5[0], acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b6[0], acc[6], 0, 0, 0);
    };

    auto image_bound = [&](int64_t img) { return fmaf(in_ks, __uint_as_float(range_row[img]), in_kb); };
    auto next_buf = [](int bo) { return bo + STAGE1 == NBUF * STAGE1 ? 0 : bo + STAGE1; };

    int64_t u = blockIdx.x;
    if (u >= units) return;
#ifdef MIRX_DF_STAMPS
    unsigned long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    unsigned long long role[2] = {0, 0}, rprev = 0, role2 = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    base_cur = unit_base(u);
    bytes_cur = unit_bytes(u);
    {
        const int64_t un0 = u + gridDim.x < units ? u + gridDim.x : u;
        base_next = unit_base(un0);
        bytes_next = unit_bytes(un0);
    }
    if (consumer) {
#pragma unroll
        for (int d = 0; d < D; ++d) load_a(d, wa[d]);
    } else {
#pragma unroll
        for (int d = 0; d < DP; ++d) load_x(d, ra[d]);                   // (nk >= DP: at most one unit boundary inside the ring)
    }
    bool first = true;            // later units find their stages in slots (d + nk) % depth: rotate (below)

    for (; u < units; u += gridDim.x) {
        const int64_t img0 = u * IPW;
        const int64_t un = u + gridDim.x < units ? u + gridDim.x : u;   // the unit whose first stages the rings run into
        base_next = unit_base(un);
        bytes_next = unit_bytes(un);
        // every read of the range row happens BEFORE the barriers of this unit's stages (here and in the 1x1 epilogue); the
        // unit's own maxima are folded in only behind the last stage barrier, so no wave sees a half-updated range
        float yinv3 = 1.f;                                              // 2^-t of the 3x3 lane's image
        {
            const int64_t img = img0 + pj;
            float inv_, ys_;
            range_scales(image_bound(img < n_img ? img : img0), x_scale, inv_);
            if (!(p_live && img < n_img)) x_scale = 0.f;
            const int64_t i3 = img0 + j3 < n_img ? img0 + j3 : img0;
            range_scales(fmaf(y_ks, image_bound(i3), y_kb), ys_, yinv3);
        }
        // the rings were filled while the previous unit's 3x3 phase ran, this unit's stage d into slot (nk + d) % depth: rotate
        // now, long after those loads have landed (a rotation right behind the K loop would wait for them)
        if (!first) {
            if (consumer) {
                if (nk & 2) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        u32x4 t0 = wa[0][j]; wa[0][j] = wa[2][j]; wa[2][j] = t0;
                        u32x4 t1 = wa[1][j]; wa[1][j] = wa[3][j]; wa[3][j] = t1;
                    }
                }
            } else {
                const int rot = nk & 7;                                  // 0, 2, 4 or 6 (nk is even): new slot d = old slot (d + rot) % 8
                if (rot & 2) {                                           // by two
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const u32x2 t0 = ra[0][j], t1 = ra[1][j];
#pragma unroll
                        for (int d = 0; d < 6; ++d) ra[d][j] = ra[d + 2][j];
                        ra[6][j] = t0;
                        ra[7][j] = t1;
                    }
                }
                if (rot & 4) {                                           // by four
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const u32x2 t = ra[d][j];
                            ra[d][j] = ra[d + 4][j];
                            ra[d + 4][j] = t;
                        }
                }
            }
        }
        first = false;
        if (IPW == 1) {
            // the unit's image's 2^s into the norm1 table (read by the producers' fetch_bn from here on; the previous unit's last
            // read of it lies behind that unit's stage barriers)
            float xs_, inv_;
            range_scales(image_bound(img0), xs_, inv_);
            float *s_bnx = reinterpret_cast<float *>(sm + BNX_OFF);
            for (int i = tid; i < 2 * cin; i += 512) s_bnx[i] = s_bn[i] * xs_;
            __syncthreads();
        }

        // ---- 1x1 phase.  Stage t lives in ring slot t % 4 and LDS buffer t % 3; one barrier per step.  At step s the consumers
        // multiply stage s (and read the head of stage s + 1, staged at step s - 1) while the producers write stage s + 2 into
        // the buffer whose last readers finished at step s - 1 -------------------------------------------------------------
        MIRX_DF_T(0)
        if (consumer) {
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
            __syncthreads();                                            // stages 0 and 1 are staged
            MIRX_DF_RB(stg, 0, b0) MIRX_DF_RB(stg, 1, b1)
#ifdef MIRX_DF_STAMPS
            rprev = __builtin_amdgcn_s_memtime();
#endif
            int bo = 0;
#define MIRX_DF_STEP(I)                                                                           \
    {                                                                                             \
        const int bn_ = next_buf(bo);                                                             \
        compute1(bo, bn_, kt + (I) + 1 < nk, wa[(I)]);                                            \
        load_a(kt + (I) + D, wa[(I)]);                                                            \
        bo = bn_;                                                                                 \
        MIRX_DF_SYNC()                                                                            \
    }
            int kt = 0;
            for (; kt + 4 <= nk; kt += 4) {
                MIRX_DF_STEP(0)
                MIRX_DF_STEP(1)
                MIRX_DF_STEP(2)
                MIRX_DF_STEP(3)
            }
            if (kt < nk) {                                               // nk is even: two stages are left
                MIRX_DF_STEP(0)
                MIRX_DF_STEP(1)
            }
#undef MIRX_DF_STEP
        } else {
            fetch_bn(0);
            store_x(0, ra[0]);
            fetch_bn(1);
            store_x(STAGE1, ra[1]);
            fetch_bn(2);
            __syncthreads();
#ifdef MIRX_DF_STAMPS
            rprev = __builtin_amdgcn_s_memtime();
#endif
            // step s: stage s + 2 -> LDS (its norm1 constants were fetched at step s - 1), those of stage s + 3 requested, stage
            // s + 4 (wrapping into the next unit) -> ring slot s % 4, whose stage went to LDS at step s - 2.  Straight-line code.
            int bo = 2 * STAGE1;
#define MIRX_DF_STEP(I)                                                                           \
    {                                                                                             \
        store_x(bo, ra[((I) + 2) & 7]);                                                           \
        fetch_bn(kt + (I) + 3);                                                                   \
        MIRX_DF_MID()                                                                             \
        load_x(kt + (I) + DP, ra[(I)]);                                                           \
        bo = next_buf(bo);                                                                        \
        MIRX_DF_SYNC()                                                                            \
    }
#define MIRX_DF_TAIL(I)          /* the last two steps: nothing left to stage */                  \
    {                                                                                             \
        load_x(kt + (I) + DP, ra[(I)]);                                                           \
        MIRX_DF_SYNC()                                                                            \
    }
            int kt = 0;
            for (; kt + 10 <= nk; kt += 8) {                             // steps kt .. kt + 7 all stage a stage (kt + 7 + 2 < nk)
                MIRX_DF_STEP(0) MIRX_DF_STEP(1) MIRX_DF_STEP(2) MIRX_DF_STEP(3)
                MIRX_DF_STEP(4) MIRX_DF_STEP(5) MIRX_DF_STEP(6) MIRX_DF_STEP(7)
            }
            const int left = nk - kt;                                    // 2, 4, 6 or 8 steps (nk is even); the last two stage nothing
            if (left == 8) {
                MIRX_DF_STEP(0) MIRX_DF_STEP(1) MIRX_DF_STEP(2) MIRX_DF_STEP(3) MIRX_DF_STEP(4) MIRX_DF_STEP(5)
                MIRX_DF_TAIL(6) MIRX_DF_TAIL(7)
            } else if (left == 6) {
                MIRX_DF_STEP(0) MIRX_DF_STEP(1) MIRX_DF_STEP(2) MIRX_DF_STEP(3)
                MIRX_DF_TAIL(4) MIRX_DF_TAIL(5)
            } else if (left == 4) {
                MIRX_DF_STEP(0) MIRX_DF_STEP(1)
                MIRX_DF_TAIL(2) MIRX_DF_TAIL(3)
            } else {
                MIRX_DF_TAIL(0) MIRX_DF_TAIL(1)
            }
#undef MIRX_DF_STEP
#undef MIRX_DF_TAIL
        }
        MIRX_DF_T(1)

        // ---- 3x3 weights of stages 0 and 1 are requested now; the epilogue below covers their latency -------------------
        // 1152 chunks of 16 bytes per stage over 512 threads: every thread issues three loads (the third round re-reads a valid
        // chunk where it has none -- a load behind a branch would make the waits conservative), waves 0 and 1 store the third
        u32x4 r3[3][3];
        const int w3_v2 = (tid + 1024 < W3STAGE / 16 ? tid + 1024 : tid) * 16;
        auto load3 = [&](int st, u32x4 (&r)[3]) {
            r[0] = __builtin_amdgcn_raw_buffer_load_b128(w3rs, tid * 16, st * W3STAGE, 0);
            r[1] = __builtin_amdgcn_raw_buffer_load_b128(w3rs, (tid + 512) * 16, st * W3STAGE, 0);
            r[2] = __builtin_amdgcn_raw_buffer_load_b128(w3rs, w3_v2, st * W3STAGE, 0);
        };
        auto store3 = [&](int bufi, const u32x4 (&r)[3]) {
            *reinterpret_cast<u32x4 *>(stg + bufi * W3STAGE + w3_lds) = r[0];
            *reinterpret_cast<u32x4 *>(stg + bufi * W3STAGE + w3_lds + 8 * 1024) = r[1];
            if (wave < 2) *reinterpret_cast<u32x4 *>(stg + bufi * W3STAGE + w3_lds + 16 * 1024) = r[2];
        };
        load3(0, r3[0]);
        load3(1, r3[1]);
        load3(2, r3[2]);

        // ---- 1x1 epilogue: bias + relu2, scale, split into fp16 terms, into the LDS image of the 3x3 conv ------------------
        // group g = 2 (row block) + half holds exactly the 16 channels a lane owns in one accumulator tile (the consumer's
        // weights are permuted to that order: mirx.model.YTERMS_CHANNEL_ORDER); 16-byte chunk c of a pixel sits at slot
        // c ^ ((pixel >> 3) & 1)
        if (consumer) {
            float osc_r[16], bias_r[16];                                // this lane's 16 channels: the same for every column block
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                osc_r[r] = sOsc[ch];
                bias_r[r] = sBias[ch];
            }
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb) {
                const int px = nb * 32 + n;
                const int64_t img = img0 + px / HW;
                char *dst = ylds + (2 * wave + half) * YGROUP + px * 32;
                if (px < NLIVE && img < n_img) {
                    const float xb = image_bound(img);
                    float xs_, x_inv, y_scale, y_inv;
                    range_scales(xb, xs_, x_inv);
                    range_scales(fmaf(y_ks, xb, y_kb), y_scale, y_inv);
                    const int sw = (px >> 3) & 1;
                    u32x4 h0, h1, l0, l1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        f32x2 v;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int r = 2 * j + e;
                            float t = fmaf(acc[nb][r], osc_r[r] * x_inv, bias_r[r]);
                            t = t < 0.f ? 0.f : t;                      // keeps a NaN
                            v[e] = t * y_scale;
                        }
                        const f16x2 hh = __builtin_convertvector(v, f16x2);
                        const f32x2 r1 = v - __builtin_convertvector(hh, f32x2);
                        const f16x2 ll = __builtin_convertvector(r1, f16x2);
                        if (j < 4) { h0[j] = __builtin_bit_cast(unsigned, hh); l0[j] = __builtin_bit_cast(unsigned, ll); }
                        else { h1[j - 4] = __builtin_bit_cast(unsigned, hh); l1[j - 4] = __builtin_bit_cast(unsigned, ll); }
                    }
                    *reinterpret_cast<u32x4 *>(dst + (sw << 4)) = h0;
                    *reinterpret_cast<u32x4 *>(dst + ((sw ^ 1) << 4)) = h1;
                    *reinterpret_cast<u32x4 *>(dst + YPLANE + (sw << 4)) = l0;
                    *reinterpret_cast<u32x4 *>(dst + YPLANE + ((sw ^ 1) << 4)) = l1;
                } else if (px < NLIVE) {
                    // an image beyond the batch: its pixels read as zero (nothing of it is stored)
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(dst) = z;
                    *reinterpret_cast<u32x4 *>(dst + 16) = z;
                    *reinterpret_cast<u32x4 *>(dst + YPLANE) = z;
                    *reinterpret_cast<u32x4 *>(dst + YPLANE + 16) = z;
                }
            }
        }
        store3(0, r3[0]);
        __syncthreads();                       // the bottleneck image and the first weight stage are complete
        MIRX_DF_T(2)

        // ---- 3x3 phase ---------------------------------------------------------------------------------------------
        f32x16 acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#define MIRX_DF_READA(DST, TAP)                                                                    \
    {                                                                                              \
        DST[0] = *reinterpret_cast<const f16x8 *>(wb + (2 * (TAP)) * 1024 + a3_off);               \
        DST[1] = *reinterpret_cast<const f16x8 *>(wb + (2 * (TAP) + 1) * 1024 + a3_off);           \
    }
#define MIRX_DF_READB(DST, TAP)                                                                    \
    {                                                                                              \
        DST[0] = *reinterpret_cast<const f16x8 *>(yb + toff[TAP]);                                 \
        DST[1] = *reinterpret_cast<const f16x8 *>(yb + toff[TAP] + YPLANE);                        \
    }
#define MIRX_DF_MFMA(A, B)                                                                         \
    {                                                                                              \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], B[0], acc3, 0, 0, 0);                  \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[1], acc3, 0, 0, 0);                  \
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], B[0], acc3, 0, 0, 0);                  \
    }
#define MIRX_DF_STAGE3(S)                                                                          \
    {                                                                                              \
        if ((S) + 1 < NST3) store3(((S) + 1) & 1, r3[((S) + 1) % 3]);                              \
        if ((S) + 3 < NST3) load3((S) + 3, r3[(S) % 3]);                                           \
        if (wave < NBLK) {                                                                         \
            const char *wb = stg + ((S) & 1) * W3STAGE;                                            \
            const char *yb = ylds + (S) * YGROUP;                                                  \
            f16x8 a0[2], a1[2], b0_[2], b1_[2];                                                    \
            MIRX_DF_READA(a0, 0)                                                                   \
            MIRX_DF_READB(b0_, 0)                                                                  \
            _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                  \
                if (tap + 1 < 9) {                                                                 \
                    if (tap & 1) { MIRX_DF_READB(b0_, tap + 1) MIRX_DF_READA(a0, tap + 1) }        \
                    else { MIRX_DF_READB(b1_, tap + 1) MIRX_DF_READA(a1, tap + 1) }                \
                }                                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                                 \
                if (tap & 1) { MIRX_DF_MFMA(a1, b1_) } else { MIRX_DF_MFMA(a0, b0_) }              \
            }                                                                                      \
        }                                                                                          \
        __syncthreads();                                                                           \
    }
        MIRX_DF_STAGE3(0) MIRX_DF_STAGE3(1) MIRX_DF_STAGE3(2) MIRX_DF_STAGE3(3)
        MIRX_DF_STAGE3(4) MIRX_DF_STAGE3(5) MIRX_DF_STAGE3(6) MIRX_DF_STAGE3(7)
        MIRX_DF_T(3)
#undef MIRX_DF_STAGE3
#undef MIRX_DF_MFMA
#undef MIRX_DF_READB
#undef MIRX_DF_READA

        // ---- outputs: register r = channel 8 (r >> 2) + (r & 3) + 4 half, lane = pixel -> the tile [32 channels][196 pixels]
        // in LDS (the staging space is free behind the barrier that closed stage 7) -> 16-byte stores of channel rows by all
        // eight waves: 1568 stores per unit instead of 3136 four-byte ones from seven waves ------------------------------------
        float *otile = reinterpret_cast<float *>(stg);
        if (wave < NBLK) {                                              // wave-uniform
            const int64_t img = img0 + j3;
            const bool st_ok = live3 && img < n_img;
            const int pimg = (int)(st_ok ? img : img0);                 // a lane that carries nothing: a valid image, vmax 0
            float vmax = 0.f;
            if (live3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oc = 8 * (r >> 2) + (r & 3) + 4 * half;
                    const float v = acc3[r] * (sC3[oc] * yinv3);
                    if (st_ok) vmax = range_max(vmax, v);
                    otile[oc * OPITCH + p3] = v;
                }
            }
            if (IPW == 1) range_publish(range_row, pimg, vmax, lane);
            else range_publish_lanes(range_row, pimg, vmax, lane);
        }
        __syncthreads();
        {
            // the 32 new channels of an image are ONE contiguous slab of 32 x HW floats (16-byte aligned: cin % 32 == 0): a linear
            // copy, 16 bytes per lane
            constexpr int SLAB4 = 8 * HW;                               // 16-byte pieces per image
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)base_cur, 0, bytes_cur, 0x00020000);
            for (int g = tid; g < IPW * SLAB4; g += 512) {
                const int j = g / SLAB4, f = 4 * (g % SLAB4);
                u32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(otile[((f + e) / HW) * OPITCH + j * HW + (f + e) % HW]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_out, (int)(((int64_t)j * bs + (int64_t)cin * HW + f) * 4), 0, 0);
            }
        }
        __syncthreads();                       // the tile is read out before the next unit stages into the same space
        base_cur = base_next;
        bytes_cur = bytes_next;
        MIRX_DF_T(4)
#ifdef MIRX_DF_STAMPS
        if (tid == 0) stamp[5] += 1;
#endif
    }
#ifdef MIRX_DF_STAMPS
    if (tid == 0 && blockIdx.x < 256) {
        stamp[6] = __builtin_amdgcn_s_memrealtime() - rt0;           // 100 MHz ticks
        for (int i = 0; i < 8; ++i) g_df_stamps[blockIdx.x * 8 + i] = stamp[i];
        g_df_roles[blockIdx.x * 8 + 0] = role[0];
        g_df_roles[blockIdx.x * 8 + 1] = role[1];
    }
    if (tid == 256 && blockIdx.x < 256) {
        g_df_roles[blockIdx.x * 8 + 2] = role[0];
        g_df_roles[blockIdx.x * 8 + 3] = role[1];
        g_df_roles[blockIdx.x * 8 + 4] = role2;
    }
#endif
}
#undef MIRX_DF_RB
#undef MIRX_DF_M3

template <int W>
__global__ __launch_bounds__(512, 2) void k_dense_fused(float *__restrict__ buf, int64_t bs, int cin,
                                                        const float *__restrict__ scale, const float *__restrict__ shift,
                                                        const uint16_t *__restrict__ w2, const float *__restrict__ oscale,
                                                        const float *__restrict__ bias, const uint16_t *__restrict__ w3,
                                                        const float *__restrict__ c3osc, int64_t n_img,
                                                        unsigned *__restrict__ range_row, float in_ks, float in_kb,
                                                        float y_ks, float y_kb) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    __shared__ float sBias[CM], sOsc[CM], sC3[32];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4)
        dense_fused_body<W, true>(sm, sBias, sOsc, sC3, buf, bs, cin, scale, shift, w2, oscale, bias, w3, c3osc, n_img, range_row,
                                  in_ks, in_kb, y_ks, y_kb);
    else
        dense_fused_body<W, false>(sm, sBias, sOsc, sC3, buf, bs, cin, scale, shift, w2, oscale, bias, w3, c3osc, n_img, range_row,
                                   in_ks, in_kb, y_ks, y_kb);
}

}  // namespace

#ifdef MIRX_DF_STAMPS
extern "C" int mirx_debug_df_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_df_stamps), sizeof(g_df_stamps));
}
extern "C" int mirx_debug_df_roles(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_df_roles), sizeof(g_df_roles));
}
#endif

hipError_t launch_dense_fused(float *buf, int64_t bs, int cin, const float *scale, const float *shift, const uint16_t *w2,
                              const float *oscale, const float *bias, const uint16_t *w3, const float *c3osc, int64_t n,
                              int side, float *range_row, float in_ks, float in_kb, float y_ks, float y_kb, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if ((side != 14 && side != 7) || cin % 32 || cin < DP * KC || cin > MAX_CIN) return hipErrorInvalidValue;
    const int n_cu = current_device_cus();
    unsigned *rr = reinterpret_cast<unsigned *>(range_row);
#define MIRX_DF_LAUNCH(WW)                                                                                        \
    {                                                                                                             \
        static unsigned long long attr_devs = 0;                                                                             \
        if (first_use_on_device(attr_devs)) {                                                                                          \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_fused<WW>),                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);            \
            if (e != hipSuccess) return e;                                                                        \
        }                                                                                                         \
        const int64_t units = (n + (196 / (WW * WW)) - 1) / (196 / (WW * WW));                                    \
        const unsigned grid = (unsigned)(units < n_cu ? units : n_cu);      /* one persistent workgroup per CU */ \
        hipLaunchKernelGGL((k_dense_fused<WW>), dim3(grid), dim3(512), LDS_BYTES, st, buf, bs, cin, scale, shift, w2, oscale, \
                           bias, w3, c3osc, n, rr, in_ks, in_kb, y_ks, y_kb);                                     \
    }
    if (side == 14) MIRX_DF_LAUNCH(14) else MIRX_DF_LAUNCH(7)
#undef MIRX_DF_LAUNCH
    return hipGetLastError();
}

}  // namespace mirx
