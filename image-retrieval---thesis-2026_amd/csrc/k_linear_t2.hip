// k_linear_t2.hip -- token-major Linear on two fp16 terms per operand (k_linear_h2's arithmetic: three MFMAs per product
// block, xl wh + xh wl + xh wh, fp32 accumulation), restructured so that the matrix pipe, not the operand delivery, paces it.
//
// Why a second kernel: k_linear_h2 takes fp32 activations, splits them in registers while it stages 16 features at a time and
// synchronises once per 16-feature stage.  A diagnostic build WITHOUT its MFMAs takes 75 % of the full kernel's time (DESIGN
// 6.4): loads, splits, LDS stores and barriers are the kernel; the MFMAs hide behind them.  Here
//   * the ACTIVATIONS ARRIVE ALREADY SPLIT ("terms rows", below): the producer (LayerNorm, the previous Linear's epilogue,
//     attention, or k_rows_to_terms) writes hi | lo fp16 instead of fp32 -- the same 4 bytes per element -- so both operands
//     go global -> LDS by `buffer_load ... lds` DMA in full 128-byte lines: no staging registers, no VALU, no LDS stores;
//   * the tile is 256 tokens x 256 outputs per CU (one 8-wave workgroup, wave tile 128 tokens x 64 outputs): an L2-served
//     LDS fill sustains about 70 GB/s per CU (MI355X_MICROARCH.md, gather into LDS), and a first version of this kernel on
//     two independent 128 x 128 workgroups per CU needed 71 GB/s to keep the matrix pipe fed -- it measured 39 GB/s, a
//     stage as long as the fill's latency, 0.50-0.56 matrix-busy.  This tile needs 35 GB/s;
//   * a stage is 32 features (one 128-byte line per row: 96 MFMAs per wave per barrier instead of 12), two 64-KiB LDS
//     buffers, v_mfma_f32_16x16x32_f16 (the chip holds a higher clock on this shape than on 32x32x16);
//   * the barrier sits INSIDE a stage's MFMA stream, after the wave's last LDS read of the stage (token tile 4 of 8): the
//     rest runs from registers while the DMA of the stage after next is issued and the next stage's first fragments are read;
//   * the two waves of a SIMD have different jobs after the barrier (ROLES below): one issues the pair's DMA, one multiplies;
//   * a launch's last, partly filled round of tiles is cut along K (TAIL SPLIT below): also what fills the chip at small batches.
//
// TERMS ROWS: a [rows][K] matrix scaled by a power of two s, stored as rows of ceil(K / 32) lines of 128 bytes:
//   line g of a row = fp16 hi(s x[32 g .. 32 g + 31]) | fp16 lo(..)   with hi = fp16(s x), lo = fp16(s x - hi)
// (features beyond K are zero).  A line is exactly where the fp32 values of those 32 features would have lived.
//
// LDS image of a stage (the distance GEMM's, k_gemm.hip): operand tile rows of 128 B; 16-byte chunk c of row r sits at
// r * 128 + ((c ^ ((r >> 1) & 7)) << 4) -- the XOR is applied to the DMA's SOURCE address, the LDS side of a DMA piece is
// lane-linear -- chunks 0-3 = hi, 4-7 = lo; a 16x16x32 fragment read (16 rows x 4 chunks of one term) is conflict-free.
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int TM = 256;                  // tokens per workgroup
constexpr int TN = 256;                  // outputs per workgroup
constexpr int LINE = 128;                // bytes of one row of one stage: 32 features x (hi | lo)
constexpr int X_BYTES = TM * LINE;       // 32 KiB
constexpr int W_BYTES = TN * LINE;       // 32 KiB
constexpr int STAGE = X_BYTES + W_BYTES; // 64 KiB
constexpr int LDS_BYTES = 2 * STAGE;

// Diagnostic builds (results wrong), -DMIRX_LT2_EXP=<bit mask>: 1 no MFMAs, 2 no DMA inside the K loop, 4 plain-store
// epilogue, 8 no workgroup barrier in the K loop, 16 no LDS reads in the K loop, 32 print cycle stamps (K loop cycles per
// stage, epilogue cycles, the clock the kernel held) from a few workgroups
#ifndef MIRX_LT2_EXP
#define MIRX_LT2_EXP 0
#endif

#if MIRX_LT2_EXP & 32
__device__ unsigned long long g_lt2_stamps[4096 * 8];      // per workgroup (mod 4096): see the kernel's last lines
#endif

// ACT: 0 none, 1 GELU (erf), 2 GELU (tanh).  RES: y = res + gamma * v.  TOUT: the result is written as terms rows.
// The epilogue of four consecutive outputs `col ..` of token `row` (v = raw accumulators): shared by the tile kernel and by
// the fix-up kernel of split tiles, so that both round identically.
template <int ACT, bool RES, bool TOUT, bool MAIN = true>
__device__ inline void finish4(f32x4 v, int64_t row, int col, const f32x4 &bv, const f32x4 &gv, float out_scale, int n,
                               const float *res, float *y, char *yt, float y_scale, int np) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = v[e] * out_scale + bv[e];
        if (!(MIRX_LT2_EXP & 4)) {
            if (ACT == 1) t = gelu_erf(t);
            if (ACT == 2) t = gelu_tanh(t);
        }
        v[e] = t;
    }
#if MIRX_LT2_EXP & 64        // diagnostic (wrong layout): every load / store instruction of the epilogue covers ONE contiguous KiB
    if (MAIN) {
        const int l_ = threadIdx.x & 63;
        const int64_t rb_ = row - (l_ & 15);
        const int cb_ = col - 4 * (l_ >> 4);
        if (RES) {
            const f32x4 r = *reinterpret_cast<const f32x4 *>(res + rb_ * n + cb_ + l_ * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = r[e] + gv[e] * v[e];
        }
        if (TOUT) {
            unsigned h0, l0, h1, l1;
            split2h_pair(v[0] * y_scale, v[1] * y_scale, h0, l0);
            split2h_pair(v[2] * y_scale, v[3] * y_scale, h1, l1);
            const u32x2 hi = {h0, h1}, lo = {l0, l1};
            char *dst = yt + rb_ * ((int64_t)np * 4) + (cb_ >> 5) * LINE + l_ * 8;
            *reinterpret_cast<u32x2 *>(dst) = hi;
            *reinterpret_cast<u32x2 *>(dst + 512) = lo;
        } else {
            *reinterpret_cast<f32x4 *>(y + rb_ * n + cb_ + l_ * 4) = v;
        }
        return;
    }
#endif
    if (RES && !(MIRX_LT2_EXP & 4)) {
        const f32x4 r = *reinterpret_cast<const f32x4 *>(res + row * n + col);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e] + gv[e] * v[e];
    }
    if (TOUT) {
        unsigned h0, l0, h1, l1;
        split2h_pair(v[0] * y_scale, v[1] * y_scale, h0, l0);
        split2h_pair(v[2] * y_scale, v[3] * y_scale, h1, l1);
        // (a pair of lanes swapping halves so that each writes 16 bytes -- eight lanes a whole 128-byte line per instruction --
        // measured 1 043 vs 1 048 img/s on DINOv2, 327 vs 328 on MedSigLIP: no gain over the two 8-byte stores, not kept)
        const u32x2 hi = {h0, h1}, lo = {l0, l1};
        char *dst = yt + row * ((int64_t)np * 4) + (col >> 5) * LINE + (col & 31) * 2;
        *reinterpret_cast<u32x2 *>(dst) = hi;
        *reinterpret_cast<u32x2 *>(dst + 64) = lo;
    } else {
        *reinterpret_cast<f32x4 *>(y + row * n + col) = v;
    }
}

// tile number -> (token tile, output tile).  The workgroups of an XCD run ~32 consecutive tile numbers at a time: bands of 8
// token tiles are walked output tile by output tile, so that those 32 are 8 token tiles x 4 output tiles -- 12 operand tiles
// come over the fabric per 32 products.  (Token-major numbering made them ~2 token tiles x every output tile: MedSigLIP's fc1,
// 17 output tiles, fetched 19 per 32 -- 9 % more fabric traffic over its forward and 2 % slower; DINOv2, 3-12 output tiles:
// equal.)  Any bijection is correct; the K-split tail takes the last numbers either way.
__device__ __forceinline__ void lt2_tile_mn(int64_t tile, int ntn, int64_t mt, int64_t &tm, int &tn) {
    const int64_t g = tile / (8 * (int64_t)ntn), base = g * 8;
    const int gm = (int)(mt - base < 8 ? mt - base : 8);
    const int i = (int)(tile - g * 8 * ntn);
    tm = base + i % gm;
    tn = i / gm;
}

// TAIL SPLIT.  One tile occupies one CU for its whole K loop, so a launch of T tiles on P CUs takes ceil(T / P) rounds: 516
// tiles on 256 CUs (DINOv2's proj at 32 images) run as long as 768 would.  The launcher therefore runs only the first
// floor(T / P) * P tiles whole ("plain"); each of the r remaining tiles is cut along K into `parts` = min(stages, P / r) pieces
// that run side by side as the launch's last, short round, store their raw accumulators into `ws` [r * parts][256][256] and are
// summed (in piece order: deterministic) and finished by k_linear_t2_fix.  parts == 1 means no split.
template <int ACT, bool RES, bool TOUT>
__global__ __launch_bounds__(512, 2) void k_linear_t2(const char *__restrict__ xt, int64_t m, int kp,
                                                      const char *__restrict__ wt, const float *__restrict__ bias, int n,
                                                      const float *res, const float *__restrict__ gamma, float out_scale,
                                                      float *y, char *yt, float y_scale, int np, int ntn,
                                                      int64_t plain_tiles, int64_t per_xcd, int parts, float *ws) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    // workgroups [0, 8 per_xcd): plain tiles, dealt so that each XCD gets a contiguous run of them; beyond: pieces of the
    // split tiles
    int64_t tile;
    int piece = -1;
    if (blockIdx.x < 8 * per_xcd) {
        tile = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        if (tile >= plain_tiles) return;
    } else {
        piece = (int)(blockIdx.x - 8 * per_xcd);
        tile = plain_tiles + piece / parts;
    }
    int tn;
    int64_t tm_;
    lt2_tile_mn(tile, ntn, (m + TM - 1) / TM, tm_, tn);
    const int64_t m0 = tm_ * TM;
    const int n0 = tn * TN;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wtok = wave >> 2, wout = wave & 3;              // this wave: tokens 128 wtok .., outputs 64 wout ..
    const int pitch = kp * 4;                                 // bytes of a terms row
    int k_first = 0, nk = kp >> 5;                            // this workgroup's stages: [k_first, k_first + nk)
    if (piece >= 0) {
        const int p = piece % parts, all = kp >> 5;
        k_first = (int)((int64_t)p * all / parts);
        nk = (int)((int64_t)(p + 1) * all / parts) - k_first;
    }

    // ---- DMA addressing: piece p = tile rows 8 p .. 8 p + 7 (1 KiB).  Only waves 0-3 move data: wave w owns pieces w,
    // w + 4, .. w + 28 of each operand (see ROLES below).  Lane l: row 8 p + (l >> 3), LDS slot l & 7 <- source chunk
    // (l & 7) ^ ((row >> 1) & 7).  Rows beyond the matrix are outside the descriptor: they read as zero.
    const int grp = wave >> 2;
    const int prow = (wave & 3) * 8 + (lane >> 3);
    const int pchunk = (lane & 7) ^ ((prow >> 1) & 7);
    const int vo0 = prow * pitch + pchunk * 16;               // piece i of this wave: + i * 32 rows
    const int64_t rows_here = m - m0 < TM ? m - m0 : TM;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(xt + m0 * pitch + k_first * LINE), 0, (int)rows_here * pitch - k_first * LINE, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(wt + (int64_t)n0 * pitch + k_first * LINE), 0, TN * pitch - k_first * LINE, 0x00020000);
    // one stage (both operands) into buffer BUF: 16 pieces of this wave, back to back
#define DMA_STAGE(BUF, KT)                                                                                             \
    {                                                                                                                  \
        int vo_ = vo0;                                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, LDS_PTR(sm + (BUF) * STAGE + (i_ * 4 + (wave & 3)) * 1024), 16, vo_, \
                                                     (KT) * LINE, 0, 0);                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, LDS_PTR(sm + (BUF) * STAGE + X_BYTES + (i_ * 4 + (wave & 3)) * 1024), \
                                                     16, vo_, (KT) * LINE, 0, 0);                                      \
            vo_ += 32 * pitch;                                                                                         \
            asm volatile("" : "+v"(vo_)); /* one running offset, not eight live ones */                                 \
        }                                                                                                              \
    }

    // ---- fragment addressing: lane -> row (lane & 15), chunk (lane >> 4) of the term; lo = hi ^ 64 ------------------
    const int sw = (lane & 15) >> 1;
    const int fr = (((lane >> 4) ^ sw) << 4);
    const int xh = (wtok * 128 + (lane & 15)) * LINE + fr, xl = xh ^ 64;
    const int wh = X_BYTES + (wout * 64 + (lane & 15)) * LINE + fr, wl = wh ^ 64;
    // buffer 1 sits 64 KiB up -- beyond the 16-bit offset field of ds_read -- so its bases are registers of their own
    const int xh1 = xh + STAGE, xl1 = xl + STAGE, wh1 = wh + STAGE, wl1 = wl + STAGE;
#define LDXR(BUF, TI, T) (*reinterpret_cast<const f16x8 *>(sm + ((BUF) ? ((T) ? xl1 : xh1) : ((T) ? xl : xh)) + (TI) * 16 * LINE))
#define LDWR(BUF, OI, T) (*reinterpret_cast<const f16x8 *>(sm + ((BUF) ? ((T) ? wl1 : wh1) : ((T) ? wl : wh)) + (OI) * 16 * LINE))
#if MIRX_LT2_EXP & 16
#define LDX(BUF, TI, T) fx[(TI) & 3][T]
#define LDW(BUF, OI, T) fw[OI][T]
#else
#define LDX(BUF, TI, T) LDXR(BUF, TI, T)
#define LDW(BUF, OI, T) LDWR(BUF, OI, T)
#endif

    // accumulator register r of tile (oi, ti): output 64 wout + 16 oi + 4 (lane >> 4) + r, token 128 wtok + 16 ti + (lane & 15)
    f32x4 acc[4][8];
#pragma unroll
    for (int oi = 0; oi < 4; ++oi)
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) acc[oi][ti] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragments: the four output tiles of the stage stay in registers (fw), the eight token tiles roll through four slots
    // (fx[ti & 3]), read up to three token tiles ahead of their MFMAs
    f16x8 fw[4][2], fx[4][2];

    // the three products of one accumulator tile, smallest terms first
#if MIRX_LT2_EXP & 1
#define T3(OI, TI) acc[OI][TI][0] += (float)fw[OI][1][0] + (float)fx[(TI) & 3][1][1] + (float)fw[OI][0][2] + (float)fx[(TI) & 3][0][3];
#else
#define T3(OI, TI)                                                                                                   \
    acc[OI][TI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[OI][1], fx[(TI) & 3][0], acc[OI][TI], 0, 0, 0);          \
    acc[OI][TI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[OI][0], fx[(TI) & 3][1], acc[OI][TI], 0, 0, 0);          \
    acc[OI][TI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[OI][0], fx[(TI) & 3][0], acc[OI][TI], 0, 0, 0);
#endif
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#if MIRX_LT2_EXP & 32
    unsigned long long st_wait = 0, st_bar = 0, st_p2 = 0, st_p1 = 0, st_t = 0;
#define STAMP(ACC)                                                   \
    {                                                                \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        ACC += now_ - st_t;                                          \
        st_t = now_;                                                 \
    }
#else
#define STAMP(ACC)
#endif
#if MIRX_LT2_EXP & 8
#define KBARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#elif MIRX_LT2_EXP & 32
#define KBARRIER()                                                   \
    do {                                                             \
        STAMP(st_p1)                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        STAMP(st_wait)                                               \
        __builtin_amdgcn_s_barrier();                                \
        STAMP(st_bar)                                                \
    } while (0)
#else
#define KBARRIER()                                                   \
    do {                                                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  \
        __builtin_amdgcn_s_barrier();                                \
    } while (0)
#endif
    // A stage (buffer BUF) runs as two parts with the workgroup barrier between them:
    //   PART1: token tiles 0-4 against the four output tiles (fw and fx of token tile 0 are in registers on entry); token
    //          tiles 1-3 are read at once, then tile ti + 3 while tile ti multiplies -- with tile 4's read (tile 7) the wave
    //          has all it needs from BUF
    //   barrier: every wave has read all it needs of BUF, and the other buffer's DMA (issued one stage ago) has landed
    //   PART2: token tiles 5-7 from registers, output tile by output tile, so that each fw fragment can be re-read for the
    //          NEXT stage (other buffer) after its last use; the next stage's token tile 0 is read into the one free slot;
    //          the DMA of stage KT + 2 goes into BUF
    // The K loop's body is PART2(stage) PART1(stage + 1), so its back edge falls on the barrier, where every counter is
    // drained anyway.  `more` (wave-uniform) guards only DMA instructions: no branch encloses an MFMA.
#if MIRX_LT2_EXP & 2
#define DMA_ON false
#else
#define DMA_ON true
#endif
#define P1_READ(BUF, TI)                                                        \
    fx[(TI) & 3][0] = LDX(BUF, TI, 0);                                          \
    fx[(TI) & 3][1] = LDX(BUF, TI, 1);
#define P1_MUL(TI)                                                              \
    T3(0, TI) T3(1, TI) T3(2, TI) T3(3, TI)                                     \
    FENCE();
#define PART1(BUF)                                                              \
    {                                                                           \
        P1_READ(BUF, 1) P1_READ(BUF, 2) P1_READ(BUF, 3) P1_MUL(0)               \
        P1_READ(BUF, 4) P1_MUL(1)                                               \
        P1_READ(BUF, 5) P1_MUL(2)                                               \
        P1_READ(BUF, 6) P1_MUL(3)                                               \
        P1_READ(BUF, 7) P1_MUL(4)                                               \
    }
    // ROLES.  A DMA piece stalls the issuing wave for 60-180 cycles (MI355X_MICROARCH.md), and right after the barrier every
    // wave wants to issue its pieces: with the pieces spread over all eight waves, both waves of a SIMD stall together and the
    // matrix pipe idles (measured: 3 975 cycles per stage for 3 072 of MFMA work).  So the two waves of a SIMD (w and w + 4)
    // get different jobs for the time after the barrier: wave w + 4 (raised priority) multiplies straight on -- PART2, then
    // PART1 of the next stage -- while wave w first issues ALL of the SIMD pair's DMA (16 pieces back to back, stalling
    // beside its partner's MFMAs) and multiplies afterwards, when the partner waits at the next barrier.
#define PART2(BUF, KT)                                                          \
    {                                                                           \
        const bool more = DMA_ON && (KT) + 2 < nk;                              \
        KBARRIER();                                                             \
        FENCE();                                                                \
        fx[0][0] = LDX((BUF) ^ 1, 0, 0);                                        \
        fx[0][1] = LDX((BUF) ^ 1, 0, 1);                                        \
        if (more && grp == 0) DMA_STAGE(BUF, (KT) + 2)                          \
        FENCE();                                                                \
        T3(0, 5) T3(0, 6) T3(0, 7)                                              \
        FENCE();                                                                \
        T3(1, 5) T3(1, 6)                                                       \
        fw[0][0] = LDW((BUF) ^ 1, 0, 0);                                        \
        fw[0][1] = LDW((BUF) ^ 1, 0, 1);                                        \
        FENCE();                                                                \
        T3(1, 7) T3(2, 5) T3(2, 6)                                              \
        fw[1][0] = LDW((BUF) ^ 1, 1, 0);                                        \
        fw[1][1] = LDW((BUF) ^ 1, 1, 1);                                        \
        FENCE();                                                                \
        T3(2, 7) T3(3, 5) T3(3, 6)                                              \
        fw[2][0] = LDW((BUF) ^ 1, 2, 0);                                        \
        fw[2][1] = LDW((BUF) ^ 1, 2, 1);                                        \
        FENCE();                                                                \
        T3(3, 7)                                                                \
        FENCE();                                                                \
        fw[3][0] = LDW((BUF) ^ 1, 3, 0);                                        \
        fw[3][1] = LDW((BUF) ^ 1, 3, 1);                                        \
        FENCE();                                                                \
        STAMP(st_p2)                                                            \
    }

    // ---- prologue: stages 0 and 1 in flight, the first fragments of stage 0 in registers ----------------------------
    if (grp == 0) DMA_STAGE(0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 0 && nk > 1) DMA_STAGE(1, 1)
#pragma unroll
    for (int oi = 0; oi < 4; ++oi) {
        fw[oi][0] = LDWR(0, oi, 0);
        fw[oi][1] = LDWR(0, oi, 1);
    }
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) {                         // (slots 1-3 are read again by PART1; set here for the diagnostic builds)
        fx[ti][0] = LDXR(0, ti, 0);
        fx[ti][1] = LDXR(0, ti, 1);
    }
    // waves 4-7 are the younger partners on their SIMDs and lose the issue arbitration against waves 0-3, which then wait for
    // them at every barrier: a static priority evens the two out (k_gemm.hip)
    if (grp) __builtin_amdgcn_s_setprio(3);

#if MIRX_LT2_EXP & 32
    const unsigned long long cy0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    st_t = cy0;
#endif
    PART1(0)
    int kt = 0;
#pragma unroll 1
    for (; kt + 2 < nk; kt += 2) {
        PART2(0, kt)
        PART1(1)
        PART2(1, kt + 1)
        PART1(0)
    }
    if (kt + 2 == nk) {                                      // two stages left
        PART2(0, kt)
        PART1(1)
        PART2(1, kt + 1)                                     // (its "next stage" reads land in registers nobody uses)
    } else {
        PART2(0, kt)
    }
#undef PART1
#undef PART2
#undef P1_READ
#undef P1_MUL
#undef DMA_ON
#undef DMA_STAGE
#undef T3
#undef LDX
#undef LDW
#undef LDXR
#undef LDWR

#if MIRX_LT2_EXP & 32
    const unsigned long long cy1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef MIRX_LT2_EPI_DIRECT
    // ---- epilogue straight from the accumulators: a lane holds 4 consecutive outputs of one token per tile --------------
    if (piece >= 0) {                                        // a piece of a split tile: raw sums -> ws[piece][token][output]
        float *wp = ws + (int64_t)piece * (TM * TN);
#pragma unroll
        for (int oi = 0; oi < 4; ++oi)
#pragma unroll
            for (int ti = 0; ti < 8; ++ti)
                *reinterpret_cast<f32x4 *>(wp + (wtok * 128 + 16 * ti + (lane & 15)) * TN + wout * 64 + 16 * oi + 4 * (lane >> 4)) =
                    acc[oi][ti];
        return;
    }
    const int colmax = TOUT ? np : n;
#pragma unroll
    for (int oi = 0; oi < 4; ++oi) {
        const int col = n0 + wout * 64 + 16 * oi + 4 * (lane >> 4);
        if (col >= colmax) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
        if (bias && col < n) bv = *reinterpret_cast<const f32x4 *>(bias + col);
        if (RES && gamma) gv = *reinterpret_cast<const f32x4 *>(gamma + col);
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            const int64_t row = m0 + wtok * 128 + 16 * ti + (lane & 15);
            if (row >= m) continue;
            finish4<ACT, RES, TOUT>(acc[oi][ti], row, col, bv, gv, out_scale, n, res, y, yt, y_scale, np);
        }
    }
#else
    // ---- epilogue through a wave-private LDS transpose ------------------------------------------------------------------
    // Straight from the accumulators a lane holds 4 consecutive outputs of ONE token per tile, so a store (or residual load)
    // instruction touched 16 rows with 64 bytes each (32 for terms rows); a diagnostic that wrote the same bytes as whole KiB
    // measured the epilogue-only shapes 2.2-2.5x faster and the ViT-B layer set 11 % faster.  So every token tile (16 tokens x
    // the wave's 64 outputs = 4 KiB) goes through LDS once -- written as the accumulators hold it (row = token, 16-byte chunk
    // c = 4 oi + (lane >> 4) at slot c ^ token: conflict-free both ways), read back with lane -> (token 4 j + (lane >> 4),
    // outputs 4 (lane & 15) ..) -- and an instruction then covers 4 rows x 256 contiguous bytes.  The staging buffers are free
    // by now (the barrier below: every wave has read its last fragments); the same wave writes and reads its 4 KiB, in order.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int l16 = lane & 15, q4 = lane >> 4;
        char *scr = sm + wave * 4096;
        const int colr = n0 + wout * 64 + 4 * l16;            // this lane's 4 outputs once the tile is read back
        const int colmax = piece >= 0 ? n0 + TN : (TOUT ? np : n);
        const bool col_ok = colr < colmax;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
        if (piece < 0 && col_ok) {
            if (bias && colr < n) bv = *reinterpret_cast<const f32x4 *>(bias + colr);
            if (RES && gamma) gv = *reinterpret_cast<const f32x4 *>(gamma + colr);
        }
        float *wp = piece >= 0 ? ws + (int64_t)piece * (TM * TN) : nullptr;
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
#pragma unroll
            for (int oi = 0; oi < 4; ++oi)
                *reinterpret_cast<f32x4 *>(scr + l16 * 256 + (((4 * oi + q4) ^ l16) << 4)) = acc[oi][ti];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tr = 4 * j + q4;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(scr + tr * 256 + ((l16 ^ tr) << 4));
                const int trow = wtok * 128 + 16 * ti + tr;
                if (piece >= 0) {                            // a piece of a split tile: raw sums -> ws[piece][token][output]
                    *reinterpret_cast<f32x4 *>(wp + trow * TN + wout * 64 + 4 * l16) = v;
                } else if (col_ok && m0 + trow < m) {
                    finish4<ACT, RES, TOUT>(v, m0 + trow, colr, bv, gv, out_scale, n, res, y, yt, y_scale, np);
                }
            }
        }
    }
#endif
#if MIRX_LT2_EXP & 32
    const unsigned long long cy2 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) {
        unsigned long long *o = g_lt2_stamps + (blockIdx.x & 4095) * 8;
        o[0] = cy1 - cy0;                 // K loop cycles
        o[1] = cy2 - cy1;                 // epilogue cycles
        o[2] = rt1 - rt0;                 // K loop in 100 MHz ticks
        o[3] = nk;
        o[4] = st_p1;
        o[5] = st_wait;
        o[6] = st_bar;
        o[7] = st_p2;
    }
#endif
}

// Split tiles: sum the `parts` pieces of tile `plain_tiles + blockIdx.x / 64` (piece order) and finish them.  One workgroup =
// 4 token rows x 256 outputs, one 16-byte vector per thread; the pieces are fetched eight at a time and added in order.
template <int ACT, bool RES, bool TOUT>
__global__ __launch_bounds__(256) void k_linear_t2_fix(const float *__restrict__ ws, int parts, int64_t plain_tiles, int ntn,
                                                       int64_t m, const float *__restrict__ bias, int n, const float *res,
                                                       const float *__restrict__ gamma, float out_scale, float *y, char *yt,
                                                       float y_scale, int np) {
    const int64_t st = blockIdx.x >> 6;
    const int64_t tile = plain_tiles + st;
    int tn_;
    int64_t tm_;
    lt2_tile_mn(tile, ntn, (m + TM - 1) / TM, tm_, tn_);
    const int n0 = tn_ * TN;
    const int64_t m0 = tm_ * TM;
    const int col = n0 + 4 * (threadIdx.x & 63);
    const int rl = (blockIdx.x & 63) * 4 + (threadIdx.x >> 6);
    const int64_t row = m0 + rl;
    if (col >= (TOUT ? np : n) || row >= m) return;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
    if (bias && col < n) bv = *reinterpret_cast<const f32x4 *>(bias + col);
    if (RES && gamma) gv = *reinterpret_cast<const f32x4 *>(gamma + col);
    const float *src = ws + st * parts * (int64_t)(TM * TN) + rl * TN + 4 * (threadIdx.x & 63);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < parts; p0 += 8) {
        f32x4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            a[j] = p0 + j < parts ? *reinterpret_cast<const f32x4 *>(src + (p0 + j) * (int64_t)(TM * TN)) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (p0 + j < parts) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (p0 + j) ? v[e] + a[j][e] : a[j][e];
            }
    }
    finish4<ACT, RES, TOUT, false>(v, row, col, bv, gv, out_scale, n, res, y, yt, y_scale, np);
}

// fp32 rows [m][k] (row stride ldx floats) -> terms rows [m][kp / 32] lines, scaled by the power of two `scale`.
// thread -> 4 consecutive features: 8 threads write one 128-byte line.
__global__ __launch_bounds__(256) void k_rows_to_terms(const float *__restrict__ x, int64_t m, int k, int64_t ldx, float scale,
                                                       char *__restrict__ xt, int kp) {
    const int per_row = kp >> 2;
    const int64_t total = m * per_row;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
        const int64_t row = it / per_row;
        const int c = (int)(it - row * per_row) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < k) {
            v = *reinterpret_cast<const f32x4 *>(x + row * ldx + c);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < k) v[e] = x[row * ldx + c + e];
        }
        unsigned h0, l0, h1, l1;
        split2h_pair(v[0] * scale, v[1] * scale, h0, l0);
        split2h_pair(v[2] * scale, v[3] * scale, h1, l1);
        const u32x2 hi = {h0, h1}, lo = {l0, l1};
        char *dst = xt + row * ((int64_t)kp * 4) + (c >> 5) * LINE + (c & 31) * 2;
        *reinterpret_cast<u32x2 *>(dst) = hi;
        *reinterpret_cast<u32x2 *>(dst + 64) = lo;
    }
}

}  // namespace

#if MIRX_LT2_EXP & 32
// reads the stamps back AND clears them: a shape with fewer workgroups than the one measured before it would otherwise
// carry that one's stamps in its upper entries (round 3's r03_linear_bench.txt: contaminated means)
extern "C" int mirx_debug_lt2_stamps(unsigned long long *out) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lt2_stamps), sizeof(g_lt2_stamps));
    if (e != hipSuccess) return (int)e;
    static unsigned long long zeros[4096 * 8];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lt2_stamps), zeros, sizeof(zeros));
}
#endif

hipError_t launch_rows_to_terms(const float *x, int64_t m, int k, int64_t ldx, float scale, void *xt, hipStream_t st) {
    if (m <= 0) return hipSuccess;
    if (k < 1 || ldx < k || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return hipErrorInvalidValue;
    const int kp = (k + 31) / 32 * 32;
    const int64_t total = m * (kp >> 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_rows_to_terms, dim3((unsigned)blocks), dim3(256), 0, st, x, m, k, ldx, scale,
                       reinterpret_cast<char *>(xt), kp);
    return hipGetLastError();
}

// how a launch of `tiles` tiles is cut on `cus` CUs (see TAIL SPLIT above): plain tiles, split tiles, pieces per split tile
static void lt2_plan(int64_t tiles, int stages, int cus, int64_t *plain, int64_t *split, int *parts) {
    *plain = tiles / cus * cus;
    *split = tiles - *plain;
    int64_t g = *split > 0 ? cus / *split : 1;
    if (g > stages) g = stages;
    if (g < 2) {                      // the last round is more than half full (or there is none): whole tiles
        *plain = tiles;
        *split = 0;
        g = 1;
    }
    *parts = (int)g;
}

static int lt2_cus() { return current_device_cus(); }

size_t linear_t2_workspace_bytes(int64_t m, int k, int n) {
    if (m <= 0 || k < 1 || n < 1) return 0;
    int64_t plain, split;
    int parts;
    lt2_plan(((m + TM - 1) / TM) * ((n + TN - 1) / TN), (k + 31) / 32, lt2_cus(), &plain, &split, &parts);
    return (size_t)(split * parts) * TM * TN * sizeof(float);
}

hipError_t launch_linear_t2(const void *xt, int64_t m, int k, const void *wt, const float *bias, int n, int act,
                            const float *res, const float *gamma, float out_scale, float *y, void *yt, float y_scale,
                            void *workspace, size_t workspace_bytes, hipStream_t st) {
    if (m <= 0) return hipSuccess;
    if (k < 1 || n < 4 || (n & 3) || act < 0 || act > 2 || (!y && !yt) || (y && yt) || (yt && res) || (gamma && !res))
        return hipErrorInvalidValue;
    const int kp = (k + 31) / 32 * 32, np = (n + 31) / 32 * 32;
    const int ntn = (n + TN - 1) / TN;                 // wt holds ntn * 256 rows, zero beyond n
    const int64_t total = ((m + TM - 1) / TM) * ntn;
    int64_t plain, split;
    int parts;
    lt2_plan(total, kp / 32, lt2_cus(), &plain, &split, &parts);
    if (split > 0 && (!workspace || workspace_bytes < (size_t)(split * parts) * TM * TN * sizeof(float))) {
        plain = total;                                 // no (or too small a) workspace: whole tiles only
        split = 0;
        parts = 1;
    }
    const int64_t per_xcd = (plain + 7) / 8;
    const int64_t blocks = per_xcd * 8 + split * parts;
    if (blocks > 0x7fffffff || (int64_t)TM * kp * 4 > 0x7fffffff) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    float *ws = reinterpret_cast<float *>(workspace);
#define MIRX_T2(A, R, T)                                                                                               \
    {                                                                                                                  \
        static unsigned long long attr_devs = 0;                                                                                  \
        if (first_use_on_device(attr_devs)) {                                                                                               \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_linear_t2<A, R, T>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);                 \
            if (e != hipSuccess) return e;                                                                             \
        }                                                                                                              \
        hipLaunchKernelGGL((k_linear_t2<A, R, T>), grid, dim3(512), LDS_BYTES, st, reinterpret_cast<const char *>(xt), m, kp, \
                           reinterpret_cast<const char *>(wt), bias, n, res, gamma, out_scale, y,                      \
                           reinterpret_cast<char *>(yt), y_scale, np, ntn, plain, per_xcd, parts, ws);                 \
        if (split > 0)                                                                                                 \
            hipLaunchKernelGGL((k_linear_t2_fix<A, R, T>), dim3((unsigned)(split * 64)), dim3(256), 0, st, ws, parts, plain, ntn, \
                               m, bias, n, res, gamma, out_scale, y, reinterpret_cast<char *>(yt), y_scale, np);       \
    }
    if (yt) {
        if (act == 2) MIRX_T2(2, false, true) else if (act == 1) MIRX_T2(1, false, true) else MIRX_T2(0, false, true)
    } else if (res) {
        if (act) return hipErrorInvalidValue;
        MIRX_T2(0, true, false)
    } else {
        if (act == 2) MIRX_T2(2, false, false) else if (act == 1) MIRX_T2(1, false, false) else MIRX_T2(0, false, false)
    }
#undef MIRX_T2
    return hipGetLastError();
}

}  // namespace mirx
