// k_gemm.hip -- tier-1 candidate generation: bf16 MFMA distance GEMM with a fused epilogue.
//
// Replaces the score matrix of the reference's brute force (test.py:1080 `-torch.cdist`,
// test_nonclip.py:151 `e @ e.t()`, eval_medsiglip.py:238) -- but the [gallery x query] score
// matrix never reaches HBM: the epilogue either keeps the rows whose score beats the query's
// threshold (filter mode) or reduces sampled rows to group maxima (group-max mode, used to
// pick the thresholds).
//
// Shape: S^T[gallery row, query] = G[rows, K] * Q[queries, K]^T, K = dimp, both operands
// K-contiguous bf16.  The gallery is the M operand so that in the 32x32 accumulator layout a
// lane owns ONE query column (lane & 31) and its registers walk gallery rows: the threshold
// test is one running max per lane and one compare, ~0.5 VALU per score.
//
// Tile: 256 gallery rows x BN queries (BN = 64/128/256), BK = 64, 8 waves (512 threads),
// v_mfma_f32_32x32x16_bf16, LDS double buffer filled by `buffer_load_dwordx4 ... lds` (16 B per
// lane, wave-linear LDS image; one per-lane voffset for the whole kernel, everything else of the
// address is scalar).  Bank conflicts: tile rows are 128 B; the 16-B chunk index is XORed with
// (row >> 1) & 7 on the SOURCE address and on the ds_read_b128 address, which makes every
// 16-lane ds_read_b128 group hit 16 distinct slots (DESIGN.md "LDS image").
//
// K loop (one barrier per K-step, LDS latency hidden inside the wave):
//   * DMA of tile kt+2 is issued right after the barrier that ends tile kt's LDS reads;
//   * fragments roll: the A fragment of row-tile mi is re-read (next k-slice) right after the
//     two MFMAs that consume it have issued; B fragments are double-buffered; so every ds_read
//     has several MFMAs of cover before its first use;
//   * the barrier sits in front of the LAST slice of a tile: that slice's fragments are already
//     in registers, so its MFMAs cover the DMA issue and the first reads of tile kt+1.
#include <cstdio>
#include <cstdlib>

#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 256;
constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;             // 128 B of K per tile row
constexpr int A_TILE_BYTES = BM * ROW_BYTES;  // 32 KiB

// Diagnostic builds (results wrong; used to attribute time, see DESIGN.md "GEMM time attribution")
#ifdef MIRX_EXP_NODMA   // no DMA after the first two K-tiles
#define MIRX_EXP_COND && (A.dimp < 0)
#else
#define MIRX_EXP_COND
#endif
#ifdef MIRX_EXP_NOBAR   // no workgroup barrier in the K loop
#define MIRX_KBARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
// vmcnt(0) first: a K-tile is published only after this wave's `buffer_load ... lds` pieces have landed (the
// compiler's own wait before s_barrier covers registers, not the asynchronous LDS writes)
#define MIRX_KBARRIER()                                      \
    do {                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     \
        __syncthreads();                                     \
    } while (0)
#endif
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// Stage ROWS tile rows of a K-contiguous bf16 matrix into LDS through a buffer descriptor.
// Wave w issues pieces w, w+8, ...; piece = 64 lanes x 16 B = 8 tile rows.  `voff` is the
// per-lane byte offset of (row-in-piece-group, swizzled chunk); `soff0` the scalar byte offset
// of (tile row 0, k0); `piece_stride` the bytes between rows 64 apart.
template <int ROWS>
__device__ inline void stage_tile(char *lds_tile, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff0,
                                  int piece_stride, int wave) {
#pragma unroll
    for (int i = 0; i < ROWS / 64; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + (i * 8 + wave) * 1024), 16, voff,
                                                 soff0 + i * piece_stride, 0, 0);
}

// ---- the persistent schedule, shared by the kernel and by gemm_plan() on the host --------------
// A workgroup is bound to ONE query tile for its whole life and walks the gallery tiles
// ph, ph + nph, ph + 2 nph, ...   ("unit" = (query tile, phase ph)).  Units are numbered
// query group (4 tiles) > phase > tile in group and dealt to workgroups so that unit U runs on XCD
// label U / (grid/8): the 32 workgroups of an XCD label then share 4 query tiles (2 MiB, L2
// resident) and stream 8 gallery tiles at a time.  Speed only -- any assignment is correct.
#ifndef MIRX_PLAN_GROUP
#define MIRX_PLAN_GROUP 4      // query tiles that share a stream of gallery tiles (speed only; A/B arm of tools/gemm_ab.py)
#endif
struct Plan {
    int64_t ngt;     // gallery tiles
    int nqt;         // query tiles
    int nph;         // phases (= producer workgroups) per query tile
    int grid;        // workgroups launched (multiple of 8)
    __host__ __device__ int units() const { return nqt * nph; }
    __host__ __device__ void unit(int u, int &qt, int &ph) const {
        constexpr int G = MIRX_PLAN_GROUP;
        const int full_groups = nqt / G, per_group = G * nph;
        if (u < full_groups * per_group) {
            const int g = u / per_group, l = u % per_group;
            ph = l / G;
            qt = g * G + (l % G);
        } else {
            const int l = u - full_groups * per_group, gsz = nqt - G * full_groups;
            ph = l / gsz;
            qt = full_groups * G + l % gsz;
        }
    }
};

__host__ __device__ inline Plan make_plan(int64_t n_rows, int64_t nq_pad, int bn, int grid_cap) {
    Plan p;
    p.ngt = (n_rows + BM - 1) / BM;
    p.nqt = (int)(nq_pad / bn);
    int nph = p.nqt > 0 ? grid_cap / p.nqt : 1;
    if (nph < 1) nph = 1;
    if (nph > p.ngt) nph = (int)(p.ngt > 0 ? p.ngt : 1);
    p.nph = nph;
    p.grid = (p.nqt * nph + 7) / 8 * 8;
    return p;
}

// candidate slots of one (query, producer wave) region: about 1024 slots per query in total
__host__ __device__ inline int region_slots(int regions) {
    int s = 2048 / (regions > 0 ? regions : 1);
    return s < 8 ? 8 : (s > 64 ? 64 : s);
}

// MODE 0: threshold filter.  MODE 1: group maxima of sampled rows.
template <int BN, int MODE, bool L2>
__global__ __launch_bounds__(512, 2) void k_gemm(GemmArgs A) {
    constexpr int WARPS_N = BN / 64;
    constexpr int WARPS_M = 8 / WARPS_N;
    constexpr int WM_ROWS = BM / WARPS_M;        // gallery rows per wave
    constexpr int M_REP = WM_ROWS / 32;
    constexpr int N_REP = 2;
    constexpr int B_TILE_BYTES = BN * ROW_BYTES;
    constexpr int LDS_B0 = 2 * A_TILE_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;
    const int nk = A.dimp / BK;                   // even (dimp is a multiple of 128)
    const int ld_bytes = A.dimp * 2;

    // ---- which unit am I ------------------------------------------------------------------------
    Plan plan;
    plan.ngt = (A.n_rows + BM - 1) / BM;
    plan.nqt = (int)(A.nq_pad / BN);
    plan.nph = A.nph;
    plan.grid = gridDim.x;
    const int u = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (u >= plan.units()) return;
    int qt, ph;
    plan.unit(u, qt, ph);
    const int64_t q_row0 = (int64_t)qt * BN;
    int *lcnt = reinterpret_cast<int *>(smem + LDS_B0 + 2 * B_TILE_BYTES);   // [BN][WARPS_M] region fill counts
    if (MODE == 0) {
        for (int i = threadIdx.x; i < BN * WARPS_M; i += 512) lcnt[i] = 0;
    }

    // ---- DMA addressing: descriptor per operand tile, one voffset per lane ---------------------
    const int prow = wave * 8 + (lane >> 3);                       // row inside a 64-row piece group
    const int pchunk = (lane & 7) ^ ((prow >> 1) & 7);             // source chunk for LDS slot lane&7
    const int rs = (int)A.row_stride;
    const int voff_a = prow * rs * ld_bytes + pchunk * 16;
    const int voff_b = prow * ld_bytes + pchunk * 16;
    const int pstride_a = 64 * rs * ld_bytes;
    const int pstride_b = 64 * ld_bytes;
    auto make_rsrc_a = [&](int64_t gt_) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(A.g16 + gt_ * BM * A.row_stride * A.dimp), 0,
                                                 BM * rs * ld_bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc((void *)(A.q16 + q_row0 * A.dimp), 0, BN * ld_bytes, 0x00020000);

    // ---- LDS fragment addressing: byte offset for k-slice kk is base ^ (kk << 5) ----------------
    // (row r, chunk c) lives at r*128 + ((c ^ ((r>>1)&7)) << 4); for this lane r = const + (lane&31)
    // with const a multiple of 32, c = 2*kk + (lane>>5)  ->  ((lane>>5) ^ sw) << 4 XOR kk << 5.
    const int sw = ((lane & 31) >> 1) & 7;
    const int frag_lo = (((lane >> 5) ^ sw) << 4);
    const int a_base = (wm * WM_ROWS + (lane & 31)) * ROW_BYTES + frag_lo;
    const int b_base = LDS_B0 + (wn * 64 + (lane & 31)) * ROW_BYTES + frag_lo;
    int a_addr[4], b_addr[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_addr[kk] = a_base ^ (kk << 5);
        b_addr[kk] = b_base ^ (kk << 5);
    }

    // ---- per-lane constants of the epilogue: this lane's two query columns ----------------------
    const int half = lane >> 5;
    float tau[N_REP];
    int64_t qcol[N_REP];
#pragma unroll
    for (int ni = 0; ni < N_REP; ++ni) {
        qcol[ni] = q_row0 + wn * 64 + ni * 32 + (lane & 31);
#ifdef MIRX_EXP_NOEPI
        tau[ni] = INFINITY;
#else
        tau[ni] = MODE == 0 ? A.tau[qcol[ni]] : 0.0f;
#endif
    }
    const int region = ph * WARPS_M + wm;                          // private to this wave

    f32x16 acc[M_REP][N_REP];

    // accumulator register r of tile (mi, ni): gallery row (r&3) + 8*(r>>2) + 4*(lane>>5) of the
    // 32-row tile, query column lane & 31.
    auto epilogue = [&](int64_t gt_) {
        const int64_t tile_row0 = gt_ * BM + wm * WM_ROWS;   // in units of sampled rows
        if (L2) {
            // bias of the 16 rows of each 32-row tile, 8 loads in flight at a time (more would push
            // the kernel over 256 VGPRs while the 128 accumulators are live)
#pragma unroll
            for (int mi = 0; mi < M_REP; ++mi) {
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float b[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        b[r] = A.gbias[(tile_row0 + mi * 32 + ((r0 + r) & 3) + 8 * ((r0 + r) >> 2) + 4 * half) * A.row_stride];
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int ni = 0; ni < N_REP; ++ni) acc[mi][ni][r0 + r] += b[r];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int ni = 0; ni < N_REP; ++ni) {
            float mx = -INFINITY;
#pragma unroll
            for (int mi = 0; mi < M_REP; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[mi][ni][r]);
            if (MODE == 1) {
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (half == 0) A.groupmax[qcol[ni] * A.ngroups + gt_ * WARPS_M + wm] = mx;
            } else if (__any(mx > tau[ni])) {
                // Rare path (about one score per wave tile passes).  No global atomics: the region
                // (query, this workgroup's phase, this wave row) belongs to this wave alone and its
                // fill count lives in LDS.  Lanes l and l+32 hold the same query: they take turns.
                // Rows beyond the region's slots go to the query's shared overflow list.
                const int cidx = (wn * 64 + ni * 32 + (lane & 31)) * WARPS_M + wm;
                Cand *dst = A.cand + (qcol[ni] * A.regions + region) * A.slots;
#pragma unroll
                for (int mi = 0; mi < M_REP; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[mi][ni][r];
                        const int64_t grow = (tile_row0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * A.row_stride;
                        const bool pass = v > tau[ni] && grow < A.n_rows;
                        if (pass) {
                            // one LDS atomic per passing score: lanes l and l + 32 (the same query) get distinct slots
                            // whatever the compiler does with the surrounding code (a plain read-modify-write of the
                            // counter shared by two lanes is a data race it may legally merge)
                            const int c = atomicAdd(&lcnt[cidx], 1);
                            Cand cd;
                            cd.s = v;
                            cd.row = (int32_t)grow;
                            if (c < A.slots) {
                                dst[c] = cd;
                            } else {
                                const int p = atomicAdd(&A.ovf_cnt[qcol[ni]], 1);
                                if (p < CAND_OVF) A.ovf[qcol[ni] * CAND_OVF + p] = cd;
                            }
                        }
                    }
            }
        }
    };

    bf16x8 fa[M_REP], fb0[N_REP], fb1[N_REP];

#ifdef MIRX_EXP_NOLDS   // diagnostic: fragments are never refreshed (results wrong)
#define MIRX_LDA(KK, CUR, MI) fa[MI]
#define MIRX_LDB(KK, CUR, NI) fb0[NI]
#else
#define MIRX_LDA(KK, CUR, MI) (*reinterpret_cast<const bf16x8 *>(smem + a_addr[KK] + (CUR) * A_TILE_BYTES + (MI) * 32 * ROW_BYTES))
#define MIRX_LDB(KK, CUR, NI) (*reinterpret_cast<const bf16x8 *>(smem + b_addr[KK] + (CUR) * B_TILE_BYTES + (NI) * 32 * ROW_BYTES))
#endif
#define MIRX_MFMA2(MI, FB_CUR)                                                                    \
    acc[MI][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[MI], FB_CUR[0], acc[MI][0], 0, 0, 0); \
    acc[MI][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[MI], FB_CUR[1], acc[MI][1], 0, 0, 0);
    // one k-slice: MFMAs with (fa, FB_CUR); refresh fa[] and fill FB_NXT from buffer NCUR, slice NKK.
    // sched_group_barrier pins "2 MFMAs, then the reads that refresh what they consumed".
#define MIRX_SLICE(FB_CUR, FB_NXT, NCUR, NKK)                                     \
    {                                                                             \
        MIRX_MFMA2(0, FB_CUR)                                                     \
        fa[0] = MIRX_LDA(NKK, NCUR, 0);                                           \
        FB_NXT[0] = MIRX_LDB(NKK, NCUR, 0);                                       \
        if constexpr (M_REP == 1) FB_NXT[1] = MIRX_LDB(NKK, NCUR, 1);             \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                        \
        if constexpr (M_REP == 1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0); \
        else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                   \
        if constexpr (M_REP > 1) {                                                \
            MIRX_MFMA2(1, FB_CUR)                                                 \
            fa[1] = MIRX_LDA(NKK, NCUR, 1);                                       \
            FB_NXT[1] = MIRX_LDB(NKK, NCUR, 1);                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                    \
        }                                                                         \
        if constexpr (M_REP > 2) {                                                \
            MIRX_MFMA2(2, FB_CUR)                                                 \
            fa[2] = MIRX_LDA(NKK, NCUR, 2);                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
            MIRX_MFMA2(3, FB_CUR)                                                 \
            fa[3] = MIRX_LDA(NKK, NCUR, 3);                                       \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    \
        }                                                                         \
    }
    // one K-tile held in buffer CUR; KT (runtime) only steers which K-tile the DMA fetches next:
    // K-tile KT+2 of this gallery tile, or -- in its last two K-tiles -- K-tile 0/1 of the NEXT
    // gallery tile, so the stream of K-tiles never drains at a tile boundary.  The body is
    // branch-free around the MFMAs (a branch there makes hipcc copy the 128 accumulator registers
    // and spill); after the very last K-tile the barrier and the "next" fragment reads still run,
    // on LDS data nobody consumes.
#define MIRX_KTILE(CUR, KT)                                                                        \
    MIRX_SLICE(fb0, fb1, CUR, 1)                                                                   \
    MIRX_SLICE(fb1, fb0, CUR, 2)                                                                   \
    MIRX_SLICE(fb0, fb1, CUR, 3)                                                                   \
    MIRX_KBARRIER(); /* my DMA of the next K-tile landed + every read of buffer CUR has returned */ \
    if ((KT) + 2 < nk MIRX_EXP_COND) {                                                             \
        stage_tile<BM>(smem + (CUR) * A_TILE_BYTES, rsrc_a, voff_a, ((KT) + 2) * ROW_BYTES, pstride_a, wave); \
        stage_tile<BN>(smem + LDS_B0 + (CUR) * B_TILE_BYTES, rsrc_b, voff_b, ((KT) + 2) * ROW_BYTES, pstride_b, wave); \
    } else if (have_next MIRX_EXP_COND) {                                                          \
        stage_tile<BM>(smem + (CUR) * A_TILE_BYTES, rsrc_a_nx, voff_a, ((KT) + 2 - nk) * ROW_BYTES, pstride_a, wave); \
        stage_tile<BN>(smem + LDS_B0 + (CUR) * B_TILE_BYTES, rsrc_b, voff_b, ((KT) + 2 - nk) * ROW_BYTES, pstride_b, wave); \
    }                                                                                              \
    MIRX_SLICE(fb1, fb0, (CUR) ^ 1, 0)

#ifdef MIRX_EXP_PRIO32
    if (wave >= 4) __builtin_amdgcn_s_setprio(3);
#endif
    int64_t gt = ph;
    __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc_a(gt);

    stage_tile<BM>(smem, rsrc_a, voff_a, 0, pstride_a, wave);
    stage_tile<BN>(smem + LDS_B0, rsrc_b, voff_b, 0, pstride_b, wave);
    MIRX_KBARRIER();                                   // K-tile 0 visible (and lcnt zeroed)
    stage_tile<BM>(smem + A_TILE_BYTES, rsrc_a, voff_a, ROW_BYTES, pstride_a, wave);
    stage_tile<BN>(smem + LDS_B0 + B_TILE_BYTES, rsrc_b, voff_b, ROW_BYTES, pstride_b, wave);
#pragma unroll
    for (int mi = 0; mi < M_REP; ++mi)
        fa[mi] = *reinterpret_cast<const bf16x8 *>(smem + a_addr[0] + mi * 32 * ROW_BYTES);
    fb0[0] = *reinterpret_cast<const bf16x8 *>(smem + b_addr[0]);
    fb0[1] = *reinterpret_cast<const bf16x8 *>(smem + b_addr[0] + 32 * ROW_BYTES);
    fb1[0] = fb0[0];
    fb1[1] = fb0[1];

    for (;;) {
        const int64_t gt_nx = gt + plan.nph;
        const bool have_next = gt_nx < plan.ngt;
        const __amdgpu_buffer_rsrc_t rsrc_a_nx = make_rsrc_a(have_next ? gt_nx : gt);

#pragma unroll
        for (int mi = 0; mi < M_REP; ++mi)
#pragma unroll
            for (int ni = 0; ni < N_REP; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

        for (int kt = 0; kt < nk; kt += 2) {
            MIRX_KTILE(0, kt)
            MIRX_KTILE(1, kt + 1)
        }

        epilogue(gt);
        if (!have_next) break;
        gt = gt_nx;
        rsrc_a = rsrc_a_nx;
    }
    if (MODE == 0) {
        // publish this workgroup's region fill counts (each (query, region) has exactly one writer)
        __syncthreads();
        for (int i = threadIdx.x; i < BN * WARPS_M; i += 512) {
            const int c = lcnt[i];
            if (c > 0) A.region_cnt[(q_row0 + i / WARPS_M) * A.regions + ph * WARPS_M + (i % WARPS_M)] = c;
        }
    }
#undef MIRX_KTILE
#undef MIRX_SLICE
#undef MIRX_MFMA2
#undef MIRX_LDA
#undef MIRX_LDB
}

// ================================================================================================
// The same kernel on v_mfma_f32_16x16x32_bf16 (BN = 256 only).  Same tile, schedule, LDS image and
// epilogue contract; the wave tile (128 gallery rows x 64 queries) is 8 x 4 accumulator tiles of
// 16 x 16 and a K-tile is two 32-deep slices.  Under load the chip holds a higher clock on this shape
// (MI355X_MICROARCH.md, DVFS (7)): same cycles per FLOP, more FLOP/s.
//   A fragment: lane -> gallery row (lane & 15), K chunk 4*slice + (lane >> 4)
//   B fragment: lane -> query     (lane & 15), same K chunk
//   accumulator register r of tile (mi, ni): gallery row 16 mi + 4 (lane >> 4) + r, query 16 ni + (lane & 15)
// ================================================================================================
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct RingEntry {     // a lane's pair of accumulator tiles that holds a passing score, parked in LDS by k_gemm16's K loop
    f32x4 a0, a1;      // tiles (mi, n0) and (mi, n0 + 1): rows 16 mi + 4 (lane >> 4) + r, queries 16 n + (lane & 15)
    unsigned tag;      // lane | mi << 6 | (n0 >> 1) << 9 | gallery tile << 10
    unsigned pad[3];   // 48 bytes: the two tiles go out as one ds_write_b128 each
};
constexpr int RING_SLOTS = 64;       // per wave: 3 KiB, 24 KiB per workgroup (LDS: 131 + 3 + 24 = 158 of 160 KiB)
constexpr int SPILL_SLOTS = 2048;    // per wave, global (96 KiB): what the ring overflowed into, kept until the workgroup's end
constexpr int SPILL_PER_TILE = 1024; // the most one gallery tile can park: every (lane, tile pair) of the wave

__device__ inline float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ inline float vmax2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

template <int MODE, bool L2>
__global__ __launch_bounds__(512, 2) void k_gemm16(GemmArgs A) {
    constexpr int BN = 256;
    constexpr int WARPS_N = 4, WARPS_M = 2;
    constexpr int WM_ROWS = BM / WARPS_M;        // 128 gallery rows per wave
    constexpr int M_REP = WM_ROWS / 16;          // 8
    constexpr int N_REP = 4;                     // 64 queries per wave
    constexpr int B_TILE_BYTES = BN * ROW_BYTES;
    constexpr int LDS_B0 = 2 * A_TILE_BYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int wm = wave / WARPS_N, wn = wave % WARPS_N;
    const int nk = A.dimp / BK;                   // even
    const int ld_bytes = A.dimp * 2;

    Plan plan;
    plan.ngt = (A.n_rows + BM - 1) / BM;
    plan.nqt = (int)(A.nq_pad / BN);
    plan.nph = A.nph;
    plan.grid = gridDim.x;
    const int u = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (u >= plan.units()) return;
    int qt, ph;
    plan.unit(u, qt, ph);
    const int64_t q_row0 = (int64_t)qt * BN;
    int *lcnt = reinterpret_cast<int *>(smem + LDS_B0 + 2 * B_TILE_BYTES);   // [BN][WARPS_M] region fill counts
    if (MODE == 0) {
        for (int i = threadIdx.x; i < BN * WARPS_M; i += 512) lcnt[i] = 0;
    }

    const int prow = wave * 8 + (lane >> 3);
    const int pchunk = (lane & 7) ^ ((prow >> 1) & 7);
    const int rs = (int)A.row_stride;
    const int voff_a = prow * rs * ld_bytes + pchunk * 16;
    const int voff_b = prow * ld_bytes + pchunk * 16;
    const int pstride_a = 64 * rs * ld_bytes;
    const int pstride_b = 64 * ld_bytes;
    auto make_rsrc_a = [&](int64_t gt_) {
#ifdef MIRX_EXP_SAMEA   // diagnostic (results wrong): every gallery tile is the workgroup's first one, i.e. L2-resident
        gt_ = ph;
#endif
        return __builtin_amdgcn_make_buffer_rsrc((void *)(A.g16 + gt_ * BM * A.row_stride * A.dimp), 0,
                                                 BM * rs * ld_bytes, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc((void *)(A.q16 + q_row0 * A.dimp), 0, BN * ld_bytes, 0x00020000);

    // (row r, chunk c) lives at r*128 + ((c ^ ((r>>1)&7)) << 4); r = 16*tile + (lane & 15),
    // c = 4*slice + (lane >> 4)  ->  (((lane >> 4) ^ sw) << 4) XOR (slice << 6)
    const int sw = ((lane & 15) >> 1) & 7;
    const int frag_lo = (((lane >> 4) ^ sw) << 4);
    const int a_base = (wm * WM_ROWS + (lane & 15)) * ROW_BYTES + frag_lo;
    const int b_base = LDS_B0 + (wn * 64 + (lane & 15)) * ROW_BYTES + frag_lo;
    // fragment byte address of (buffer, slice): base ^ (buffer << 15) ^ (slice << 6); the loop keeps the
    // CURRENT buffer's base in acur / bcur and flips bit 15 once per K-tile
    int acur = a_base, bcur = b_base;

    // thresholds of this workgroup's 256 queries: LDS, read back per gallery tile by the epilogue
    float *ltau = reinterpret_cast<float *>(lcnt + BN * WARPS_M);
    // this wave's candidate ring (filter mode): passing scores are parked here by the K loop and handed to the global
    // candidate regions in bulk -- the K loop then holds no global store (whose completion its per-K-tile vmcnt(0) would
    // wait for) and no LDS atomic with a result to wait for
    RingEntry *ring = reinterpret_cast<RingEntry *>(ltau + BN) + wave * RING_SLOTS;
    int rcnt = 0, scnt = 0;                        // entries in the ring / in this wave's global spill area (wave-uniform: SGPRs)
#ifdef MIRX_EXP_NOEPI      // diagnostic (results wrong): no score passes, the candidate path never runs
    if (MODE == 0 && threadIdx.x < BN) ltau[threadIdx.x] = INFINITY;
#else
    if (MODE == 0 && threadIdx.x < BN) ltau[threadIdx.x] = A.tau[q_row0 + threadIdx.x];
#endif

    f32x4 acc[M_REP][N_REP];

    // FUSE (cosine filter): the threshold test of gallery tile t runs INSIDE the first K-slice of tile t + 1 -- right
    // before the MFMA pair that overwrites an accumulator tile pair, its 8 values are reduced (4 VALU), compared
    // with the two thresholds and, rarely, handed to emit_tile() -- so the matrix pipe does not idle for the
    // ~4 000 cycles per gallery tile that the stand-alone epilogue took while all 8 waves ran it together.
    // The last gallery tile of a workgroup (and every tile of the L2 / group-max kernels) uses epilogue().
#ifdef MIRX_GEMM_NOFUSE
    constexpr bool FUSE = false;
#else
    constexpr bool FUSE = MODE == 0 && !L2;
#endif

    // One parked entry -> the global candidate regions: its eight values against the two thresholds (the K loop only knows
    // that at least one passes).  No global atomics: the region (query, this workgroup's phase, this wave row) belongs to
    // this wave alone and its fill count lives in LDS; lanes that hold values of the same query claim slots with one LDS
    // atomic each (a plain load / store of the shared counter would be a data race across lanes).  Rows beyond the region's
    // slots go to the query's overflow list.
    auto push_entry = [&](const f32x4 e0, const f32x4 e1, unsigned tag) __attribute__((always_inline)) {
        const int el = tag & 63, mi = (tag >> 6) & 7, n0 = ((tag >> 9) & 1) * 2;
        const int64_t row0 = (int64_t)(tag >> 10) * BM + wm * WM_ROWS + mi * 16 + 4 * (el >> 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = j < 4 ? e0[j & 3] : e1[j & 3];
            const int qi = wn * 64 + (n0 + (j >> 2)) * 16 + (el & 15);
            const int64_t grow = (row0 + (j & 3)) * A.row_stride;
            // (rows beyond the gallery exist in its last, partial tile)
            if (v > ltau[qi] && grow < A.n_rows) {
                const int64_t qc = q_row0 + qi;
                const int c = atomicAdd(&lcnt[qi * WARPS_M + wm], 1);
                Cand cd;
                cd.s = v;
                cd.row = (int32_t)grow;
                if (c < A.slots) {
                    A.cand[(qc * A.regions + (ph * WARPS_M + wm)) * A.slots + c] = cd;
                } else {
                    const int p = atomicAdd(&A.ovf_cnt[qc], 1);
                    if (p < CAND_OVF) A.ovf[qc * CAND_OVF + p] = cd;
                }
            }
        }
    };
    // Everything parked so far goes out: at the END of the workgroup's life (about 500 entries per wave on the bench
    // workload: the K loop itself only parks), or between two gallery tiles should the spill area be more than half full.
    // The spill area is this wave's own and was written by this wave: its stores are complete after vmcnt(0), and it is
    // read past the vector L1.
    RingEntry *gspill = reinterpret_cast<RingEntry *>(A.spill) + ((int64_t)blockIdx.x * 8 + wave) * SPILL_SLOTS;
    auto drain = [&]() __attribute__((always_inline)) {
        if (__builtin_expect(scnt != 0, 0)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int i = lane; i < scnt; i += 64) {
                const unsigned *src = reinterpret_cast<const unsigned *>(gspill + i);
                unsigned w[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) w[k] = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f32x4 e0, e1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e0[k] = __uint_as_float(w[k]);
                    e1[k] = __uint_as_float(w[4 + k]);
                }
                push_entry(e0, e1, w[8]);
            }
            scnt = 0;
        }
        if (lane < rcnt) push_entry(ring[lane].a0, ring[lane].a1, ring[lane].tag);
        rcnt = 0;
    };

    // Candidate path of one PAIR of accumulator tiles (row tile mi, query tiles n0 and n0 + 1) in which `hit` lanes hold a
    // passing score: about two of a wave's 2 048 scores pass per gallery tile, but the slowest of the eight waves sets the
    // pace of a K-tile, so this is kept to a dozen instructions: the hit lanes park both tiles and a tag in the wave's ring
    // (ballot-ranked slots, two ds_write_b128 + one ds_write_b32); everything else happens in ring_flush().
    auto emit_pair = [&](int64_t gt_, int mi, int n0, const f32x4 a0, const f32x4 a1, bool hit, unsigned long long bm) __attribute__((always_inline)) {
        const int nh = __builtin_popcountll(bm);
        if (__builtin_expect(rcnt + nh > RING_SLOTS, 0)) {     // wave-uniform, rare: the ring moves to the wave's spill area
            if (lane < rcnt) gspill[scnt + lane] = ring[lane];
            scnt += rcnt;
            rcnt = 0;
        }
        if (hit) {
            RingEntry *e = ring + rcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bm, 0));
            e->a0 = a0;
            e->a1 = a1;
            e->tag = (unsigned)lane | (unsigned)mi << 6 | (unsigned)(n0 >> 1) << 9 | (unsigned)gt_ << 10;
        }
        rcnt = __builtin_amdgcn_readfirstlane(rcnt + nh);
    };
    // largest of an accumulator tile's four values in two instructions (fmaxf() would first canonicalise each
    // operand with a v_max_f32 x, x: four more instructions per tile in the fused check's issue budget)
#define MIRX_TILEMAX(T) vmax2(vmax3((T)[0], (T)[1], (T)[2]), (T)[3])

    auto epilogue = [&](int64_t gt_, int n_first) __attribute__((always_inline)) {
        int el = lane;
        asm volatile("" : "+v"(el));
        const int quad = el >> 4, col = el & 15;
        const int64_t tile_row0 = gt_ * BM + wm * WM_ROWS;
        float tau[N_REP];
#pragma unroll
        for (int ni = 0; ni < N_REP; ++ni) tau[ni] = MODE == 0 ? ltau[wn * 64 + ni * 16 + col] : 0.0f;
        if (L2) {
#pragma unroll
            for (int m0 = 0; m0 < M_REP; m0 += 2) {
                float b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    b[j] = A.gbias[(tile_row0 + (m0 + (j >> 2)) * 16 + 4 * quad + (j & 3)) * A.row_stride];
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int ni = 0; ni < N_REP; ++ni) acc[m0 + (j >> 2)][ni][j & 3] += b[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int ni = 0; ni < N_REP; ++ni) {
                float mx = -INFINITY;
#pragma unroll
                for (int mi = 0; mi < M_REP; ++mi) mx = fmaxf(mx, MIRX_TILEMAX(acc[mi][ni]));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (quad == 0)
                    A.groupmax[(q_row0 + wn * 64 + ni * 16 + col) * A.ngroups + gt_ * WARPS_M + wm] = mx;
            }
        } else {
#pragma unroll
            for (int n0 = 0; n0 < N_REP; n0 += 2) {
                if (n0 < n_first) continue;
                // maxima per row tile first: the candidate path then looks only inside row tiles that
                // hold a passing score (about two scores per wave and gallery tile pass, out of 2048)
                float mr0[M_REP], mr1[M_REP];
                float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
                for (int mi = 0; mi < M_REP; ++mi) {
                    mr0[mi] = MIRX_TILEMAX(acc[mi][n0]);
                    mr1[mi] = MIRX_TILEMAX(acc[mi][n0 + 1]);
                    mx0 = fmaxf(mx0, mr0[mi]);
                    mx1 = fmaxf(mx1, mr1[mi]);
                }
                if (__builtin_expect(__any((mx0 > tau[n0]) | (mx1 > tau[n0 + 1])), 0)) {
#pragma unroll
                    for (int mi = 0; mi < M_REP; ++mi) {
                        const bool hit = (mr0[mi] > tau[n0]) | (mr1[mi] > tau[n0 + 1]);
                        const unsigned long long bm = __ballot(hit);
                        if (bm) emit_pair(gt_, mi, n0, acc[mi][n0], acc[mi][n0 + 1], hit, bm);
                    }
                }
            }
        }
    };

    // Fragments: 8 A + 4 B, single-buffered and rolling (48 VGPRs).  A 32-deep slice runs as two halves:
    //   H1: all row tiles x query tiles 0,1; meanwhile B fragments 2,3 of THIS slice are read;
    //   H2: all row tiles x query tiles 2,3; after row tile mi's pair its A fragment is re-read for the
    //       NEXT slice, and B fragments 0,1 of the next slice are read.
    // Every ds_read has >= 12 MFMAs (192 cycles) between issue and first use.
    bf16x8 fa[M_REP], fb[N_REP];

    static_assert(A_TILE_BYTES == 32768 && BN * ROW_BYTES == 32768, "buffer toggle assumes 32 KiB tiles");
    // OTHER = 0: the current buffer, 1: the other one
#ifdef MIRX_EXP_NOLDS      // diagnostic (results wrong): fragments are never refreshed
#define MIRX_LDA(S, OTHER, MI) fa[MI]
#define MIRX_LDB(S, OTHER, NI) fb[NI]
#else
#define MIRX_LDA(S, OTHER, MI) (*reinterpret_cast<const bf16x8 *>(smem + (acur ^ ((OTHER) << 15) ^ ((S) << 6)) + (MI) * 16 * ROW_BYTES))
#define MIRX_LDB(S, OTHER, NI) (*reinterpret_cast<const bf16x8 *>(smem + (bcur ^ ((OTHER) << 15) ^ ((S) << 6)) + (NI) * 16 * ROW_BYTES))
#endif
    // Z = 1: the first slice of a gallery tile starts its 32 accumulator tiles from the constant 0 (no
    // 128 v_mov per tile)
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    // FUSE: where the threshold tests of a gallery tile's 32 accumulator tiles sit.  A SITE tests one tile pair (row tile mi,
    // query tiles n0 and n0 + 1): 4 VALU for the two maxima and two ballots against the two thresholds.  The sites of query
    // tiles 0,1 ride in the LAST K-tile's last H2 (L = 1: those tiles are final once its H1 has run, and nothing writes them
    // before the next gallery tile's first H1); the sites of query tiles 2,3 ride in the NEXT gallery tile's first H1 (Z = 1,
    // which writes only query tiles 0,1).  Inside its phase a tested tile pair is therefore NOT overwritten, so the branch on a
    // site's ballots is taken ONE SITE LATER (MIRX_TEST_PREV): the chain max -> compare -> SGPR -> scalar OR -> branch parks an
    // in-order wave for ~90 cycles when the branch follows the compare directly (measured: 87 cycles per site, both waves of a
    // SIMD doing the same), and costs an issue slot when the next site's maxima and MFMAs sit in between.  Variants measured
    // and dropped: all sixteen sites in one K-tile (it stretched from 2 400 to 4 150 cycles before a single score passed);
    // branch-free sites that record a bit per site and one flush per phase (the bookkeeping is the same SGPR round trip per
    // site); running maxima per phase or per group of row tiles with one flush that re-visits the sites (cheap sites, but the
    // re-visit runs in two phases of three on the slowest of eight waves: 6.2 vs 5.9 ms per launch).
    float ftau[N_REP] = {0.0f, 0.0f, 0.0f, 0.0f};
    int64_t gt_prev = 0;
    unsigned long long pend_a = 0, pend_b = 0;     // ballots of the previous site of the phase (wave-uniform)
#define MIRX_SITE_TEST(GT, MI, N0)                                                                               \
    if (__builtin_expect((pend_a | pend_b) != 0, 0)) {                                                           \
        const bool hit_ = (MIRX_TILEMAX(acc[MI][N0]) > ftau[N0]) | (MIRX_TILEMAX(acc[MI][(N0) + 1]) > ftau[(N0) + 1]); \
        emit_pair(GT, MI, N0, acc[MI][N0], acc[MI][(N0) + 1], hit_, pend_a | pend_b);                            \
    }
    // site MI: its maxima and ballots; then the test of site MI - 1 (whose ballots are a site old by now)
#define MIRX_CHECK2(GT, MI, N0)                                                                                  \
    {                                                                                                            \
        const float m0_ = MIRX_TILEMAX(acc[MI][N0]);                                                             \
        const float m1_ = MIRX_TILEMAX(acc[MI][(N0) + 1]);                                                       \
        const unsigned long long na_ = __ballot(m0_ > ftau[N0]), nb_ = __ballot(m1_ > ftau[(N0) + 1]);           \
        if constexpr ((MI) > 0) MIRX_SITE_TEST(GT, (MI) - 1, N0)                                                 \
        pend_a = na_;                                                                                            \
        pend_b = nb_;                                                                                            \
    }
    // the phase's last site (row tile 7) is tested behind the phase
#define MIRX_FLUSH_HITS(GT, N0)                                                                                  \
    {                                                                                                            \
        MIRX_SITE_TEST(GT, M_REP - 1, N0)                                                                        \
        pend_a = pend_b = 0;                                                                                     \
    }
#define MIRX_MFMA2Z(MI, N0, Z)                                                                         \
    if constexpr ((Z) && FUSE && (N0) == 0) MIRX_CHECK2(gt_prev, MI, 2)                                \
    acc[MI][N0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[MI], fb[N0], (Z) ? zero4 : acc[MI][N0], 0, 0, 0); \
    acc[MI][N0 + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[MI], fb[N0 + 1], (Z) ? zero4 : acc[MI][N0 + 1], 0, 0, 0);
#define MIRX_MFMA2(MI, N0) MIRX_MFMA2Z(MI, N0, 0)
    // L = 1 (last K-tile of a gallery tile, its last H2): query tiles 0,1 of row tile MI are final -- test them here
#define MIRX_MFMA2L(MI, L)                                                                             \
    if constexpr ((L) && FUSE) MIRX_CHECK2(gt, MI, 0)                                                  \
    MIRX_MFMA2Z(MI, 2, 0)
    // ---- LDS-DMA of the K-tile after next, one 1-KiB piece at a time --------------------------------
    // A wave's issue stalls for 60-180 cycles on every `buffer_load ... lds` piece, and right after the
    // workgroup barrier all eight waves would stall together while the MFMA pipes drain.  So the eight
    // pieces a wave owns per K-tile (4 of the gallery tile, 4 of the query tile) are spread over the ~30
    // MFMAs that follow the barrier, one piece per four MFMAs, and the two waves that share a SIMD
    // (wave w and w + 4) take alternate slots: while one stalls on a piece the other feeds the matrix
    // pipe.  A slot is a wave-uniform branch, which also pins it between the MFMA pairs around it.
    // The source of the K-tile in flight (this gallery tile, or the next one for the last two K-tiles)
    // is chosen at the barrier without a branch; past the very last K-tile the pieces re-read the
    // current tile into a buffer nobody consumes.
    const int grp = wave >> 2;
    __amdgpu_buffer_rsrc_t dma_rsrc;               // gallery-tile descriptor of the K-tile in flight
    int dma_koff;                                  // its K byte offset
    char *dma_a, *dma_b;                           // this wave's first piece slot of the destination buffers
#define MIRX_DMA_AT_BARRIER(KT)                                                     \
    {                                                                               \
        const int k2 = (KT) + 2;                                                    \
        const bool wrap = k2 >= nk;                                                 \
        dma_rsrc = wrap ? rsrc_a_nx : rsrc_a;                                       \
        dma_koff = (wrap ? k2 - nk : k2) * ROW_BYTES;                               \
        dma_a = smem + cur * A_TILE_BYTES + wave * 1024;                            \
        dma_b = smem + LDS_B0 + cur * B_TILE_BYTES + wave * 1024;                   \
    }
    // diagnostic builds: -DMIRX_EXP_SLOTS=0 no DMA slots at all, =1 slot branches present but never taken
    // (results wrong either way); -DMIRX_EXP_CYCLES prints K-loop cycles per K-tile
#ifndef MIRX_GEMM_A_AUX
#define MIRX_GEMM_A_AUX 0          // cache policy of the gallery DMA (2 = nt: stream past the L2-resident query tiles)
#endif
#if defined(MIRX_EXP_SLOTS) && MIRX_EXP_SLOTS == 1
#define MIRX_SLOT_COND(G) (grp == (G) && A.dimp < 0)
#else
#define MIRX_SLOT_COND(G) (grp == (G))
#endif
#if defined(MIRX_EXP_SLOTS) && MIRX_EXP_SLOTS == 0
#define MIRX_SLOT_A(G, P)
#define MIRX_SLOT_B(G, P)
#else
#define MIRX_SLOT_A_(G, P)                                                          \
    if (MIRX_SLOT_COND(G))                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dma_rsrc, LDS_PTR(dma_a + (P) * 8192), 16, voff_a, \
                                                 dma_koff + (P) * pstride_a, 0, MIRX_GEMM_A_AUX);
#define MIRX_SLOT_B_(G, P)                                                          \
    if (MIRX_SLOT_COND(G))                                                          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, LDS_PTR(dma_b + (P) * 8192), 16, voff_b, \
                                                 dma_koff + (P) * pstride_b, 0, 0);
#ifdef MIRX_EXP_NOA         // diagnostic (results wrong): no gallery DMA inside the loop
#define MIRX_SLOT_A(G, P)
#else
#define MIRX_SLOT_A(G, P) MIRX_SLOT_A_(G, P)
#endif
#ifdef MIRX_EXP_NOB         // diagnostic (results wrong): no query DMA inside the loop
#define MIRX_SLOT_B(G, P)
#else
#define MIRX_SLOT_B(G, P) MIRX_SLOT_B_(G, P)
#endif
#endif

    // H1 of a slice: query tiles 0,1 of every row tile; B fragments 2,3 of the same slice are read early
#define MIRX_H1_HEAD(OTHER, S, Z)                         \
    MIRX_MFMA2Z(0, 0, Z)                                  \
    fb[2] = MIRX_LDB(S, OTHER, 2);                        \
    fb[3] = MIRX_LDB(S, OTHER, 3);                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    // the A fragment of row tile MI is re-read one row tile late (after the NEXT pair of MFMAs), so the
    // ds_read never overwrites a register an MFMA issued just before is still reading
#define MIRX_H2_ROWZ(MI, NCUR, NS, Z)                     \
    MIRX_MFMA2Z(MI, 2, Z)                                 \
    fa[MI - 1] = MIRX_LDA(NS, NCUR, MI - 1);              \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#define MIRX_H2_ROWL(MI, NCUR, NS, L)                     \
    MIRX_MFMA2L(MI, L)                                    \
    fa[MI - 1] = MIRX_LDA(NS, NCUR, MI - 1);              \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    // H2 of a slice (query tiles 2,3) without DMA slots
#define MIRX_H2(NCUR, NS, Z)                              \
    MIRX_MFMA2Z(0, 2, Z)                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    \
    MIRX_H2_ROWZ(1, NCUR, NS, Z)                          \
    MIRX_H2_ROWZ(2, NCUR, NS, Z)                          \
    fb[0] = MIRX_LDB(NS, NCUR, 0);                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
    MIRX_H2_ROWZ(3, NCUR, NS, Z)                          \
    MIRX_H2_ROWZ(4, NCUR, NS, Z)                          \
    fb[1] = MIRX_LDB(NS, NCUR, 1);                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    \
    MIRX_H2_ROWZ(5, NCUR, NS, Z)                          \
    MIRX_H2_ROWZ(6, NCUR, NS, Z)                          \
    MIRX_H2_ROWZ(7, NCUR, NS, Z)                          \
    fa[7] = MIRX_LDA(NS, NCUR, 7);                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    // H2 of a K-tile's last slice: reads the next K-tile's first fragments and carries eight DMA slots
#define MIRX_H2_DMA(NCUR, NS, L)                          \
    MIRX_MFMA2L(0, L)                                     \
    MIRX_SLOT_A(0, 2)                                     \
    MIRX_H2_ROWL(1, NCUR, NS, L)                          \
    MIRX_SLOT_A(1, 2)                                     \
    MIRX_H2_ROWL(2, NCUR, NS, L)                          \
    fb[0] = MIRX_LDB(NS, NCUR, 0);                        \
    MIRX_SLOT_A(0, 3)                                     \
    MIRX_H2_ROWL(3, NCUR, NS, L)                          \
    MIRX_SLOT_A(1, 3)                                     \
    MIRX_H2_ROWL(4, NCUR, NS, L)                          \
    fb[1] = MIRX_LDB(NS, NCUR, 1);                        \
    MIRX_SLOT_B(0, 0)                                     \
    MIRX_H2_ROWL(5, NCUR, NS, L)                          \
    MIRX_SLOT_B(1, 0)                                     \
    MIRX_H2_ROWL(6, NCUR, NS, L)                          \
    MIRX_SLOT_B(0, 1)                                     \
    MIRX_H2_ROWL(7, NCUR, NS, L)                          \
    fa[7] = MIRX_LDA(NS, NCUR, 7);                        \
    MIRX_SLOT_B(1, 1)
    // one K-tile (held in buffer `cur`):
    //   slice 0: H1 (with the last four DMA slots of the K-tile in flight), H2
    //   slice 1: H1 | barrier: buffer `cur` is free, the other buffer's DMA has landed | H2 with DMA slots,
    //            reading the first fragments of the next K-tile from the other buffer.
    // No branch encloses an MFMA (see k_gemm).
#ifdef MIRX_EXP_CYCLES
#define MIRX_SEG(Z, I)                                                       \
    {                                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
        seg_sum[(Z) ? 0 : 1][I] += now_ - seg_t;                             \
        seg_t = now_;                                                        \
    }
#else
#define MIRX_SEG(Z, I)
#endif
#define MIRX_KTILE(KT, Z, L)                                                                       \
    MIRX_SEG(Z, 0)                                                                                 \
    MIRX_H1_HEAD(0, 0, Z)                                                                          \
    MIRX_SLOT_B(0, 2)                                                                              \
    MIRX_MFMA2Z(1, 0, Z)                                                                           \
    MIRX_SLOT_B(1, 2)                                                                              \
    MIRX_MFMA2Z(2, 0, Z)                                                                           \
    MIRX_SLOT_B(0, 3)                                                                              \
    MIRX_MFMA2Z(3, 0, Z)                                                                           \
    MIRX_SLOT_B(1, 3)                                                                              \
    MIRX_MFMA2Z(4, 0, Z) MIRX_MFMA2Z(5, 0, Z) MIRX_MFMA2Z(6, 0, Z) MIRX_MFMA2Z(7, 0, Z)            \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                             \
    MIRX_SEG(Z, 1)                                                                                 \
    if constexpr ((Z) && FUSE) MIRX_FLUSH_HITS(gt_prev, 2)                                         \
    MIRX_H2(0, 1, Z)                                                                               \
    MIRX_SEG(Z, 2)                                                                                 \
    MIRX_H1_HEAD(0, 1, 0)                                                                          \
    MIRX_MFMA2(1, 0) MIRX_MFMA2(2, 0) MIRX_MFMA2(3, 0) MIRX_MFMA2(4, 0)                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                             \
    MIRX_SEG(Z, 3)                                                                                 \
    MIRX_KBARRIER(); /* last reads of `cur` returned; the other buffer's DMA (mine) landed */      \
    MIRX_SEG(Z, 4)                                                                                 \
    MIRX_DMA_AT_BARRIER(KT)                                                                        \
    MIRX_SLOT_A(0, 0)                                                                              \
    MIRX_MFMA2(5, 0)                                                                               \
    MIRX_SLOT_A(1, 0)                                                                              \
    MIRX_MFMA2(6, 0)                                                                               \
    MIRX_SLOT_A(0, 1)                                                                              \
    MIRX_MFMA2(7, 0)                                                                               \
    MIRX_SLOT_A(1, 1)                                                                              \
    MIRX_H2_DMA(1, 0, L)                                                                           \
    if constexpr ((L) && FUSE) MIRX_FLUSH_HITS(gt, 0)                                              \
    MIRX_SEG(Z, 5)                                                                                 \
    acur ^= 32768;                                                                                 \
    bcur ^= 32768;                                                                                 \
    cur ^= 1;

    int64_t gt = ph;
    __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc_a(gt);

    stage_tile<BM>(smem, rsrc_a, voff_a, 0, pstride_a, wave);
    stage_tile<BN>(smem + LDS_B0, rsrc_b, voff_b, 0, pstride_b, wave);
    MIRX_KBARRIER();
    stage_tile<BM>(smem + A_TILE_BYTES, rsrc_a, voff_a, ROW_BYTES, pstride_a, wave);
    stage_tile<BN>(smem + LDS_B0 + B_TILE_BYTES, rsrc_b, voff_b, ROW_BYTES, pstride_b, wave);
#pragma unroll
    for (int mi = 0; mi < M_REP; ++mi) fa[mi] = MIRX_LDA(0, 0, mi);
#pragma unroll
    for (int ni = 0; ni < N_REP; ++ni) fb[ni] = MIRX_LDB(0, 0, ni);
    int cur = 0;                                       // nk is even: every gallery tile starts in buffer 0
    // "K-tile in flight" before the first barrier: K-tile 1 -> buffer 1, which the prologue above has
    // already requested in full; the first iteration's four late slots fetch the same bytes again
    dma_rsrc = rsrc_a;
    dma_koff = ROW_BYTES;
    dma_a = smem + A_TILE_BYTES + wave * 1024;
    dma_b = smem + LDS_B0 + B_TILE_BYTES + wave * 1024;
#ifdef MIRX_EXP_CYCLES
    unsigned long long cy_sum = 0, cy_n = 0, ep_sum = 0, ep_n = 0, k0_sum = 0, seg_sum[2][6] = {}, seg_t = 0;
    const unsigned long long life0 = __builtin_amdgcn_s_memtime();
    seg_t = life0;
#endif
    // Waves 4-7 are the younger SIMD partners and lose every issue arbitration against waves 0-3, which
    // then wait for them at each barrier: a static priority for the younger half evens the two out
    // (bench: 5.75 -> 5.54 ms per 4096-query launch).
    if (wave >= 4) __builtin_amdgcn_s_setprio(3);

    if constexpr (FUSE) {
        // this lane's four thresholds: constant for the workgroup's life (read once: a read per gallery tile waited for the
        // twelve fragment reads queued in front of it, ~500 cycles)
#pragma unroll
        for (int ni = 0; ni < N_REP; ++ni) ftau[ni] = ltau[wn * 64 + ni * 16 + (lane & 15)];
        // nothing of "the tile before the first" passes a threshold
        const f32x4 ninf4 = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int mi = 0; mi < M_REP; ++mi)
#pragma unroll
            for (int ni = 0; ni < N_REP; ++ni) acc[mi][ni] = ninf4;
    }

    for (;;) {
        const int64_t gt_nx = gt + plan.nph;
        const bool have_next = gt_nx < plan.ngt;
        const __amdgpu_buffer_rsrc_t rsrc_a_nx = make_rsrc_a(have_next ? gt_nx : gt);

#ifdef MIRX_EXP_CYCLES
        const unsigned long long cy0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (MODE == 0) {
            // (never on the bench workload: a tile parks ~2 entries per wave; the area holds 2048)
            if (__builtin_expect(scnt > SPILL_SLOTS - SPILL_PER_TILE, 0)) drain();
        }
        MIRX_KTILE(0, 1, 0)                            // the first slice initialises the accumulators
#ifdef MIRX_EXP_CYCLES
        k0_sum += __builtin_amdgcn_s_memtime() - cy0;
#endif
#pragma unroll 1
        for (int kt = 1; kt < nk - 1; ++kt) {
            MIRX_KTILE(kt, 0, 0)
        }
        MIRX_KTILE(nk - 1, 0, 1)                       // nk >= 2: the last K-tile is never the first
#ifdef MIRX_EXP_CYCLES
        cy_sum += __builtin_amdgcn_s_memtime() - cy0;
        cy_n += nk;
#endif

#ifdef MIRX_EXP_CYCLES
        const unsigned long long ep0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (!FUSE) epilogue(gt, 0);
#ifdef MIRX_EXP_CYCLES
        ep_sum += __builtin_amdgcn_s_memtime() - ep0;
        ep_n += 1;
#endif
        if (!have_next) break;
        gt_prev = gt;
        gt = gt_nx;
        rsrc_a = rsrc_a_nx;
    }
    if constexpr (FUSE) epilogue(gt, 2);               // the last gallery tile of this workgroup: its query tiles 2,3 are still untested
#ifdef MIRX_EXP_CYCLES
    if (lane == 0 && (blockIdx.x % 61) == 0 && cy_n > 1000 && (wave == 0 || wave == 4))
        printf("wg %d wave %d segments (loop top/prev end -> H1(s0) start | H1(s0) | H2(s0) | H1(s1) to barrier | barrier wait | rest): first K-tile %.0f %.0f %.0f %.0f %.0f %.0f ; other K-tiles %.0f %.0f %.0f %.0f %.0f %.0f\n",
               (int)blockIdx.x, wave, (double)seg_sum[0][0] / ep_n, (double)seg_sum[0][1] / ep_n, (double)seg_sum[0][2] / ep_n,
               (double)seg_sum[0][3] / ep_n, (double)seg_sum[0][4] / ep_n, (double)seg_sum[0][5] / ep_n,
               (double)seg_sum[1][0] / (cy_n - ep_n), (double)seg_sum[1][1] / (cy_n - ep_n), (double)seg_sum[1][2] / (cy_n - ep_n),
               (double)seg_sum[1][3] / (cy_n - ep_n), (double)seg_sum[1][4] / (cy_n - ep_n), (double)seg_sum[1][5] / (cy_n - ep_n));
    if (lane == 0 && (blockIdx.x % 61) == 0 && cy_n > 1000 && (wave == 0 || wave == 4))
        printf("wg %d wave %d: %.0f cycles per K-tile in the K loop (%llu K-tiles); first K-tile of a gallery tile (with the fused checks) %.0f, the others %.0f; kernel %.0f cycles per K-tile\n",
               (int)blockIdx.x, wave, (double)cy_sum / cy_n, cy_n, (double)k0_sum / ep_n,
               (double)(cy_sum - k0_sum) / (cy_n - ep_n), (double)(__builtin_amdgcn_s_memtime() - life0) / cy_n);
#endif
    if (MODE == 0) {
        drain();
        __syncthreads();
        for (int i = threadIdx.x; i < BN * WARPS_M; i += 512) {
            const int c = lcnt[i];
            if (c > 0) A.region_cnt[(q_row0 + i / WARPS_M) * A.regions + ph * WARPS_M + (i % WARPS_M)] = c;
        }
    }
#undef MIRX_KTILE
#undef MIRX_SEG
#undef MIRX_H1_HEAD
#undef MIRX_H2
#undef MIRX_H2_DMA
#undef MIRX_SLOT_A
#undef MIRX_SLOT_B
#undef MIRX_DMA_AT_BARRIER
#undef MIRX_H2_ROWL
#undef MIRX_MFMA2L
#undef MIRX_MFMA2
#undef MIRX_MFMA2Z
#undef MIRX_CHECK2
#undef MIRX_SITE_TEST
#undef MIRX_FLUSH_HITS
#undef MIRX_TILEMAX
#undef MIRX_H2_ROWZ
#undef MIRX_LDA
#undef MIRX_LDB
}

// MIRX_GEMM_MFMA=32 selects the 32x32x16 kernel for the 256-query tile (development A/B switch)
bool use_mfma16() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MIRX_GEMM_MFMA");
        v = (e && e[0] == '3') ? 0 : 1;
    }
    return v == 1;
}

int device_cus() { return current_device_cus(); }

template <int BN, int MODE>
hipError_t launch_bn(const GemmArgs &a0, hipStream_t st) {
    GemmArgs a = a0;
    const size_t lds = 2 * (size_t)A_TILE_BYTES + 2 * (size_t)BN * ROW_BYTES +
                       (MODE == 0 ? (size_t)BN * (8 / (BN / 64)) * sizeof(int) + (size_t)BN * sizeof(float) +
                                        (BN == 256 ? (size_t)8 * RING_SLOTS * sizeof(RingEntry) : 0) : 0);
    const Plan plan = make_plan(a.n_rows, a.nq_pad, BN, device_cus() / 8 * 8);
    if (plan.ngt <= 0 || plan.nqt <= 0) return hipSuccess;
    a.nph = plan.nph;
    if (MODE == 0 && (a.regions != plan.nph * (8 / (BN / 64)) || a.slots != region_slots(a.regions)))
        return hipErrorInvalidValue;
    // a tile's byte span must fit the 32-bit buffer offsets
    if ((int64_t)BM * a.row_stride * a.dimp * 2 > 0x7FFFFFFF) return hipErrorInvalidValue;
    if (plan.ngt >= ((int64_t)1 << 22)) return hipErrorInvalidValue;      // k_gemm16's ring tags hold the gallery tile in 22 bits
    if (MODE == 0 && BN == 256 && !a.spill) return hipErrorInvalidValue;
    hipError_t e;
    if constexpr (BN == 256) {
        if (use_mfma16()) {
            if (a.gbias) {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm16<MODE, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((k_gemm16<MODE, true>), dim3((unsigned)plan.grid), dim3(512), lds, st, a);
            } else {
                e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm16<MODE, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL((k_gemm16<MODE, false>), dim3((unsigned)plan.grid), dim3(512), lds, st, a);
            }
            return hipGetLastError();
        }
    }
    if (a.gbias) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm<BN, MODE, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_gemm<BN, MODE, true>), dim3((unsigned)plan.grid), dim3(512), lds, st, a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm<BN, MODE, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_gemm<BN, MODE, false>), dim3((unsigned)plan.grid), dim3(512), lds, st, a);
    }
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_mode(const GemmArgs &a, int bn, hipStream_t st) {
    if (a.dimp % (2 * BK) || a.nq_pad % bn) return hipErrorInvalidValue;
    switch (bn) {
        case 64: return launch_bn<64, MODE>(a, st);
        case 128: return launch_bn<128, MODE>(a, st);
        case 256: return launch_bn<256, MODE>(a, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace

size_t gemm_spill_bytes() { return (size_t)(device_cus() / 8 * 8) * 8 * SPILL_SLOTS * sizeof(RingEntry); }

int gemm_query_tile(int64_t nq) { return nq <= 64 ? 64 : (nq <= 128 ? 128 : 256); }
int gemm_groups_per_tile(int bn) { return 8 / (bn / 64); }

void gemm_plan(int64_t n_rows, int64_t nq_pad, int bn, int *regions, int *slots) {
    const Plan plan = make_plan(n_rows, nq_pad, bn, device_cus() / 8 * 8);
    *regions = plan.nph * (8 / (bn / 64));
    *slots = region_slots(*regions);
}

hipError_t launch_gemm_filter(const GemmArgs &a, int bn, hipStream_t st) { return launch_mode<0>(a, bn, st); }
hipError_t launch_gemm_groupmax(const GemmArgs &a, int bn, hipStream_t st) { return launch_mode<1>(a, bn, st); }

}  // namespace mirx
