// k_conv3x3_d2s.hip -- the dense layer's 3x3 convolution of k_conv3x3_d2h.hip (k_conv3x3_d2p) for SMALL launches: the
// reference's own batch sizes (one image per query in milvus/milvus_retrieval.py:53-66, 32 in ingest_embeddings.py:465-469,
// 64 in test.py:1513).
//
// k_conv3x3_d2p gives a 4-wave workgroup a strip of 196-224 output pixels (a whole 14 x 14 image, four 7 x 7 images) and
// walks 8 stages x 9 taps behind a barrier per stage: at B = 1 a layer of the 14 x 14 / 7 x 7 maps is ONE workgroup
// (12 us, 58 layers per forward) and at B = 64 a quarter of the chip.
//
// Here ONE WAVE owns one block of 32 output pixels of one image (all 32 output channels = one 32 x 32 accumulator tile)
// and runs the same 72 (stage, tap) steps with no LDS and no barrier: per step a lane loads 16 bytes of each weight term
// (row lane & 31 of the (stage, tap, term) plane, channel half lane >> 5) and 16 bytes of each activation term (the record
// of its pixel shifted by the tap; a buffer load past num_records returns the zero of the padding ring), six steps ahead
// in a register ring.  7 x more workgroups than images on the 14 x 14 map.
//
// Bit-identical to k_conv3x3_d2p by construction: an output is accumulated by the same v_mfma_f32_32x32x16_f16 sequence
// (stages in order, taps in order, per tap lo_w hi_x, hi_w lo_x, hi_w hi_x) on the same fragments, and the epilogue is the
// same multiply.  tests/test_model_gpu.py runs both on one input and compares bits.
#include <atomic>

#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int CIN = 128, COUT = 32, KC = 16, NST = CIN / KC;
constexpr int DEPTH = 6;                        // (stage, tap) steps of loads in flight
constexpr int NSTEP = NST * 9;                  // 72
static_assert(NSTEP % (3 * DEPTH) == 0 && (3 * DEPTH) % 9 == 0, "the unrolled body covers whole stages and whole ring turns");

template <int W>
__global__ __launch_bounds__(64) void k_conv3x3_d2s(const uint16_t *__restrict__ yt, const uint16_t *__restrict__ w3,
                                                    const float *__restrict__ oscale, float *__restrict__ out, int64_t out_bs,
                                                    const float *__restrict__ in_inv, unsigned *__restrict__ out_range,
                                                    int64_t out_ps) {
    constexpr int HW = W * W;
    constexpr unsigned IMG_BYTES = 16u * HW * 32u;           // 8 groups x 2 terms x HW pixels x 32 B
    constexpr unsigned STAGE_BYTES = 2u * HW * 32u;
    constexpr unsigned TERM_BYTES = HW * 32u;
    const int lane = threadIdx.x & 63, half = lane >> 5, n = lane & 31;
    const int64_t img = blockIdx.y;
    int p = blockIdx.x * 32 + n;
    const bool live = p < HW;
    if (!live) p = HW - 1;                                   // idle lanes shadow a valid pixel (never stored)
    const int py = p / W, px = p % W;

    const __amdgpu_buffer_rsrc_t yrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(yt + img * (int64_t)(IMG_BYTES / 2)), 0, IMG_BYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)w3, 0, NST * 9 * 2 * COUT * KC * 2, 0x00020000);
    // this lane's activation offset per tap inside a (stage, term) plane, or "out of range" for the zero ring
    unsigned xoff[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = py + tap / 3 - 1, ix = px + tap % 3 - 1;
        const bool inside = iy >= 0 && iy < W && ix >= 0 && ix < W;
        xoff[tap] = inside ? (unsigned)((iy * W + ix) * 32 + half * 16) : 0x80000000u;      // + a stage offset: still out of range, no wrap
    }
    const unsigned woff = (unsigned)(n * 32 + half * 16);    // inside a (stage, tap, term) plane of 1 KiB

    u32x4 ra[DEPTH][2], rb[DEPTH][2];
    auto issue = [&](int step, int slot) __attribute__((always_inline)) {
        const int st = step / 9, tap = step % 9;
        const unsigned wbase = (unsigned)((st * 9 + tap) * 2) * 1024u;
        ra[slot][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, wbase, 0));
        ra[slot][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, wbase + 1024u, 0));
        // the stage offset goes into the per-lane offset (what the range check certainly covers); the term into soffset
        const unsigned xo = xoff[tap] + (unsigned)st * STAGE_BYTES;
        rb[slot][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, xo, 0, 0));
        rb[slot][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, xo, TERM_BYTES, 0));
    };
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) issue(s, s);
    __builtin_amdgcn_sched_barrier(0);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // (sched_barrier: hipcc's scheduler otherwise sinks every load to just in front of its use -- one round trip per step)
#pragma unroll 1
    for (int s0 = 0; s0 < NSTEP; s0 += 3 * DEPTH) {
#pragma unroll
        for (int u = 0; u < 3 * DEPTH; ++u) {
            const int slot = u % DEPTH;
            const f16x8 a0 = __builtin_bit_cast(f16x8, ra[slot][0]), a1 = __builtin_bit_cast(f16x8, ra[slot][1]);
            const f16x8 b0 = __builtin_bit_cast(f16x8, rb[slot][0]), b1 = __builtin_bit_cast(f16x8, rb[slot][1]);
            // refill the slot (clamped: a load is never behind a branch; the last steps re-load the last one)
            const int nx = s0 + u + DEPTH;
            // (s0 advances by 18 = two whole stages, so (s0 + u) % 9 == u % 9 at compile time)
            {
                const int stn = nx < NSTEP ? nx / 9 : NST - 1;
                const int tapn = (u + DEPTH) % 9;
                const unsigned wbase = (unsigned)((stn * 9 + (nx < NSTEP ? tapn : 8)) * 2) * 1024u;
                const unsigned xo = (nx < NSTEP ? xoff[tapn] : xoff[8]) + (unsigned)stn * STAGE_BYTES;
                ra[slot][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, wbase, 0));
                ra[slot][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff, wbase + 1024u, 0));
                rb[slot][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, xo, 0, 0));
                rb[slot][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrsrc, xo, TERM_BYTES, 0));
            }
            // smallest terms first, as k_conv3x3_d2p
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- outputs straight from the accumulator: register r = channel 8 (r >> 2) + (r & 3) + 4 half, lane = pixel ----
    float vmax = 0.f;
    if (live) {
        const float x_inv = in_inv[img];                                   // 2^-t of this image (oscale * 2^-t is exact)
        float *op = out + img * out_bs + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = 8 * (r >> 2) + (r & 3) + 4 * half;
            const float v = acc[r] * (oscale[oc] * x_inv);
            vmax = range_max(vmax, v);
            op[(int64_t)oc * out_ps] = v;
        }
    }
    if (out_range) range_publish(out_range, (int)img, vmax, lane);
}

std::atomic<int> g_small_max_wg{96};

}  // namespace

void set_conv3x3_small_max_wg(int v) { g_small_max_wg.store(v < 0 ? 0 : v); }
int conv3x3_small_max_wg() { return g_small_max_wg.load(std::memory_order_relaxed); }

// called by launch_conv3x3_d2p (k_conv3x3_d2h.hip) for launches that would leave most CUs without a workgroup
hipError_t launch_conv3x3_d2s(const uint16_t *yt, const uint16_t *w2, const float *oscale, int64_t n, int side, float *out,
                              int64_t out_bs, const float *in_inv, float *out_range, int64_t out_ps, hipStream_t st) {
    unsigned *orr = reinterpret_cast<unsigned *>(out_range);
    const dim3 grid((unsigned)((side * side + 31) / 32), (unsigned)n);
#define MIRX_D2S(Wd) hipLaunchKernelGGL((k_conv3x3_d2s<Wd>), grid, dim3(64), 0, st, yt, w2, oscale, out, out_bs, in_inv, orr, out_ps)
    if (side == 56) MIRX_D2S(56);
    else if (side == 28) MIRX_D2S(28);
    else if (side == 14) MIRX_D2S(14);
    else if (side == 7) MIRX_D2S(7);
    else return hipErrorInvalidValue;
#undef MIRX_D2S
    return hipGetLastError();
}

}  // namespace mirx
