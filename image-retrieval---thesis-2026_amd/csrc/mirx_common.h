// mirx_common.h -- shared host/device helpers for libmirx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/mirx.h"

// ---- diagnostic builds ----------------------------------------------------------------------------------------------
// A few kernels carry arms that give WRONG RESULTS on purpose (a phase skipped, raw bits stored instead of values, operands
// never refreshed) or add cycle stamps / printf, to attribute a kernel's time.  The macros below select them, and they are
// honoured only together with -DMIRX_DIAG: the shipped library never defines it (csrc/Makefile refuses a CXXFLAGS that
// carries any -DMIRX_ switch; the diag-* targets build such objects into exp/ under other names).
#if !defined(MIRX_DIAG) &&                                                                                             \
    (defined(MIRX_EXP_NODMA) || defined(MIRX_EXP_NOBAR) || defined(MIRX_EXP_NOEPI) || defined(MIRX_EXP_NOLDS) ||         \
     defined(MIRX_EXP_SLOTS) || defined(MIRX_EXP_SAMEA) || defined(MIRX_EXP_NOA) || defined(MIRX_EXP_NOB) ||             \
     defined(MIRX_EXP_CYCLES) || defined(MIRX_C1H2_EXP_SKIP) || defined(MIRX_C1H2_EXP_SPLIT) ||                          \
     defined(MIRX_C1H2_EXP_ONE_MFMA) || defined(MIRX_C1H2_STAMPS) || defined(MIRX_LT2_EXP) || defined(MIRX_LH2_EXP) ||   \
     defined(MIRX_DF_EXP) || defined(MIRX_DF_STAMPS) || defined(MIRX_DW_STAMPS) || defined(MIRX_STEM_EXP) ||             \
     defined(MIRX_W3_CYCLES) || defined(MIRX_D2P_WAVES) || defined(MIRX_STEM_PLAIN_ORDER) || defined(MIRX_ATT_EXP))
#error "a MIRX diagnostic switch without -DMIRX_DIAG: these arms give wrong results or alter timing and must never reach libmirx.so"
#endif

namespace mirx {

// ---- error plumbing -------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

#define MIRX_HIP(expr)                                                                     \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess)                                                             \
            return ::mirx::fail(MIRX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

#define MIRX_CHECK(cond, msg)                                   \
    do {                                                        \
        if (!(cond)) return ::mirx::fail(MIRX_EINVAL, (msg));   \
    } while (0)

// ---- sizes ----------------------------------------------------------------------------
constexpr int WAVE = 64;
constexpr int DIM_ALIGN = 128;       // rows are stored padded to a multiple of 128 elements (2 GEMM K-steps)
constexpr int ROW_ALIGN = 256;       // gallery capacity is a multiple of the GEMM M tile
constexpr int CAND_CAP = 1024;       // per-query candidate capacity of the threshold filter
constexpr int CAND_OVF = 256;        // per-query shared overflow list (global atomics, rare)
constexpr int MAX_K = 1024;
constexpr int MAX_DIMP = 4096;

struct Cand {          // one row that passed the bf16 threshold filter
    float s;           // approximate score (bf16 MFMA, fp32 accumulate [+ bias])
    int32_t row;       // gallery row
};

struct Hit {           // one exactly scored row
    double s;          // fp64 ranking score (lane-tree order)
    int64_t id;
};

// `a` ranks before `b`: higher score first, then lower id.
__host__ __device__ inline bool hit_before(double sa, int64_t ia, double sb, int64_t ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// CUs of the CURRENT device, cached per device (a process that drives several GPUs sizes every persistent grid for the one it
// launches on: ADVICE r3)
inline int current_device_cus() {
    static int cache[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cache[dev] == 0) {
        int v = 256;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        cache[dev] = v > 0 ? v : 256;
    }
    return cache[dev];
}
// "has this launcher's function attribute been set on the current device": one bit per device in a per-call-site mask
inline bool first_use_on_device(unsigned long long &mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;      // unknown: set the attribute again (cheap)
    if (mask >> dev & 1ull) return false;
    mask |= 1ull << dev;
    return true;
}

#ifdef __HIPCC__
// ---- device helpers -------------------------------------------------------------------
__device__ inline int lane_id() { return threadIdx.x & 63; }

__device__ inline double wave_butterfly_sum(double v) {
    // s[l] = s[l] + s[l ^ off], off = 32..1 : the order search_ref.c pins.
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// fp32 -> bf16, round to nearest even (finite inputs).
__device__ inline uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// Lane-tree fp64 score of two fp32 rows of `dimp` (multiple of 64) elements.
// METRIC 0: sum q*g ; METRIC 1: -(sum (q-g)^2).  All lanes return the same value.
template <int METRIC>
__device__ inline double lane_tree_score(const float *__restrict__ q, const float *__restrict__ g,
                                         int dimp) {
    const int lane = lane_id();
    const int nchunk = dimp >> 2;
    double acc = 0.0;
    for (int c = lane; c < nchunk; c += WAVE) {
        const float4 a = *reinterpret_cast<const float4 *>(q + 4 * c);
        const float4 b = *reinterpret_cast<const float4 *>(g + 4 * c);
        if (METRIC == 0) {
            acc = fma((double)a.x, (double)b.x, acc);
            acc = fma((double)a.y, (double)b.y, acc);
            acc = fma((double)a.z, (double)b.z, acc);
            acc = fma((double)a.w, (double)b.w, acc);
        } else {
            double d;
            d = (double)a.x - (double)b.x; acc = fma(d, d, acc);
            d = (double)a.y - (double)b.y; acc = fma(d, d, acc);
            d = (double)a.z - (double)b.z; acc = fma(d, d, acc);
            d = (double)a.w - (double)b.w; acc = fma(d, d, acc);
        }
    }
    acc = wave_butterfly_sum(acc);
    return METRIC == 0 ? acc : -acc;
}
// tanh-form GELU (transformers "gelu_pytorch_tanh", the SigLIP MLP activation): 0.5 v (1 + tanh(u)), u = sqrt(2/pi) (v + 0.044715 v^3).
// 1 + tanh(u) = 2 / (1 + e^(-2u)) exactly, so the value is v / (1 + 2^(v (K1 + K2 v^2))) with K1 = -2 sqrt(2/pi) log2(e), K2 =
// 0.044715 K1: one v_exp_f32 and one v_rcp_f32 instead of tanhf's ~25 instructions (the Linear epilogues are VALU-bound on their
// activation).  Against float64 over [-12, 12]: 7.4e-7 absolute, 1.3e-6 relative where |value| > 1e-3 -- the tanhf form measures
// 6.7e-7 and 5e-5 (it cancels in 1 + tanh for negative arguments).
__device__ inline float gelu_tanh(float v) {
    const float a = fmaf(v * v, -0.10294324159622192f, -2.302208185195923f);
    return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * a));
}

// erf-form GELU (torch.nn.GELU()).  A branch-free erf (two fitted polynomials evaluated for every value so that neighbours pair
// into packed fp32 instructions, tools/fit_gelu.py) was built and measured: DINOv2 1 031 -> 1 032 img/s, ConvNeXtV2 2 049 ->
// 2 021 -- ocml's erff mostly runs ONE of its branches per wave (|z| < 1 for most activations), which is cheaper than both
// polynomials at half price.  Kept: erff.
__device__ inline float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }

// ---- value ranges that travel with activations (two-fp16-term kernels) ---------------------------------
// PER IMAGE: a buffer's range is one fp32 per image (a "range row" [n], zeroed once per forward).  A producer folds the
// largest |value| it wrote for image b into row[b] with an unsigned atomic max: the bit patterns of non-negative floats
// order like the floats, +inf and NaN sort above every finite value, so a non-finite activation makes THAT image's
// consumer scale NaN and its embedding NaN (loud, never a silently wrong finite number) and leaves its batch mates
// untouched.  A consumer reads row[b] and derives the power-of-two staging scale of image b: an image's arithmetic does
// not depend on what else is in the batch.
__device__ inline float range_max(float m, float v) {
    // max(m, |v|) that keeps a NaN (fmaxf would drop it)
    const float a = fabsf(v);
    return (a > m || a != a) ? a : m;
}

// every lane of the wave belongs to image `img` (wave-uniform): one atomic per wave
__device__ inline void range_publish(unsigned *__restrict__ row, int img, float vmax, int lane) {
    unsigned a = __float_as_uint(vmax);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)a, off, 64);
        a = o > a ? o : a;
    }
    if (lane == 0 && a) atomicMax(row + img, a);
}

// lanes may belong to different images (a pixel tile that straddles images; a lane that carries nothing passes a valid
// image index and vmax = 0): one atomic per DISTINCT image of the wave -- the images are peeled off one at a time (lowest
// pending lane's image, masked wave maximum), two or three rounds at most for any tile geometry in this library
__device__ inline void range_publish_lanes(unsigned *__restrict__ row, int img, float vmax, int lane) {
    const unsigned a = __float_as_uint(vmax);
    bool pending = true;
    for (;;) {
        const unsigned long long bm = __ballot(pending);           // wave-uniform
        if (!bm) break;
        const int src = __ffsll((long long)bm) - 1;
        const int cur = __shfl(img, src, 64);
        const bool mine = pending && img == cur;
        unsigned m = mine ? a : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)m, off, 64);
            m = o > m ? o : m;
        }
        if (lane == src && m) atomicMax(row + cur, m);
        pending = pending && !mine;
    }
}

// bound >= every |value|: x_scale = 2^(14 - floor(log2 bound)) puts bound * x_scale in [2^14, 2^15) (fp16 overflows at
// 65504); inv = 1 / x_scale.  bound == 0 (or subnormal) -> 1; non-finite -> NaN.
__device__ inline void range_scales(float bound, float &x_scale, float &inv) {
    const unsigned u = __float_as_uint(bound);
    const int e = (int)((u >> 23) & 0xffu) - 127;
    if (!(bound < 3.0e38f) || (u >> 31)) {
        x_scale = inv = __uint_as_float(0x7fc00000u);
    } else if (e < -100) {
        x_scale = inv = 1.f;
    } else {
        x_scale = __uint_as_float((unsigned)(127 + 14 - e) << 23);
        inv = __uint_as_float((unsigned)(127 - 14 + e) << 23);
    }
}

// The two fp16 terms of a pair of fp32 values: hi = RNE(v) (one v_cvt_pk_f16_f32), lo = RNE(v - hi) where v - hi comes from
// v_fma_mix_f32, which reads the fp16 half in place (exact: the difference of a float and its fp16 rounding is a float).  The
// compiler's form of `v - float(hi)` converts hi back with an SDWA instruction per value and subtracts with a packed fp32 op:
// five instructions per pair, two of them the kind that cost 10+ cycles beside an MFMA stream; this is four plain ones.
__device__ inline void split2h_pair(float v0, float v1, unsigned &hi, unsigned &lo) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_;
    const f32x2_ vv = {v0, v1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(vv, f16x2_));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(v1));
    const f32x2_ rr = {r0, r1};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(rr, f16x2_));
}

#endif  // __HIPCC__

}  // namespace mirx
