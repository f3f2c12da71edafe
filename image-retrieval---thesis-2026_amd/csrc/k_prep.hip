// k_prep.hip -- streaming (HBM-bound) kernels: gallery ingest, query preparation,
// L2 normalisation and the fused BN+ReLU+GAP+L2-norm embedding head.
//
// Reference behaviour replaced (paths into /root/reference):
//   F.normalize(x, dim=1)                       model.py:83,116,493,634; milvus_retrieval.py:63
//   norm5 -> relu -> AdaptiveAvgPool2d(1) -> flatten -> normalize      model.py:59-60,73-74,83
// All kernels: one 64-lane wave per row, 16-byte loads, wave-shuffle reductions.
#include "mirx_kernels.h"

#include <algorithm>

namespace mirx {

namespace {

// Sum of squares of a row in the lane-tree order (fp64), any dim (elements past dim = 0).
__device__ inline double row_sumsq_lane_tree(const float *__restrict__ x, int dim) {
    const int lane = lane_id();
    const int nchunk = (dim + 3) >> 2;
    const bool vec = ((dim & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    double acc = 0.0;
    for (int c = lane; c < nchunk; c += WAVE) {
        float v[4];
        if (vec) {
            const float4 t = *reinterpret_cast<const float4 *>(x + 4 * c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (4 * c + e < dim) ? x[4 * c + e] : 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fma((double)v[e], (double)v[e], acc);
    }
    return wave_butterfly_sum(acc);
}

// fp32 upper bound of sqrt(ss) (ss >= 0): round to nearest, then one ulp-ish up.
__device__ inline float norm_upper(double ss) {
    float f = (float)sqrt(ss);
    return f * 1.0000002f + 1e-30f;
}

__global__ __launch_bounds__(256) void k_ingest(const float *__restrict__ src, int64_t n, int dim,
                                                int dimp, float *__restrict__ g32,
                                                uint16_t *__restrict__ g16,
                                                float *__restrict__ gbias,
                                                unsigned *__restrict__ gnorm_max_bits, int metric) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = lane_id();
    const float *s = src + row * dim;
    float *d32 = g32 + row * dimp;
    uint16_t *d16 = g16 + row * dimp;
    for (int e = lane; e < dimp; e += WAVE) {
        const float v = e < dim ? s[e] : 0.0f;
        d32[e] = v;
        d16[e] = f32_to_bf16(v);
    }
    const double ss = row_sumsq_lane_tree(s, dim);
    if (lane == 0) {
        gbias[row] = metric == MIRX_METRIC_NEG_L2 ? (float)(-0.5 * ss) : 0.0f;
        atomicMax(gnorm_max_bits, __float_as_uint(norm_upper(ss)));
    }
}

__global__ __launch_bounds__(256) void k_prep_queries(const float *__restrict__ q, int64_t nq,
                                                      int64_t nq_pad, int dim, int dimp,
                                                      float *__restrict__ q32p,
                                                      uint16_t *__restrict__ q16,
                                                      float *__restrict__ qnorm) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nq_pad) return;
    const int lane = lane_id();
    float *d32 = q32p + row * dimp;
    uint16_t *d16 = q16 + row * dimp;
    if (row >= nq) {
        for (int e = lane; e < dimp; e += WAVE) { d32[e] = 0.0f; d16[e] = 0; }
        if (lane == 0) qnorm[row] = 0.0f;
        return;
    }
    const float *s = q + row * dim;
    for (int e = lane; e < dimp; e += WAVE) {
        const float v = e < dim ? s[e] : 0.0f;
        d32[e] = v;
        d16[e] = f32_to_bf16(v);
    }
    const double ss = row_sumsq_lane_tree(s, dim);
    if (lane == 0) qnorm[row] = norm_upper(ss);
}

__global__ __launch_bounds__(256) void k_l2_normalize(float *__restrict__ x, int64_t n, int dim) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float *r = x + row * dim;
    const double ss = row_sumsq_lane_tree(r, dim);
    double nrm = sqrt(ss);
    if (nrm < 1e-12) nrm = 1e-12;
    for (int e = lane_id(); e < dim; e += WAVE) r[e] = (float)((double)r[e] / nrm);
}

// One workgroup per image.  Pass 1: one wave per channel sums relu(x*scale+shift) over the
// hw contiguous pixels (coalesced, shuffle reduce).  Pass 2: L2-normalise the c means.
constexpr int HEAD_THREADS = 1024;     // 16 waves: at one workgroup per image the channel loop is the launch's latency (B = 64: 144 -> ~40 us)
__global__ __launch_bounds__(HEAD_THREADS) void k_head(const float *__restrict__ x,
                                              const float *__restrict__ scale,
                                              const float *__restrict__ shift, int c, int hw,
                                              int normalize, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float mean_s[];   // [cpad] floats + 1 double
    const int64_t b = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const float *xb = x + b * (int64_t)c * hw;
    const float inv = 1.0f / (float)hw;
    // a wave takes 8 channels at a time so that 8 loads are in flight per lane (one channel per trip made a B = 1 forward wait
    // 256 dependent L2 round trips here: 120 us of its 2.4 ms).  relu as a compare: a NaN stays a NaN (fmaxf would drop it and
    // a poisoned image would come out as a finite embedding).
    constexpr int UC = 8;
    for (int ch0 = wave * UC; ch0 < c; ch0 += (HEAD_THREADS / WAVE) * UC) {
        float acc[UC];
#pragma unroll
        for (int u = 0; u < UC; ++u) {
            const int ch = ch0 + u < c ? ch0 + u : c - 1;
            const float sc = scale ? scale[ch] : 1.0f, sh = shift ? shift[ch] : 0.0f;
            const float *p = xb + (int64_t)ch * hw;
            acc[u] = 0.0f;
            for (int i = lane; i < hw; i += WAVE) {
                const float t = fmaf(p[i], sc, sh);
                acc[u] += t < 0.0f ? 0.0f : t;
            }
        }
#pragma unroll
        for (int u = 0; u < UC; ++u) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc[u] += __shfl_xor(acc[u], off, 64);
            if (lane == 0 && ch0 + u < c) mean_s[ch0 + u] = acc[u] * inv;
        }
    }
    const int cpad = (c + 3) & ~3;
    double *nrm_s = reinterpret_cast<double *>(mean_s + cpad);
    for (int i = c + threadIdx.x; i < cpad; i += HEAD_THREADS) mean_s[i] = 0.0f;
    __syncthreads();
    if (normalize) {
        if (wave == 0) {
            const double ss = row_sumsq_lane_tree(mean_s, c);
            double nrm = sqrt(ss);
            if (nrm < 1e-12) nrm = 1e-12;
            if (lane == 0) *nrm_s = nrm;
        }
        __syncthreads();
        const double nrm = *nrm_s;
        for (int i = threadIdx.x; i < c; i += HEAD_THREADS) y[b * c + i] = (float)((double)mean_s[i] / nrm);
    } else {
        for (int i = threadIdx.x; i < c; i += HEAD_THREADS) y[b * c + i] = mean_s[i];
    }
}

// relu(x*scale+shift) over the channel-prefix slab of each image: the slab [c*hw] is contiguous,
// so lanes stream it with 16-B loads; the channel of an element is e / hw.
__global__ __launch_bounds__(256) void k_bn_relu(const float *__restrict__ x, int64_t xbs,
                                                 const float *__restrict__ scale,
                                                 const float *__restrict__ shift, int c, int hw,
                                                 float *__restrict__ y) {
    const int64_t b = blockIdx.y;
    const int64_t slab = (int64_t)c * hw;                 // multiple of 4 (checked by the launcher)
    const float *xb = x + b * xbs;
    float *yb = y + b * slab;
    const float inv_hw = 1.0f / (float)hw;
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < slab; e += (int64_t)gridDim.x * 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(xb + e);
        int ch = (int)((float)e * inv_hw);                // estimate, then fix up by at most one
        while ((int64_t)(ch + 1) * hw <= e) ++ch;
        while ((int64_t)ch * hw > e) --ch;
        float in[4] = {v.x, v.y, v.z, v.w}, out[4];
        int64_t next = (int64_t)(ch + 1) * hw;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (e + j >= next) { ++ch; next += hw; }
            out[j] = fmaxf(fmaf(in[j], scale[ch], shift[ch]), 0.0f);
        }
        *reinterpret_cast<float4 *>(yb + e) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// avgpool2x2(relu(bn(x))): one thread per pair of horizontally adjacent outputs = two 16-byte loads and
// one 8-byte store; items are numbered (channel, output row, output column pair) inside an image (blockIdx.y).
// W > 0: the map is W x W with W known at compile time (DenseNet's 56 / 28 / 14: the index arithmetic is then a few
// 32-bit multiplies; with run-time sizes its three 64-bit divisions cost more than the two loads), W = 0: any even h, w.
template <int W>
__global__ __launch_bounds__(256) void k_bn_relu_avgpool2(const float *__restrict__ x, int64_t xbs,
                                                          const float *__restrict__ scale,
                                                          const float *__restrict__ shift, int c, int h_,
                                                          int w_, unsigned items, float *__restrict__ y, int64_t xps,
                                                          int64_t ybs) {
    const int h = W ? W : h_, w = W ? W : w_;
    const int oh = h >> 1, ow = w >> 1, pw = (ow + 1) >> 1;
    const unsigned it = blockIdx.x * 256u + threadIdx.x;
    if (it >= items) return;
    const int64_t b = blockIdx.y;
    const int px = (int)(it % (unsigned)pw);
    const unsigned t1 = it / (unsigned)pw;
    const int oy = (int)(t1 % (unsigned)oh);
    const int ch = (int)(t1 / (unsigned)oh);
    const float sc = scale[ch], sh = shift[ch];
    const float *xp = x + b * xbs + (int64_t)ch * xps + (2 * oy) * w + 4 * px;    // xps: channel-plane stride (>= h w)
    float *yp = y + b * ybs + ((int64_t)ch * oh + oy) * ow + 2 * px;            // ybs: batch stride of y (>= c oh ow)
    auto act = [&](float v) { return fmaxf(fmaf(v, sc, sh), 0.0f); };
    if (2 * px + 1 < ow) {
        float4 r0, r1;
        if ((w & 3) == 0) {                                  // rows start 16-byte aligned
            r0 = *reinterpret_cast<const float4 *>(xp);
            r1 = *reinterpret_cast<const float4 *>(xp + w);
        } else {
            const float2 a0 = *reinterpret_cast<const float2 *>(xp), a1 = *reinterpret_cast<const float2 *>(xp + 2);
            const float2 b0 = *reinterpret_cast<const float2 *>(xp + w), b1 = *reinterpret_cast<const float2 *>(xp + w + 2);
            r0 = make_float4(a0.x, a0.y, a1.x, a1.y);
            r1 = make_float4(b0.x, b0.y, b1.x, b1.y);
        }
        const float s0 = act(r0.x) + act(r0.y) + act(r1.x) + act(r1.y);
        const float s1 = act(r0.z) + act(r0.w) + act(r1.z) + act(r1.w);
        if ((ow & 1) == 0) {
            *reinterpret_cast<float2 *>(yp) = make_float2(s0 * 0.25f, s1 * 0.25f);
        } else {
            yp[0] = s0 * 0.25f;
            yp[1] = s1 * 0.25f;
        }
    } else {
        const float2 r0 = *reinterpret_cast<const float2 *>(xp);
        const float2 r1 = *reinterpret_cast<const float2 *>(xp + w);
        yp[0] = (act(r0.x) + act(r0.y) + act(r1.x) + act(r1.y)) * 0.25f;
    }
}

}  // namespace

hipError_t launch_bn_relu_nchw(const float *x, int64_t x_batch_stride, const float *scale,
                               const float *shift, int64_t n, int c, int hw, float *y, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const int64_t slab = (int64_t)c * hw;
    if ((slab & 3) || (x_batch_stride & 3) || n > 65535) return hipErrorInvalidValue;
    const unsigned gx = (unsigned)std::min<int64_t>((slab / 4 + 255) / 256, 4096);
    hipLaunchKernelGGL(k_bn_relu, dim3(gx, (unsigned)n), dim3(256), 0, st, x, x_batch_stride, scale, shift,
                       c, hw, y);
    return hipGetLastError();
}

hipError_t launch_bn_relu_avgpool2(const float *x, int64_t x_batch_stride, const float *scale,
                                   const float *shift, int64_t n, int c, int h, int w, float *y,
                                   int64_t x_plane_stride, hipStream_t st, int64_t y_batch_stride) {
    if (n <= 0) return hipSuccess;
    if (!x_plane_stride) x_plane_stride = (int64_t)h * w;
    if (!y_batch_stride) y_batch_stride = (int64_t)c * (h / 2) * (w / 2);
    if (y_batch_stride < (int64_t)c * (h / 2) * (w / 2) || (y_batch_stride & 1)) return hipErrorInvalidValue;
    if ((h & 1) || (w & 1) || (x_batch_stride & 1) || n > 65535 || c > 65535) return hipErrorInvalidValue;
    if (x_plane_stride < (int64_t)h * w || (x_plane_stride & 3)) return hipErrorInvalidValue;   // rows stay 16-byte aligned
    const int64_t items = (int64_t)c * (h / 2) * ((w / 2 + 1) / 2);      // per image
    if (items > 0x7fffff00LL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((items + 255) / 256), (unsigned)n);
#define MIRX_POOL_LAUNCH(W_)                                                                                            \
    hipLaunchKernelGGL(k_bn_relu_avgpool2<W_>, grid, dim3(256), 0, st, x, x_batch_stride, scale, shift, c, h, w,       \
                       (unsigned)items, y, x_plane_stride, y_batch_stride)
    if (h == w && w == 56) MIRX_POOL_LAUNCH(56);
    else if (h == w && w == 28) MIRX_POOL_LAUNCH(28);
    else if (h == w && w == 14) MIRX_POOL_LAUNCH(14);
    else MIRX_POOL_LAUNCH(0);
#undef MIRX_POOL_LAUNCH
    return hipGetLastError();
}

hipError_t launch_ingest(const float *src, int64_t n, int dim, int dimp, float *g32,
                         uint16_t *g16, float *gbias, unsigned *gnorm_max_bits, int metric,
                         hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ingest, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, src, n, dim, dimp,
                       g32, g16, gbias, gnorm_max_bits, metric);
    return hipGetLastError();
}

hipError_t launch_prep_queries(const float *q, int64_t nq, int64_t nq_pad, int dim, int dimp,
                               float *q32p, uint16_t *q16, float *qnorm, hipStream_t st) {
    if (nq_pad <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_prep_queries, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, st, q, nq,
                       nq_pad, dim, dimp, q32p, q16, qnorm);
    return hipGetLastError();
}

hipError_t launch_l2_normalize(float *x, int64_t n, int dim, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_l2_normalize, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, x, n, dim);
    return hipGetLastError();
}

hipError_t launch_bn_relu_gap_l2norm(const float *x, const float *scale, const float *shift,
                                     int64_t n, int c, int hw, int normalize, float *y,
                                     hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const size_t lds = (size_t)((c + 3) & ~3) * sizeof(float) + 16;
    hipLaunchKernelGGL(k_head, dim3((unsigned)n), dim3(HEAD_THREADS), lds, st, x, scale, shift, c, hw,
                       normalize, y);
    return hipGetLastError();
}

}  // namespace mirx
