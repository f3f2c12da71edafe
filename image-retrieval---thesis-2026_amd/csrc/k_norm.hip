// k_norm.hip -- the memory-bound glue of the token-major backbones (ConvNeXtV2 / DINOv2 / SigLIP), so that no library
// kernel (MIOpen convolution find + naive_conv, ATen layer_norm) is left in their forward:
//
//   k_layernorm_rows   LayerNorm over the last axis of [m, c] rows (timm / transformers nn.LayerNorm): one wavefront per
//                      row, two-pass mean / variance in fp32 with wave-shuffle reductions, 16-byte loads; optionally the
//                      result is written channels-first ([image][c][tokens]: the ConvNeXt stem's LayerNorm2d) through an
//                      LDS transpose so that both sides stay coalesced.
//   k_patchify         non-overlapping p x p patches of an NCHW tensor -> token-major rows [B * (H/p) * (W/p), kpad] whose
//                      feature order (c, ky, kx) is the flattening of a conv weight [Cout, C, p, p]: a stride-p p x p
//                      convolution (ViT patch embedding 14 x 14, ConvNeXt stem 4 x 4 and downsample 2 x 2) becomes
//                      patchify + the MFMA Linear kernel.  Optional LayerNorm2d prologue (normalise every pixel over its
//                      C channels first: ConvNeXt's downsample = LayerNorm2d + conv 2x2/2); columns beyond C p p are zero
//                      (K padded to the Linear kernel's multiple of 16).
//   k_attention_small  softmax(q k^T * scale [+ key mask]) v for SHORT query sets: one wavefront per (image, head, query),
//                      lanes over keys, online softmax per lane, butterfly combine.  fp32 VALU (exact products).  Used where
//                      the MFMA flash kernels do not fit: the SigLIP text tower (64 tokens, key-padding mask) and the
//                      SigLIP attention-pooling head (1 probe query over 1024 keys).
#include "mirx_kernels.h"

namespace mirx {

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- LayerNorm over rows ----------------------------------------------------------------------------------------
// grid: ceil(m / 4) workgroups of 4 waves; wave w normalises row 4 * blockIdx.x + w.  c % 4 == 0.
// TERMS: the result is written as "terms rows" (k_linear_t2.hip: line g of a row = fp16 hi | lo of scale * y[32 g .. 32 g + 31],
// the input format of the DMA-fed Linear) instead of fp32 -- the same bytes, and the Linear that follows needs no split.
// The row is read ONCE and kept in registers (RV 16-byte vectors per lane: c <= 256 RV features; the launcher picks RV): mean,
// variance (two passes over the registers, the reference's formula) and the result come from the same loads.
// LW = lanes per row: 64, or 32 for rows of at most 128 features (two rows per wavefront -- with 64 lanes half of them
// would idle on ConvNeXtV2's 128-channel maps, its largest LayerNorms by bytes); reductions stay inside the LW lanes.
template <int LW>
__device__ inline float row_sum(float v) {
#pragma unroll
    for (int off = LW / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <bool TERMS, int RV, int LW = 64>
__global__ __launch_bounds__(256) void k_layernorm_rows(const float *__restrict__ x, int64_t m, int c,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float eps, float *__restrict__ y, char *__restrict__ yt, float scale,
                                                        int cp, int patch_w, int patch_h) {
    const int lane = threadIdx.x & (LW - 1);
    const int64_t row = (int64_t)blockIdx.x * (256 / LW) + threadIdx.x / LW;
    if (row >= m) return;
    const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + row * c);
    const int nv = c >> 2;
    f32x4 v[RV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < RV; ++j) {
        const int i = lane + LW * j;
        v[j] = i < nv ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = row_sum<LW>(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < RV; ++j) {
        if (lane + LW * j < nv) {
            const float d0 = v[j][0] - mean, d1 = v[j][1] - mean, d2 = v[j][2] - mean, d3 = v[j][3] - mean;
            q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    }
    const float rstd = 1.0f / sqrtf(row_sum<LW>(q) / (float)c + eps);
    // patch_w > 0: the rows are the pixels of channels-last maps [image][patch_h][patch_w][c] and the result goes to the
    // 2 x 2 patch rows of a stride-2 convolution, [image][patch_h / 2][patch_w / 2][(ky, kx, c)] (ConvNeXt downsample)
    int64_t orow = row * c;
    if (!TERMS && patch_w > 0) {
        const int64_t hw = (int64_t)patch_w * patch_h, im = row / hw;
        const int rem = (int)(row - im * hw), py = rem / patch_w, px = rem - py * patch_w;
        orow = (((im * (patch_h >> 1) + (py >> 1)) * (patch_w >> 1) + (px >> 1)) * 4 + ((py & 1) * 2 + (px & 1))) * c;
    }
    f32x4 *yr = reinterpret_cast<f32x4 *>(y + orow);
    char *tr = yt + row * ((int64_t)cp * 4);
#pragma unroll
    for (int j = 0; j < RV; ++j) {
        const int i = lane + LW * j;
        if (i >= (TERMS ? cp >> 2 : nv)) continue;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = gamma ? gamma[4 * i + e] : 1.f, b = beta ? beta[4 * i + e] : 0.f;
                o[e] = (v[j][e] - mean) * rstd * g + b;
            }
        }
        if (TERMS) {
            unsigned h0, l0, h1, l1;
            split2h_pair(o[0] * scale, o[1] * scale, h0, l0);
            split2h_pair(o[2] * scale, o[3] * scale, h1, l1);
            const u32x2 hi = {h0, h1}, lo = {l0, l1};
            char *dst = tr + (i >> 3) * 128 + (i & 7) * 8;
            *reinterpret_cast<u32x2 *>(dst) = hi;
            *reinterpret_cast<u32x2 *>(dst + 64) = lo;
        } else {
            yr[i] = o;
        }
    }
}

// Channels-first output: y[(img * c + j) * tpi + t] for token t of image img.  One workgroup = 64 consecutive tokens;
// wave w normalises tokens 16 w .. 16 w + 15 into LDS ([c][65] floats), then the tile is written with lanes walking
// tokens.  c <= 512 (LDS 133 KiB at 512).
__global__ __launch_bounds__(256) void k_layernorm_rows_nchw(const float *__restrict__ x, int64_t m, int c, int tpi,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps,
                                                             float *__restrict__ y) {
    extern __shared__ float sm[];                        // [c][65]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * 64;
    for (int r = 0; r < 16; ++r) {
        const int tl = wave * 16 + r;
        const int64_t row = t0 + tl;
        if (row >= m) break;                             // wave-uniform
        const float *xr = x + row * c;
        float s = 0.f;
        for (int i = lane; i < c; i += 64) s += xr[i];
        const float mean = wave_sum(s) / (float)c;
        float q = 0.f;
        for (int i = lane; i < c; i += 64) {
            const float d = xr[i] - mean;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c + eps);
        for (int i = lane; i < c; i += 64)
            sm[i * 65 + tl] = (xr[i] - mean) * rstd * (gamma ? gamma[i] : 1.f) + (beta ? beta[i] : 0.f);
    }
    __syncthreads();
    // item = (channel, token): lanes walk the 64 tokens of the tile
    for (int it = threadIdx.x; it < c * 64; it += 256) {
        const int j = it >> 6, tl = it & 63;
        const int64_t row = t0 + tl;
        if (row < m) {
            const int64_t img = row / tpi, t = row - img * tpi;
            y[(img * c + j) * tpi + t] = sm[j * 65 + tl];
        }
    }
}

// ---- patchify (+ optional LayerNorm2d) ---------------------------------------------------------------------------
// grid: (H / p patch rows, B).  A workgroup owns one row of patches: p input rows x W pixels of every channel.
// out[((b * gh + py) * gw + px) * kpad + (ch * p + ky) * p + kx] = norm(x)[b, ch, py * p + ky, px * p + kx]
template <bool LN>
__global__ __launch_bounds__(256) void k_patchify(const float *__restrict__ x, int c, int h, int w, int p,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                  float eps, float *__restrict__ out, int kpad) {
    extern __shared__ float sm[];                        // LN: mean[p * w], rstd[p * w]
    const int gh = h / p, gw = w / p;
    const int py = blockIdx.x;
    const int64_t b = blockIdx.y;
    const float *xb = x + (b * c) * (int64_t)h * w + (int64_t)(py * p) * w;      // + ch * h * w + ky * w + col
    const int npx = p * w;                               // pixels of the strip
    if (LN) {
        float *s_mean = sm, *s_rstd = sm + npx;
        for (int i = threadIdx.x; i < npx; i += 256) {
            const int ky = i / w, col = i - ky * w;
            const float *px = xb + (int64_t)ky * w + col;
            float s = 0.f;
            for (int ch = 0; ch < c; ++ch) s += px[(int64_t)ch * h * w];
            const float mean = s / (float)c;
            float q = 0.f;
            for (int ch = 0; ch < c; ++ch) {
                const float d = px[(int64_t)ch * h * w] - mean;
                q += d * d;
            }
            s_mean[i] = mean;
            s_rstd[i] = 1.0f / sqrtf(q / (float)c + eps);
        }
        __syncthreads();
    }
    float *ob = out + ((b * gh + py) * (int64_t)gw) * kpad;
    const int kk = c * p * p;
    // item = (ch, ky, col): lanes walk columns -> coalesced reads; a patch's kx run is contiguous in the output
    const int per_ch = p * w;
    for (int it = threadIdx.x; it < c * per_ch; it += 256) {
        const int ch = it / per_ch, r = it - ch * per_ch;
        const int ky = r / w, col = r - ky * w;
        float v = xb[(int64_t)ch * h * w + (int64_t)ky * w + col];
        if (LN) v = (v - sm[r]) * sm[npx + r] * gamma[ch] + beta[ch];
        const int px = col / p, kx = col - px * p;
        if (px < gw) ob[(int64_t)px * kpad + (ch * p + ky) * p + kx] = v;
    }
    // zero the K padding of every token of the row
    const int padn = kpad - kk;
    for (int it = threadIdx.x; it < gw * padn; it += 256) ob[(int64_t)(it / padn) * kpad + kk + it % padn] = 0.f;
}

// ---- attention for short query sets --------------------------------------------------------------------------------
// q[(b * nq + i) * q_rs + hd * DH + d], k / v[(b * nk + j) * kv_rs + hd * DH + d], out[(b * nq + i) * heads * DH + hd * DH + d]
// key_mask[b * nk + j] != 0 -> key j takes part (null: all keys).  One wave per (b, hd, i): grid = ceil(B heads nq / 4).
template <int DH>
__global__ __launch_bounds__(256) void k_attention_small(const float *__restrict__ q, int64_t q_rs,
                                                         const float *__restrict__ k, const float *__restrict__ v,
                                                         int64_t kv_rs, const uint8_t *__restrict__ key_mask,
                                                         int64_t batch, int heads, int nq, int nk, float scale,
                                                         float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= batch * heads * nq) return;
    const int i = (int)(unit % nq);
    const int hd = (int)((unit / nq) % heads);
    const int64_t b = unit / ((int64_t)nq * heads);
    float qr[DH];
    {
        const float *qp = q + (b * nq + i) * q_rs + hd * DH;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(qp + d);
            qr[d] = t[0] * scale; qr[d + 1] = t[1] * scale; qr[d + 2] = t[2] * scale; qr[d + 3] = t[3] * scale;
        }
    }
    float mx = -INFINITY, den = 0.f, acc[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) acc[d] = 0.f;
    for (int j = lane; j < nk; j += 64) {
        if (key_mask && !key_mask[b * nk + j]) continue;
        const float *kp = k + (b * nk + j) * kv_rs + hd * DH;
        const float *vp = v + (b * nk + j) * kv_rs + hd * DH;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(kp + d);
            s = fmaf(qr[d], t[0], s); s = fmaf(qr[d + 1], t[1], s); s = fmaf(qr[d + 2], t[2], s); s = fmaf(qr[d + 3], t[3], s);
        }
        const float nm = fmaxf(mx, s);
        const float a = __expf(mx - nm), pj = __expf(s - nm);       // exp(-inf) = 0 on the first key
        den = den * a + pj;
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(vp + d);
            acc[d] = fmaf(pj, t[0], acc[d] * a); acc[d + 1] = fmaf(pj, t[1], acc[d + 1] * a);
            acc[d + 2] = fmaf(pj, t[2], acc[d + 2] * a); acc[d + 3] = fmaf(pj, t[3], acc[d + 3] * a);
        }
        mx = nm;
    }
    // combine the 64 partial softmaxes
    float gm = mx;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) gm = fmaxf(gm, __shfl_xor(gm, off, 64));
    const float w = (mx == -INFINITY) ? 0.f : __expf(mx - gm);      // lanes without a key contribute nothing
    const float dsum = wave_sum(den * w);
    float *op = out + ((b * nq + i) * (int64_t)heads + hd) * DH;
    const float inv = dsum > 0.f ? 1.0f / dsum : 0.f;                // every key masked: zeros
#pragma unroll
    for (int d = 0; d < DH; ++d) {
        const float t = wave_sum(acc[d] * w);
        if (lane == (d & 63)) op[d] = t * inv;
    }
}

}  // namespace

hipError_t launch_layernorm_rows(const float *x, int64_t m, int c, const float *gamma, const float *beta, float eps,
                                 float *y, int tokens_per_image, hipStream_t st, void *yt, float scale, int patch_w, int patch_h) {
    if (m <= 0) return hipSuccess;
    if (c < 4 || c % 4) return hipErrorInvalidValue;
    if (yt) {
        if (tokens_per_image > 0 || y) return hipErrorInvalidValue;
        const int64_t blocks = (m + 3) / 4;
        if (blocks > 0x7fffffff) return hipErrorInvalidValue;
        const int cp = (c + 31) / 32 * 32;
        if (cp > 8192) return hipErrorInvalidValue;
#define MIRX_LNT(RV) hipLaunchKernelGGL((k_layernorm_rows<true, RV>), dim3((unsigned)blocks), dim3(256), 0, st, x, m, c, gamma, beta, eps, \
                                        nullptr, reinterpret_cast<char *>(yt), scale, cp, 0, 0)
        if (cp <= 128) {
            hipLaunchKernelGGL((k_layernorm_rows<true, 1, 32>), dim3((unsigned)((m + 7) / 8)), dim3(256), 0, st, x, m, c, gamma, beta, eps,
                               nullptr, reinterpret_cast<char *>(yt), scale, cp, 0, 0);
            return hipGetLastError();
        }
        if (cp <= 512) MIRX_LNT(2); else if (cp <= 1024) MIRX_LNT(4); else if (cp <= 2048) MIRX_LNT(8); else if (cp <= 4096) MIRX_LNT(16); else MIRX_LNT(32);
#undef MIRX_LNT
        return hipGetLastError();
    }
    if (tokens_per_image > 0) {
        if (c > 512) return hipErrorInvalidValue;
        const size_t lds = (size_t)c * 65 * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_layernorm_rows_nchw),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_layernorm_rows_nchw, dim3((unsigned)((m + 63) / 64)), dim3(256), lds, st, x, m, c,
                           tokens_per_image, gamma, beta, eps, y);
    } else {
        const int64_t blocks = (m + 3) / 4;
        if (blocks > 0x7fffffff) return hipErrorInvalidValue;
        if (c > 8192) return hipErrorInvalidValue;
#define MIRX_LN(RV) hipLaunchKernelGGL((k_layernorm_rows<false, RV>), dim3((unsigned)blocks), dim3(256), 0, st, x, m, c, gamma, beta, eps, \
                                       y, nullptr, 1.f, 0, patch_w, patch_h)
        if (c <= 128) {
            hipLaunchKernelGGL((k_layernorm_rows<false, 1, 32>), dim3((unsigned)((m + 7) / 8)), dim3(256), 0, st, x, m, c, gamma, beta, eps,
                               y, nullptr, 1.f, 0, patch_w, patch_h);
            return hipGetLastError();
        }
        if (c <= 512) MIRX_LN(2); else if (c <= 1024) MIRX_LN(4); else if (c <= 2048) MIRX_LN(8); else if (c <= 4096) MIRX_LN(16); else MIRX_LN(32);
#undef MIRX_LN
    }
    return hipGetLastError();
}

hipError_t launch_patchify(const float *x, int64_t n, int c, int h, int w, int p, const float *gamma, const float *beta,
                           float eps, float *out, int kpad, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (p < 1 || h < p || w < p || kpad < c * p * p || n > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(h / p), (unsigned)n);
    if (gamma) {
        const size_t lds = (size_t)2 * p * w * sizeof(float);
        hipLaunchKernelGGL(k_patchify<true>, grid, dim3(256), lds, st, x, c, h, w, p, gamma, beta, eps, out, kpad);
    } else {
        hipLaunchKernelGGL(k_patchify<false>, grid, dim3(256), 0, st, x, c, h, w, p, gamma, beta, eps, out, kpad);
    }
    return hipGetLastError();
}

hipError_t launch_attention_small(const float *q, int64_t q_rs, const float *k, const float *v, int64_t kv_rs,
                                  const uint8_t *key_mask, int64_t batch, int heads, int head_dim, int nq, int nk,
                                  float scale, float *out, hipStream_t st) {
    if (batch <= 0 || nq <= 0) return hipSuccess;
    const int64_t units = batch * heads * nq;
    const int64_t blocks = (units + 3) / 4;
    if (blocks > 0x7fffffff || nk < 1) return hipErrorInvalidValue;
#define MIRX_AS(D)                                                                                                      \
    hipLaunchKernelGGL(k_attention_small<D>, dim3((unsigned)blocks), dim3(256), 0, st, q, q_rs, k, v, kv_rs, key_mask,  \
                       batch, heads, nq, nk, scale, out)
    if (head_dim == 72) MIRX_AS(72);
    else if (head_dim == 64) MIRX_AS(64);
    else if (head_dim == 32) MIRX_AS(32);
    else if (head_dim == 16) MIRX_AS(16);
    else return hipErrorInvalidValue;
#undef MIRX_AS
    return hipGetLastError();
}

}  // namespace mirx
