// Development micro-benchmark: sustained whole-chip fp16 MFMA throughput under the power limit, per instruction shape.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_power.hip -o gpurun_out/mfma_power && gpurun_out/mfma_power
// Every CU runs 8 waves (2 per SIMD) of in-register MFMAs on random data for ~100 ms; the wall-clock rate is what the
// power management lets the shape sustain (bare issue rate is the same 1024 FLOP / cycle / SIMD for both shapes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// SHAPE 0: v_mfma_f32_32x32x16_f16, 8 independent accumulators (4 A x 2 B fragments)
// SHAPE 1: v_mfma_f32_16x16x32_f16, 32 independent accumulators (8 A x 4 B fragments): the same FLOP per loop trip
// SHAPE 2: v_mfma_f32_16x16x32_bf16, as 1 (the distance GEMM's instruction)
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float *in, float *out, int iters) {
    f16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)in[(threadIdx.x * 7 + i * 8 + j) & 4095];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (_Float16)in[(threadIdx.x * 3 + i * 8 + j + 1000) & 4095];
    float s = 0;
    if (SHAPE == 0) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 4 * kk], b[j + 2 * kk], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else if (SHAPE == 2) {
        bf16x8 ab[8], bb[4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) ab[i][j] = (__bf16)(float)a[i][j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) bb[i][j] = (__bf16)(float)b[i][j];
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[i], bb[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float *in, *out;
    std::vector<float> h(4096);
    srand(1);
    for (auto &v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    (void)hipMalloc(&in, 4096 * 4);
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 60000;                              // 60000 trips x 16 x 32768 FLOP x 8 waves x 256 CUs
    const double flop = (double)iters * 16 * 32768 * 8 * 256;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 3; ++shape) {
            (void)hipEventRecord(e0);
            for (int l = 0; l < 3; ++l) {
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, in, out, iters);
                else if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, in, out, iters);
            }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f ms for 3 launches, %.0f TFLOP/s sustained\n", shape == 0 ? "32x32x16_f16" : shape == 1 ? "16x16x32_f16" : "16x16x32_bf16", ms,
                   3 * flop / ms / 1e9);
        }
    return 0;
}
