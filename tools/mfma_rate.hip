// Development micro-benchmark (hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o exp/mfma_rate):
// cycles per v_mfma_f32_16x16x32_bf16 in the MFMA stream of k_gemm16, alone, with a workgroup barrier per
// 32/64/128 MFMAs, and with ds_read_b128 interleaved -- the hardware ceilings DESIGN.md quotes.
// cycles per v_mfma_f32_16x16x32_bf16
// (32 in-place MFMAs, 8 A fragments x 4 B fragments), accumulators in VGPRs ("v") or AGPRs ("a")
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define ROW(C, MI)                                                                  \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"                      \
                 "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"                      \
                 "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\t"                      \
                 "v_mfma_f32_16x16x32_bf16 %3, %4, %8, %3"                          \
                 : "+" C(acc[MI][0]), "+" C(acc[MI][1]), "+" C(acc[MI][2]), "+" C(acc[MI][3]) \
                 : "v"(a[MI]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));

template <int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float *in, float *out, unsigned long long *cyc, int iters) {
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)in[(threadIdx.x + i * 8 + j) & 1023];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)in[(threadIdx.x * 3 + i * 8 + j) & 1023];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) { ROW("v", 0) ROW("v", 1) ROW("v", 2) ROW("v", 3) ROW("v", 4) ROW("v", 5) ROW("v", 6) ROW("v", 7) }
        else           { ROW("a", 0) ROW("a", 1) ROW("a", 2) ROW("a", 3) ROW("a", 4) ROW("a", 5) ROW("a", 6) ROW("a", 7) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// 8 waves (2 per SIMD): NROWS*4 MFMAs per wave, then a workgroup barrier
template <int REPS>
__global__ __launch_bounds__(512) void kb(const float *in, float *out, unsigned long long *cyc, int iters) {
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)in[(threadIdx.x + i * 8 + j) & 1023];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)in[(threadIdx.x * 3 + i * 8 + j) & 1023];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REPS; ++r) { ROW("v", 0) ROW("v", 1) ROW("v", 2) ROW("v", 3) ROW("v", 4) ROW("v", 5) ROW("v", 6) ROW("v", 7) }
        asm volatile("s_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

// 64 MFMAs per wave per iteration with NRD ds_read_b128 per wave spread between the MFMA rows, + barrier
template <int NRD>
__global__ __launch_bounds__(512) void kl(const float *in, float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float *>(lds)[i] = in[i & 1023];
    __syncthreads();
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[i][j] = (__bf16)in[(threadIdx.x + i * 8 + j) & 1023];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) b[i][j] = (__bf16)in[(threadIdx.x * 3 + i * 8 + j) & 1023];
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;      // conflict-free, wave-linear
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define RD(D, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(D) : "v"(addr));
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            ROW("v", 0) if (NRD >= 8) RD(d0, 0)    if (NRD >= 24) { RD(d1, 1024) RD(d2, 2048) }
            ROW("v", 1) if (NRD >= 8) RD(d0, 0)    if (NRD >= 24) { RD(d1, 1024) RD(d2, 2048) }
            ROW("v", 2) if (NRD >= 8) RD(d0, 0)    if (NRD >= 24) { RD(d1, 1024) RD(d2, 2048) }
            ROW("v", 3) if (NRD >= 8) RD(d0, 0)    if (NRD >= 24) { RD(d1, 1024) RD(d2, 2048) }
            ROW("v", 4) if (NRD >= 16) RD(d0, 0)
            ROW("v", 5) if (NRD >= 16) RD(d0, 0)
            ROW("v", 6) if (NRD >= 16) RD(d0, 0)
            ROW("v", 7) if (NRD >= 16) RD(d0, 0)
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = d0[0] + d1[1] + d2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NRD>
void runl(float *in, float *out, unsigned long long *cyc) {
    const int iters = 10000;
    unsigned long long c[8];
    hipLaunchKernelGGL((kl<NRD>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    const int nrd = NRD == 0 ? 0 : (NRD == 8 ? 8 : (NRD == 16 ? 16 : 32));
    printf("64 MFMAs + %2d ds_read_b128 per wave + barrier: %.0f cycles per iteration (wave 0), %.0f (wave 4)\n", nrd,
           (double)c[0] / iters, (double)c[4] / iters);
}

template <int REPS>
void runb(float *in, float *out, unsigned long long *cyc) {
    const int iters = 10000;
    unsigned long long c[8];
    hipLaunchKernelGGL((kb<REPS>), dim3(256), dim3(512), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%d MFMAs per wave then s_barrier, 2 waves/SIMD: %.0f cycles per iteration (wave 0), %.0f (wave 4); MFMA work per SIMD %d cycles\n",
           32 * REPS, (double)c[0] / iters, (double)c[4] / iters, 2 * 32 * REPS * 16);
}

template <int KIND, int WAVES>
void run(const char *name, float *in, float *out, unsigned long long *cyc) {
    const int iters = 20000;
    unsigned long long c;
    hipLaunchKernelGGL((k<KIND, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.2f cycles per MFMA per wave (%.2f per SIMD)\n", name, (double)c / iters / 32, (double)c / iters / 32 / (WAVES / 4));
}

int main() {
    float *in, *out; unsigned long long *cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
    float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    run<0, 4>("16x16x32 VGPR acc, 1 wave/SIMD ", in, out, cyc);
    run<1, 4>("16x16x32 AGPR acc, 1 wave/SIMD ", in, out, cyc);
    run<0, 8>("16x16x32 VGPR acc, 2 waves/SIMD", in, out, cyc);
    run<1, 8>("16x16x32 AGPR acc, 2 waves/SIMD", in, out, cyc);
    runb<1>(in, out, cyc);
    runb<2>(in, out, cyc);
    runb<4>(in, out, cyc);
    runl<0>(in, out, cyc);
    runl<8>(in, out, cyc);
    runl<16>(in, out, cyc);
    runl<24>(in, out, cyc);
    return 0;
}
