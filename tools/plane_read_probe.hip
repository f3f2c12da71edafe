// Development micro-benchmark (hipcc -O3 --offload-arch=gfx950 tools/plane_read_probe.hip -o exp/plane_read_probe):
// how fast the chip streams a DenseNet block buffer [n][c][hw] fp32 the way k_conv1x1_h2 reads it -- a workgroup = 256
// consecutive pixels, stages of 16 channel planes -- with the kernel's request shape (thread = one pixel: 16 dword loads per
// stage, a wave-load = 256 contiguous bytes) against wide requests (thread = 4 consecutive pixels of 4 channels: 4 b128
// loads per stage, a wave-load = 1 KiB contiguous).  Same bytes, same loads in flight (two stages), nothing else in the loop:
// the ceiling the 1x1 conv's activation stream could reach with either shape.  Third arm: the kernel's own stage top -- every load
// in flight is drained (s_waitcnt vmcnt(0), which its LDS-DMA'd weights need) and the workgroup synchronises before the next
// stage's loads are issued -- to see what that alone does to the stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int WIDE, int DRAIN>
__global__ __launch_bounds__(256, 2) void k_read(const float *__restrict__ x, int64_t n, int c, int hw, float *out) {
    const int64_t p0 = (int64_t)blockIdx.x * 256;
    const int64_t total = n * hw;
    float acc = 0.f;
    const int nk = c / 16;
    if (WIDE) {
        const int quad = threadIdx.x & 63, cg = threadIdx.x >> 6;
        int64_t pp = p0 + 4 * quad;
        if (pp >= total) pp = total - 4;
        const int64_t img = pp / hw, off = pp - img * hw;
        const float *src = x + (img * c + 4 * cg) * (int64_t)hw + off;
        f32x4 ra[4], rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4 *>(src + (int64_t)j * hw);
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            const int k1 = kt + 1 < nk ? kt + 1 : kt, k2 = kt + 2 < nk ? kt + 2 : kt;
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const f32x4 *>(src + ((int64_t)k1 * 16 + j) * hw);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += (ra[j][0] + ra[j][1]) + (ra[j][2] + ra[j][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const f32x4 *>(src + ((int64_t)k2 * 16 + j) * hw);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += (rb[j][0] + rb[j][1]) + (rb[j][2] + rb[j][3]);
        }
    } else {
        const int px = threadIdx.x & 127, kg = threadIdx.x >> 7;
        float ra[2][8], rb[2][8];
        const float *src[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int64_t pp = p0 + u * 128 + px;
            if (pp >= total) pp = total - 1;
            const int64_t img = pp / hw, off = pp - img * hw;
            src[u] = x + (img * c + 8 * kg) * (int64_t)hw + off;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) ra[u][j] = src[u][(int64_t)j * hw];
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            const int k1 = kt + 1 < nk ? kt + 1 : kt, k2 = kt + 2 < nk ? kt + 2 : kt;
            if (DRAIN) {                                     // k_conv1x1_h2's stage top: everything in flight is drained, then a barrier
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) rb[u][j] = src[u][((int64_t)k1 * 16 + j) * hw];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += ra[u][j];
            if (DRAIN) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) ra[u][j] = src[u][((int64_t)k2 * 16 + j) * hw];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += rb[u][j];
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 2048;
    const struct { int side, c; } shapes[] = {{56, 256}, {28, 512}, {14, 1024}, {7, 1024}};
    float *out;
    hipMalloc(&out, 4);
    for (auto s : shapes) {
        const int hw = s.side * s.side;
        const size_t bytes = (size_t)n * s.c * hw * 4;
        float *x;
        if (hipMalloc(&x, bytes) != hipSuccess) return 1;
        hipMemset(x, 0, bytes);
        const unsigned grid = (unsigned)((n * hw + 255) / 256);
        for (int wide = 0; wide < 3; ++wide) {
            if (wide == 1 && hw % 4) continue;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int it = 0; it < 4; ++it) {
                if (it == 1) hipEventRecord(e0);
                if (wide == 1) hipLaunchKernelGGL((k_read<1, 0>), dim3(grid), dim3(256), 0, 0, x, n, s.c, hw, out);
                else if (wide == 2) hipLaunchKernelGGL((k_read<0, 1>), dim3(grid), dim3(256), 0, 0, x, n, s.c, hw, out);
                else hipLaunchKernelGGL((k_read<0, 0>), dim3(grid), dim3(256), 0, 0, x, n, s.c, hw, out);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            printf("side %2d, %4d channels, %lld images (%.2f GB): %s  %.3f ms  %.2f TB/s\n", s.side, s.c, (long long)n, bytes / 1e9,
                   wide == 1 ? "thread = 4 px x 4 ch, b128 loads (1 KiB per wave-load)"
                   : wide == 2 ? "thread = 1 px x 8 ch, dword loads, vmcnt(0) + barrier before every stage's loads (the kernel's stage top)"
                               : "thread = 1 px x 8 ch, dword loads (256 B per wave-load)",
                   ms, bytes / ms / 1e9);
        }
        hipFree(x);
    }
    return 0;
}
