// Development probe (hipcc -O3 --offload-arch=gfx950 tools/dma_order_probe.hip -o exp/dma_order_probe):
// may a wave that has issued `buffer_load ... lds` (LDS-DMA) pieces and, BEHIND them, ordinary loads to registers wait for the
// DMA with a COUNTED s_waitcnt vmcnt(N) (N = the younger loads) instead of vmcnt(0)?  k_conv1x1_h2 drains everything at every
// stage top because a counted wait gave state-dependent results in round 2; this probe isolates the question.
// Every workgroup: per iteration each wave DMAs a fresh 1-KiB piece (values = their global index) into one of two LDS buffers,
// issues NY loads from a far-away array behind it, waits (variant), synchronises, and checks the piece written by ANOTHER wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int NY = 8;

// VAR 0: vmcnt(0).  1: vmcnt(NY) + __syncthreads().  2: vmcnt(NY) + raw s_barrier.  3: vmcnt(NY) + lgkmcnt(0) + raw s_barrier.
// HOT: the DMA source is 3 small patterns per workgroup (12 KiB, cache-resident after the first pass; 3 patterns against 2 LDS
// buffers, so a stale buffer shows) while the loads behind the DMA miss to HBM -- the 1x1 conv's situation (weights hot, activations
// cold): the DMA's data returns long before the younger loads'.
template <int VAR, int HOT>
__global__ __launch_bounds__(256, 2) void k_probe(const unsigned *__restrict__ src, const unsigned *__restrict__ far_, int iters,
                                                  int far_stride, unsigned long long *bad, unsigned *sink) {
    __shared__ __attribute__((aligned(16))) unsigned sm[2][4][256];            // [buffer][wave][1 KiB]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7ffffff0, 0x00020000);
    unsigned acc = 0;
    unsigned long long nbad = 0;
    const unsigned *fp = far_ + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        // piece of (workgroup, iteration, wave): 256 words starting at word index base
        const unsigned base = HOT ? (((unsigned)blockIdx.x * 3u + (unsigned)(it % 3)) * 4u + (unsigned)wave) * 256u
                                  : (((unsigned)blockIdx.x * (unsigned)iters + (unsigned)it) * 4u + (unsigned)wave) * 256u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(&sm[buf][wave][0]), 16, lane * 16, (int)(base * 4u), 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        unsigned y[NY];
#pragma unroll
        for (int j = 0; j < NY; ++j) y[j] = fp[(size_t)(it * NY + j) * far_stride];
        __builtin_amdgcn_sched_barrier(0);
        if (VAR == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else if (VAR == 1) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __syncthreads();
        } else if (VAR == 2) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        // check the piece of the next wave
        const int ow = (wave + 1) & 3;
        const unsigned obase = HOT ? (((unsigned)blockIdx.x * 3u + (unsigned)(it % 3)) * 4u + (unsigned)ow) * 256u
                                   : (((unsigned)blockIdx.x * (unsigned)iters + (unsigned)it) * 4u + (unsigned)ow) * 256u;
        const u32x4 got = *reinterpret_cast<const u32x4 *>(&sm[buf][ow][lane * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) nbad += got[e] != obase + lane * 4 + e;
#pragma unroll
        for (int j = 0; j < NY; ++j) acc += y[j];
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // everyone has checked before the other buffer's next piece... (two buffers)
    }
    if (nbad) atomicAdd(bad, nbad);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int grid = 512;
    const size_t nsrc = (size_t)grid * iters * 4 * 256;
    unsigned *src, *far_, *sink;
    unsigned long long *bad;
    if (hipMalloc(&src, nsrc * 4) != hipSuccess) return 1;
    const int far_stride = 1 << 18;                        // words: 1 MiB apart
    const size_t nfar = (size_t)iters * NY * far_stride + (size_t)grid * 256 * 4 + 16;
    if (hipMalloc(&far_, nfar * 4) != hipSuccess) return 1;
    (void)hipMalloc(&sink, 4);
    (void)hipMalloc(&bad, 8);
    unsigned *h = (unsigned *)malloc(nsrc * 4);
    for (size_t i = 0; i < nsrc; ++i) h[i] = (unsigned)i;
    (void)hipMemcpy(src, h, nsrc * 4, hipMemcpyHostToDevice);
    (void)hipMemset(far_, 0, nfar * 4);
    for (int hot = 0; hot < 2; ++hot)
    for (int var = 0; var < 4; ++var) {
        for (int rep = 0; rep < 3; ++rep) {                // rep 0 cold, 1-2 warm
            (void)hipMemset(bad, 0, 8);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0);
            (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
#define LAUNCH(V, H) hipLaunchKernelGGL((k_probe<V, H>), dim3(grid), dim3(256), 0, 0, src, far_, iters, far_stride, bad, sink)
            if (hot == 0) { if (var == 0) LAUNCH(0, 0); if (var == 1) LAUNCH(1, 0); if (var == 2) LAUNCH(2, 0); if (var == 3) LAUNCH(3, 0); }
            else          { if (var == 0) LAUNCH(0, 1); if (var == 1) LAUNCH(1, 1); if (var == 2) LAUNCH(2, 1); if (var == 3) LAUNCH(3, 1); }
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            unsigned long long nb = 0;
            (void)hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
            static const char *names[] = {"vmcnt(0) + __syncthreads", "vmcnt(8) + __syncthreads", "vmcnt(8) + s_barrier",
                                          "vmcnt(8) lgkmcnt(0) + s_barrier"};
            printf("%s DMA source, %-34s run %d: %8.3f ms, %llu wrong words of %zu\n", hot ? "hot " : "cold", names[var], rep, ms, nb,
                   (size_t)grid * iters * 1024);
        }
    }
    return 0;
}
