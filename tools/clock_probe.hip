// clock_probe.hip -- measures the shader clock the chip actually holds while another kernel runs.
// One wave spins for `spin_us` microseconds of the constant 100 MHz reference counter (s_memrealtime) and reports how
// many shader-clock ticks (s_memtime) passed: clock [MHz] = 100 * d(memtime) / d(memrealtime).  Launched on its own
// stream next to the kernel under test (tools/clock_probe.py).  Build: hipcc --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_clock_probe(unsigned long long *out, unsigned long long spin_ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
}

extern "C" int clock_probe_launch(unsigned long long *out_dev, unsigned long long spin_us, void *stream) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out_dev, spin_us * 100ull);
    return (int)hipGetLastError();
}
