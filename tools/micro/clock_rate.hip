// How fast __builtin_readcyclecounter() (s_memtime) ticks: against wall_clock64() (the 100 MHz constant clock).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/clock_rate tools/micro/clock_rate.hip && tools/micro/clock_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(unsigned long long *out)
{
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    while (wall_clock64() - w0 < 100000ull) { }          // 1 ms
    out[0] = wall_clock64() - w0;
    out[1] = __builtin_readcyclecounter() - c0;
}

int main()
{
    unsigned long long *d, h[2];
    if (hipMalloc(&d, 16) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    if (hipMemcpy(h, d, 16, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    printf("wall clock ticks %llu (100 MHz), cycle counter ticks %llu: %.1f MHz\n", h[0], h[1], 100.0 * (double)h[1] / (double)h[0]);
    return 0;
}
