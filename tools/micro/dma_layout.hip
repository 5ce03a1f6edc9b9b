// Where do the bytes of a 12- and a 16-byte LDS-DMA load (global_load_lds_dwordx3/x4) land in LDS?
// hipcc --offload-arch=gfx950 -O3 -o dma_layout dma_layout.hip && ./dma_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define PROBE(NAME, SIZE) \
__global__ void NAME(const unsigned *src, unsigned *out, int stride_dwords) \
{ \
    __shared__ unsigned lds[1024]; \
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xDEAD0000u + i; \
    __syncthreads(); \
    __builtin_amdgcn_global_load_lds((glb_void *)(src + threadIdx.x * stride_dwords), (lds_void *)lds, SIZE, 0, 0); \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    __syncthreads(); \
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i]; \
}
PROBE(probe12, 12)
PROBE(probe16, 16)

int main()
{
    unsigned h[4096], *d, *o, r[1024];
    for (int i = 0; i < 4096; i++) h[i] = i;           // dword i holds i: lane t's triple = 3t, 3t+1, 3t+2
    hipMalloc(&d, sizeof h);
    hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    for (int size : {12, 16}) {
        if (size == 12) probe12<<<1, 64>>>(d, o, 3); else probe16<<<1, 64>>>(d, o, 3);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        printf("size %d, per-lane source stride 12 bytes; first 48 LDS dwords:\n", size);
        for (int i = 0; i < 48; i++) printf("%s%u", i % 12 ? " " : "\n  ", r[i] >= 0xDEAD0000u ? 99999u : r[i]);
        int last = 0;
        for (int i = 0; i < 1024; i++) if (r[i] < 0xDEAD0000u) last = i;
        printf("\n  last written dword: %d\n", last);
    }
    return 0;
}
