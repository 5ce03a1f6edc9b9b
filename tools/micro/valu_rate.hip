// Issue rate of packed and plain binary32 vector instructions on one SIMD, by waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ void __launch_bounds__(256) spin(float *out, int iters)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 * 1.5f, a2 = a0 + 3.f, a3 = a0 - 1.f, a4 = a0 * 0.5f, a5 = a0 + 7.f, a6 = a0 * 3.f, a7 = a0 - 9.f;
    const v2 m = {1.0000001f, 0.9999999f};
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // packed multiply, 8 independent chains
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 1) {   // plain multiply, 8 independent chains
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x));
        } else if (KIND == 2) {   // compare to a scalar pair + scalar popcount (the hit-count idiom)
            unsigned long long k0, k1, k2, k3;
            asm volatile("v_cmp_le_f32 %0, %4, %8\n v_cmp_le_f32 %1, %5, %8\n v_cmp_le_f32 %2, %6, %8\n v_cmp_le_f32 %3, %7, %8\n"
                         : "=s"(k0), "=s"(k1), "=s"(k2), "=s"(k3) : "v"(a0.x), "v"(a1.x), "v"(a2.x), "v"(a3.x), "v"(m.x));
            a4.x += __popcll(k0) + __popcll(k1);
            a5.x += __popcll(k2) + __popcll(k3);
        } else {                  // packed add with a negated operand (the subtraction idiom)
            asm volatile("v_pk_add_f32 %0, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x == 123.456f) out[threadIdx.x] = s.y;
}

template <int KIND>
static void run(const char *name, int per_instr)
{
    float *d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        // 256 CUs; a 256-thread block is 4 waves = one per SIMD; waves_per_simd blocks per CU
        const int blocks = 256 * waves_per_simd;
        spin<KIND><<<blocks, 256>>>(d, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        spin<KIND><<<blocks, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * per_instr * waves_per_simd;
        printf("%-28s waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name,
               waves_per_simd, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
    hipFree(d);
}

int main()
{
    run<0>("v_pk_mul_f32", 8);
    run<1>("v_mul_f32", 8);
    run<3>("v_pk_add_f32 (neg)", 8);
    run<2>("v_cmp_le_f32 -> s_bcnt1 x4", 4);
    return 0;
}
