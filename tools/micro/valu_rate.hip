// Issue rate of plain and packed binary32 vector instructions on one SIMD, by waves per SIMD, measured INSIDE the
// kernel: every wave stamps the shader clock (s_memtime) and the 100 MHz reference clock (s_memrealtime) around
// its loop, so the result is in shader cycles whatever clock the chip holds under this load, and the clock itself
// is reported (delta s_memtime / delta s_memrealtime x 100 MHz).  ONE workgroup per CU (grid = CUs) of 256 k threads:
// its 4 k waves go to the CU's four SIMDs in turn, k to each (k = 1, 2, 4; for 8 per SIMD two such workgroups of 1024
// threads per CU).  The cycles of a wave's loop divided by the instructions its SIMD's k waves issued meanwhile is the
// issue interval of that SIMD; the wall time of the launch must agree with it (both are printed).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o valu_rate valu_rate.hip && ./valu_rate [out.json]
//
// Kinds: plain v_mul_f32, packed v_pk_mul_f32, packed v_pk_add_f32 with a negated operand (the subtraction), and the
// two forms of the pair test's distance arithmetic -- nine plain instructions per test, and the packed form the
// merged-window kernel uses (v_pk_add, v_sub, v_pk_mul, v_mul, v_add, v_add, v_cmp: seven instructions per test).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float v2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(1024) spin(float *out, unsigned long long *stamps, int iters)
{
    v2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = a0 * 1.5f, a2 = a0 + 3.f, a3 = a0 - 1.f, a4 = a0 * 0.5f, a5 = a0 + 7.f, a6 = a0 * 3.f, a7 = a0 - 9.f;
    const v2 m = {1.0000001f, 0.9999999f};
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int rep = 0; rep < 8; rep++) {           // (64 to 144 vector instructions per turn: the loop's own branch does not show)
        if (KIND == 0) {          // packed multiply, 8 independent chains
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else if (KIND == 1) {   // plain multiply, 8 independent chains
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x));
        } else if (KIND == 2) {   // packed add with a negated operand (the subtraction idiom)
            asm volatile("v_pk_add_f32 %0, %0, %8 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else {
            // two tests per turn against candidates the compiler cannot see through (the empty asm marks them changed)
            float cx = a0.x, cy = a1.x, cz = a2.x, ex = a3.x, ey = a4.x, ez = a5.x;
            asm volatile("" : "+v"(cx), "+v"(cy), "+v"(cz), "+v"(ex), "+v"(ey), "+v"(ez));
            float s0, s1;
            if (KIND == 3) {      // nine plain instructions per test: 3 sub, 3 mul, 2 add, 1 compare
                const float dx = m.x - cx, dy = m.y - cy, dz = a6.x - cz, fx = m.x - ex, fy = m.y - ey, fz = a6.x - ez;
                s0 = (dx * dx + dy * dy) + dz * dz;
                s1 = (fx * fx + fy * fy) + fz * fz;
            } else {              // the packed form: v_pk_add (x,y), v_sub (z), v_pk_mul, v_mul, v_add, v_add, compare
                const v2 d = m - v2{cx, cy}, f = m - v2{ex, ey};
                const float dz = a6.x - cz, fz = a6.x - ez;
                const v2 dd = d * d, ff = f * f;
                s0 = (dd.x + dd.y) + dz * dz;
                s1 = (ff.x + ff.y) + fz * fz;
            }
            acc += __popcll(__builtin_amdgcn_ballot_w64(s0 <= a7.x)) + __popcll(__builtin_amdgcn_ballot_w64(s1 <= a7.x));
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + acc == 123.456f) out[threadIdx.x] = s.y;
}

static FILE *g_json;
static bool g_first = true;

template <int KIND>
static void run(const char *name, int vector_instr_per_iter, int tests_per_iter)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *d;
    unsigned long long *stamps;
    (void)hipMalloc(&d, 4096);
    (void)hipMalloc(&stamps, (size_t)cus * 2 * 16 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 5000, reps = 8;
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        const int threads = waves_per_simd <= 4 ? 256 * waves_per_simd : 1024, blocks = cus * (waves_per_simd <= 4 ? 1 : 2);
        for (int warm = 0; warm < 3; warm++) spin<KIND><<<blocks, threads>>>(d, stamps, iters);       // the clock settles under this load
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        spin<KIND><<<blocks, threads>>>(d, stamps, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const int waves = blocks * threads / 64;
        std::vector<unsigned long long> h((size_t)waves * 2);
        (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int w = 0; w < waves; w++) {
            cyc.push_back((double)h[2 * w]);
            ghz.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 0.1);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(ghz.begin(), ghz.end());
        const double cycles = cyc[cyc.size() / 2], clock = ghz[ghz.size() / 2];
        const double instr_per_simd = (double)iters * reps * vector_instr_per_iter * waves_per_simd;
        const double cyc_per_instr = cycles / instr_per_simd, ns_per_instr = ms * 1e6 / instr_per_simd;
        // whole chip: 4 SIMDs per CU, 64 lanes per wave-instruction
        const double lane_ops_per_s = 64.0 * 4 * cus / (ns_per_instr * 1e-9);
        printf("%-34s waves/SIMD %d: %7.3f ms  %5.2f cycles in-kernel, %4.2f ns = %5.2f cycles by the launch's wall time, per wave-instruction per SIMD; clock %.2f GHz",
               name, waves_per_simd, ms, cyc_per_instr, ns_per_instr, ns_per_instr * clock, clock);
        double tests_per_s = 0;
        if (tests_per_iter) {
            tests_per_s = 64.0 * 4 * cus * (double)iters * reps * tests_per_iter * waves_per_simd / (ms * 1e-3);
            printf(", %.2fe12 tests/s", tests_per_s / 1e12);
        }
        printf("\n");
        if (g_json) {
            fprintf(g_json, "%s\n  {\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave_instruction_per_simd\": %.3f, "
                            "\"ns_per_wave_instruction_per_simd\": %.4f, \"clock_ghz\": %.3f, \"lane_ops_per_s\": %.4e, \"tests_per_s\": %.4e}",
                    g_first ? "" : ",", name, waves_per_simd, ms, cyc_per_instr, ns_per_instr, clock, lane_ops_per_s, tests_per_s);
            g_first = false;
        }
    }
    (void)hipFree(d);
    (void)hipFree(stamps);
}

int main(int argc, char **argv)
{
    if (argc > 1) {
        g_json = fopen(argv[1], "w");
        if (g_json) fprintf(g_json, "{\"what\": \"vector issue rates measured in-kernel (s_memtime / s_memrealtime), tools/micro/valu_rate.hip\", \"rows\": [");
    }
    run<1>("v_mul_f32", 8, 0);
    run<0>("v_pk_mul_f32", 8, 0);
    run<2>("v_pk_add_f32 (neg)", 8, 0);
    run<3>("pair test, 9 plain instructions", 18, 2);
    run<4>("pair test, packed (7 instructions)", 14, 2);
    if (g_json) {
        fprintf(g_json, "\n]}\n");
        fclose(g_json);
    }
    return 0;
}
