// Rate of the matrix-core distance filter's inner loop on one SIMD, by waves per SIMD.  Per 32 x 32 tile: one
// 16-byte LDS read (the candidates' operand), two v_mfma_f32_32x32x16_f16 on the same operands but for one
// element pair that carries the threshold (E1 = d2 - (T - eps), E2 = d2 - (T + eps), results in vector registers),
// and per result register one v_alignbit_b32 that shifts its sign bit into a bit field; two population counts give
// the lane's hits and possible hits, one compare says whether any result lies inside the error band.  Form 0 is
// the obvious loop instead (one MFMA, two compares per result whose masks go to the scalar unit).
// Prints ns per tile per SIMD and the tests per second of the whole chip.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_filter_rate mfma_filter_rate.hip && ./mfma_filter_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
#define TILES 16

__device__ __forceinline__ f16 mfma_v(h8 a, h8 b)
{
    const f16 zero = {0};
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, zero, 0, 0, 0);
}

template <int FORM>
__global__ void __launch_bounds__(64) spin(unsigned long long *out, int iters, float tlo, float thi)
{
    __shared__ h8 recs[TILES * 64];
    const int lane = threadIdx.x;
    for (int t = 0; t < TILES; t++) {
        h8 v;
        for (int j = 0; j < 8; j++) v[j] = (_Float16)(0.01f * ((lane * 7 + t * 3 + j) % 50) - 0.2f);
        recs[t * 64 + lane] = v;
    }
    h8 a, a2;
    for (int j = 0; j < 8; j++) a[j] = (_Float16)(0.02f * ((lane + j) % 40) - 0.3f);
    a2 = a;
    a2[7] = (_Float16)((float)a[7] + 0.001f);
    __syncthreads();
    unsigned long long hits = 0, amb = 0;
    unsigned int vhits = 0, vamb = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll 2
        for (int t = 0; t < TILES; t++) {
            const h8 b = recs[t * 64 + lane];
            if (FORM == 0) {
                f16 acc = {0};
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned long long m1 = __builtin_amdgcn_ballot_w64(acc[r] <= tlo);
                    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(acc[r] <= thi);
                    hits += __builtin_popcountll(m1);
                    amb |= m1 ^ m2;
                }
            } else if (FORM == 3) {
                // as form 2 with four independent chains of sign-bit shifts and two of minima
                const f16 e1 = mfma_v(a, b);
                unsigned int sa = 0, sb = 0, sc = 0, sd = 0;
                float ma = 3.0e38f, mb = 3.0e38f;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    sa = __builtin_amdgcn_alignbit(sa, __float_as_uint(e1[r]), 31);
                    sb = __builtin_amdgcn_alignbit(sb, __float_as_uint(e1[4 + r]), 31);
                    sc = __builtin_amdgcn_alignbit(sc, __float_as_uint(e1[8 + r]), 31);
                    sd = __builtin_amdgcn_alignbit(sd, __float_as_uint(e1[12 + r]), 31);
                }
#pragma unroll
                for (int r = 0; r < 8; r += 2) {
                    ma = __builtin_fminf(__builtin_fminf(__builtin_fabsf(e1[r]), __builtin_fabsf(e1[r + 1])), ma);
                    mb = __builtin_fminf(__builtin_fminf(__builtin_fabsf(e1[8 + r]), __builtin_fabsf(e1[9 + r])), mb);
                }
                vhits += __builtin_popcount(sa | (sb << 4) | (sc << 8) | (sd << 12));
                vamb += __builtin_fminf(ma, mb) < thi ? 1u : 0u;
            } else if (FORM == 2) {
                // one matrix instruction; the sign bits as before, and the smallest |E| of the lane's results through
                // three-input minima with absolute-value modifiers (one instruction per two results)
                const f16 e1 = mfma_v(a, b);
                unsigned int s1 = 0;
                float m = 3.0e38f;
#pragma unroll
                for (int r = 0; r < 16; r++) s1 = __builtin_amdgcn_alignbit(s1, __float_as_uint(e1[r]), 31);
#pragma unroll
                for (int r = 0; r < 16; r += 2) m = __builtin_fminf(__builtin_fminf(__builtin_fabsf(e1[r]), __builtin_fabsf(e1[r + 1])), m);
                vhits += __builtin_popcount(s1);
                vamb += m < thi ? 1u : 0u;
            } else {
                const f16 e1 = mfma_v(a, b), e2 = mfma_v(a2, b);
                unsigned int s1 = 0, s2 = 0;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    s1 = __builtin_amdgcn_alignbit(s1, __float_as_uint(e1[r]), 31);
                    s2 = __builtin_amdgcn_alignbit(s2, __float_as_uint(e2[r]), 31);
                }
                vhits += __builtin_popcount(s1);
                vamb |= s1 ^ s2;
            }
        }
    }
    if (FORM == 0) { if (lane == 0) out[blockIdx.x] = hits + amb; }
    else out[blockIdx.x * 64 + lane] = vhits + vamb;
}

int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned long long *out;
    (void)hipMalloc(&out, 8 * 65536 * 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int form = 2; form <= 3; form++)
        for (int k = 1; k <= 8; k *= 2) {
            const int grid = cus * 4 * k;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                (void)hipEventRecord(e0);
                if (form == 0) hipLaunchKernelGGL(spin<0>, dim3(grid), dim3(64), 0, 0, out, iters, 0.5f, 0.6f);
                else if (form == 1) hipLaunchKernelGGL(spin<1>, dim3(grid), dim3(64), 0, 0, out, iters, 0.5f, 0.6f);
                else if (form == 2) hipLaunchKernelGGL(spin<2>, dim3(grid), dim3(64), 0, 0, out, iters, 0.5f, 0.6f);
                else hipLaunchKernelGGL(spin<3>, dim3(grid), dim3(64), 0, 0, out, iters, 0.5f, 0.6f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double tiles_per_simd = (double)iters * TILES * k;
            const double ns = best * 1e6 / tiles_per_simd;
            printf("form %d  waves/SIMD %d: %.3f ms  %.1f ns per tile per SIMD (%.0f cycles at 2.35 GHz)  %.3e tests/s\n", form, k,
                   best, ns, ns * 2.35, 1024.0 * cus * 4 / (ns * 1e-9));
        }
    return 0;
}
