// Device-side arithmetic shared by every kernel of the pair-counting path.
// gfx950 only (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/paircount.h"   // PC_NBB, PC_NCELLS, PC_NLAYERS
#define PC_NBB2 (PC_NBB * PC_NBB)
#define PC_WAVE 64

// Geometry of a run, passed to every kernel by value.  The reference hard-codes it
// (main.c:30-31,134-136,165: domain from -10, cell edge BBSIZE 0.05, threshold 0.0025);
// pc_geom_default() is exactly that, and other values go through the same arithmetic.
// The table keeps its 403 cells per axis whatever ncell is: cells from ncell on are
// simply never occupied (a point that maps there is outside the domain).
struct pc_geom {
    double add;        // minus the lower domain bound: cell = (int)(((double)f + add) / cell)
    double cell;       // cell edge
    float addf, invf;  // binary32 shortcut: floor((f + addf) * invf) away from cell boundaries
    float hi;          // the shortcut is used for quotients in [0, hi)
    int ncell;         // occupied cells per axis, at most PC_NLAYERS
    float d2max;       // squared cutoff: a pair counts if (dx*dx + dy*dy) + dz*dz <= d2max
};

// Cell index along one axis: reference main.c:134-136,
//   (int)((f - (-10.)) / 0.05)   with f promoted to double.
// f - (-10.0) and f + 10.0 are the same IEEE operation; the divide must stay a
// correctly rounded binary64 division (no reciprocal multiply).
// A NaN or a quotient beyond int converts to INT_MIN, as cvttsd2si does in the
// reference's build (the GPU conversion would give 0 for NaN: a valid cell).
__device__ __forceinline__ int pc_cell_index_exact(float f, const pc_geom &g)
{
    const double q = ((double)f + g.add) / g.cell;
    if (!(q > -2147483649.0 && q < 2147483648.0)) return (int)0x80000000;
    return (int)q;
}

// The same value at binary32 cost for all but about 2 points in 1000.
// u = fl32(fl32(f + 10) * 20) differs from the exact quotient q by less than
// 3.5e-5 for 0 <= u < 401 (half an ulp of f + 10 <= 2^-20, times 20, plus half
// an ulp of the product <= 2^-16; q's own two binary64 roundings and the
// representation error of 0.05 are below 1e-12), so whenever u lies at least
// 1e-3 away from an integer, floor(u) == floor(q) == the reference's index.
// Points closer to a cell boundary, negative u and everything outside the grid
// take the reference's formula itself.  pc_selfcheck_cell_index_device compares
// the two functions over all 2^32 bit patterns.  (For another geometry the host
// checks that the same error budget holds -- the rounding of addf and invf enter it --
// and otherwise sets hi = 0, which sends every point through the exact formula.)
__device__ __forceinline__ int pc_cell_index(float f, const pc_geom &g)
{
    const float u = __fmul_rn(__fadd_rn(f, g.addf), g.invf);
    const float fl = floorf(u);
    const float fr = __fsub_rn(u, fl);
    int c = (int)fl;
    if (!(fr >= 1.0e-3f && fr <= 0.999f && u >= 0.0f && u < g.hi)) c = pc_cell_index_exact(f, g);
    return c;
}

// Linear key: reference main.c:32 INDEX(i,j,k).
__device__ __forceinline__ uint32_t pc_key(int i, int j, int k)
{
    return (uint32_t)(i * PC_NBB2 + j * PC_NBB + k);
}

__device__ __forceinline__ bool pc_in_grid(int i, int j, int k, const pc_geom &g)
{
    const unsigned nc = (unsigned)g.ncell;
    return ((unsigned)i < nc) & ((unsigned)j < nc) & ((unsigned)k < nc);
}

// A call of the whole path may be served by the fused pair test (pc_count_fused.h), which needs neither
// the z-sorted array nor the table: fgate[0] = largest pencil neighbourhood of the slab (written by the
// level-Y scan), fgate[1] = what the fused kernel holds.  The kernels of the classic route return at once
// when it serves the call (fgate == nullptr: there is no such alternative in this call).
__device__ __forceinline__ bool pc_fused_serves(const uint32_t *fgate)
{
    return fgate != nullptr && fgate[0] <= fgate[1];
}

// Distance predicate: reference main.c:85-91 (dpps imm 0x71) and main.c:165.
// Three products, summed as (dx*dx + dy*dy) + dz*dz, every step a separately
// rounded binary32 operation -- __fmul_rn/__fadd_rn are never contracted into
// an FMA.  (double)d2 <= 0.0025 is equivalent to d2 <= 0.0025f because
// 0.0025f < 0.0025 < nextafterf(0.0025f, 1).
__device__ __forceinline__ bool pc_hit(float xi, float yi, float zi, float xj, float yj, float zj, float d2max)
{
    float dx = __fsub_rn(xi, xj), dy = __fsub_rn(yi, yj), dz = __fsub_rn(zi, zj);
    float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    return s <= d2max;
}

// Orders one wave's LDS traffic: LDS operations of a wave complete in issue
// order, so draining the counter makes every lane's earlier LDS writes and
// atomics visible to every lane's later reads.  No other wave touches this LDS.
__device__ __forceinline__ void pc_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Inclusive prefix sum over the 64 lanes with DPP row shifts and row broadcasts
// (six VALU operations, no LDS traffic).
__device__ __forceinline__ uint32_t pc_wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// Sum of a 32-bit value over the wave (DPP row ops + cross-row broadcast).
__device__ __forceinline__ uint32_t pc_wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum of per-thread 64-bit partials, one atomic per block.
// NWAVES = blockDim.x / 64.  Every thread of the block must call it.
template <int NWAVES>
__device__ __forceinline__ void pc_block_add_u64(unsigned long long part, unsigned long long *dst)
{
    __shared__ unsigned long long wsum[NWAVES];
    unsigned long long s = part;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned l2 = __shfl_xor((unsigned)s, off, 64);
        unsigned h2 = __shfl_xor((unsigned)(s >> 32), off, 64);
        s += ((unsigned long long)h2 << 32) | l2;
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; w++) t += wsum[w];
        if (t) atomicAdd(dst, t);
    }
}
