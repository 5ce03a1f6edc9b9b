// Pair test, variant 0 ("global"): one thread per home particle, candidates
// read straight from global memory.  Restates counter_thread's loop nest
// (reference main.c:210-229) literally on the sorted array: the 13 forward
// neighbour cells of key c are the five contiguous key ranges
//   c+1 | c+402..c+404 | c+162005..c+162007 | c+162408..c+162410 | c+162811..c+162813
// (offsets of main.c:195-209; the 403-wide padding keeps face aliasing empty),
// and "later points of the same cell" are the sorted slots after the home
// particle.  No staging and no per-window bookkeeping: the fastest form for sparse
// inputs (below about two thirds of a point per cell), where it is the default, and the
// cross-check for the staged variants elsewhere.  pc_set_variant("count", 0).
// The grid is capped by the host (grid-stride loop): every workgroup ends with
// one atomic on the same address, and those serialise.
#pragma once
#include "pc_device.h"

// Pairs of the home particle in sorted slot p with its candidates (the loop nest of
// main.c:210-229 for one point): later slots of its cell, and the 13 forward cells.
// Slots the binning left unwritten (it drops out-of-domain points and flags the status
// word) still hold whatever the scratch use of the array put there: a point outside
// the grid is never a home particle, and its key must not index the table.
__device__ __forceinline__ uint32_t pc_count_one_home(const float *__restrict__ pts, const uint32_t *__restrict__ T, uint32_t p, const pc_geom &geo)
{
    struct pt3 { float x, y, z; };                 // one 12-byte load per point
    const pt3 *P = reinterpret_cast<const pt3 *>(pts);
    const pt3 me = P[p];
    const float x = me.x, y = me.y, z = me.z;
    const int ci = pc_cell_index(x, geo), cj = pc_cell_index(y, geo), ck = pc_cell_index(z, geo);
    if (!pc_in_grid(ci, cj, ck, geo)) return 0;
    const uint32_t c = pc_key(ci, cj, ck);
    uint32_t cnt = 0;
    // the five candidate ranges: same cell (later slots) with cell c+1 = slots (p, T[c+2]), then the
    // three-cell runs of the four neighbour pencils; all ten table reads are issued before any is used
    const uint32_t base[5] = {0u, 402u, 162005u, 162408u, 162811u};
    uint32_t lo[5], hi[5];
    lo[0] = p + 1;
    hi[0] = T[c + 2];
#pragma unroll
    for (int r = 1; r < 5; r++) {
        lo[r] = T[c + base[r]];
        hi[r] = T[c + base[r] + 3];
    }
    // two candidates per turn, packed arithmetic (v_pk_add/mul_f32); the second one of an odd tail is masked
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f ax = {x, x}, ay = {y, y}, az = {z, z};
#pragma unroll
    for (int r = 0; r < 5; r++) {
        for (uint32_t q = lo[r]; q < hi[r]; q += 2) {
            const pt3 v0 = P[q], v1 = P[min(q + 1, hi[r] - 1)];
            const v2f cx = {v0.x, v1.x}, cy = {v0.y, v1.y}, cz = {v0.z, v1.z};
            const v2f dx = ax - cx, dy = ay - cy, dz = az - cz;
            const v2f sq = (dx * dx + dy * dy) + dz * dz;            // unfused: see the build flags
            cnt += (sq.x <= geo.d2max) + ((sq.y <= geo.d2max) & (q + 1 < hi[r]));
        }
    }
    return cnt;
}

__global__ void __launch_bounds__(256)
pc_c0_count_global(const float *__restrict__ pts, const uint32_t *__restrict__ T,
                   uint32_t c_lo, uint32_t c_hi,   // home keys [c_lo, c_hi)
                   unsigned long long *__restrict__ out, const pc_geom geo, const uint32_t *__restrict__ fgate)
{
    if (pc_fused_serves(fgate)) return;
    unsigned long long acc = 0;
    const uint32_t first = T[c_lo], last = T[c_hi];   // their slots in sorted order
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t p64 = (uint64_t)first + blockIdx.x * blockDim.x + threadIdx.x; p64 < last; p64 += stride)
        acc += pc_count_one_home(pts, T, (uint32_t)p64, geo);
    pc_block_add_u64<4>(acc, out);
}

// dist2() call count of the reference for home cells [c_first, c_last):
//   sum_c  n_c(n_c-1)/2 + n_c * (points in the 13 forward neighbour cells)
__global__ void __launch_bounds__(256)
pc_count_tests(const uint32_t *__restrict__ T, uint32_t c_first, uint32_t c_last,
               unsigned long long *__restrict__ out)
{
    unsigned long long acc = 0;
    uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t c64 = (uint64_t)c_first + blockIdx.x * blockDim.x + threadIdx.x; c64 < c_last; c64 += stride) {
        uint32_t c = (uint32_t)c64;
        uint32_t b = T[c], e = T[c + 1];
        unsigned long long nc = e - b;
        if (nc == 0) continue;
        unsigned long long nb = (unsigned long long)(T[c + 2] - e)
                              + (T[c + 405] - T[c + 402])
                              + (T[c + 162008] - T[c + 162005])
                              + (T[c + 162411] - T[c + 162408])
                              + (T[c + 162814] - T[c + 162811]);
        acc += nc * (nc - 1) / 2 + nc * nb;
    }
    pc_block_add_u64<4>(acc, out);
}
