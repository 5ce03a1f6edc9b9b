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
    // three-cell runs of the four neighbour pencils.  A run's first and last table entry lie 12 bytes apart:
    // one 16-byte load per run, and all five table reads are issued before any is used.
    const uint32_t base[5] = {0u, 402u, 162005u, 162408u, 162811u};
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const uint32_t own_hi = T[c + 2];
    u4 run[5];
    // (written as one block: the compiler would load the first and the last entry with two 4-byte loads, and it
    // must not touch the destination registers before the loads have returned)
    asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                 "global_load_dwordx4 %1, %5, off\n\t"
                 "global_load_dwordx4 %2, %6, off\n\t"
                 "global_load_dwordx4 %3, %7, off\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(run[1]), "=&v"(run[2]), "=&v"(run[3]), "=&v"(run[4])
                 : "v"(T + (c + base[1])), "v"(T + (c + base[2])), "v"(T + (c + base[3])), "v"(T + (c + base[4]))
                 : "memory");
    // The five ranges as ONE list (one loop whose length is the lane's total, not five loops that each run as long
    // as the wave's longest): list index t is sorted slot t + shift[r] for the range r it falls into.
    uint32_t first[6], shift[5];
    first[0] = 0;
    shift[0] = p + 1;
    // (after a domain error the binning leaves slots unwritten, and such a slot may hold an older call's point of
    // some other cell: its row-0 range is then empty or meaningless, never negative)
    first[1] = own_hi > p + 1 ? own_hi - (p + 1) : 0u;
#pragma unroll
    for (int r = 1; r < 5; r++) {
        shift[r] = run[r].x - first[r];
        first[r + 1] = first[r] + (run[r].w - run[r].x);
    }
    const uint32_t total = first[5];
    // two candidates per turn, packed arithmetic (v_pk_add/mul_f32); the second one of an odd tail is masked
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f ax = {x, x}, ay = {y, y}, az = {z, z};
    for (uint32_t t = 0; t < total; t += 2) {
        const uint32_t t1 = min(t + 1, total - 1);
        uint32_t s0 = shift[0], s1 = shift[0];
#pragma unroll
        for (int r = 1; r < 5; r++) {
            s0 = t >= first[r] ? shift[r] : s0;
            s1 = t1 >= first[r] ? shift[r] : s1;
        }
        const pt3 v0 = P[t + s0], v1 = P[t1 + s1];
        const v2f cx = {v0.x, v1.x}, cy = {v0.y, v1.y}, cz = {v0.z, v1.z};
        const v2f dx = ax - cx, dy = ay - cy, dz = az - cz;
        const v2f sq = (dx * dx + dy * dy) + dz * dz;            // unfused: see the build flags
        cnt += (sq.x <= geo.d2max) + ((sq.y <= geo.d2max) & (t + 1 < total));
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
