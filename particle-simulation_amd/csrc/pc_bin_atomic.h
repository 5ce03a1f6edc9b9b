// Binning variant 0 ("atomic"): a direct counting sort over the full key space.
//   A1 keys + per-cell histogram (global atomics)
//   A2 inclusive scan of the histogram (3 kernels: reduce / spine / apply)
//   A3 scatter with a per-cell atomic cursor
// Simple and obviously correct; every point costs two scattered global atomics,
// so it runs far below the HBM roofline.  Kept as the cross-check for the
// LDS-staged variant and selectable with pc_set_variant("bin", 0).
//
// Table convention (shared by all variants): T has PC_NCELLS+2 entries and,
// when binning is done, T[c] is the first sorted slot of key c (T[c+1] its end).
// Variant 0 gets there in place: the histogram is accumulated at T[key+2], an
// inclusive scan turns T[c+1] into start(c), and the scatter's atomic cursor on
// T[key+1] leaves start(c+1) there.
#pragma once
#include "pc_device.h"

#define PC_SCAN_TPB 256
#define PC_SCAN_EPT 16
#define PC_SCAN_TILE (PC_SCAN_TPB * PC_SCAN_EPT)

// replaces main.c:134-137 + the counting side of hashmap.h:94-135
__global__ void __launch_bounds__(256)
pc_a1_keys_hist(const float *__restrict__ xyz, uint64_t n, uint32_t *__restrict__ keys,
                uint32_t *__restrict__ table, int32_t *__restrict__ status, const pc_geom geo)
{
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        float x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
        int i = pc_cell_index(x, geo), j = pc_cell_index(y, geo), k = pc_cell_index(z, geo);
        if (!pc_in_grid(i, j, k, geo)) { bad = true; keys[p] = 0xFFFFFFFFu; continue; }
        uint32_t key = pc_key(i, j, k);
        keys[p] = key;
        atomicAdd(&table[key + 2], 1u);
    }
    if (bad) atomicOr(status, 1);
}

// every float bit pattern: shortcut index against the reference expression
__global__ void __launch_bounds__(256)
pc_selfcheck_cell_index(unsigned long long *__restrict__ mismatches, const pc_geom geo)
{
    unsigned int bad = 0;
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * 256u;     // grid 65536 x 256 threads x 256 patterns = 2^32
    for (uint32_t q = 0; q < 256u; q++) {
        const float f = __uint_as_float(first + q);
        if (f != f) continue;                                          // NaN: the conversion to int is not defined by the reference
        bad += pc_cell_index(f, geo) != pc_cell_index_exact(f, geo);
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// keys only (the stage entry point pc_cell_keys_device)
__global__ void __launch_bounds__(256)
pc_keys_only(const float *__restrict__ xyz, uint64_t n, uint32_t *__restrict__ keys,
             int32_t *__restrict__ status, const pc_geom geo)
{
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        float x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
        int i = pc_cell_index(x, geo), j = pc_cell_index(y, geo), k = pc_cell_index(z, geo);
        if (!pc_in_grid(i, j, k, geo)) bad = true;
        keys[p] = (uint32_t)(i * PC_NBB2 + j * PC_NBB + k);   // INDEX(), main.c:32
    }
    if (bad) atomicOr(status, 1);
}

// ---- inclusive scan over m uint32 (in place) ------------------------------
__device__ __forceinline__ uint32_t pc_block_excl_scan(uint32_t v, uint32_t *total)
{
    // exclusive scan of one value per thread over a 256-thread block
    __shared__ uint32_t wtot[PC_SCAN_TPB / 64];
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < PC_SCAN_TPB / 64; w++) {
        if (w < wave) base += wtot[w];
        tot += wtot[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ void __launch_bounds__(PC_SCAN_TPB)
pc_a2_scan_reduce(const uint32_t *__restrict__ a, uint64_t m, uint32_t *__restrict__ tile_sums)
{
    uint64_t base = (uint64_t)blockIdx.x * PC_SCAN_TILE + (uint64_t)threadIdx.x * PC_SCAN_EPT;
    uint32_t s = 0;
    if (base + PC_SCAN_EPT <= m) {
        const uint4 *v = reinterpret_cast<const uint4 *>(a + base);
#pragma unroll
        for (int q = 0; q < PC_SCAN_EPT / 4; q++) { uint4 t = v[q]; s += t.x + t.y + t.z + t.w; }
    } else {
        for (int e = 0; e < PC_SCAN_EPT; e++) if (base + e < m) s += a[base + e];
    }
    uint32_t tot;
    pc_block_excl_scan(s, &tot);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block: exclusive scan of the tile sums, in place
__global__ void __launch_bounds__(PC_SCAN_TPB)
pc_a2_scan_spine(uint32_t *__restrict__ tile_sums, uint32_t ntiles)
{
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < ntiles; b0 += PC_SCAN_TPB) {
        uint32_t idx = b0 + threadIdx.x;
        uint32_t v = idx < ntiles ? tile_sums[idx] : 0u;
        uint32_t tot;
        uint32_t ex = pc_block_excl_scan(v, &tot);
        if (idx < ntiles) tile_sums[idx] = carry + ex;
        carry += tot;
    }
}

__global__ void __launch_bounds__(PC_SCAN_TPB)
pc_a2_scan_apply(uint32_t *__restrict__ a, uint64_t m, const uint32_t *__restrict__ tile_offs)
{
    uint64_t base = (uint64_t)blockIdx.x * PC_SCAN_TILE + (uint64_t)threadIdx.x * PC_SCAN_EPT;
    uint32_t v[PC_SCAN_EPT];
    bool full = base + PC_SCAN_EPT <= m;
    if (full) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a + base);
#pragma unroll
        for (int q = 0; q < PC_SCAN_EPT / 4; q++) {
            uint4 t = src[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int e = 0; e < PC_SCAN_EPT; e++) v[e] = base + e < m ? a[base + e] : 0u;
    }
    uint32_t s = 0;
#pragma unroll
    for (int e = 0; e < PC_SCAN_EPT; e++) { s += v[e]; v[e] = s; }   // thread-local inclusive
    uint32_t tot;
    uint32_t ex = pc_block_excl_scan(s, &tot) + tile_offs[blockIdx.x];
    if (full) {
        uint4 *dst = reinterpret_cast<uint4 *>(a + base);
#pragma unroll
        for (int q = 0; q < PC_SCAN_EPT / 4; q++)
            dst[q] = make_uint4(v[4 * q] + ex, v[4 * q + 1] + ex, v[4 * q + 2] + ex, v[4 * q + 3] + ex);
    } else {
#pragma unroll
        for (int e = 0; e < PC_SCAN_EPT; e++) if (base + e < m) a[base + e] = v[e] + ex;
    }
}

// replaces atomic_insert (hashmap.h:69-91): claim a slot, store the point
__global__ void __launch_bounds__(256)
pc_a3_scatter(const float *__restrict__ xyz, const uint32_t *__restrict__ keys, uint64_t n,
              uint32_t *__restrict__ table, float *__restrict__ sorted)
{
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        uint32_t key = keys[p];
        if (key == 0xFFFFFFFFu) continue;
        float x = xyz[3 * p], y = xyz[3 * p + 1], z = xyz[3 * p + 2];
        uint32_t slot = atomicAdd(&table[key + 1], 1u);
        sorted[3 * (uint64_t)slot] = x;
        sorted[3 * (uint64_t)slot + 1] = y;
        sorted[3 * (uint64_t)slot + 2] = z;
    }
}
