// Binning variant 1 ("lds"): most-significant-digit bucket sort in three levels
// that follow the key's own structure (reference main.c:32, key = i*403^2 + j*403 + k):
//
//   level X  points -> 403 x layers      L1 histogram, L2 scan, L3 partition
//   level Y  layer  -> 403 pencils (i,j) L4 histogram, L5 scan, L6 partition
//   level Z  pencil -> 403 cells         L7: one wave sorts one pencil in LDS and
//                                        writes its slice of the cell-start table
//
// Replaces insert/atomic_insert/flatten (reference hashmap.h:69-153) and supersedes
// variant 0's per-point global atomics: every histogram and every slot claim is an
// LDS atomic; global atomics are one per (tile, non-empty bin).  Each partition
// pass sorts a tile of points by bin inside LDS and then writes every bin's run
// with 12-byte stores from consecutive lanes, so all three passes stream at HBM rate.
//
// L3 also records every point's y cell (2 bytes, in level-X order), so the level-Y
// histogram reads 2 bytes per point instead of the point itself.
// Data movement per point: L1 12 B read; L3 12 B read + 14 B write; L4 2 B read;
// L6 12 B read + 12 B write; L7 12 B read + 12 B write  => 88 B/point, plus 4 B per
// cell of table written by L7.
// The order of points inside a cell is unspecified (as in the reference).
#pragma once
#include "pc_device.h"

#define PC_L_TPB 256
#ifndef PC_L_PPT
#define PC_L_PPT 16
#endif
#define PC_L_TILE (PC_L_TPB * PC_L_PPT)        // points of a tile
#define PC_L_LDS_POINTS 4096                   // points of a tile that the partition kernels hold in LDS at a time: 48 KB
#define PC_L7H_TILE 4096                       // tile of the tile-parallel sort of huge pencils (L7H)
static_assert(PC_L_TILE % PC_L_LDS_POINTS == 0, "a tile is written out in whole passes of PC_L_LDS_POINTS points");
#ifndef PC_L_PTPB
#define PC_L_PTPB 512                          // partition kernels: threads per tile (level Y and the huge pencils) ...
#endif
#ifndef PC_L_XTPB
#define PC_L_XTPB 1024                         // ... and in level X (measured: 0.55 ms against 0.58 with 512; level Y 0.48 against 0.50 the other way)
#endif
#define PC_L_PPPT (PC_L_TILE / PC_L_PTPB)      // and points per thread
#define PC_L_NPENCILS (PC_NBB * PC_NBB)        // 162409, padding pencils included
#define PC_L_BINS 404

struct pc_pt3 { float x, y, z; };              // one point of the arrays: 12 bytes, 4-byte aligned

// Streaming accesses.  A load or store can carry the non-temporal hint (the `nt` bit of the instruction).  What it
// does to a kernel AND to the kernel after it (which finds more or less of its input in the 256 MB memory-side
// cache) was measured at 100 M uniform points, kernel by kernel (profiles/r3e_cache_hints.txt):
//   L1's loads          0.208 -> 0.190 ms, nothing else moves: always on
//   L7's loads + stores 0.489 -> 0.478 ms at 100 M points, but 0.067 -> 0.074 ms on a 12.7 M-point slab, whose
//                       arrays the cache holds whole: on for inputs above PC_L_STREAM_POINTS only (run time)
//   partition loads     the kernel itself 1 % faster, the kernels after it 3-20 % slower: off
//   partition stores    3.2x slower (the runs' partial lines no longer merge in the L2): off
//   L4's loads, L7's table stores   slower: off
// PC_L_NT switches them for measurements (bit 0: loads of the partition kernels, 1: their stores, 2: loads of L7,
// 3: its stores, 4: loads of L1, 5: loads of level X's partition only, 6: of level Y's only, 7: loads of L4,
// 8: L7's table stores); the product build has bit 4 set and gives L7 its hint at run time.
#ifndef PC_L_NT
#define PC_L_NT 16
#endif
#define PC_L_STREAM_POINTS 40000000ull         // L7 streams (nt loads and stores) above this many points

// Direction in which a kernel walks its input.  A kernel that starts where the kernel before it stopped writing
// finds those lines in the memory-side cache: L7 takes the pencils from the last to the first, because level Y's
// partition wrote the last layers last (0.476 -> 0.464 ms at 100 M points).  Measured and left alone
// (profiles/r3e_cache_hints.txt): level Y's partition, L4 and the merged-window pair test backwards (no gain),
// level X's partition backwards behind an L1 without the non-temporal hint (L1 loses more than L3 gains).
// PC_L_REV, a bit per kernel, for measurements: 1 level Y's partition, 2 L7, 4 the merged-window pair test,
// 8 level X's partition, 16 L4.
#ifndef PC_L_REV
#define PC_L_REV 2
#endif
typedef float pc_v3f_u __attribute__((ext_vector_type(3), aligned(4)));
typedef float pc_v4f_a __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ pc_pt3 pc_load_pt(const pc_pt3 *p)
{
    if constexpr (NT) {
        const pc_v3f_u v = __builtin_nontemporal_load(reinterpret_cast<const pc_v3f_u *>(p));
        return pc_pt3{v.x, v.y, v.z};
    } else
        return *p;
}
template <bool NT>
__device__ __forceinline__ void pc_store_pt(pc_pt3 *p, const pc_pt3 &v)
{
    if constexpr (NT) {
        const pc_v3f_u w = {v.x, v.y, v.z};
        __builtin_nontemporal_store(w, reinterpret_cast<pc_v3f_u *>(p));
    } else
        *p = v;
}

#define PC_L_NREP 8                            // levels X and Y keep one cursor set per XCD group (tile % 8)
#define PC_L_PSTRIDE (PC_L_NPENCILS + 7)        // pencil counters per replica (padded)

// workspace carve-up (uint32 units after the N*12-byte temp array)
struct pc_lds_ws {
    float *tmp;               // N*3 floats
    uint32_t *layer_count;    // [NREP][404] zeroed each call   } contiguous, one memset
    uint32_t *pencil_count;   // [NREP][PSTRIDE] zeroed each call }
    uint32_t *layer_start;    // [404]
    uint32_t *layer_cursor;   // [NREP][404]
    uint32_t *layer_tile;     // [404]  prefix of tiles per layer (level Y)
    uint32_t *tile_layer;     // [tiles+404] layer of each level-Y tile
    uint32_t *pencil_start;   // [162409+1]
    uint32_t *pencil_cursor;  // [NREP][PSTRIDE]
    uint16_t *ycell;          // [N] y cell of every point of the level-X output (0xFFFF: outside the grid)
    uint32_t *big_count;      // [8] zeroed with the counters: [0] entries of big_list, [1] of huge_list, [2] largest pencil neighbourhood of the
                              //     home layers and [3] the fused pair test's capacity (pc_fused_serves), [4] entries of fused_list
    uint32_t *big_list;       // [162409] pencils with more points than one wave of L7 takes
    uint32_t *huge_list;      // [n/8192+2] pencils with more points than one workgroup of L7B takes (count: big_count[1])
    uint32_t *huge_count;     // [n/8192+2][404] their z-cell counts, then ...
    uint32_t *huge_cursor;    // [n/8192+2][404] ... the write cursors of their cells
    uint32_t *fused_list;     // [162409] home pencils the fused pair test hands to its second instantiation
};

// ---- L1: histogram of x layers ---------------------------------------------
// Tile t of the input (4096 points) belongs to replica t % NREP.  A layer's
// region of the level-X output is split into NREP sub-regions, one per replica:
// the partition kernel runs tile t as workgroup t, workgroups t and t+8 share an
// XCD (round-robin dispatch), so each sub-region is written by one XCD only and
// its partial lines complete in that XCD's L2; it also spreads the cursor
// atomics over NREP words per layer.  The grid is a multiple of NREP, so every
// tile a workgroup visits has the same replica.
__global__ void __launch_bounds__(PC_L_TPB)
pc_l1_hist_x(const float *__restrict__ xyz, uint64_t n, uint32_t *__restrict__ layer_count,
             int32_t *__restrict__ status, const pc_geom geo)
{
    __shared__ uint32_t h[PC_L_BINS];
    for (int b = threadIdx.x; b < PC_L_BINS; b += PC_L_TPB) h[b] = 0;
    __syncthreads();
    bool bad = false;
    const uint64_t ntiles = (n + PC_L_TILE - 1) / PC_L_TILE;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint64_t first = t * PC_L_TILE;
        const uint32_t cnt = (uint32_t)min((uint64_t)PC_L_TILE, n - first);
        float xs[PC_L_PPT];
#pragma unroll
        for (int j = 0; j < PC_L_PPT; j++)
            xs[j] = (PC_L_NT & 16) ? __builtin_nontemporal_load(xyz + 3 * (first + min((uint32_t)(threadIdx.x + j * PC_L_TPB), cnt - 1)))
                                   : xyz[3 * (first + min((uint32_t)(threadIdx.x + j * PC_L_TPB), cnt - 1))];
#pragma unroll
        for (int j = 0; j < PC_L_PPT; j++) {
            if ((uint32_t)(threadIdx.x + j * PC_L_TPB) < cnt) {
                const int i = pc_cell_index(xs[j], geo);
                if ((unsigned)i < (unsigned)geo.ncell) atomicAdd(&h[i], 1u);
                else bad = true;
            }
        }
    }
    __syncthreads();
    uint32_t *mine = layer_count + (blockIdx.x % PC_L_NREP) * PC_L_BINS;
    for (int b = threadIdx.x; b < 400; b += PC_L_TPB)
        if (h[b]) atomicAdd(&mine[b], h[b]);
    if (bad) atomicOr(status, 1);
}

// L1 for a range [first, end) of the array, run as the points arrive (streaming interface): the counts
// of the ranges add up to what pc_l1_hist_x counts over the whole array, replica by replica -- a point's
// replica is that of its tile in the whole array, so a range may start and end inside a tile.  The grid is
// a multiple of NREP: workgroup w takes the tiles t0 + w, t0 + w + grid, ..., all of one replica.
__global__ void __launch_bounds__(PC_L_TPB)
pc_l1_hist_x_range(const float *__restrict__ xyz, uint64_t first, uint64_t end, uint32_t *__restrict__ layer_count,
                   int32_t *__restrict__ status, const pc_geom geo)
{
    __shared__ uint32_t h[PC_L_BINS];
    for (int b = threadIdx.x; b < PC_L_BINS; b += PC_L_TPB) h[b] = 0;
    __syncthreads();
    bool bad = false;
    const uint64_t t0 = first / PC_L_TILE, t1 = (end + PC_L_TILE - 1) / PC_L_TILE;
    for (uint64_t t = t0 + blockIdx.x; t < t1; t += gridDim.x) {
        const uint64_t lo = max(first, t * PC_L_TILE), hi = min(end, (t + 1) * PC_L_TILE);
        for (uint64_t p = lo + threadIdx.x; p < hi; p += PC_L_TPB) {
            const int i = pc_cell_index(xyz[3 * p], geo);
            if ((unsigned)i < (unsigned)geo.ncell) atomicAdd(&h[i], 1u);
            else bad = true;
        }
    }
    __syncthreads();
    uint32_t *mine = layer_count + ((t0 + blockIdx.x) % PC_L_NREP) * PC_L_BINS;
    for (int b = threadIdx.x; b < 400; b += PC_L_TPB)
        if (h[b]) atomicAdd(&mine[b], h[b]);
    if (bad) atomicOr(status, 1);
}

// ---- L2: scan of the 403 layer counts (one workgroup) ------------------------
__global__ void __launch_bounds__(512)
pc_l2_scan_x(const uint32_t *__restrict__ layer_count, uint32_t *__restrict__ layer_start,
             uint32_t *__restrict__ layer_cursor, uint32_t *__restrict__ layer_tile,
             uint32_t *__restrict__ tile_layer, uint32_t *__restrict__ fused_cap_word, uint32_t fused_cap,
             unsigned int *__restrict__ zero_words, uint32_t zero_n, unsigned long long *__restrict__ zero_out)
{
    if (threadIdx.x == 0) *fused_cap_word = fused_cap;   // second word of the switch that pc_fused_serves reads
    // A whole-path call clears the pair test's work counters and its result here instead of with two fill
    // launches of their own (every launch costs about 6 us, a tenth of a small slab's step).
    for (uint32_t k = threadIdx.x; k < zero_n; k += 512) zero_words[k] = 0u;
    if (zero_out != nullptr && threadIdx.x == 0) *zero_out = 0ull;
    __shared__ uint32_t a[512], t[512];
    const int i = threadIdx.x;
    uint32_t rep[PC_L_NREP], c = 0;
#pragma unroll
    for (int g = 0; g < PC_L_NREP; g++) {
        rep[g] = i < PC_NBB ? layer_count[g * PC_L_BINS + i] : 0u;
        c += rep[g];
    }
    const uint32_t my_tiles = (c + PC_L_TILE - 1) / PC_L_TILE;
    a[i] = c;
    t[i] = my_tiles;
    __syncthreads();
    for (int off = 1; off < 512; off <<= 1) {          // Hillis-Steele, inclusive
        uint32_t va = i >= off ? a[i - off] : 0u, vt = i >= off ? t[i - off] : 0u;
        __syncthreads();
        a[i] += va;
        t[i] += vt;
        __syncthreads();
    }
    if (i < PC_L_BINS) {
        uint32_t run = a[i] - c;                       // first slot of layer i
        layer_start[i] = run;
#pragma unroll
        for (int g = 0; g < PC_L_NREP; g++) {          // replica sub-regions, in replica order
            layer_cursor[g * PC_L_BINS + i] = run;
            run += rep[g];
        }
        const uint32_t tile0 = t[i] - my_tiles;
        layer_tile[i] = tile0;
        for (uint32_t k = 0; k < my_tiles; k++) tile_layer[tile0 + k] = (uint32_t)i;
    }
}

// ---- L4: histogram of pencils inside each layer ------------------------------
// Level-Y tile b is chunk (b - layer_tile[layer]) of layer tile_layer[b].
__global__ void __launch_bounds__(PC_L_TPB)
pc_l4_hist_y(const uint16_t *__restrict__ ycell, const uint32_t *__restrict__ layer_start,
             const uint32_t *__restrict__ layer_tile, const uint32_t *__restrict__ tile_layer,
             uint32_t *__restrict__ pencil_count, int32_t *__restrict__ status, const pc_geom geo)
{
    __shared__ uint32_t h[PC_L_BINS];
    if (blockIdx.x >= layer_tile[PC_NBB]) return;
    const uint32_t tile = (PC_L_REV & 16) ? layer_tile[PC_NBB] - 1u - blockIdx.x : blockIdx.x;
    const int layer = (int)tile_layer[tile];
    const uint32_t chunk = tile - layer_tile[layer];
    const uint32_t first = layer_start[layer] + chunk * PC_L_TILE;
    const uint32_t cnt = min((uint32_t)PC_L_TILE, layer_start[layer + 1] - first);
    for (int b = threadIdx.x; b < PC_L_BINS; b += PC_L_TPB) h[b] = 0;
    uint32_t ys[PC_L_PPT];
#pragma unroll
    for (int j = 0; j < PC_L_PPT; j++)
        ys[j] = (PC_L_NT & 128) ? __builtin_nontemporal_load(ycell + (uint64_t)first + min((uint32_t)(threadIdx.x + j * PC_L_TPB), cnt - 1))
                                : ycell[(uint64_t)first + min((uint32_t)(threadIdx.x + j * PC_L_TPB), cnt - 1)];
    __syncthreads();
    bool bad = false;
#pragma unroll
    for (int j = 0; j < PC_L_PPT; j++) {
        if ((uint32_t)(threadIdx.x + j * PC_L_TPB) < cnt) {
            if (ys[j] < (unsigned)geo.ncell) atomicAdd(&h[ys[j]], 1u);
            else bad = true;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < 400; b += PC_L_TPB)
        if (h[b]) atomicAdd(&pencil_count[(tile % PC_L_NREP) * PC_L_PSTRIDE + layer * PC_NBB + b], h[b]);
    if (bad) atomicOr(status, 1);
}

// ---- L5: scan of the pencil counts, one workgroup per layer -------------------
// A layer's pencils start where the layer starts (known from L2), so the 403
// rows scan independently: pencil_start[i*403+j] = layer_start[i] + sum_{j'<j}.
// The workgroups of the home layers [x_lo, x_hi) also look at the layer above and
// record (atomic maximum into *max5) the largest number of points any home pencil
// (i,j) of theirs has in its neighbourhood (i,j) (i,j+1) (i+1,j-1..j+1): what the
// fused pair test must hold in LDS at once, and the switch between it and the classic kernels.
// Pencils too big for a wave of L7 go on big_list, those too big for a workgroup of L7B also on huge_list.
__global__ void __launch_bounds__(512)
pc_l5_scan_y(const uint32_t *__restrict__ pencil_count, const uint32_t *__restrict__ layer_start,
             uint32_t *__restrict__ pencil_start, uint32_t *__restrict__ pencil_cursor,
             uint32_t wave_cap, uint32_t *__restrict__ big_count, uint32_t *__restrict__ big_list,
             int x_lo, int x_hi, uint32_t *__restrict__ max5,
             int sort_lo, int sort_hi, uint32_t huge_above, uint32_t *__restrict__ huge_list, uint32_t *__restrict__ huge_count)
{
    __shared__ uint32_t a[512], own[512], above[512], most;
    const int j = threadIdx.x, layer = blockIdx.x;
    uint32_t rep[PC_L_NREP], c = 0, cn = 0;
#pragma unroll
    for (int g = 0; g < PC_L_NREP; g++) {
        rep[g] = j < PC_NBB ? pencil_count[g * PC_L_PSTRIDE + layer * PC_NBB + j] : 0u;
        c += rep[g];
    }
    const bool home_layer = layer >= x_lo && layer < x_hi;
    if (home_layer && layer + 1 < PC_NBB && j < PC_NBB)
#pragma unroll
        for (int g = 0; g < PC_L_NREP; g++) cn += pencil_count[g * PC_L_PSTRIDE + (layer + 1) * PC_NBB + j];
    a[j] = c;
    own[j] = c;
    above[j] = cn;
    if (j == 0) most = 0;
    __syncthreads();
    if (home_layer && j < PC_NLAYERS && c != 0) {
        const uint32_t s5 = c + own[j + 1] + (j > 0 ? above[j - 1] : 0u) + cn + above[j + 1];
        atomicMax(&most, s5);
    }
    for (int off = 1; off < 512; off <<= 1) {
        uint32_t v = j >= off ? a[j - off] : 0u;
        __syncthreads();
        a[j] += v;
        __syncthreads();
    }
    const uint32_t ex = layer_start[layer] + a[j] - c;
    if (j < PC_NBB) {
        pencil_start[layer * PC_NBB + j] = ex;
        if (c > wave_cap) big_list[atomicAdd(big_count, 1u)] = (uint32_t)(layer * PC_NBB + j);   // sorted by L7B
        // more points than a workgroup of L7B sorts, in a layer [sort_lo, sort_hi] whose pencils are wanted sorted:
        // on the list of the tile-parallel sort (L7H), with its row of cell counts cleared (many points in one place: rare)
        if (c > huge_above && layer >= sort_lo && layer <= sort_hi) {
            const uint32_t slot = atomicAdd(big_count + 1, 1u);
            huge_list[slot] = (uint32_t)(layer * PC_NBB + j);
            for (int k = 0; k < PC_L_BINS; k++) huge_count[(uint64_t)slot * PC_L_BINS + k] = 0u;
        }
        uint32_t run = ex;
#pragma unroll
        for (int g = 0; g < PC_L_NREP; g++) {          // replica sub-regions of the pencil, as in level X
            pencil_cursor[g * PC_L_PSTRIDE + layer * PC_NBB + j] = run;
            run += rep[g];
        }
    }
    if (layer == PC_NBB - 1 && j == PC_NBB) pencil_start[PC_L_NPENCILS] = ex;   // total
    if (j == 0 && most != 0) atomicMax(max5, most);
}

#ifdef PC_L_TRACE
// Measurement build only: shader-clock cycles that the workgroups of the partition kernels spend in each phase,
// kept in registers by thread 0 and added at the workgroup's end to one of 64 rows (atomics on a single address
// would serialise the kernel); [level][row][phase], phase 7 = workgroups counted.  Read with pc_trace_read.
#define PC_L_TRACE_ROWS 64
__device__ unsigned long long pc_l_trace[2][PC_L_TRACE_ROWS][8];
#define PC_L_MARK(i)                                                                         \
    do {                                                                                     \
        if (tid == 0) {                                                                      \
            const unsigned long long now_ = __builtin_readcyclecounter();                   \
            spent_[i] = now_ - mark_;                                                        \
            mark_ = now_;                                                                    \
        }                                                                                    \
    } while (0)
#else
#define PC_L_MARK(i) do { } while (0)
#endif

// ---- L3 / L6: partition a tile of points into 403 bins -----------------------
// LEVEL 0: tile = points [blockIdx*TILE, +TILE) of the input, bin = x cell,
//          cursors = layer_cursor; writes ycell[] beside the points.
// LEVEL 1: tile = a chunk of one layer, bin = y cell,
//          cursors = that layer's row of pencil_cursor.
template <int LEVEL>
__global__ void __launch_bounds__(LEVEL == 0 ? PC_L_XTPB : PC_L_PTPB)
pc_l36_partition(const float *__restrict__ src, uint64_t n, float *__restrict__ dst,
                 const uint32_t *__restrict__ layer_start, const uint32_t *__restrict__ layer_tile,
                 const uint32_t *__restrict__ tile_layer, uint32_t *__restrict__ cursor, uint16_t *__restrict__ ycell, const pc_geom geo)
{
    constexpr int TPB = LEVEL == 0 ? PC_L_XTPB : PC_L_PTPB, PPT = PC_L_TILE / TPB;
    __shared__ float buf[PC_L_LDS_POINTS * 3];        // a tile's points in bin order, PC_L_LDS_POINTS at a time: 48 KB, with the
                                                      // three small arrays 54 000 bytes, so that three workgroups share a CU's 160 KB
    constexpr int PASSES = PC_L_TILE / PC_L_LDS_POINTS;
    __shared__ uint32_t hist[PC_L_BINS];
    __shared__ uint32_t lstart[PC_L_BINS];
    __shared__ uint32_t delta[PC_L_BINS];             // global slot of bin's first point minus lstart
    const int tid = threadIdx.x;

    uint64_t first;
    uint32_t cnt;
    if (LEVEL == 0) {
        const uint32_t tile = (PC_L_REV & 8) ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
        first = (uint64_t)tile * PC_L_TILE;
        if (first >= n) return;
        cnt = (uint32_t)min((uint64_t)PC_L_TILE, n - first);
        cursor += (tile % PC_L_NREP) * PC_L_BINS;             // this tile's replica (see L1)
    } else {
        if (blockIdx.x >= layer_tile[PC_NBB]) return;
        const uint32_t tile = (PC_L_REV & 1) ? layer_tile[PC_NBB] - 1u - blockIdx.x : blockIdx.x;
        const int layer = (int)tile_layer[tile];
        const uint32_t chunk = tile - layer_tile[layer];
        first = (uint64_t)layer_start[layer] + (uint64_t)chunk * PC_L_TILE;
        cnt = min((uint32_t)PC_L_TILE, layer_start[layer + 1] - (uint32_t)first);
        cursor += (tile % PC_L_NREP) * PC_L_PSTRIDE + layer * PC_NBB;
    }
#ifdef PC_L_TRACE
    unsigned long long mark_ = __builtin_readcyclecounter(), spent_[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
    for (int b = tid; b < PC_L_BINS; b += TPB) hist[b] = 0;
    // the tile -> registers: thread t takes points t, t+TPB, ... (12-byte loads,
    // consecutive lanes on consecutive points); all loads are issued before any use
    const pc_pt3 *g = reinterpret_cast<const pc_pt3 *>(src) + first;
    pc_pt3 pt[PPT];
#pragma unroll
    for (int j = 0; j < PPT; j++) pt[j] = pc_load_pt<(PC_L_NT & (LEVEL == 0 ? 1 | 32 : 1 | 64)) != 0>(g + min((uint32_t)(tid + j * TPB), cnt - 1));
#ifdef PC_L_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __syncthreads();
    PC_L_MARK(0);                                      // the tile has arrived

    // bin and rank inside the bin (LDS atomic) of each point
    uint32_t tag[PPT];                           // bin | rank << 10, or ~0 for "skip"
#pragma unroll
    for (int j = 0; j < PPT; j++) {
        tag[j] = 0xFFFFFFFFu;
        if ((uint32_t)(tid + j * TPB) < cnt) {
            const uint32_t b = (uint32_t)pc_cell_index(LEVEL == 0 ? pt[j].x : pt[j].y, geo);
            if (b < (unsigned)geo.ncell) tag[j] = b | (atomicAdd(&hist[b], 1u) << 10);
        }
    }
    __syncthreads();
    PC_L_MARK(1);                                      // ranked

    // exclusive scan of the 404 bins by wave 0 (7 bins per lane)
    if (tid < 64) {
        uint32_t c[7], s = 0;
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const int b = tid * 7 + q;
            c[q] = b < PC_L_BINS ? hist[b] : 0u;
            s += c[q];
        }
        uint32_t run = pc_wave_incl_scan(s) - s;
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const int b = tid * 7 + q;
            if (b < PC_L_BINS) lstart[b] = run;
            run += c[q];
        }
    }
    __syncthreads();
    PC_L_MARK(2);                                      // scanned
    // claim global space: one atomic per non-empty bin; its answer is awaited only after the tile has been
    // copied into LDS (the claims take about as long as that copy)
    constexpr int CLAIMS = (PC_L_BINS + TPB - 1) / TPB;
    uint32_t claimed[CLAIMS];
#pragma unroll
    for (int q = 0; q < CLAIMS; q++) {
        const int b = tid + q * TPB;
        claimed[q] = 0;
        if (b < PC_L_BINS && hist[b]) claimed[q] = atomicAdd(&cursor[b], hist[b]);
    }
    // The tile in bin order -> LDS -> global.  A tile larger than the LDS buffer goes in passes: pass q takes the
    // points whose place in the tile's bin order is [q, q + 1) * PC_L_LDS_POINTS (they wait in registers until then),
    // so a bin's run in global memory is as long as the whole tile makes it, not as long as the buffer would.
#pragma unroll
    for (int pass = 0; pass < PASSES; pass++) {
        const uint32_t base = (uint32_t)pass * PC_L_LDS_POINTS;
        if (pass > 0) __syncthreads();                 // the previous pass's reads of buf are done
#pragma unroll
        for (int j = 0; j < PPT; j++) {
            if (tag[j] != 0xFFFFFFFFu) {
                const uint32_t b = tag[j] & 1023u, pos = lstart[b] + (tag[j] >> 10) - base;
                if (PASSES == 1 || pos < PC_L_LDS_POINTS) {   // (unsigned: also false below the pass's range)
                    buf[3 * pos] = pt[j].x;
                    buf[3 * pos + 1] = pt[j].y;
                    buf[3 * pos + 2] = pt[j].z;
                }
            }
        }
        if (pass == 0) {
#pragma unroll
            for (int q = 0; q < CLAIMS; q++) {
                const int b = tid + q * TPB;
                if (b < PC_L_BINS) delta[b] = claimed[q] - lstart[b];
            }
#ifdef PC_L_TRACE
            PC_L_MARK(3);                              // tile scattered into LDS (this thread's share), claims issued
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        }
        __syncthreads();
        if (pass == 0) PC_L_MARK(4);                   // claims answered, LDS complete
        const uint32_t kept = lstart[PC_L_BINS - 1] + hist[PC_L_BINS - 1];
        const uint32_t here = kept > base ? min(kept - base, (uint32_t)PC_L_LDS_POINTS) : 0u;
        // runs -> global, a point per lane (12-byte stores, consecutive lanes on consecutive slots of a run); the bin
        // is computed again from the coordinate, which leaves the LDS to the points alone
        for (uint32_t e = tid; e < here; e += TPB) {
            const pc_pt3 p = {buf[3 * e], buf[3 * e + 1], buf[3 * e + 2]};
            const uint32_t b = (uint32_t)pc_cell_index(LEVEL == 0 ? p.x : p.y, geo);
#ifdef PC_L_EXP_LINEAR                                 // measurement build: the tile goes back where it came from, as one contiguous run
            const uint32_t slot = (uint32_t)first + base + e + 0u * delta[b];
#else
            const uint32_t slot = delta[b] + base + e;
#endif
            pc_store_pt<(PC_L_NT & 2) != 0>(reinterpret_cast<pc_pt3 *>(dst) + slot, p);
            if (LEVEL == 0) {
                const int cy = pc_cell_index(p.y, geo);
                const uint16_t yc = (unsigned)cy < (unsigned)geo.ncell ? (uint16_t)cy : (uint16_t)0xFFFFu;
                if (PC_L_NT & 2) __builtin_nontemporal_store(yc, ycell + slot);
                else ycell[slot] = yc;
            }
        }
    }
#ifdef PC_L_TRACE
    PC_L_MARK(5);                                      // stores issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PC_L_MARK(6);                                      // stores acknowledged
    if (tid == 0) {
        unsigned long long *row = pc_l_trace[LEVEL][blockIdx.x % PC_L_TRACE_ROWS];
        for (int i = 0; i < 7; i++) atomicAdd(&row[i], spent_[i]);
        atomicAdd(&row[7], 1ull);
    }
#endif
}

// ---- L7: one wave sorts one pencil by z cell and writes its table slice --------
// No workgroup barrier and no staging buffer: a pencil's sorted points land in a
// window of a few KB that this one wave writes completely, so the 12-byte stores,
// scattered inside that window, merge into full lines in the XCD's L2.
#define PC_L7_WAVES 2
#define PC_L7_CAP 768                                  // points a wave keeps in registers and stages in LDS ...
#define PC_L7_CAP_SPARSE 192                           // ... and for sparse inputs (less LDS per wave, more waves per CU)


// exclusive scan of the wave's 404 bin counts in place (lane l owns bins 7l..7l+6)
__device__ __forceinline__ void pc_l7_scan_bins(uint32_t *hist, int lane)
{
    uint32_t c[7], s = 0;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const int b = lane * 7 + q;
        c[q] = b < PC_L_BINS ? hist[b] : 0u;
        s += c[q];
    }
    uint32_t run = pc_wave_incl_scan(s) - s;
#pragma unroll
    for (int q = 0; q < 7; q++) {
        const int b = lane * 7 + q;
        if (b < PC_L_BINS) hist[b] = run;
        run += c[q];
    }
}

// ---- L7H: pencils too big for one workgroup, sorted by all of them -------------
// A pencil with more than PC_L7B_CAP_LARGE points (many points in one place) is one more
// level of the same bucket sort: tiles of 4096 points, histogram of z cells (LDS, then
// one global atomic per tile and cell), scan, and a partition pass that sorts each tile
// by cell in LDS and writes every cell's run with coalesced stores.  The pencils are
// listed by L5; tile c of list entry h is taken by workgroup (c + 7 h) mod (workgroups of
// the role), so that many entries with few tiles each spread out as well as one entry with
// thousands.
// The three steps are not launches of their own: an input without such pencils (nearly
// every input) would pay three launches that find nothing to do, about 6 us each.  They
// ride as extra workgroups ("roles") at the end of the grids of kernels that run anyway
// and in the order the steps need: the histogram with L7, the scan with the small L7B
// instantiation, the partition with the large one.  A role's workgroups are numbered
// blk = 0 .. nblk-1.
#define PC_L7H_PPPT (PC_L7H_TILE / PC_L_PTPB)          // points per thread of the partition role
#define PC_L7H_GRID 1024                               // workgroups of the histogram and of the partition role
#define PC_L7H_SCAN_GRID 64                            // and of the scan role (one wave of each works)
__device__ __forceinline__ uint32_t pc_l7h_first_tile(uint32_t h, uint32_t blk, uint32_t nblk)
{
    return (blk + nblk - (7u * h) % nblk) % nblk;
}

// histogram role: TPB threads, h_lds = PC_L_BINS words of LDS
template <int TPB>
__device__ __forceinline__ void pc_l7h_hist_role(uint32_t blk, uint32_t nblk, uint32_t *h_lds, const float *__restrict__ src,
                                                 const uint32_t *__restrict__ pencil_start, const uint32_t *__restrict__ huge_n,
                                                 const uint32_t *__restrict__ huge_list, uint32_t *__restrict__ huge_count,
                                                 int32_t *__restrict__ status, const pc_geom &geo)
{
    const uint32_t entries = *huge_n;
    if (entries == 0) return;
    constexpr int BATCH = 8, PPT = PC_L7H_TILE / TPB;
    static_assert(PPT % BATCH == 0, "tile / threads must be a multiple of the load batch");
    const int tid = threadIdx.x;
    bool bad = false;
    for (uint32_t h = 0; h < entries; h++) {
        const uint32_t pencil = huge_list[h];
        const uint32_t base = pencil_start[pencil], n = pencil_start[pencil + 1] - base;
        const uint32_t ntiles = (n + PC_L7H_TILE - 1) / PC_L7H_TILE;
        const pc_pt3 *g = reinterpret_cast<const pc_pt3 *>(src) + base;
        for (uint32_t t = pc_l7h_first_tile(h, blk, nblk); t < ntiles; t += nblk) {
            const uint32_t first = t * PC_L7H_TILE, cnt = min((uint32_t)PC_L7H_TILE, n - first);
            __syncthreads();
            for (int b = tid; b < PC_L_BINS; b += TPB) h_lds[b] = 0;
            __syncthreads();
            for (int j0 = 0; j0 < PPT; j0 += BATCH) {
                float zs[BATCH];
#pragma unroll
                for (int j = 0; j < BATCH; j++) zs[j] = g[first + min((uint32_t)(tid + (j0 + j) * TPB), cnt - 1)].z;
#pragma unroll
                for (int j = 0; j < BATCH; j++) {
                    if ((uint32_t)(tid + (j0 + j) * TPB) < cnt) {
                        const int k = pc_cell_index(zs[j], geo);
                        if ((unsigned)k < (unsigned)geo.ncell) atomicAdd(&h_lds[k], 1u);
                        else bad = true;
                    }
                }
            }
            __syncthreads();
            for (int b = tid; b < 400; b += TPB)
                if (h_lds[b]) atomicAdd(&huge_count[(uint64_t)h * PC_L_BINS + b], h_lds[b]);
        }
    }
    if (bad) atomicOr(status, 1);
}

// scan role: wave 0 of each workgroup (hist = PC_L_BINS words of LDS); also writes the pencils' table slices
__device__ __forceinline__ void pc_l7h_scan_role(uint32_t blk, uint32_t nblk, uint32_t *hist, const uint32_t *__restrict__ pencil_start,
                                                 const uint32_t *__restrict__ huge_n, const uint32_t *__restrict__ huge_list,
                                                 const uint32_t *__restrict__ huge_count, uint32_t *__restrict__ huge_cursor,
                                                 uint32_t *__restrict__ table)
{
    const uint32_t entries = *huge_n;
    if (threadIdx.x >= 64) return;
    for (uint32_t h = blk; h < entries; h += nblk) {
        const uint32_t pencil = huge_list[h], base = pencil_start[pencil];
        for (int k = threadIdx.x; k < PC_L_BINS; k += 64) hist[k] = huge_count[(uint64_t)h * PC_L_BINS + k];
        pc_wave_lds_sync();
        pc_l7_scan_bins(hist, threadIdx.x);                // exclusive starts, in place
        pc_wave_lds_sync();
        uint32_t *slice = table + (uint64_t)pencil * PC_NBB;
        for (int k = threadIdx.x; k < PC_L_BINS; k += 64) {
            huge_cursor[(uint64_t)h * PC_L_BINS + k] = base + hist[k];
            if (k < PC_NBB) slice[k] = base + hist[k];
        }
        pc_wave_lds_sync();
    }
}

// partition role: PC_L_PTPB threads; LDS: buf = a tile of points (PC_L7H_TILE * 3 floats), bin_of (PC_L7H_TILE), three
// arrays of PC_L_BINS words
__device__ __forceinline__ void pc_l7h_partition_role(uint32_t blk, uint32_t nblk, float *buf, uint16_t *bin_of, uint32_t *hist,
                                                      uint32_t *lstart, uint32_t *delta, const float *__restrict__ src,
                                                      float *__restrict__ dst, const uint32_t *__restrict__ pencil_start,
                                                      const uint32_t *__restrict__ huge_n, const uint32_t *__restrict__ huge_list,
                                                      uint32_t *__restrict__ huge_cursor, const pc_geom &geo)
{
    const int tid = threadIdx.x;
    const uint32_t entries = *huge_n;
    for (uint32_t h = 0; h < entries; h++) {
        const uint32_t pencil = huge_list[h];
        const uint32_t base = pencil_start[pencil], n = pencil_start[pencil + 1] - base;
        const uint32_t ntiles = (n + PC_L7H_TILE - 1) / PC_L7H_TILE;
        uint32_t *cursor = huge_cursor + (uint64_t)h * PC_L_BINS;
        for (uint32_t t = pc_l7h_first_tile(h, blk, nblk); t < ntiles; t += nblk) {
            const uint32_t first = t * PC_L7H_TILE, cnt = min((uint32_t)PC_L7H_TILE, n - first);
            const pc_pt3 *g = reinterpret_cast<const pc_pt3 *>(src) + base + first;
            __syncthreads();                           // the previous tile's LDS reads are done
            for (int b = tid; b < PC_L_BINS; b += PC_L_PTPB) hist[b] = 0;
            pc_pt3 pt[PC_L7H_PPPT];
#pragma unroll
            for (int j = 0; j < PC_L7H_PPPT; j++) pt[j] = g[min((uint32_t)(tid + j * PC_L_PTPB), cnt - 1)];
            __syncthreads();
            uint32_t tag[PC_L7H_PPPT];                   // cell | rank << 10, or ~0 for "skip"
#pragma unroll
            for (int j = 0; j < PC_L7H_PPPT; j++) {
                tag[j] = 0xFFFFFFFFu;
                if ((uint32_t)(tid + j * PC_L_PTPB) < cnt) {
                    const uint32_t b = (uint32_t)pc_cell_index(pt[j].z, geo);
                    if (b < (unsigned)geo.ncell) tag[j] = b | (atomicAdd(&hist[b], 1u) << 10);
                }
            }
            __syncthreads();
            if (tid < 64) {                            // exclusive scan of the tile's cell counts
                uint32_t c[7], sum = 0;
#pragma unroll
                for (int q = 0; q < 7; q++) {
                    const int b = tid * 7 + q;
                    c[q] = b < PC_L_BINS ? hist[b] : 0u;
                    sum += c[q];
                }
                uint32_t run = pc_wave_incl_scan(sum) - sum;
#pragma unroll
                for (int q = 0; q < 7; q++) {
                    const int b = tid * 7 + q;
                    if (b < PC_L_BINS) lstart[b] = run;
                    run += c[q];
                }
            }
            __syncthreads();
            for (int b = tid; b < PC_L_BINS; b += PC_L_PTPB) {      // claim global space: one atomic per non-empty cell
                const uint32_t c = hist[b];
                if (c) delta[b] = atomicAdd(&cursor[b], c) - lstart[b];
            }
#pragma unroll
            for (int j = 0; j < PC_L7H_PPPT; j++) {
                if (tag[j] != 0xFFFFFFFFu) {
                    const uint32_t b = tag[j] & 1023u, pos = lstart[b] + (tag[j] >> 10);
                    buf[3 * pos] = pt[j].x;
                    buf[3 * pos + 1] = pt[j].y;
                    buf[3 * pos + 2] = pt[j].z;
                    bin_of[pos] = (uint16_t)b;
                }
            }
            __syncthreads();
            const uint32_t kept = lstart[PC_L_BINS - 1] + hist[PC_L_BINS - 1];
            for (uint32_t d = tid; d < 3 * kept; d += PC_L_PTPB) {   // runs -> global, coalesced dwords
                const uint32_t e = d / 3u;
                const uint32_t slot = delta[bin_of[e]] + e;
                dst[3 * (uint64_t)slot + (d - 3u * e)] = buf[d];
            }
        }
    }
}

template <int CAP, bool STREAM>
__global__ void __launch_bounds__(PC_L7_WAVES * 64)
pc_l7_sort_pencils(const float *__restrict__ src, float *__restrict__ dst,
                   const uint32_t *__restrict__ pencil_start, uint32_t *__restrict__ table,
                   uint32_t first_pencil, uint32_t end_pencil, int32_t *__restrict__ status, const pc_geom geo,
                   const uint32_t *__restrict__ fgate, uint32_t sort_blocks, const uint32_t *__restrict__ huge_n,
                   const uint32_t *__restrict__ huge_list, uint32_t *__restrict__ huge_count)
{
    if (pc_fused_serves(fgate)) return;
    __shared__ uint32_t hist_all[PC_L7_WAVES][PC_L_BINS];
    if (blockIdx.x >= sort_blocks) {                   // the workgroups behind the pencils': L7H's histogram (normally nothing to do)
        pc_l7h_hist_role<PC_L7_WAVES * 64>(blockIdx.x - sort_blocks, gridDim.x - sort_blocks, hist_all[0], src, pencil_start, huge_n,
                                           huge_list, huge_count, status, geo);
        return;
    }
    constexpr int PPL = CAP / 64;                      // points per lane
    __shared__ __attribute__((aligned(16))) float stage_all[PC_L7_WAVES][CAP * 3 + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hist = hist_all[wave];
    // pencils [first_pencil, end_pencil): all of them, or the layers one slab's pair test reads
    const uint32_t pencil = (PC_L_REV & 2) ? end_pencil - 1u - (blockIdx.x * PC_L7_WAVES + wave) : first_pencil + blockIdx.x * PC_L7_WAVES + wave;
    if (pencil >= end_pencil || pencil < first_pencil) return;
    const uint32_t base = pencil_start[pencil], n = pencil_start[pencil + 1] - base;
    uint32_t *slice = table + (uint64_t)pencil * PC_NBB;
    if (pencil == PC_L_NPENCILS - 1 && lane < 2) table[(uint64_t)PC_L_NPENCILS * PC_NBB + lane] = base + n;
    if (n > CAP) return;                               // on the big-pencil list: a whole workgroup of L7B sorts it
    if (n == 0) {
        for (int k = lane; k < PC_NBB; k += 64) slice[k] = base;
        return;
    }
    for (int k = lane; k < PC_L_BINS; k += 64) hist[k] = 0;
    pc_wave_lds_sync();
    const pc_pt3 *g = reinterpret_cast<const pc_pt3 *>(src) + base;
    pc_pt3 *o = reinterpret_cast<pc_pt3 *>(dst) + base;
    bool bad = false;

    {
        const int rounds = (int)((n + 63) >> 6);       // wave-uniform
        pc_pt3 pt[PPL];
        uint32_t tag[PPL];
#pragma unroll
        for (int j = 0; j < PPL; j++)            // all loads first: one memory round trip
            if (j < rounds) pt[j] = pc_load_pt<STREAM || (PC_L_NT & 4) != 0>(g + min(lane + 64u * j, n - 1));
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            tag[j] = 0xFFFFFFFFu;
            if (j < rounds && lane + 64u * j < n) {
                const int k = pc_cell_index(pt[j].z, geo);
                if ((unsigned)k < (unsigned)geo.ncell) tag[j] = (uint32_t)k | (atomicAdd(&hist[k], 1u) << 10);
                else bad = true;
            }
        }
        pc_wave_lds_sync();
        pc_l7_scan_bins(hist, lane);                   // hist now holds the exclusive starts
        pc_wave_lds_sync();
        for (int k = lane; k < PC_NBB; k += 64) {
            if (PC_L_NT & 256) __builtin_nontemporal_store(base + hist[k], slice + k);
            else slice[k] = base + hist[k];
        }
        // The pencil in cell order goes through LDS and leaves as 16-byte stores.
        // `shift` gives the staged copy the same position modulo 16 bytes as its
        // destination, so every aligned 16-byte LDS read feeds an aligned global store.
        float *stage = stage_all[wave];
        float *od = reinterpret_cast<float *>(o);
        const uint32_t shift = (uint32_t)((reinterpret_cast<uintptr_t>(od) >> 2) & 3u);
        uint32_t kept = 0;
#pragma unroll
        for (int j = 0; j < PPL; j++) {
            if (j < rounds && tag[j] != 0xFFFFFFFFu) {
                const uint32_t pos = shift + 3u * (hist[tag[j] & 1023u] + (tag[j] >> 10));
                stage[pos] = pt[j].x; stage[pos + 1] = pt[j].y; stage[pos + 2] = pt[j].z;
                kept++;
            }
        }
        kept = pc_wave_sum(kept);
        pc_wave_lds_sync();
        const uint32_t nd = 3u * kept;
        const uint32_t head = min((4u - shift) & 3u, nd);          // dwords before the first 16-byte boundary
        const uint32_t nbody = (nd - head) >> 2;
        if ((uint32_t)lane < head) od[lane] = stage[shift + lane];
        for (uint32_t q = lane; q < nbody; q += 64) {
            if (STREAM || (PC_L_NT & 8))
                __builtin_nontemporal_store(*reinterpret_cast<const pc_v4f_a *>(stage + shift + head + 4u * q),
                                            reinterpret_cast<pc_v4f_a *>(od + head + 4u * q));
            else
                *reinterpret_cast<float4 *>(od + head + 4u * q) = *reinterpret_cast<const float4 *>(stage + shift + head + 4u * q);
        }
        const uint32_t t0 = head + 4u * nbody;
        if ((uint32_t)lane < nd - t0) od[t0 + lane] = stage[shift + t0 + lane];
    }
    if (bad) atomicOr(status, 1);
}

// ---- L7B: one workgroup sorts one big pencil ------------------------------------
// Pencils with more points than a wave of L7 keeps (clustered inputs, or more than
// about 2 points per cell everywhere) are listed by L5.  Its points go through
// registers and an LDS copy in cell order like L7's.  Three instantiations share the
// list: a small one (several workgroups per CU) takes the pencils of at most
// PC_L7B_CAP_SMALL points, a medium and a large one those of up to PC_L7B_CAP_MEDIUM and PC_L7B_CAP_LARGE; what is
// larger still is on L5's second list and sorted tile by tile (L7H above).  The
// first sort_blocks workgroups of the grid sort; the rest take a step of L7H.
#define PC_L7B_CAP_SMALL 2048
#define PC_L7B_TPB_SMALL 256
#define PC_L7B_CAP_MEDIUM 4096                     // (48 KB of LDS: three workgroups per CU where the large one gets one)
#define PC_L7B_TPB_MEDIUM 512
#define PC_L7B_CAP_LARGE 8192
#define PC_L7B_TPB_LARGE 512
#define PC_L7B_ROLE_NONE 0                         // what the workgroups behind the sorting ones do (a step of L7H)
#define PC_L7B_ROLE_SCAN 1
#define PC_L7B_ROLE_PARTITION 2

template <int PC_L7B_CAP, int PC_L7B_TPB, int PC_L7B_ROLE>
__global__ void __launch_bounds__(PC_L7B_TPB)
pc_l7b_sort_big_pencils(const float *__restrict__ src, float *__restrict__ dst,
                        const uint32_t *__restrict__ pencil_start, uint32_t *__restrict__ table,
                        const uint32_t *__restrict__ big_count, const uint32_t *__restrict__ big_list,
                        uint32_t first_pencil, uint32_t end_pencil, uint32_t n_above, uint32_t n_upto,
                        uint32_t sort_blocks, const uint32_t *__restrict__ huge_n, const uint32_t *__restrict__ huge_list,
                        const uint32_t *__restrict__ huge_count, uint32_t *__restrict__ huge_cursor,
                        int32_t *__restrict__ status, const pc_geom geo, const uint32_t *__restrict__ fgate)
{
    if (pc_fused_serves(fgate)) return;
    constexpr int PC_L7B_PPT = PC_L7B_CAP / PC_L7B_TPB;
    __shared__ uint32_t hist[PC_L_BINS];
    __shared__ float stage[PC_L7B_CAP * 3];
    if (blockIdx.x >= sort_blocks) {                   // the workgroups behind the sorting ones: a step of L7H (normally nothing to do)
        const uint32_t blk = blockIdx.x - sort_blocks, nblk = gridDim.x - sort_blocks;
        if constexpr (PC_L7B_ROLE == PC_L7B_ROLE_SCAN)
            pc_l7h_scan_role(blk, nblk, hist, pencil_start, huge_n, huge_list, huge_count, huge_cursor, table);
        else if constexpr (PC_L7B_ROLE == PC_L7B_ROLE_PARTITION) {
            static_assert(PC_L7B_ROLE != PC_L7B_ROLE_PARTITION || PC_L7B_TPB == PC_L_PTPB, "the partition role runs with PC_L_PTPB threads");
            static_assert(PC_L7B_ROLE != PC_L7B_ROLE_PARTITION ||
                          PC_L7B_CAP * 3 >= PC_L7H_TILE * 3 + PC_L7H_TILE / 2 + 2 * PC_L_BINS, "the partition role's arrays lie inside `stage`");
            uint32_t *words = reinterpret_cast<uint32_t *>(stage + PC_L7H_TILE * 3 + PC_L7H_TILE / 2);
            pc_l7h_partition_role(blk, nblk, stage, reinterpret_cast<uint16_t *>(stage + PC_L7H_TILE * 3), hist, words, words + PC_L_BINS,
                                  src, dst, pencil_start, huge_n, huge_list, huge_cursor, geo);
        }
        return;
    }
    const int tid = threadIdx.x;
    const uint32_t nbig = *big_count;
    bool bad = false;
    for (uint32_t b = blockIdx.x; b < nbig; b += sort_blocks) {
        const uint32_t pencil = big_list[b];
        if (pencil < first_pencil || pencil >= end_pencil) continue;          // workgroup-uniform
        const uint32_t base = pencil_start[pencil], n = pencil_start[pencil + 1] - base;
        if (n <= n_above) continue;                                           // the other instantiation's
        if (n > n_upto) continue;                                             // the other instantiation's, or L7H's (listed by L5)
        uint32_t *slice = table + (uint64_t)pencil * PC_NBB;
        const pc_pt3 *g = reinterpret_cast<const pc_pt3 *>(src) + base;
        __syncthreads();                                                     // previous pencil's LDS reads are done
        for (int k = tid; k < PC_L_BINS; k += PC_L7B_TPB) hist[k] = 0;
        __syncthreads();
        {   // (n <= PC_L7B_CAP: larger pencils were handed over above)
            pc_pt3 pt[PC_L7B_PPT];
            uint32_t tag[PC_L7B_PPT];
#pragma unroll
            for (int q = 0; q < PC_L7B_PPT; q++) pt[q] = g[min((uint32_t)(tid + q * PC_L7B_TPB), n - 1)];
#pragma unroll
            for (int q = 0; q < PC_L7B_PPT; q++) {
                tag[q] = 0xFFFFFFFFu;
                if ((uint32_t)(tid + q * PC_L7B_TPB) < n) {
                    const int k = pc_cell_index(pt[q].z, geo);
                    if ((unsigned)k < (unsigned)geo.ncell) tag[q] = (uint32_t)k | (atomicAdd(&hist[k], 1u) << 10);
                    else bad = true;
                }
            }
            __syncthreads();
            if (tid < 64) pc_l7_scan_bins(hist, tid);                        // exclusive starts, in place
            __syncthreads();
            for (int k = tid; k < PC_NBB; k += PC_L7B_TPB) slice[k] = base + hist[k];
#pragma unroll
            for (int q = 0; q < PC_L7B_PPT; q++) {
                if (tag[q] != 0xFFFFFFFFu) {
                    const uint32_t pos = 3u * (hist[tag[q] & 1023u] + (tag[q] >> 10));
                    stage[pos] = pt[q].x; stage[pos + 1] = pt[q].y; stage[pos + 2] = pt[q].z;
                }
            }
            __syncthreads();
            const uint32_t nd = 3u * hist[PC_L_BINS - 1];                     // bins 400..403 are empty: start of bin 403 = points kept
            float *od = dst + 3 * (uint64_t)base;
            for (uint32_t d = tid; d < nd; d += PC_L7B_TPB) od[d] = stage[d];
        }
    }
    if (bad) atomicOr(status, 1);
}
