// Pair test, variant 1 ("tile"): LDS-staged half-stencil over pencil strips.
//
// Replaces counter_thread/compare/count_chunk/dist2 (reference main.c:85-91,
// 161-234).  Keys are x-major (main.c:32), so a pencil -- all cells (i,j,*) --
// is one contiguous run of the sorted array and the 13 forward neighbour cells
// of (i,j,k) (main.c:195-209) are cells k-1..k+1 of four neighbour pencils
// (i,j+1) (i+1,j-1) (i+1,j) (i+1,j+1) plus cell k+1 of the home pencil.
//
// One workgroup owns a strip of S consecutive home pencils (i, j0..j0+S-1).
// It stages in LDS the table slices of the 2S+3 pencils involved and then walks
// the strip in z-chunks: each chunk's candidate particles (home cells' z range
// widened by one cell, all 2S+3 pencils) are copied to LDS as SoA with
// coalesced dword loads, and every thread takes home particles and tests them
// against five contiguous LDS ranges.  A chunk is as long as fits the LDS
// candidate buffer; a single home cell whose candidates do not fit (very dense
// data) is processed straight from global memory, where its lanes share
// candidates and the loads broadcast.
//
// Workgroup -> tile mapping is XCD-aware: workgroups with equal blockIdx%8 share
// an XCD (round-robin dispatch) and are given one contiguous band of strips in
// y, walked layer by layer, so neighbour pencils are re-read from that XCD's L2.
// The mapping affects speed only; every tile is processed exactly once.
//
// Per-wave hit counts come from the compare mask (ballot + popcount on the
// scalar unit); one 64-bit atomic per workgroup.
#pragma once
#include "pc_device.h"

#define PC_T_S 4                       // home pencils per strip
#define PC_T_ROWS (2 * PC_T_S + 3)     // pencils whose particles a strip touches
#define PC_T_NS (PC_NLAYERS / PC_T_S)  // strips per x layer (400 cells in y)
#define PC_T_SLICE 404                 // table entries per pencil: cells 0..402 + end
#define PC_T_CAP 2891                  // candidate points per chunk; CAP % 32 == 11 keeps the
                                       // x/y/z planes on disjoint LDS banks during the SoA fill
#define PC_T_TPB 256

static_assert(PC_NLAYERS % PC_T_S == 0, "strips must tile the 400 occupied y cells");
static_assert(PC_T_CAP % 32 == 11, "bank-offset rule for the SoA fill");

struct pc_tile_lds {
    uint32_t slice[PC_T_ROWS][PC_T_SLICE];   // global sorted slots: start of cell k in pencil r
    float x[PC_T_CAP], y[PC_T_CAP], z[PC_T_CAP];
    uint32_t row_lo[PC_T_ROWS];              // first global slot of the row's chunk window
    uint32_t row_base[PC_T_ROWS + 1];        // LDS index of that slot (prefix over rows)
    uint32_t home_base[PC_T_S + 1];          // prefix of home particle counts in the chunk
    int z1;                                  // chunk end chosen by the planner
};

// count hits of (x,y,z) against LDS candidates [lo,hi)
__device__ __forceinline__ uint32_t pc_tile_scan(const pc_tile_lds &L, float x, float y, float z,
                                                 uint32_t lo, uint32_t hi, float d2max)
{
    uint32_t c = 0;
    for (uint32_t q = lo; q < hi; q++) c += pc_hit(x, y, z, L.x[q], L.y[q], L.z[q], d2max);
    return c;
}

__device__ __forceinline__ uint32_t pc_global_scan(const float *__restrict__ pts, float x, float y, float z,
                                                   uint32_t lo, uint32_t hi, float d2max)
{
    uint32_t c = 0;
    for (uint32_t q = lo; q < hi; q++)
        c += pc_hit(x, y, z, pts[3 * (uint64_t)q], pts[3 * (uint64_t)q + 1], pts[3 * (uint64_t)q + 2], d2max);
    return c;
}

__global__ void __launch_bounds__(PC_T_TPB)
pc_c1_count_tiles(const float *__restrict__ pts, const uint32_t *__restrict__ T, int x_lo, int x_hi,
                  unsigned long long *__restrict__ out, const pc_geom geo)
{
    __shared__ pc_tile_lds L;
    const int tid = threadIdx.x;

    // ---- tile of this workgroup (XCD-aware) --------------------------------
    const int g = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int s0 = g * PC_T_NS / 8, s1 = (g + 1) * PC_T_NS / 8, nsg = s1 - s0;
    const int layer = x_lo + q / nsg;
    const int strip = s0 + q % nsg;
    unsigned long long acc = 0;
    if (layer < x_hi) {
        const int j0 = strip * PC_T_S;
        const uint32_t p0 = (uint32_t)layer * PC_NBB + j0;                 // first home pencil
        const uint32_t home_total = T[(p0 + PC_T_S) * PC_NBB] - T[p0 * PC_NBB];
        if (home_total != 0) {
            // ---- table slices of the 2S+3 rows -> LDS ----------------------
            // rows 0..S: layer, y=j0+r ; rows S+1..2S+2: layer+1, y=j0-1+(r-S-1)
            for (int e = tid; e < PC_T_ROWS * PC_T_SLICE; e += PC_T_TPB) {
                int r = e / PC_T_SLICE, k = e - r * PC_T_SLICE;
                int rx = r <= PC_T_S ? layer : layer + 1;
                int ry = r <= PC_T_S ? j0 + r : j0 - 1 + (r - PC_T_S - 1);
                uint32_t v = 0;                                            // invalid row: empty everywhere
                if (rx < PC_NLAYERS && ry >= 0 && ry < PC_NLAYERS)
                    v = T[((uint32_t)rx * PC_NBB + ry) * PC_NBB + k];
                L.slice[r][k] = v;
            }
            __syncthreads();

            int z0 = 0;
            while (z0 < PC_NLAYERS) {
                // ---- planner: longest chunk [z0,z1) whose candidates fit --
                if (tid == 0) L.z1 = z0;                                   // z0 = "nothing fits"
                __syncthreads();
                const int zl = z0 > 0 ? z0 - 1 : 0;
                for (int c = z0 + 1 + tid; c <= PC_NLAYERS; c += PC_T_TPB) {
                    uint32_t tot = 0;
#pragma unroll
                    for (int r = 0; r < PC_T_ROWS; r++) tot += L.slice[r][c + 1] - L.slice[r][zl];
                    if (tot <= PC_T_CAP) atomicMax(&L.z1, c);
                }
                __syncthreads();
                const int z1 = L.z1;

                if (z1 == z0) {
                    // ---- one home cell too dense for LDS: global path ------
                    uint32_t hb[PC_T_S + 1];
                    hb[0] = 0;
#pragma unroll
                    for (int h = 0; h < PC_T_S; h++) hb[h + 1] = hb[h] + (L.slice[h][z0 + 1] - L.slice[h][z0]);
                    for (uint32_t gidx = tid; gidx < hb[PC_T_S]; gidx += PC_T_TPB) {
                        int h = 0;
#pragma unroll
                        for (int t = 1; t < PC_T_S; t++) h += gidx >= hb[t];
                        uint32_t i = L.slice[h][z0] + (gidx - hb[h]);
                        float x = pts[3 * (uint64_t)i], y = pts[3 * (uint64_t)i + 1], z = pts[3 * (uint64_t)i + 2];
                        uint32_t c = pc_global_scan(pts, x, y, z, i + 1, L.slice[h][z0 + 2], geo.d2max);
                        const int rows[4] = {h + 1, PC_T_S + 1 + h, PC_T_S + 2 + h, PC_T_S + 3 + h};
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            c += pc_global_scan(pts, x, y, z, L.slice[rows[t]][zl], L.slice[rows[t]][z0 + 2], geo.d2max);
                        acc += c;
                    }
                    z0 += 1;
                    __syncthreads();
                    continue;
                }

                // ---- chunk bookkeeping ------------------------------------
                if (tid == 0) {
                    uint32_t b = 0;
                    for (int r = 0; r < PC_T_ROWS; r++) {
                        L.row_lo[r] = L.slice[r][zl];
                        L.row_base[r] = b;
                        b += L.slice[r][z1 + 1] - L.slice[r][zl];
                    }
                    L.row_base[PC_T_ROWS] = b;
                    uint32_t hbase = 0;
                    for (int h = 0; h < PC_T_S; h++) {
                        L.home_base[h] = hbase;
                        hbase += L.slice[h][z1] - L.slice[h][z0];
                    }
                    L.home_base[PC_T_S] = hbase;
                }
                __syncthreads();
                const uint32_t nhome = L.home_base[PC_T_S];
                if (nhome != 0) {
                    // ---- SoA fill: each row's window is one contiguous dword run
                    for (int r = 0; r < PC_T_ROWS; r++) {
                        const uint32_t lo = L.row_lo[r], base = L.row_base[r];
                        const uint32_t nd = 3u * (L.row_base[r + 1] - base);
                        const float *src = pts + 3 * (uint64_t)lo;
                        for (uint32_t d = tid; d < nd; d += PC_T_TPB) {
                            uint32_t pidx = d / 3u, comp = d - 3u * pidx;
                            float v = src[d];
                            float *dst = comp == 0 ? L.x : (comp == 1 ? L.y : L.z);
                            dst[base + pidx] = v;
                        }
                    }
                    __syncthreads();

                    // ---- pair tests ---------------------------------------
                    for (uint32_t gidx = tid; gidx < nhome; gidx += PC_T_TPB) {
                        int h = 0;
#pragma unroll
                        for (int t = 1; t < PC_T_S; t++) h += gidx >= L.home_base[t];
                        const uint32_t i = L.slice[h][z0] + (gidx - L.home_base[h]);     // global slot
                        const uint32_t li = L.row_base[h] + (i - L.row_lo[h]);           // LDS index
                        const float x = L.x[li], y = L.y[li], z = L.z[li];
                        const int k = pc_cell_index(z, geo);
                        const int kl = k > 0 ? k - 1 : 0;
                        // same cell, later slots + cell k+1 of the home pencil (main.c:215-221, (0,0,1))
                        uint32_t c = pc_tile_scan(L, x, y, z, li + 1,
                                                  L.row_base[h] + (L.slice[h][k + 2] - L.row_lo[h]), geo.d2max);
                        const int rows[4] = {h + 1, PC_T_S + 1 + h, PC_T_S + 2 + h, PC_T_S + 3 + h};
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const int r = rows[t];
                            const uint32_t off = L.row_base[r] - L.row_lo[r];
                            c += pc_tile_scan(L, x, y, z, off + L.slice[r][kl], off + L.slice[r][k + 2], geo.d2max);
                        }
                        acc += c;
                    }
                }
                z0 = z1;
                __syncthreads();
            }
        }
    }
    pc_block_add_u64<PC_T_TPB / 64>(acc, out);
}
