// Pair test, variant 2 ("wave"): one wavefront owns one home pencil; no
// workgroup barriers anywhere in the main loop.
//
// Replaces counter_thread/compare/count_chunk/dist2 (reference main.c:85-91,
// 161-234).  For home pencil (i,j) the 13 forward neighbour cells of (i,j,k)
// (main.c:195-209) are cells k-1..k+1 of the four pencils (i,j+1) (i+1,j-1)
// (i+1,j) (i+1,j+1) plus cell k+1 of the home pencil, and each pencil is one
// contiguous run of the key-sorted array.
//
// A wave keeps in its private LDS region the table slices of its five pencils
// and walks the home pencil in z-chunks.  For a chunk it copies the candidate
// particles (home cells' z range widened by one cell, five pencils) into LDS as
// 16-byte records (one 12-byte global load and one 16-byte LDS store per point)
// and then each lane takes home particles and tests them against five
// contiguous LDS ranges, one ds_read_b128 per candidate.  Waves never wait for
// each other, so the memory latency of one wave's fill is covered by the pair
// tests of the other waves on the SIMD.  A home cell whose candidates exceed
// the buffer is processed against the candidates in buffer-sized pieces.
//
// Workgroup -> pencil mapping is XCD-aware: workgroups with equal blockIdx%8
// share an XCD (round-robin dispatch) and are given one band of 50 pencils in
// y, walked layer by layer, so neighbour pencils are re-read from that XCD's
// L2.  Speed only: every home pencil is processed exactly once.
#pragma once
#include "pc_device.h"

#define PC_W_WAVES 4                    // waves (home pencils) per workgroup
#define PC_W_CAP 768                    // candidate points per wave buffer
#define PC_W_SLICE 402                  // table entries kept per pencil: cells 0..401
#define PC_W_BAND (PC_NLAYERS / 8)      // 50 pencils of a layer per XCD group
#define PC_W_WG_PER_BAND ((PC_W_BAND + PC_W_WAVES - 1) / PC_W_WAVES)   // 13

struct pc_wave_lds {
    uint32_t slice[5][PC_W_SLICE];      // global sorted slot where cell k of row r starts
    float4 cand[PC_W_CAP];
};

// hits of (x,y,z) against LDS records [lo,hi)
__device__ __forceinline__ uint32_t pc_w_scan(const float4 *__restrict__ cand, float x, float y, float z,
                                              uint32_t lo, uint32_t hi, float d2max)
{
    // four candidates per trip so that four LDS reads are in flight; slots past
    // hi are read (clamped into the buffer) and masked out
    uint32_t c = 0;
    for (uint32_t q = lo; q < hi; q += 4) {
        const float4 v0 = cand[q];
        const float4 v1 = cand[min(q + 1, (uint32_t)PC_W_CAP - 1)];
        const float4 v2 = cand[min(q + 2, (uint32_t)PC_W_CAP - 1)];
        const float4 v3 = cand[min(q + 3, (uint32_t)PC_W_CAP - 1)];
        c += pc_hit(x, y, z, v0.x, v0.y, v0.z, d2max);
        c += (q + 1 < hi) & pc_hit(x, y, z, v1.x, v1.y, v1.z, d2max);
        c += (q + 2 < hi) & pc_hit(x, y, z, v2.x, v2.y, v2.z, d2max);
        c += (q + 3 < hi) & pc_hit(x, y, z, v3.x, v3.y, v3.z, d2max);
    }
    return c;
}

// copy sorted points [lo, lo+cnt) to LDS records [dst, dst+cnt)
__device__ __forceinline__ void pc_w_fill(float4 *__restrict__ cand, uint32_t dst,
                                          const float *__restrict__ pts, uint32_t lo, uint32_t cnt, int lane)
{
    for (uint32_t e = lane; e < cnt; e += 64) {
        const float *p = pts + 3 * (uint64_t)(lo + e);
        cand[dst + e] = make_float4(p[0], p[1], p[2], 0.0f);
    }
}

__global__ void __launch_bounds__(PC_W_WAVES * 64)
pc_c2_count_waves(const float *__restrict__ pts, const uint32_t *__restrict__ T, int x_lo, int x_hi,
                  unsigned long long *__restrict__ out, const pc_geom geo)
{
    __shared__ pc_wave_lds lds_all[PC_W_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pc_wave_lds &L = lds_all[wave];
    unsigned long long acc = 0;

    // ---- home pencil of this wave (XCD-aware) ------------------------------
    const int g = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int layer = x_lo + q / PC_W_WG_PER_BAND;
    const int jb = (q % PC_W_WG_PER_BAND) * PC_W_WAVES + wave;      // index inside the band
    const int j = g * PC_W_BAND + jb;
    if (layer < x_hi && jb < PC_W_BAND) {
        const uint32_t tb0 = ((uint32_t)layer * PC_NBB + j) * PC_NBB;
        const uint32_t home_lo = T[tb0], home_hi = T[tb0 + PC_NBB];
        if (home_hi != home_lo) {
            // rows: 0 home (i,j); 1 (i,j+1); 2..4 (i+1, j-1..j+1)
            const int ri[5] = {layer, layer, layer + 1, layer + 1, layer + 1};
            const int rj[5] = {j, j + 1, j - 1, j, j + 1};
#pragma unroll
            for (int r = 0; r < 5; r++) {
                const bool valid = ri[r] < PC_NLAYERS && rj[r] >= 0 && rj[r] < PC_NLAYERS;
                const uint32_t tb = ((uint32_t)ri[r] * PC_NBB + rj[r]) * PC_NBB;
                for (int k = lane; k < PC_W_SLICE; k += 64) L.slice[r][k] = valid ? T[tb + k] : 0u;
            }
            pc_wave_lds_sync();

            int z0 = 0;
            // skip the empty head of the home pencil
            {
                int first = PC_NLAYERS;
                for (int k = lane; k < PC_NLAYERS; k += 64)
                    if (L.slice[0][k + 1] != home_lo) { first = k; break; }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
                z0 = first;
            }
            while (z0 < PC_NLAYERS && L.slice[0][z0] != home_hi) {
                const int zl = z0 > 0 ? z0 - 1 : 0;
                // ---- planner: lanes probe chunk ends z0+1 .. ; longest that fits
                uint32_t lo_r[5];
#pragma unroll
                for (int r = 0; r < 5; r++) lo_r[r] = L.slice[r][r == 0 ? z0 : zl];   // home row: no cell below z0 needed
                int z1 = z0;
                for (int probe0 = z0; probe0 < PC_NLAYERS; probe0 += 64) {
                    const int c = min(probe0 + 1 + lane, PC_NLAYERS);
                    uint32_t tot = 0;
#pragma unroll
                    for (int r = 0; r < 5; r++) tot += L.slice[r][c + 1] - lo_r[r];
                    const unsigned long long fit = __ballot(tot <= PC_W_CAP);
                    const int nfit = __popcll(fit);            // feasible ends form a prefix of the lanes
                    if (nfit) z1 = min(probe0 + nfit, PC_NLAYERS);
                    if (nfit < 64) break;
                }

                if (z1 > z0) {
                    // ---- candidates of the chunk -> LDS -------------------
                    uint32_t base_r[5], b = 0;
#pragma unroll
                    for (int r = 0; r < 5; r++) {
                        base_r[r] = b;
                        const uint32_t cnt = L.slice[r][z1 + 1] - lo_r[r];
                        pc_w_fill(L.cand, b, pts, lo_r[r], cnt, lane);
                        b += cnt;
                    }
                    pc_wave_lds_sync();
                    // ---- pair tests: lanes take the chunk's home particles
                    const uint32_t h0 = L.slice[0][z0], h1 = L.slice[0][z1];
                    for (uint32_t i = h0 + lane; i < h1; i += 64) {
                        const uint32_t li = i - lo_r[0];
                        const float4 me = L.cand[li];
                        const int k = pc_cell_index(me.z, geo);
                        const int kl = k > 0 ? k - 1 : 0;
                        uint32_t c = pc_w_scan(L.cand, me.x, me.y, me.z, li + 1, L.slice[0][k + 2] - lo_r[0], geo.d2max);
#pragma unroll
                        for (int r = 1; r < 5; r++) {
                            const uint32_t off = base_r[r] - lo_r[r];
                            c += pc_w_scan(L.cand, me.x, me.y, me.z, off + L.slice[r][kl], off + L.slice[r][k + 2], geo.d2max);
                        }
                        acc += c;
                    }
                    pc_wave_lds_sync();
                    z0 = z1;
                } else {
                    // ---- home cell z0 alone exceeds the buffer: stream its
                    //      candidates through LDS in pieces --------------------
                    const uint32_t h0 = L.slice[0][z0], h1 = L.slice[0][z0 + 1];
#pragma unroll
                    for (int r = 0; r < 5; r++) {
                        // row 0: the cell itself (later slots only) and cell z0+1; others: z0-1..z0+1
                        const uint32_t c_lo = r == 0 ? h0 : lo_r[r];
                        const uint32_t c_hi = L.slice[r][z0 + 2];
                        for (uint32_t p0 = c_lo; p0 < c_hi; p0 += PC_W_CAP) {
                            const uint32_t cnt = min((uint32_t)PC_W_CAP, c_hi - p0);
                            pc_w_fill(L.cand, 0, pts, p0, cnt, lane);
                            pc_wave_lds_sync();
                            for (uint32_t i = h0 + lane; i < h1; i += 64) {
                                const float *p = pts + 3 * (uint64_t)i;
                                const float x = p[0], y = p[1], z = p[2];
                                uint32_t s = 0;
                                if (r == 0 && i + 1 > p0) s = min(i + 1 - p0, cnt);   // only slots after i
                                acc += pc_w_scan(L.cand, x, y, z, s, cnt, geo.d2max);
                            }
                            pc_wave_lds_sync();
                        }
                    }
                    z0 += 1;
                }
            }
        }
    }
    pc_block_add_u64<PC_W_WAVES>(acc, out);
}
