// Pair test, variant 6 ("dense"): the form for well-filled cells (from about eight
// points per cell), where the distance arithmetic itself decides the time.
//
// Replaces counter_thread/compare/count_chunk/dist2 (reference main.c:85-91,
// 161-234) for the segments of pencils that the planning kernel has cut (pencils of
// 2048 points and more) and for the pencil rests that the merged-window kernel hands
// over.  One wavefront takes one segment [z_begin, z_end) of a home pencil (i,j) and
// walks it cell by cell.  For home cell k everything its particles meet -- the cell
// itself and cell k+1 of the pencil, cells k-1..k+1 of the pencils (i,j+1) (i+1,j-1)
// (i+1,j) (i+1,j+1): the offsets of main.c:195-209 -- is five contiguous ranges of the
// sorted array.  Their concatenation is the cell's CANDIDATE LIST (own cell first), and
// it is held in registers, lane = candidate, 64 per chunk, up to PC_D_MAXCH chunks at
// a time.  The home particles are then taken one after the other: the coordinates of
// home particle i reach the scalar registers through three lane reads, and every chunk
// costs nine vector instructions for 64 tests --
//
//     dx = hx - cx   dy = hy - cy   dz = hz - cz      (scalar operand: the home particle)
//     (dx*dx + dy*dy) + dz*dz <= d2max                 (each step rounded: no contraction)
//
// -- with the hit count taken from the compare mask on the scalar unit.  Home particle
// i pairs with the candidates behind it in the list only (main.c:215-221: later slots
// of the same cell), which is a shift of the compare mask of the chunks that hold
// own-cell candidates.  Unused lanes hold NaN and never match.  No LDS, no gather
// list, no workgroup barriers; the table entries are scalar loads.
//
// A cell whose home particles x candidates reach PC_G_HEAVY_COST is not processed
// here: its home particles go on the list of pc_c3_heavy, as in variant 3.
// Work is drawn from the segment list of pc_count_group.h (its own draw counters, one per XCD group) -- if most of the listed points lie in
// well-filled segments; else the group kernel takes the list (pc_segments_go_to): below about 11 points per cell
// the 64-candidate chunks are mostly empty lanes and that kernel is faster.
#pragma once
#include "pc_count_group.h"

#ifndef PC_D_MAXCH
#define PC_D_MAXCH 4                    // candidate chunks (of 64) held in registers at a time
#endif
#ifndef PC_DENSE_MIN_DEFAULT
#define PC_DENSE_MIN_DEFAULT 11         // points per cell from which a segment counts as this kernel's kind
#endif
#ifndef PC_D_WAVES
#define PC_D_WAVES 8                    // per SIMD (64 registers)
#endif

struct pc_d_chunk { float x, y, z; };

// the value lane `l` (wave-uniform) holds, as a scalar
__device__ __forceinline__ float pc_d_lane(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// chunk of the candidate list starting at list index cb: lane t holds candidate cb + t, or NaN behind the list's end.
// start[r] - first[r] turns a list index inside range r into a slot of the sorted array.
__device__ __forceinline__ pc_d_chunk pc_d_load(const pc_pt3 *__restrict__ pts, uint32_t cb, int lane, const uint32_t first[6],
                                                const uint32_t start[5])
{
    const uint32_t i = cb + (uint32_t)lane;
    uint32_t shift = start[0];                                   // (first[0] = 0)
#pragma unroll
    for (int r = 1; r < 5; r++) shift = i >= first[r] ? start[r] - first[r] : shift;
    const bool live = i < first[5];
    const pc_pt3 p = pts[live ? i + shift : start[0]];           // (a cell that is processed has a home particle: start[0] is a valid slot)
    pc_d_chunk c;
    c.x = live ? p.x : __builtin_nanf("");
    c.y = p.y;
    c.z = p.z;
    return c;
}

// home particles [i0, i1) of the cell (list indices; their coordinates are lanes i - hb of `h`) against NCH chunks
// of candidates whose first list index is cb.  MASK: 0 = no chunk holds own-cell candidates, 1 = only chunk 0 does
// and cb = hb = 0 (the usual case: at most 64 points in the cell), 2 = any chunk may (general rule).
template <int NCH, int MASK>
__device__ __forceinline__ uint32_t pc_d_homes(const pc_d_chunk &h, uint32_t hb, uint32_t i0, uint32_t i1, const pc_d_chunk *c,
                                               uint32_t cb, uint32_t n_own, float d2max)
{
    uint32_t hits = 0;
    for (uint32_t i = i0; i < i1; i++) {
        const float hx = pc_d_lane(h.x, (int)(i - hb)), hy = pc_d_lane(h.y, (int)(i - hb)), hz = pc_d_lane(h.z, (int)(i - hb));
#pragma unroll
        for (int q = 0; q < NCH; q++) {
            const float dx = hx - c[q].x, dy = hy - c[q].y, dz = hz - c[q].z;
            const float sq = (dx * dx + dy * dy) + dz * dz;
            unsigned long long b = __builtin_amdgcn_ballot_w64(sq <= d2max);
            if (MASK == 1 && q == 0) b &= ~1ull << i;            // list indices above i (i < 64)
            if (MASK == 2) {
                const uint32_t first = cb + 64u * q;             // list index of lane 0
                if (first < n_own) {                             // (uniform) own-cell candidates in this chunk
                    const uint32_t drop = i + 1 > first ? i + 1 - first : 0u;       // lanes below `drop` are at or before i
                    b = drop >= 64 ? 0ull : b & (~0ull << drop);
                }
            }
            hits += (uint32_t)__popcll(b);
        }
    }
    return hits;
}

template <int MASK>
__device__ __forceinline__ uint32_t pc_d_homes_n(int nch, const pc_d_chunk &h, uint32_t hb, uint32_t i0, uint32_t i1,
                                                 const pc_d_chunk *c, uint32_t cb, uint32_t n_own, float d2max)
{
    // (uniform branches; chunk counts above PC_D_MAXCH do not occur)
    if (nch >= PC_D_MAXCH) return pc_d_homes<PC_D_MAXCH, MASK>(h, hb, i0, i1, c, cb, n_own, d2max);
    if constexpr (PC_D_MAXCH > 5) if (nch == 5) return pc_d_homes<5, MASK>(h, hb, i0, i1, c, cb, n_own, d2max);
    if constexpr (PC_D_MAXCH > 4) if (nch == 4) return pc_d_homes<4, MASK>(h, hb, i0, i1, c, cb, n_own, d2max);
    if constexpr (PC_D_MAXCH > 3) if (nch == 3) return pc_d_homes<3, MASK>(h, hb, i0, i1, c, cb, n_own, d2max);
    if constexpr (PC_D_MAXCH > 2) if (nch == 2) return pc_d_homes<2, MASK>(h, hb, i0, i1, c, cb, n_own, d2max);
    return pc_d_homes<1, MASK>(h, hb, i0, i1, c, cb, n_own, d2max);
}

static_assert(PC_D_MAXCH >= 2 && PC_D_MAXCH <= 6, "pc_d_homes_n lists the chunk counts 1..6");

__global__ void __launch_bounds__(64, PC_D_WAVES)
pc_c6_count_dense(const float *__restrict__ pts_f, const uint32_t *__restrict__ T, int x_lo, int x_hi,
                  unsigned int *__restrict__ queue, uint32_t seg_cap, uint32_t heavy_cap,
                  int gate, uint32_t gate_limit, unsigned long long *__restrict__ out, const float d2max,
                  uint32_t dense_min)   // see pc_segments_go_to
{
    if (!pc_gate_open(queue, gate, gate_limit)) return;
    if (!pc_segments_go_to(queue, PC_SEGS_TO_DENSE, dense_min)) return;
    const pc_pt3 *__restrict__ pts = reinterpret_cast<const pc_pt3 *>(pts_f);
    const int lane = threadIdx.x;
    const int n_segments = (int)min(__builtin_amdgcn_readfirstlane(queue[PC_G_SEG_COUNT]), seg_cap);
    if (n_segments == 0) return;                               // (nothing was added to the sum: no atomic either)
    unsigned long long acc = 0;                                // lane 0 carries the wave-uniform hit counts
    // The list is cut into eight contiguous parts, one per XCD group (workgroups with equal blockIdx % 8 share an XCD
    // and its L2): neighbouring segments -- neighbouring pencils, whose points are each other's candidates -- are
    // then worked on under one L2.  Each part has its own draw counter (the word behind the group's band counter,
    // zeroed with it); a wave whose part is used up moves on to the next group's.
    const int per_group = (n_segments + 7) >> 3;
    int g = blockIdx.x & 7, groups_done = 0, m_local = 0;
    if (lane == 0) m_local = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE + 1, 1u);
    m_local = __builtin_amdgcn_readfirstlane(m_local);
    for (;;) {
        while (m_local >= min(per_group, n_segments - g * per_group) && ++groups_done < 8) {
            g = (g + 1) & 7;
            if (lane == 0) m_local = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE + 1, 1u);
            m_local = __builtin_amdgcn_readfirstlane(m_local);
        }
        if (m_local >= min(per_group, n_segments - g * per_group)) break;
        const int m = g * per_group + m_local;
        int m_next = 0;
        if (lane == 0) m_next = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE + 1, 1u);   // (its latency hides behind the segment)
        const uint32_t who = __builtin_amdgcn_readfirstlane(queue[PC_G_SEGS + 2 * m]);
        const uint32_t span = __builtin_amdgcn_readfirstlane(queue[PC_G_SEGS + 2 * m + 1]);
        if (who != 0) {                                        // (0: slot left empty by a pencil that did not fit the list)
            const int layer = (int)((who - 1) / PC_NBB), j = (int)((who - 1) % PC_NBB);
            const int z_begin = (int)(span & 0xFFFFu), z_end = (int)(span >> 16);
            // table rows of the five pencils; a pencil outside the occupied grid has no points
            const int ri[5] = {layer, layer, layer + 1, layer + 1, layer + 1};
            const int rj[5] = {j, j + 1, j - 1, j, j + 1};
            const uint32_t *row[5];
            bool valid[5];
#pragma unroll
            for (int r = 0; r < 5; r++) {
                valid[r] = ri[r] < PC_NLAYERS && rj[r] >= 0 && rj[r] < PC_NLAYERS;
                row[r] = T + (valid[r] ? ((uint32_t)ri[r] * PC_NBB + rj[r]) * PC_NBB : 0u);
            }
            for (int k = z_begin; k < z_end; k++) {
                // the cell's five ranges (scalar loads).  Row 0: cells k and k+1; rows 1-4: cells k-1 .. k+1, where
                // cell -1 is the last, always empty, padding cell of the pencil before (its start = the pencil's start).
                const uint32_t own0 = row[0][k], own1 = row[0][k + 1];
                const uint32_t n_own = own1 - own0;
                if (n_own == 0) continue;
                uint32_t start[5], first[6];
                start[0] = own0;
                first[0] = 0;
                first[1] = row[0][k + 2] - own0;
#pragma unroll
                for (int r = 1; r < 5; r++) {
                    uint32_t len = 0;
                    start[r] = own0;
                    if (valid[r]) {
                        start[r] = row[r][k - 1];
                        len = row[r][k + 2] - start[r];
                    }
                    first[r + 1] = first[r] + len;
                }
                const uint32_t n_cand = first[5];
                // too much for one wave: list the cell's home particles for pc_c3_heavy instead
                if ((unsigned long long)n_own * n_cand >= PC_G_HEAVY_COST) {
                    const uint32_t chunks = (n_own + PC_G_HEAVY_CHUNK - 1) / PC_G_HEAVY_CHUNK;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(queue + PC_G_HEAVY_COUNT, chunks);
                    base = __builtin_amdgcn_readfirstlane(base);
                    const bool room = base + chunks <= heavy_cap && base + chunks >= base;
                    const uint32_t key = (uint32_t)(row[0] - T) + (uint32_t)k;
                    // (an append that does not fit voids the slots it drew; the cell is then processed here)
                    for (uint32_t c = lane; c < chunks && base + c < heavy_cap; c += 64) {
                        queue[PC_G_HEAVY + 2 * (base + c)] = room ? key : 0xFFFFFFFFu;
                        queue[PC_G_HEAVY + 2 * (base + c) + 1] = c;
                    }
                    if (room) continue;
                }
                unsigned long long hits = 0;
                if (n_own <= 64 && n_cand <= 64 * PC_D_MAXCH) {
                    // the usual case: one chunk of home particles -- chunk 0 of the candidates itself -- and one piece
                    pc_d_chunk c[PC_D_MAXCH];
                    const int nch = (int)((n_cand + 63) >> 6);
#pragma unroll
                    for (int q = 0; q < PC_D_MAXCH; q++)
                        if (q < nch) c[q] = pc_d_load(pts, 64u * q, lane, first, start);
                    hits = pc_d_homes_n<1>(nch, c[0], 0u, 0u, n_own, c, 0u, n_own, d2max);
                } else {
                    // any size: home particles in chunks of 64, candidates in pieces of PC_D_MAXCH chunks
                    for (uint32_t hb = 0; hb < n_own; hb += 64) {
                        const uint32_t i1 = min(hb + 64u, n_own);
                        pc_d_chunk h;
                        {
                            const pc_pt3 p = pts[own0 + min(hb + (uint32_t)lane, n_own - 1)];
                            h.x = p.x; h.y = p.y; h.z = p.z;
                        }
                        // pieces that end at or before list index hb hold own-cell candidates before every home particle of the chunk
                        for (uint32_t cb = hb / (64u * PC_D_MAXCH) * (64u * PC_D_MAXCH); cb < n_cand; cb += 64u * PC_D_MAXCH) {
                            pc_d_chunk c[PC_D_MAXCH];
                            const int nch = (int)min((uint32_t)PC_D_MAXCH, (n_cand - cb + 63) >> 6);
#pragma unroll
                            for (int q = 0; q < PC_D_MAXCH; q++)
                                if (q < nch) c[q] = pc_d_load(pts, cb + 64u * q, lane, first, start);
                            hits += cb < n_own ? pc_d_homes_n<2>(nch, h, hb, hb, i1, c, cb, n_own, d2max)
                                               : pc_d_homes_n<0>(nch, h, hb, hb, i1, c, cb, n_own, d2max);
                        }
                    }
                }
                if (lane == 0) acc += hits;
            }
        }
        m_local = __builtin_amdgcn_readfirstlane(m_next);
    }
    pc_block_add_u64<1>(acc, out);
}
