// Pair test, variant 3 ("group"): one wavefront owns one home pencil; lanes are
// spread over CANDIDATES, not over home particles.
//
// Replaces counter_thread/compare/count_chunk/dist2 (reference main.c:85-91,
// 161-234).  For home pencil (i,j), the 13 forward neighbour cells of (i,j,k)
// (main.c:195-209) are cells k-1..k+1 of the pencils (i,j+1) (i+1,j-1) (i+1,j)
// (i+1,j+1) -- "rows 1-4" -- plus cell k+1 of the home pencil ("row 0").
//
// The wave walks its pencil in windows of up to 64 z cells, lane t owning cell
// t of the window for all bookkeeping (table reads, prefix sums, the gather
// list).  The points of rows 1-4 are copied to LDS MERGED CELL-MAJOR (cell t's
// points of all four rows are adjacent, padded to an even count), so the
// candidates of a home particle in cell k are ONE contiguous LDS range: cells
// k-1..k+1 of the merged array.  The copy itself is a gather through LDS-DMA
// loads (global memory -> LDS without passing registers).  A group of G lanes
// then takes one home particle and each lane NP pairs of candidates (one 8-byte
// LDS read per pair and coordinate, v_pk_add/mul_f32 on both at once), so a wave
// tests 64/G home particles per step with no per-lane loops and no divergence;
// (G, NP) is picked per window from the mean range length.  Same-pencil pairs
// (later points of the cell, and cell k+1 of row 0: main.c:215-221 and offset
// (0,0,1)) are a second short range of the same step.  A home cell whose
// candidates do not fit the LDS buffer is counted against buffer-sized pieces of
// its candidate ranges.
//
// Hit counts come from compare masks (ballot + popcount on the scalar unit);
// one 64-bit atomic per workgroup.  Waves are persistent and draw pencils from a
// work counter per XCD group (speed only).
#pragma once
#include <utility>
#include "pc_device.h"

#define PC_G_WAVES 1
#define PC_G_CAPM 448                   // merged candidates (rows 1-4) per window; a multiple of 64
#define PC_G_CAPH 128                   // row-0 points per window; a multiple of 64
#define PC_G_BAND (PC_NLAYERS / 8)
#define PC_G_QUEUE_STRIDE 32             // one work counter per XCD group, each on its own 128-byte line
// Heavy pencils (clustered inputs) are cut into z segments of roughly equal cost
// by a planning kernel and handed out one segment at a time after the whole
// pencils.  Layout of the work area (32-bit words):
//   [0, 256)                      the eight band counters (word 32 g), each followed by the dense kernel's draw counter of group g
//   [PC_G_SEG_COUNT]              segments planned        [PC_G_SEG_DRAW] segments handed out
//   [PC_G_FLAGS, +PC_G_FLAG_WORDS)   one byte per pencil: 1 = planned as segments
//   [PC_G_SEGS, +2*PC_G_SEG_CAP)     segments: pencil + 1 (0 = empty slot), z_begin | z_end << 16
#define PC_G_SEG_COUNT 256
#define PC_G_SEG_DRAW 288
#define PC_G_FLAGS 320
#define PC_G_FLAG_WORDS ((PC_NBB * PC_NBB + 3) / 4 + 3)
#define PC_G_SEGS (PC_G_FLAGS + PC_G_FLAG_WORDS)
#ifndef PC_G_SEG_CAP
#define PC_G_SEG_CAP 262144
#endif
// A single cell so full that its pairs are too much for one wave (many points in one
// place) is not processed by the wave that meets it: its home particles go, 256 at a
// time, on a list that a follow-up kernel (pc_c3_heavy, thread per home particle)
// works through with the whole GPU.
//   [PC_G_HEAVY_COUNT]               items appended
//   [PC_G_HEAVY, +2*PC_G_HEAVY_CAP)  items: cell key (0xFFFFFFFF = void), chunk of 256 home particles
#define PC_G_HEAVY_COUNT 304
// points in pencils far fuller than the average one (counted by the planning kernel): a clustered input.
// The kernels of one pair-test call take it as the switch between their two divisions of the work (PC_GATE_*).
#define PC_G_DENSE_POINTS 296
#define PC_G_SERVED 312               // set by the planning kernel when the fused pair test serves the call: every later kernel returns at once
#define PC_GATE_ALWAYS 0
#define PC_GATE_IF_EVEN 1            // run only if the dense points are at most the limit
#define PC_GATE_IF_CLUSTERED 2       // run only if they are more
// set by the planning kernel when its segments do not fit the list (an input so heavy that more than PC_G_SEG_CAP
// segments are wanted, e.g. 10^9 points on one GPU): the kernels then divide the work as for a clustered input, whose
// group kernel takes uncut pencils whole; the merged-window kernel would find the list full for what it defers and
// count those pencils a home particle per lane
#define PC_G_SEG_OVERFLOW 290
__device__ __forceinline__ bool pc_gate_open(const unsigned int *queue, int gate, uint32_t limit)
{
    if (__builtin_amdgcn_readfirstlane(queue[PC_G_SERVED]) != 0) return false;
    if (gate == PC_GATE_ALWAYS) return true;
    const bool clustered = __builtin_amdgcn_readfirstlane(queue[PC_G_DENSE_POINTS]) > limit ||
                           __builtin_amdgcn_readfirstlane(queue[PC_G_SEG_OVERFLOW]) != 0;
    return (gate == PC_GATE_IF_CLUSTERED) == clustered;
}
// Who works the segment list off on an even input is decided per call from what is on it: the points in segments of
// well-filled cells (at least dense_min points per cell on average) and the points in the others are added up by
// whoever lists a segment; pc_c6_count_dense takes the list if the former are the majority (its 64-candidate chunks
// are full there), pc_c3_count_groups<false> if not (measured: the dense kernel loses below about 11 points per cell).
#define PC_G_SEG_POINTS_DENSE 291
#define PC_G_SEG_POINTS_THIN 292
// home pencils the planning kernel has cut into segments: if that is all of them (every pencil of a dense slab holds
// thousands of points), the merged-window kernel has nothing to do and returns at once instead of visiting each
#define PC_G_PLANNED_PENCILS 293
#define PC_SEGS_TO_DENSE 1               // which of the two kernels asks pc_segments_are_dense
#define PC_SEGS_TO_GROUP 0
__device__ __forceinline__ bool pc_segment_is_dense(uint32_t points, int cells, uint32_t dense_min)
{
    return (unsigned long long)points >= (unsigned long long)dense_min * (uint32_t)cells;
}
// (dense_min = 0: the caller has switched the choice off and launches only the kernel it wants)
__device__ __forceinline__ bool pc_segments_go_to(const unsigned int *queue, int who, uint32_t dense_min)
{
    if (dense_min == 0) return true;
    const bool dense = __builtin_amdgcn_readfirstlane(queue[PC_G_SEG_POINTS_DENSE]) >= __builtin_amdgcn_readfirstlane(queue[PC_G_SEG_POINTS_THIN]);
    return dense == (who == PC_SEGS_TO_DENSE);
}
#define PC_G_HEAVY (PC_G_SEGS + 2 * PC_G_SEG_CAP)
#define PC_G_HEAVY_CAP 65536
#define PC_G_HEAVY_CHUNK 256
#ifndef PC_G_HEAVY_COST
#define PC_G_HEAVY_COST (1ull << 20)     // home particles x candidates of one cell from which on it is listed
#endif
#define PC_G_WORK_WORDS (PC_G_HEAVY + 2 * PC_G_HEAVY_CAP)
#ifndef PC_G_PLAN_MIN_POINTS
#define PC_G_PLAN_MIN_POINTS 2048        // pencils with fewer points are never cut
#endif
#ifndef PC_G_PLAN_TARGET
#define PC_G_PLAN_TARGET 70000u          // cost of one segment, in units of n_c * (n_{c-1} + n_c + n_{c+1}) (about 4.5 tests each)
#endif
#define PC_G_WG_PER_BAND ((PC_G_BAND + PC_G_WAVES - 1) / PC_G_WAVES)

typedef float pc_v2f __attribute__((ext_vector_type(2)));

struct alignas(16) pc_group_lds {
    float x[PC_G_CAPM], y[PC_G_CAPM], z[PC_G_CAPM];   // merged rows 1-4 (reads may run past the end into the next arrays)
    union {
        struct {
            float hx[PC_G_CAPH], hy[PC_G_CAPH], hz[PC_G_CAPH]; // row 0: home cells and the cell above
            uint2 meta[PC_G_CAPH];                             // per home particle: .x = lo | hi<<16 (merged range), .y = end of its row-0 range
        };
        // sorted-array slot of every merged candidate (gather list of the copy); dead once
        // the merged copy has been issued, which is before the row-0 data is written
        uint32_t src[PC_G_CAPM];
    };
};
static_assert(sizeof(uint32_t) * PC_G_CAPM <= (3 * sizeof(float) + sizeof(uint2)) * PC_G_CAPH, "the gather list must fit under the row-0 arrays");

// Asynchronous copy of one dword per lane from global memory into LDS: lane t's
// dword lands at lds_chunk[t].  The source address is per lane, the destination
// is wave-uniform base + 4*lane (global_load_lds_dword); byte_offset is added to
// both addresses by the instruction, hence the compensation on the LDS side.
typedef __attribute__((address_space(3))) void pc_lds_void;
typedef const __attribute__((address_space(1))) void pc_glb_void;
template <int BYTE_OFFSET>
__device__ __forceinline__ void pc_dma_dword(const float *g, float *lds_chunk)
{
    __builtin_amdgcn_global_load_lds((pc_glb_void *)g, (pc_lds_void *)(reinterpret_cast<char *>(lds_chunk) - BYTE_OFFSET),
                                     4, BYTE_OFFSET, 0);
}
// one point (x, y, z) per lane into three separate arrays
__device__ __forceinline__ void pc_dma_point(const float *g, float *lx, float *ly, float *lz)
{
    pc_dma_dword<0>(g, lx);
    pc_dma_dword<4>(g, ly);
    pc_dma_dword<8>(g, lz);
}
// the copies are complete (and visible to this wave's LDS reads) after this
__device__ __forceinline__ void pc_dma_wait()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Two neighbouring floats at an even index of an 8-byte aligned LDS array as ONE
// ds_read_b64.  Written as inline assembly because the compiler either assumes
// 4-byte alignment (two half-rate 4-byte reads) or, told the alignment, fuses two
// such reads into a ds_read2_b64, which the LDS serves at half the rate of two
// separate ds_read_b64.  The compiler does not track these reads: the values may
// be used only after pc_lds_pairs_arrived() has been applied to them.
template <int BYTE_OFFSET>
__device__ __forceinline__ pc_v2f pc_lds_pair(uint32_t lds_byte_addr)
{
    pc_v2f v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(BYTE_OFFSET));
    return v;
}
__device__ __forceinline__ void pc_lds_pairs_arrived(pc_v2f &a, pc_v2f &b, pc_v2f &c)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c) : : "memory");
}
// pairs c = 0..NP-1 of a lane: candidates 2*G*c and 2*G*c + 1 past the lane's first, from the three coordinate arrays
template <int G, int... C>
__device__ __forceinline__ void pc_lds_pairs_request(uint32_t ap, pc_v2f *x, pc_v2f *y, pc_v2f *z, std::integer_sequence<int, C...>)
{
    ((x[C] = pc_lds_pair<8 * G * C>(ap), y[C] = pc_lds_pair<4 * PC_G_CAPM + 8 * G * C>(ap),
      z[C] = pc_lds_pair<8 * PC_G_CAPM + 8 * G * C>(ap)), ...);
}
__device__ __forceinline__ uint32_t pc_lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// Cross-lane reads that stay off the LDS crossbar: a wave-uniform lane index is a
// v_readlane (the result is scalar), a neighbour's value is a DPP wave shift.
__device__ __forceinline__ uint32_t pc_lane_value(uint32_t v, int uniform_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(uniform_lane));
}
__device__ __forceinline__ uint32_t pc_from_lane_below(uint32_t v)      // lane t gets lane t-1's value, lane 0 gets 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t pc_from_lane_above(uint32_t v)      // lane t gets lane t+1's value, lane 63 gets 0
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

// Candidate-parallel test of the nh home particles staged in LDS.  A group of G
// lanes takes one home particle, so the wave handles 64/G of them per step; lane
// j of a group tests the candidate pairs (lo + 2j + 2G*c, +1), c = 0..NP-1, i.e.
// 2*G*NP candidates per home particle and pass, and the pass repeats while any
// group has candidates left.  Ranges start and end on even slots (cell blocks
// are padded with a never-matching point), so a pair is one aligned 8-byte LDS
// read and is live or dead as a whole.
template <int G, int NP>
__device__ __forceinline__ uint32_t pc_g_steps(const pc_group_lds &L, uint32_t h_first, uint32_t nh, int lane, float d2max)
{
    constexpr int NP0 = G >= 4 ? 1 : 4 / G;             // same-pencil pairs per lane: at least 8 candidates per group
    uint32_t hits = 0;
    // (the lane number is made opaque here so that the per-lane constants of each instantiation are
    // recomputed per window instead of being kept in registers across the whole kernel)
    asm volatile("" : "+v"(lane));
    const uint32_t j = lane & (G - 1), grp = lane / G;
    nh = __builtin_amdgcn_readfirstlane(nh);             // home particles h_first .. nh-1; wave-uniform: keep the step loop scalar
    const uint32_t ax_lds = pc_lds_addr(L.x);
    for (uint32_t base = __builtin_amdgcn_readfirstlane(h_first); base < nh; base += 64 / G) {
        const uint32_t hp = min(base + grp, nh - 1);
        const uint2 m = L.meta[hp];
        const float hx = L.hx[hp], hy = L.hy[hp], hz = L.hz[hp];
        const bool live = base + grp < nh;
        uint32_t p = (m.x & 0xFFFFu) + 2 * j;
        // candidates from p to the end of the range (even, or <= 0): pair c is live iff rem > 2*G*c
        int rem = live ? (int)(m.x >> 16) - (int)p : 0;
        const pc_v2f ax = {hx, hx}, ay = {hy, hy}, az = {hz, hz};    // splat operands of the packed math
        for (;;) {
            pc_v2f x[NP], y[NP], z[NP], sq[NP];
            const uint32_t ap = ax_lds + 4 * p;          // LDS byte address of x[p]; y and z are whole arrays further
            pc_lds_pairs_request<G>(ap, x, y, z, std::make_integer_sequence<int, NP>());
#pragma unroll
            for (int c = 0; c < NP; c++) {
                pc_lds_pairs_arrived(x[c], y[c], z[c]);
                // (dx*dx + dy*dy) + dz*dz, each step rounded separately (no contraction: see build flags)
                const pc_v2f dx = ax - x[c], dy = ay - y[c], dz = az - z[c];
                sq[c] = (dx * dx + dy * dy) + dz * dz;
            }
            // one ballot per compare, masks combined on the scalar unit (a ballot of
            // "a && b" would be materialised through a 0/1 VGPR and compared again)
#pragma unroll
            for (int c = 0; c < NP; c++) {
                const unsigned long long ok = __builtin_amdgcn_ballot_w64(rem > 2 * G * c);
                hits += __popcll(__builtin_amdgcn_ballot_w64(sq[c].x <= d2max) & ok)
                      + __popcll(__builtin_amdgcn_ballot_w64(sq[c].y <= d2max) & ok);
            }
            rem -= 2 * G * NP;
            p += 2 * G * NP;
            if (__builtin_amdgcn_ballot_w64(rem > 0) == 0) break;
        }
        // same pencil: later points of the home cell and the cell above are row-0 slots
        // hp+1 .. m.y-1 (main.c:215-221 and offset (0,0,1)); two neighbouring slots per lane and pass
        uint32_t p0 = hp + 1 + 2 * j;
        int rem0 = live ? (int)m.y - (int)p0 : 0;         // slots from p0 to the end: slot u of pair c is live iff rem0 > 2*G*c + u
        for (;;) {
            pc_v2f sq[NP0];
#pragma unroll
            for (int c = 0; c < NP0; c++) {
                const uint32_t pc = p0 + 2 * G * c;       // dead slots may lie past the row-0 arrays, still inside the LDS block
                const pc_v2f x = {L.hx[pc], L.hx[pc + 1]}, y = {L.hy[pc], L.hy[pc + 1]}, z = {L.hz[pc], L.hz[pc + 1]};
                const pc_v2f dx = ax - x, dy = ay - y, dz = az - z;
                sq[c] = (dx * dx + dy * dy) + dz * dz;
            }
#pragma unroll
            for (int c = 0; c < NP0; c++) {
                hits += __popcll(__builtin_amdgcn_ballot_w64(sq[c].x <= d2max) & __builtin_amdgcn_ballot_w64(rem0 > 2 * G * c))
                      + __popcll(__builtin_amdgcn_ballot_w64(sq[c].y <= d2max) & __builtin_amdgcn_ballot_w64(rem0 > 2 * G * c + 1));
            }
            rem0 -= 2 * G * NP0;
            p0 += 2 * G * NP0;
            if (__builtin_amdgcn_ballot_w64(rem0 > 0) == 0) break;
        }
    }
    return hits;    // wave-uniform (every lane holds the wave total)
}

// All home particles of a window with 2*G*NP candidates per home particle and
// pass.  Whole steps of 64/G home particles use the narrowest groups; a last
// step that would be less than half (quarter) full is run with groups twice (four
// times) as wide and as many fewer pairs per lane: the same candidates per pass
// at half (a quarter of) the instructions.
template <int G, int NP>
__device__ __forceinline__ uint32_t pc_g_window(const pc_group_lds &L, uint32_t nh, int lane, float d2max)
{
    constexpr uint32_t H = 64 / G;
    nh = __builtin_amdgcn_readfirstlane(nh);
    if constexpr (NP >= 4 && G <= 8) {
        const uint32_t rest = nh % H, whole = nh - rest;
        uint32_t hits = pc_g_steps<G, NP>(L, 0, rest > H / 2 ? nh : whole, lane, d2max);
        if (rest != 0 && rest <= H / 4) hits += pc_g_steps<4 * G, NP / 4>(L, whole, nh, lane, d2max);
        else if (rest != 0 && rest <= H / 2) hits += pc_g_steps<2 * G, NP / 2>(L, whole, nh, lane, d2max);
        return hits;
    } else {
        return pc_g_steps<G, NP>(L, 0, nh, lane, d2max);
    }
}

// Candidate-stationary form for dense data: the wave keeps the 128*PAIRS merged
// candidates [p0, p0 + 128*PAIRS) in registers (PAIRS packed pairs per lane) and
// tests them against every home particle h[h_lo..h_hi), whose coordinates arrive
// as LDS broadcasts.  All candidates below c_hi are valid partners, except that
// with USE_LO a home particle only pairs with candidates at or after its own
// meta.x & 0xFFFF (the "later slots of the same cell" rule of main.c:215-221).
template <bool USE_LO, int PAIRS>
__device__ __forceinline__ uint32_t pc_g_chunk(const pc_group_lds &L, uint32_t p0, uint32_t c_hi,
                                               uint32_t h_lo, uint32_t h_hi, int lane, float d2max)
{
    uint32_t hits = 0;
    const uint32_t p = p0 + lane;
    pc_v2f x[PAIRS], y[PAIRS], z[PAIRS];
    const float never = __builtin_nanf("");                  // a slot behind the range never matches: no mask per home particle
#pragma unroll
    for (int c = 0; c < PAIRS; c++) {
        x[c] = pc_v2f{p + 128 * c < c_hi ? L.x[p + 128 * c] : never, p + 128 * c + 64 < c_hi ? L.x[p + 128 * c + 64] : never};
        y[c] = pc_v2f{L.y[p + 128 * c], L.y[p + 128 * c + 64]};
        z[c] = pc_v2f{L.z[p + 128 * c], L.z[p + 128 * c + 64]};
    }
    for (uint32_t hp = h_lo; hp < h_hi; hp++) {
        const float hx = L.hx[hp], hy = L.hy[hp], hz = L.hz[hp];
        const pc_v2f ax = {hx, hx}, ay = {hy, hy}, az = {hz, hz};
        uint32_t lo = 0;
        if (USE_LO) lo = L.meta[hp].x & 0xFFFFu;
#pragma unroll
        for (int c = 0; c < PAIRS; c++) {
            const pc_v2f dx = ax - x[c], dy = ay - y[c], dz = az - z[c];
            const pc_v2f sq = (dx * dx + dy * dy) + dz * dz;
            unsigned long long b0 = __builtin_amdgcn_ballot_w64(sq.x <= d2max);
            unsigned long long b1 = __builtin_amdgcn_ballot_w64(sq.y <= d2max);
            if (USE_LO) {
                b0 &= __builtin_amdgcn_ballot_w64(p + 128 * c >= lo);
                b1 &= __builtin_amdgcn_ballot_w64(p + 128 * c + 64 >= lo);
            }
            hits += __popcll(b0) + __popcll(b1);
        }
    }
    return hits;
}

// candidates [c_lo, c_hi) in chunks of 256, the last one 128 wide if that is enough
template <bool USE_LO>
__device__ __forceinline__ uint32_t pc_g_block(const pc_group_lds &L, uint32_t c_lo, uint32_t c_hi,
                                               uint32_t h_lo, uint32_t h_hi, int lane, float d2max)
{
    uint32_t hits = 0;
    c_lo = __builtin_amdgcn_readfirstlane(c_lo); c_hi = __builtin_amdgcn_readfirstlane(c_hi);
    h_lo = __builtin_amdgcn_readfirstlane(h_lo); h_hi = __builtin_amdgcn_readfirstlane(h_hi);
    uint32_t p0 = c_lo;
    for (; p0 + 128 < c_hi; p0 += 256) hits += pc_g_chunk<USE_LO, 2>(L, p0, c_hi, h_lo, h_hi, lane, d2max);
    if (p0 < c_hi) hits += pc_g_chunk<USE_LO, 1>(L, p0, c_hi, h_lo, h_hi, lane, d2max);
    return hits;
}

// Planning: a workgroup of 16 waves per 64 home pencils of the slab, four pencils
// per wave.  A pencil with many points is cut wherever the running cost estimate
// passes a multiple of PC_G_PLAN_TARGET (the cost of one cell is capped at the
// target, so the running sum of a pencil fits 32 bits); its segments are appended to
// the list (all of them or, if the list is full, none) and its flag tells the pair
// test to leave the pencil to the segment phase.  The segments of a pencil always
// cover the cells [0, 400) exactly.  Lane l looks at the cells l, l + 64, ... so the
// table row is read in whole lines; the workgroup reserves the list slots of its 64
// pencils with one atomic (the counter is a single address: one atomic per pencil
// would serialise the whole kernel on it).
#define PC_G_PLAN_STEPS ((PC_NLAYERS + 63) / 64)
#define PC_G_PLAN_PENCILS 64             // per workgroup
#define PC_G_PLAN_TPB 1024
static_assert((unsigned long long)PC_G_PLAN_TARGET * PC_NLAYERS < 0xFFFFFFFFull, "running cost of a pencil must fit 32 bits");

__global__ void __launch_bounds__(PC_G_PLAN_TPB)
pc_c3_plan(const uint32_t *__restrict__ T, int x_lo, int x_hi, unsigned int *__restrict__ work, uint32_t seg_cap,
           uint32_t full_pencil,    // a pencil with more points than this counts as dense (several times the average)
           const uint32_t *__restrict__ fgate,
           uint32_t dense_min)      // points per cell from which a segment counts as well filled (0: not counted)
{
    if (pc_fused_serves(fgate)) {
        if (blockIdx.x == 0 && threadIdx.x == 0) work[PC_G_SERVED] = 1;
        return;
    }
    __shared__ unsigned long long ends[PC_G_PLAN_PENCILS][PC_G_PLAN_STEPS];   // per 64 cells: the cells that close a segment
    __shared__ uint32_t seg_n[PC_G_PLAN_PENCILS], seg_at[PC_G_PLAN_PENCILS];
    __shared__ unsigned long long to_plan;
    __shared__ uint32_t kind_points[2];        // points in the workgroup's thinly / well filled segments (see pc_segments_go_to)
    if (threadIdx.x < 2) kind_points[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_pencils = (x_hi - x_lo) * PC_NLAYERS, per_wave = PC_G_PLAN_PENCILS / (PC_G_PLAN_TPB / 64);
    const uint32_t target = PC_G_PLAN_TARGET;
    // the populations first, a thread per pencil: on most inputs no pencil of the workgroup needs a plan
    if (wave == 0) {
        const int t = blockIdx.x * PC_G_PLAN_PENCILS + lane;
        uint32_t population = 0;
        if (t < n_pencils) {
            const uint32_t *row = T + (uint64_t)((uint32_t)(x_lo + t / PC_NLAYERS) * PC_NBB + t % PC_NLAYERS) * PC_NBB;
            population = row[PC_NLAYERS] - row[0];
        }
        const uint32_t dense_total = pc_wave_sum(population > full_pencil ? population : 0);
        if (lane == 0 && dense_total) atomicAdd(work + PC_G_DENSE_POINTS, dense_total);
        const unsigned long long need = __builtin_amdgcn_ballot_w64(population >= PC_G_PLAN_MIN_POINTS);
        if (lane == 0) to_plan = need;
        if (!need && t < n_pencils)
            reinterpret_cast<unsigned char *>(work + PC_G_FLAGS)[(uint32_t)(x_lo + t / PC_NLAYERS) * PC_NBB + t % PC_NLAYERS] = 0;
    }
    __syncthreads();
    const unsigned long long need = to_plan;
    if (!need) return;
    for (int q = 0; q < per_wave; q++) {
        const int slot = wave * per_wave + q, t = blockIdx.x * PC_G_PLAN_PENCILS + slot;
        uint32_t segs = 0;
        if (t < n_pencils) {
            const uint32_t *row = T + (uint64_t)((uint32_t)(x_lo + t / PC_NLAYERS) * PC_NBB + t % PC_NLAYERS) * PC_NBB;
            if ((need >> slot) & 1) {
                uint32_t running = 0;
#pragma unroll
                for (int u = 0; u < PC_G_PLAN_STEPS; u++) {
                    const int c = u * 64 + lane;
                    uint32_t cost = 0;
                    if (c < PC_NLAYERS) {
                        const uint32_t r0 = row[c > 0 ? c - 1 : 0], r1 = row[c], r2 = row[c + 1], r3 = row[c + 2];
                        const unsigned long long full = (unsigned long long)(r2 - r1) * ((r1 - r0) + (r2 - r1) + (r3 - r2));
                        cost = full < target ? (uint32_t)full : target;
                    }
                    const uint32_t incl = pc_wave_incl_scan(cost) + running;
                    const bool end = c < PC_NLAYERS && (incl / target != (incl - cost) / target || c == PC_NLAYERS - 1);
                    const unsigned long long e = __builtin_amdgcn_ballot_w64(end);
                    if (lane == 0) ends[slot][u] = e;
                    segs += (uint32_t)__popcll(e);
                    running = __builtin_amdgcn_readlane(incl, 63);
                }
            }
        }
        if (lane == 0) seg_n[slot] = segs >= 2 ? segs : 0;       // a pencil of one segment stays whole
    }
    __syncthreads();
    if (wave == 0) {
        // one reservation for the workgroup's pencils, in pencil order
        const uint32_t mine = seg_n[lane];
        const uint32_t incl = pc_wave_incl_scan(mine);
        const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(work + PC_G_SEG_COUNT, total);
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t at = base + incl - mine;
        seg_at[lane] = at;
        const int t = blockIdx.x * PC_G_PLAN_PENCILS + lane;
        const bool cut = t < n_pencils && mine && at + mine <= seg_cap;
        if (t < n_pencils)
            reinterpret_cast<unsigned char *>(work + PC_G_FLAGS)[(uint32_t)(x_lo + t / PC_NLAYERS) * PC_NBB + t % PC_NLAYERS] = cut;
        const uint32_t n_cut = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(cut));
        if (lane == 0 && n_cut) atomicAdd(work + PC_G_PLANNED_PENCILS, n_cut);
        if (seg_cap != 0 && __builtin_amdgcn_ballot_w64(mine && at + mine > seg_cap) != 0 && lane == 0) work[PC_G_SEG_OVERFLOW] = 1u;
    }
    __syncthreads();
    for (int q = 0; q < per_wave; q++) {
        const int slot = wave * per_wave + q, t = blockIdx.x * PC_G_PLAN_PENCILS + slot;
        const uint32_t segs = seg_n[slot];
        uint32_t at = seg_at[slot];
        if (!segs) continue;
        if (at + segs <= seg_cap) {
            const uint32_t pencil = (uint32_t)(x_lo + t / PC_NLAYERS) * PC_NBB + t % PC_NLAYERS;
            int z_after = 0;                               // first cell after the last segment closed in earlier steps
#pragma unroll
            for (int u = 0; u < PC_G_PLAN_STEPS; u++) {
                const unsigned long long e = ends[slot][u], below = e & ((1ull << lane) - 1);
                if ((e >> lane) & 1) {
                    const int z_begin = below ? u * 64 + 64 - __clzll(below) : z_after;
                    const uint32_t w = at + (uint32_t)__popcll(below);
                    work[PC_G_SEGS + 2 * w] = pencil + 1;
                    work[PC_G_SEGS + 2 * w + 1] = (uint32_t)z_begin | ((uint32_t)(u * 64 + lane + 1) << 16);
                    if (dense_min != 0) {                  // the segment's points, to the workgroup's sum of its kind
                        const uint32_t *row = T + (uint64_t)pencil * PC_NBB;
                        const uint32_t pts_seg = row[u * 64 + lane + 1] - row[z_begin];
                        atomicAdd(&kind_points[pc_segment_is_dense(pts_seg, u * 64 + lane + 1 - z_begin, dense_min) ? 1 : 0], pts_seg);
                    }
                }
                at += (uint32_t)__popcll(e);
                if (e) z_after = u * 64 + 64 - __clzll(e);
            }
        } else {
            // the list is full: the pencil stays whole, the slots it reserved below the cap stay empty
            for (uint32_t w = at + lane; w < seg_cap; w += 64) work[PC_G_SEGS + 2 * w] = 0;
        }
    }
    if (dense_min != 0) {                                  // (one atomic per workgroup and kind: the sums are single addresses)
        __syncthreads();
        if (threadIdx.x < 2 && kind_points[threadIdx.x] != 0)
            atomicAdd(work + (threadIdx.x ? PC_G_SEG_POINTS_DENSE : PC_G_SEG_POINTS_THIN), kind_points[threadIdx.x]);
    }
}

// Candidate-stationary blocks over a JOINT list for a home cell of a staged window: first the row-0 slots from the
// cell's first home particle to the end of the cell above (row-0 arrays, from slot r0), then the merged candidates
// [c_lo, c_hi).  Home particle hp (a row-0 slot) pairs with list indices from hp - r0 + 1 on: the later slots of its
// own cell, the cell above, and every merged candidate (main.c:215-221 and the 13 offsets).  That rule is a shift of
// the compare masks on the scalar unit, and only in the first block (a window holds at most PC_G_CAPH row-0 slots).
template <int PAIRS, int MASKED>   // MASKED: how many of the block's leading 64-candidate parts hold slots of the home cell itself
__device__ __forceinline__ uint32_t pc_g_joint_chunk(const pc_group_lds &L, uint32_t p0, uint32_t n_r0, uint32_t total,
                                                     uint32_t r0, uint32_t c_lo, uint32_t h_hi, int lane, float d2max)
{
    static_assert(offsetof(pc_group_lds, hx) == 3 * PC_G_CAPM * sizeof(float), "row-0 arrays follow the merged ones");
    const float *all = L.x;                                  // every array of the block as one
    pc_v2f x[PAIRS], y[PAIRS], z[PAIRS];
    const float never = __builtin_nanf("");                  // a slot behind the list never matches: no mask per home particle
#pragma unroll
    for (int c = 0; c < PAIRS; c++) {
        float cx[2], cy[2], cz[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t v = p0 + 128 * c + 64 * u + lane;
            const uint32_t vv = min(v, total - 1);
            const bool in_r0 = vv < n_r0;
            const uint32_t at = in_r0 ? 3 * PC_G_CAPM + r0 + vv : c_lo + (vv - n_r0);
            const uint32_t stride = in_r0 ? PC_G_CAPH : PC_G_CAPM;
            cx[u] = v < total ? all[at] : never;
            cy[u] = all[at + stride];
            cz[u] = all[at + 2 * stride];
        }
        x[c] = pc_v2f{cx[0], cx[1]};
        y[c] = pc_v2f{cy[0], cy[1]};
        z[c] = pc_v2f{cz[0], cz[1]};
    }
    uint32_t hits = 0;
    for (uint32_t hp = r0; hp < h_hi; hp++) {
        const float hx = L.hx[hp], hy = L.hy[hp], hz = L.hz[hp];
        const pc_v2f ax = {hx, hx}, ay = {hy, hy}, az = {hz, hz};
#pragma unroll
        for (int c = 0; c < PAIRS; c++) {
            const pc_v2f dx = ax - x[c], dy = ay - y[c], dz = az - z[c];
            const pc_v2f sq = (dx * dx + dy * dy) + dz * dz;
            unsigned long long b0 = __builtin_amdgcn_ballot_w64(sq.x <= d2max);
            unsigned long long b1 = __builtin_amdgcn_ballot_w64(sq.y <= d2max);
            // (p0 = 0) lanes below hp - r0 + 1 - 128 c - 64 u are at or before hp
            if (2 * c < MASKED) {
                const uint32_t lo = hp - r0 + 1, d0 = lo > 128u * c ? lo - 128u * c : 0u;
                b0 = d0 >= 64 ? 0ull : b0 & (~0ull << d0);
            }
            if (2 * c + 1 < MASKED) {
                const uint32_t lo = hp - r0 + 1, d1 = lo > 128u * c + 64u ? lo - 128u * c - 64u : 0u;
                b1 = d1 >= 64 ? 0ull : b1 & (~0ull << d1);
            }
            hits += __popcll(b0) + __popcll(b1);
        }
    }
    return hits;
}

// home particles [r0, h_hi) of one cell of a staged window against row-0 slots [r0, r0_end) and merged candidates [c_lo, c_hi)
__device__ __forceinline__ uint32_t pc_g_joint(const pc_group_lds &L, uint32_t r0, uint32_t h_hi, uint32_t r0_end,
                                               uint32_t c_lo, uint32_t c_hi, int lane, float d2max)
{
    r0 = __builtin_amdgcn_readfirstlane(r0); h_hi = __builtin_amdgcn_readfirstlane(h_hi);
    c_lo = __builtin_amdgcn_readfirstlane(c_lo);
    const uint32_t n_r0 = __builtin_amdgcn_readfirstlane(r0_end - r0);
    const uint32_t total = n_r0 + __builtin_amdgcn_readfirstlane(c_hi - c_lo);
    uint32_t hits = 0, p0 = 0;
    const bool two = h_hi - r0 > 64;                           // the cell's own slots reach into the second 64 candidates
    if (total > 128)
        hits += two ? pc_g_joint_chunk<2, 2>(L, 0, n_r0, total, r0, c_lo, h_hi, lane, d2max)
                    : pc_g_joint_chunk<2, 1>(L, 0, n_r0, total, r0, c_lo, h_hi, lane, d2max);
    else
        hits += two ? pc_g_joint_chunk<1, 2>(L, 0, n_r0, total, r0, c_lo, h_hi, lane, d2max)
                    : pc_g_joint_chunk<1, 1>(L, 0, n_r0, total, r0, c_lo, h_hi, lane, d2max);
    for (p0 = 256; p0 + 128 < total; p0 += 256) hits += pc_g_joint_chunk<2, 0>(L, p0, n_r0, total, r0, c_lo, h_hi, lane, d2max);
    if (p0 < total) hits += pc_g_joint_chunk<1, 0>(L, p0, n_r0, total, r0, c_lo, h_hi, lane, d2max);
    return hits;
}

// One home cell against its whole neighbourhood, the candidates going through the merged arrays in pieces: home
// particles [h0, h1) of the sorted array, candidate ranges [c_lo[r], c_hi[r]) -- row 0 from h0 to the end of the cell
// above, rows 1-4 three cells each.  The five ranges are taken as ONE list, row 0 first (its "later slots only"
// rule, main.c:215-221, then reads "list index at least i + 1 - h0" for home particle i, and everything behind
// row 0 pairs with every home particle), cut into pieces of 384 = 256 + 128 candidates (full register blocks)
// until what is left fits the buffer: the part-filled block comes once per cell, not once per row.
__device__ __forceinline__ uint32_t pc_g_cell(pc_group_lds &L, const float *__restrict__ pts, uint32_t h0, uint32_t h1,
                                              const uint32_t c_lo[5], const uint32_t c_hi[5], int lane, float d2max)
{
    uint32_t hits = 0;
    uint32_t v_end[5];                     // list index behind each range
    v_end[0] = c_hi[0] - c_lo[0];
#pragma unroll
    for (int r = 1; r < 5; r++) v_end[r] = v_end[r - 1] + (c_hi[r] - c_lo[r]);
    const uint32_t ctot = v_end[4];
    for (uint32_t v0 = 0; v0 < ctot;) {
        const uint32_t cnt = ctot - v0 <= (uint32_t)PC_G_CAPM ? ctot - v0 : 384u;
#pragma unroll
        for (int qq = 0; qq < PC_G_CAPM / 64; qq++) {
            if (64u * qq < cnt) {
                const uint32_t v = v0 + min(lane + 64u * qq, cnt - 1);
                uint32_t slot = c_lo[0] + v;
#pragma unroll
                for (int r = 1; r < 5; r++)
                    if (v >= v_end[r - 1]) slot = c_lo[r] + (v - v_end[r - 1]);
                pc_dma_point(pts + 3 * (uint64_t)slot, L.x + 64 * qq, L.y + 64 * qq, L.z + 64 * qq);
            }
        }
        const bool own = v0 < h1 - h0;     // the piece holds slots of the home cell itself
        // the cell's home particles in batches: every lane works on the
        // piece's candidates for one home particle at a time (G = 64)
        for (uint32_t hb = h0; hb < h1; hb += PC_G_CAPH) {
            const uint32_t nhb = min((uint32_t)PC_G_CAPH, h1 - hb);
#pragma unroll
            for (int qq = 0; qq < PC_G_CAPH / 64; qq++) {
                const uint32_t e = lane + 64 * qq;
                if (64u * qq < nhb)
                    pc_dma_point(pts + 3 * (uint64_t)(e < nhb ? hb + e : hb), L.hx + 64 * qq, L.hy + 64 * qq, L.hz + 64 * qq);
                if (e < nhb) {
                    const uint32_t first = hb + e + 1 - h0;              // list index of the slot after the home particle
                    const uint32_t lo = first > v0 ? min(first - v0, cnt) : 0u;
                    L.meta[e] = make_uint2(lo | (cnt << 16), 0u);
                }
            }
            pc_dma_wait();
            pc_wave_lds_sync();
            hits += own ? pc_g_block<true>(L, 0, cnt, 0, nhb, lane, d2max) : pc_g_block<false>(L, 0, cnt, 0, nhb, lane, d2max);
            pc_wave_lds_sync();
        }
        v0 += cnt;
    }
    return hits;
}

#ifdef PC_G_TRACE
// Measurement build only: shader-clock cycles a wave of the group kernel spends in each part of its work, kept in
// scalar registers and added at the wave's end to one of 64 rows; [row][0..15] cycles, [row][16..31] times.
// 0 table reads and window bookkeeping, 1 copy into LDS, 2 steps of 2-lane groups, 3 of 4-lane groups, 4 of 8-lane
// groups, 5 register blocks of a dense window; a cell that does not fit the buffers: 7 its set-up,
// 8 the cell itself (pc_g_cell).
__device__ unsigned long long pc_g_trace[64][32];
#define PC_G_MARK(i)                                                      \
    do {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();     \
        tr_[i] += now_ - mark_;                                            \
        tr_[16 + (i)] += 1;                                                \
        mark_ = now_;                                                      \
    } while (0)
#else
#define PC_G_MARK(i) do { } while (0)
#endif

// WHOLE_PENCILS false: the whole pencils have been counted by another kernel, only the listed segments are left
template <bool WHOLE_PENCILS>
__global__ void __launch_bounds__(PC_G_WAVES * 64, 4)   // registers and LDS admit 5 waves per SIMD
pc_c3_count_groups(const float *__restrict__ pts, const uint32_t *__restrict__ T, int x_lo, int x_hi,
                   unsigned int *__restrict__ queue, uint32_t seg_cap, uint32_t heavy_cap,
                   int gate, uint32_t gate_limit, unsigned long long *__restrict__ out, const float d2max,
                   uint32_t dense_min)   // segments only (WHOLE_PENCILS false): see pc_segments_go_to
{
    if (!pc_gate_open(queue, gate, gate_limit)) return;
    if (!WHOLE_PENCILS && !pc_segments_go_to(queue, PC_SEGS_TO_GROUP, dense_min)) return;
    __shared__ pc_group_lds lds_all[PC_G_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pc_group_lds &L = lds_all[wave];
    unsigned long long acc = 0;          // per-lane partial (lane 0 also carries the wave-uniform hit counts)
#ifdef PC_G_TRACE
    unsigned long long tr_[32] = {}, mark_ = __builtin_readcyclecounter();
#endif

    // Persistent waves with a work queue per XCD group.  Workgroups with equal
    // blockIdx%8 share an XCD; that group owns the band of 50 pencils
    // y = g*50 .. g*50+49 of every layer and its waves draw the band's pencils
    // (layer-major) from the group's counter, so neighbouring pencils are in flight
    // together and a wave that finishes a light pencil simply draws the next one.
    // The next index is drawn before the current pencil is processed (its latency
    // is hidden).  When its own band is used up a wave moves on to the next
    // group's band (clustered inputs load the bands unevenly); it leaves the loop
    // once it has seen all eight counters past the end.
    //
    // Pencils the planning kernel has cut into segments are skipped in this first
    // phase; in the second phase the waves draw those segments, one at a time,
    // from a single counter.
    int g = blockIdx.x & 7, bands_done = 0;
    const int band_pencils = (x_hi - x_lo) * PC_G_BAND;
    const unsigned char *planned = reinterpret_cast<const unsigned char *>(queue + PC_G_FLAGS);
    bool segment_phase = !WHOLE_PENCILS;
    int n_segments = 0;
    int m = 0;
    if (segment_phase) {
        n_segments = (int)min(__builtin_amdgcn_readfirstlane(queue[PC_G_SEG_COUNT]), seg_cap);
        if (n_segments == 0) return;                       // (nothing was added to the sum: no atomic either)
        if (lane == 0) m = (int)atomicAdd(queue + PC_G_SEG_DRAW, 1u);
    } else if (lane == 0) m = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE, 1u);
    m = __builtin_amdgcn_readfirstlane(m);
    for (;;) {
        int layer, j, z_begin = 0, z_end = PC_NLAYERS, m_next = 0;
        uint32_t skip = 0;                                 // first phase: non-zero if the pencil is left to the segment phase
        if (!segment_phase) {
            while (m >= band_pencils && ++bands_done < 8) {
                g = (g + 1) & 7;
                if (lane == 0) m = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE, 1u);
                m = __builtin_amdgcn_readfirstlane(m);
            }
            if (m >= band_pencils) {
                segment_phase = true;
                n_segments = (int)min(__builtin_amdgcn_readfirstlane(queue[PC_G_SEG_COUNT]), seg_cap);
                if (lane == 0) m = (int)atomicAdd(queue + PC_G_SEG_DRAW, 1u);
                m = __builtin_amdgcn_readfirstlane(m);
                continue;
            }
            if (lane == 0) m_next = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE, 1u);
            layer = x_lo + m / PC_G_BAND;
            j = g * PC_G_BAND + m % PC_G_BAND;
            skip = planned[layer * PC_NBB + j];             // (requested together with the table entries below)
        } else {
            if (m >= n_segments) break;
            if (lane == 0) m_next = (int)atomicAdd(queue + PC_G_SEG_DRAW, 1u);
            const uint32_t who = __builtin_amdgcn_readfirstlane(queue[PC_G_SEGS + 2 * m]);
            const uint32_t span = __builtin_amdgcn_readfirstlane(queue[PC_G_SEGS + 2 * m + 1]);
            if (who == 0) {                                // slot left empty by a pencil that did not fit the list
                m = __builtin_amdgcn_readfirstlane(m_next);
                continue;
            }
            layer = (int)((who - 1) / PC_NBB);
            j = (int)((who - 1) % PC_NBB);
            z_begin = (int)(span & 0xFFFFu);
            z_end = (int)(span >> 16);
        }
        // table bases of the five rows; invalid rows (outside the occupied grid) are empty
        const int ri[5] = {layer, layer, layer + 1, layer + 1, layer + 1};
        const int rj[5] = {j, j + 1, j - 1, j, j + 1};
        uint32_t tb[5];
        bool valid[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            valid[r] = ri[r] < PC_NLAYERS && rj[r] >= 0 && rj[r] < PC_NLAYERS;
            tb[r] = valid[r] ? ((uint32_t)ri[r] * PC_NBB + rj[r]) * PC_NBB : 0u;
        }
        // home cells [z_begin, z_end) of the pencil (all of it unless this is a planned segment)
        const uint32_t home_lo = T[tb[0] + z_begin], home_hi = T[tb[0] + z_end];
        int z0 = z_begin;
        uint32_t done = skip ? home_hi : home_lo;          // first home slot not yet processed
        // table entries of the current window: lane t holds, per row, the start of
        // cell zl+t and of the cell after it.  The next window's entries are
        // requested as soon as its position is known, so they arrive while the
        // current window is being processed.
        uint32_t a0[5], a1[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const uint32_t *e = T + (tb[r] + (uint32_t)min((z_begin > 0 ? z_begin - 1 : 0) + lane, PC_NBB - 1));
            a0[r] = e[0];
            a1[r] = e[1];
        }
        while (done != home_hi && z0 < z_end) {
            const int zl = z0 > 0 ? z0 - 1 : 0;
            const int th0 = z0 - zl;                       // window index of the first home cell (0 or 1)
            // ---- lane t owns window cell zl + t: per-row starts and counts
            uint32_t s[5], n[5];
#pragma unroll
            for (int r = 0; r < 5; r++) {
                s[r] = valid[r] ? a0[r] : 0u;
                n[r] = (valid[r] && zl + lane < PC_NBB - 1) ? a1[r] - a0[r] : 0u;
            }
            if (lane < th0) n[0] = 0;                      // row-0 points below the first home cell are not needed
            // merged points of the cell, and its block length in the merged array: padded to
            // an even count, so every candidate range starts and ends on an even slot
            const uint32_t nm_pts = n[1] + n[2] + n[3] + n[4];
            const uint32_t nm = (nm_pts + 1) & ~1u;
            const uint32_t cum_m = pc_wave_incl_scan(nm), cum_h = pc_wave_incl_scan(n[0]);
            // longest window that fits both buffers (feasible lanes form a prefix)
            // and ends at cell z_end at the latest (the last window cell only supplies candidates)
            const int nfit = __popcll(__builtin_amdgcn_ballot_w64(cum_m <= PC_G_CAPM && cum_h <= PC_G_CAPH && zl + lane <= z_end));
            const int t_last = nfit - 1;                   // last window cell; home cells are th0 .. t_last-1
            const bool fits = t_last >= th0 + 1;
            const int z_next = fits ? zl + t_last : z0 + 1;
            {   // request the next window's table entries now
                const int zl2 = z_next > 0 ? z_next - 1 : 0;
                const int c = min(zl2 + lane, PC_NBB - 1);
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    const uint32_t *e = T + (tb[r] + (uint32_t)c);
                    a0[r] = e[0];
                    a1[r] = e[1];
                }
            }

            PC_G_MARK(0);
            if (fits) {
                const uint32_t tot_m = pc_lane_value(cum_m, t_last);
                const uint32_t tot_h = pc_lane_value(cum_h, t_last);       // row-0 points staged (home cells + cell above)
                const uint32_t nh = pc_lane_value(cum_h, t_last - 1);      // home particles in this window
                const uint32_t h_first = pc_lane_value(s[0], th0);         // global slot of the first staged row-0 point
                const uint32_t ex_m = cum_m - nm, ex_h = cum_h - n[0];
                const uint32_t lo_m = pc_from_lane_below(ex_m);            // start of cell t-1 in the merged array (0 for t = 0)
                const uint32_t hi_m = pc_from_lane_above(cum_m);           // end of cell t+1
                const uint32_t hi_h = pc_from_lane_above(cum_h);           // end of cell t+1 in the row-0 arrays
                done = pc_lane_value(s[0], t_last);                        // start of the last window cell = next first home cell
                if (nh != 0) {
                    // ---- merged copy of rows 1-4, in two stages.  Stage 1, lane t = cell t:
                    //      write the sorted-array slot of each of the cell's points to the
                    //      gather list, row after row (they land at ex_m + 0..nm-1).
                    //      A window of at most 16 cells (dense data: many points per cell, so each
                    //      of the loops below would run long on a few lanes) is spread over all
                    //      lanes instead: lane = (cell, row) writes that row's part of the cell.
                    const bool few_cells = t_last < 16;
                    if (few_cells) {
                        const int tc = lane >> 2, rr = lane & 3;
                        const uint32_t off[4] = {ex_m, ex_m + n[1], ex_m + n[1] + n[2], ex_m + n[1] + n[2] + n[3]};
                        uint32_t sv = 0, nv = 0, ov = 0;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const uint32_t a = __shfl(s[1 + r], tc, 64), b = __shfl(n[1 + r], tc, 64), c = __shfl(off[r], tc, 64);
                            if (rr == r) { sv = a; nv = b; ov = c; }
                            // one row's exchange at a time: all twelve in flight would cost the kernel a wave per SIMD in registers
                            asm volatile("" : "+v"(sv), "+v"(nv), "+v"(ov));
                        }
                        if (tc > t_last) nv = 0;
                        for (uint32_t v = 0; __builtin_amdgcn_ballot_w64(v < nv) != 0; v++)
                            if (v < nv) L.src[ov + v] = sv + v;
                        if (lane <= t_last && (nm_pts & 1)) L.src[ex_m + nm_pts] = home_lo;   // the pad slot copies any valid point ...
                    } else {
                        uint32_t o = ex_m;
#pragma unroll
                        for (int r = 1; r < 5; r++) {
                            const uint32_t nr = lane <= t_last ? n[r] : 0u;
                            for (uint32_t v = 0; __builtin_amdgcn_ballot_w64(v < nr) != 0; v++)
                                if (v < nr) L.src[o + v] = s[r] + v;
                            o += nr;
                        }
                        if (lane <= t_last && (nm_pts & 1)) L.src[o] = home_lo;   // the pad slot copies any valid point ...
                    }
                    pc_wave_lds_sync();
                    // ---- stage 2, lane = merged slot: gather the points through the list,
                    //      straight into LDS by the DMA path (no registers, no LDS stores)
                    for (uint32_t q0 = 0; q0 < tot_m; q0 += 64) {
                        const uint32_t q = q0 + lane;
                        const uint32_t src = q < tot_m ? L.src[q] : home_lo;
                        pc_dma_point(pts + 3 * (uint64_t)src, L.x + q0, L.y + q0, L.z + q0);
                    }
                    // (every lane has consumed its gather-list entries by now -- they were the DMA
                    // addresses -- so the list's space can take the row-0 data)
                    // ---- row 0: one contiguous run of at most PC_G_CAPH points
#pragma unroll
                    for (int qq = 0; qq < PC_G_CAPH / 64; qq++) {
                        const uint32_t e = lane + 64 * qq;
                        if (64u * qq < tot_h)
                            pc_dma_point(pts + 3 * (uint64_t)(e < tot_h ? h_first + e : home_lo),
                                         L.hx + 64 * qq, L.hy + 64 * qq, L.hz + 64 * qq);
                    }
                    // ---- per home particle: its merged candidate range and the end of its row-0 range
                    // (spreading these writes over four lanes per cell as well costs ten registers, i.e. the fifth wave per SIMD)
                    if (lane >= th0 && lane < t_last) {
                        for (uint32_t u = 0; u < n[0]; u++) L.meta[ex_h + u] = make_uint2(lo_m | (hi_m << 16), hi_h);
                    }
                    pc_dma_wait();
                    if (lane <= t_last && (nm_pts & 1)) L.x[ex_m + nm_pts] = __builtin_nanf("");   // ... and then never matches
                    pc_wave_lds_sync();
                    PC_G_MARK(1);
                    // ---- rows 1-4: candidate-parallel steps, group size from the mean range length
                    // candidates per pass (2*G*NP) should cover mean + 2 sigma of a Poisson range length:
                    // m + 2*sqrt(m) + 1 <= 8, 16, 32, 64, 128  <=>  m <= 3, 9, 22, 49, 106
                    // (mean = 3 * tot_m / cells, compared without dividing)
                    const uint32_t m3 = 3u * tot_m, cells = (uint32_t)(t_last + 1);
                    uint32_t hits;
                    if (m3 < 4 * cells) { hits = pc_g_window<2, 2>(L, nh, lane, d2max); PC_G_MARK(2); }
                    else if (m3 < 10 * cells) { hits = pc_g_window<2, 4>(L, nh, lane, d2max); PC_G_MARK(2); }
                    else if (m3 < 23 * cells) { hits = pc_g_window<2, 8>(L, nh, lane, d2max); PC_G_MARK(2); }
                    else if (m3 < 50 * cells) { hits = pc_g_window<4, 8>(L, nh, lane, d2max); PC_G_MARK(3); }
                    else if (m3 < 107 * cells) { hits = pc_g_window<8, 8>(L, nh, lane, d2max); PC_G_MARK(4); }
                    else {
                        // dense window: per home cell, candidates stay in registers while the cell's home
                        // particles stream by; row 0 and the merged ranges as one list (pc_g_joint)
                        hits = 0;
                        for (int t = th0; t < t_last; t++) {
                            const uint32_t n0c = pc_lane_value(n[0], t);
                            if (n0c == 0) continue;
                            const uint32_t hb = pc_lane_value(ex_h, t);
                            hits += pc_g_joint(L, hb, hb + n0c, min(pc_lane_value(hi_h, t), (uint32_t)PC_G_CAPH),
                                               pc_lane_value(lo_m, t), pc_lane_value(hi_m, t), lane, d2max);
                        }
                        PC_G_MARK(5);
                    }
                    if (lane == 0) acc += hits;
                    pc_wave_lds_sync();
                }
            } else {
                // ---- home cell z0 with its neighbours does not fit: stream the
                //      candidate ranges through LDS in pieces ---------------------
                const uint32_t h0 = pc_lane_value(s[0], th0);
                uint32_t h1 = h0 + pc_lane_value(n[0], th0);
                uint32_t c_lo[5], c_hi[5];
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    // rows 1-4: cells z0-1..z0+1 ; row 0: cells z0 (later slots) and z0+1
                    c_lo[r] = r == 0 ? h0 : pc_lane_value(s[r], 0);
                    const int te = min(th0 + 1, 63);
                    c_hi[r] = pc_lane_value(s[r], te) + pc_lane_value(n[r], te);
                }
                // too much for one wave: list the cell's home particles for pc_c3_heavy instead
                {
                    const uint32_t nhome = h1 - h0;
                    const uint32_t ncand = (c_hi[0] - c_lo[0]) + (c_hi[1] - c_lo[1]) + (c_hi[2] - c_lo[2]) +
                                           (c_hi[3] - c_lo[3]) + (c_hi[4] - c_lo[4]);
                    if ((unsigned long long)nhome * ncand >= PC_G_HEAVY_COST) {
                        const uint32_t chunks = (nhome + PC_G_HEAVY_CHUNK - 1) / PC_G_HEAVY_CHUNK;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(queue + PC_G_HEAVY_COUNT, chunks);
                        base = __builtin_amdgcn_readfirstlane(base);
                        const bool room = base + chunks <= heavy_cap && base + chunks >= base;
                        const uint32_t key = tb[0] + (uint32_t)z0;
                        // (an append that does not fit voids the slots it drew; the cell is then processed here)
                        for (uint32_t c = lane; c < chunks && base + c < heavy_cap; c += 64) {
                            queue[PC_G_HEAVY + 2 * (base + c)] = room ? key : 0xFFFFFFFFu;
                            queue[PC_G_HEAVY + 2 * (base + c) + 1] = c;
                        }
                        if (room) h1 = h0;                  // nothing left to do for this cell here
                    }
                }
                PC_G_MARK(7);
                if (h0 != h1) {
                    const uint32_t hits = pc_g_cell(L, pts, h0, h1, c_lo, c_hi, lane, d2max);
                    if (lane == 0) acc += hits;
                    PC_G_MARK(8);
                }
                done = h0 + pc_lane_value(n[0], th0);
            }
            z0 = z_next;
        }
        m = __builtin_amdgcn_readfirstlane(m_next);
    }
#ifdef PC_G_TRACE
    if (lane == 0)
        for (int i = 0; i < 32; i++) atomicAdd(&pc_g_trace[blockIdx.x % 64][i], tr_[i]);
#endif
    pc_block_add_u64<PC_G_WAVES>(acc, out);
}

// The listed home particles of over-full cells.  A workgroup takes one list item -- 256
// home particles of one cell, two per thread, in registers -- and streams the cell's
// candidates (its later slots and the cell above, then the three-cell runs of the four
// neighbour pencils: main.c:195-221) through an LDS tile of 256 points; every thread
// walks the whole tile, so the LDS reads are broadcasts and each serves two tests.
// Only the tile that holds the home particles themselves needs the "later slots only"
// rule (main.c:215-219) as a mask.  Missing home particles and tile padding are NaN
// points, which never match.
#define PC_HV_TPB 128
#define PC_HV_TILE 256
typedef float pc_hv4 __attribute__((ext_vector_type(4)));

// one candidate tile [t0, t0 + cnt) of the sorted array against the two home particles of every thread;
// MASKED: candidate slot q pairs with a home particle in slot p only if q > p (i0, i1 = p - t0)
template <bool MASKED>
__device__ __forceinline__ uint32_t pc_hv_tile(const float *__restrict__ pts, pc_hv4 *tile, uint32_t t0, uint32_t cnt,
                                               pc_v2f h0xy, float h0z, pc_v2f h1xy, float h1z, int i0, int i1, float d2max)
{
    struct pt3 { float x, y, z; };
    const pt3 *P = reinterpret_cast<const pt3 *>(pts);
    const float never = __builtin_nanf("");
    __syncthreads();                                   // the previous tile has been read
#pragma unroll
    for (int u = 0; u < PC_HV_TILE / PC_HV_TPB; u++) {
        const uint32_t e = threadIdx.x + u * PC_HV_TPB;
        pc_hv4 v = {never, never, never, never};
        if (e < cnt) { const pt3 c = P[t0 + e]; v.x = c.x; v.y = c.y; v.z = c.z; }
        tile[e] = v;
    }
    __syncthreads();
    uint32_t hits = 0;
    const int rounds = (int)((cnt + 3) & ~3u);         // (padding entries are NaN)
    for (int i = 0; i < rounds; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const pc_hv4 c = tile[i + u];               // the same address in every lane: a broadcast
            const pc_v2f cxy = {c.x, c.y};
            // (dx*dx + dy*dy) + dz*dz, each step rounded separately (no contraction: see build flags)
            const pc_v2f d0 = h0xy - cxy, d1 = h1xy - cxy;
            const float e0 = h0z - c.z, e1 = h1z - c.z;
            const pc_v2f q0 = d0 * d0, q1 = d1 * d1;
            const float s0 = (q0.x + q0.y) + e0 * e0, s1 = (q1.x + q1.y) + e1 * e1;
            unsigned long long b0 = __builtin_amdgcn_ballot_w64(s0 <= d2max), b1 = __builtin_amdgcn_ballot_w64(s1 <= d2max);
            if (MASKED) {
                b0 &= __builtin_amdgcn_ballot_w64(i + u > i0);
                b1 &= __builtin_amdgcn_ballot_w64(i + u > i1);
            }
            hits += __popcll(b0) + __popcll(b1);
        }
    }
    return hits;
}

__global__ void __launch_bounds__(PC_HV_TPB)
pc_c3_heavy(const float *__restrict__ pts, const uint32_t *__restrict__ T, const unsigned int *__restrict__ work,
            uint32_t heavy_cap, unsigned long long *__restrict__ out, const float d2max)
{
    static_assert(PC_G_HEAVY_CHUNK == 2 * PC_HV_TPB && PC_HV_TILE == PC_G_HEAVY_CHUNK, "two home particles per thread, home chunk = one tile");
    const uint32_t items = min(work[PC_G_HEAVY_COUNT], heavy_cap);
    if (items == 0 || work[PC_G_SERVED] != 0) return;
    __shared__ pc_hv4 tile[PC_HV_TILE];
    struct pt3 { float x, y, z; };
    const pt3 *P = reinterpret_cast<const pt3 *>(pts);
    const float never = __builtin_nanf("");
    unsigned long long acc = 0;            // wave-uniform hit counts: lane 0 of each wave carries them
    // The unit of work is (item, candidate tile): an item's tiles are dealt out over the whole grid -- tile t of
    // item i goes to workgroup (t + 7 i) mod gridDim -- so that a single over-full cell keeps every CU busy.
    for (uint32_t item = 0; item < items; item++) {
        const uint32_t key = work[PC_G_HEAVY + 2 * item], chunk = work[PC_G_HEAVY + 2 * item + 1];
        if (key == 0xFFFFFFFFu) continue;
        const uint32_t cell_hi = T[key + 1];
        const uint32_t h0 = T[key] + chunk * PC_G_HEAVY_CHUNK;         // first home slot of the item
        if (h0 >= cell_hi) continue;
        // the five candidate ranges and their tile counts: the same pencil from the item's own slots to the end of
        // the cell above (its first tile is the item itself), then the three-cell runs of the four neighbour pencils
        const uint32_t base[5] = {0u, 402u, 162005u, 162408u, 162811u};
        uint32_t lo[5], first_tile[6];
        first_tile[0] = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            lo[r] = r == 0 ? h0 : T[key + base[r]];
            const uint32_t hi = r == 0 ? T[key + 2] : T[key + base[r] + 3];
            first_tile[r + 1] = first_tile[r] + (hi - lo[r] + PC_HV_TILE - 1) / PC_HV_TILE;
        }
        uint32_t t = (blockIdx.x + gridDim.x - (7u * item) % gridDim.x) % gridDim.x;
        if (t >= first_tile[5]) continue;
        const uint32_t p0 = h0 + threadIdx.x, p1 = p0 + PC_HV_TPB;
        pt3 m0 = {never, never, never}, m1 = {never, never, never};
        if (p0 < cell_hi) m0 = P[p0];
        if (p1 < cell_hi) m1 = P[p1];
        const pc_v2f h0xy = {m0.x, m0.y}, h1xy = {m1.x, m1.y};
        uint32_t hits = 0;
        for (; t < first_tile[5]; t += gridDim.x) {
            int r = 0;
#pragma unroll
            for (int q = 1; q < 5; q++) r += t >= first_tile[q];
            uint32_t lo_r = lo[0], hi_r = T[key + 2], ft = 0;
#pragma unroll
            for (int q = 1; q < 5; q++)
                if (r == q) { lo_r = lo[q]; hi_r = T[key + base[q] + 3]; ft = first_tile[q]; }
            const uint32_t t0 = lo_r + (t - ft) * PC_HV_TILE, cnt = min((uint32_t)PC_HV_TILE, hi_r - t0);
            if (t == 0)
                hits += pc_hv_tile<true>(pts, tile, t0, cnt, h0xy, m0.z, h1xy, m1.z, (int)threadIdx.x, (int)threadIdx.x + PC_HV_TPB, d2max);
            else
                hits += pc_hv_tile<false>(pts, tile, t0, cnt, h0xy, m0.z, h1xy, m1.z, 0, 0, d2max);
        }
        if ((threadIdx.x & 63) == 0) acc += hits;
    }
    pc_block_add_u64<PC_HV_TPB / 64>(acc, out);
}
