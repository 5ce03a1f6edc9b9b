// Pair test, variant 5 ("merge"): the form for moderately filled grids (about
// half a point to a few tens of points per cell), where a window's bookkeeping
// and the copy into LDS, not the distance arithmetic, decide the time.
//
// Replaces counter_thread/compare/count_chunk/dist2 (reference main.c:85-91,
// 161-234).  One wavefront owns one home pencil (i,j) and walks it in windows of
// up to 62 home cells.  ALL FIVE pencils a home particle meets -- its own ("row
// 0") and (i,j+1) (i+1,j-1) (i+1,j) (i+1,j+1) ("rows 1-4": the offsets of
// main.c:195-209) -- are copied into ONE array in LDS, ordered in blocks
//
//     S_k = [ row 0, cell k | rows 1-4, cell k-1 ]
//
// so that everything a home particle in slot q of S_k has to be tested against
// lies in the single run (q, end of S_{k+2}): the later points of its own cell,
// rows 1-4 of cell k-1, row 0 of cell k+1, rows 1-4 of cell k, [row 0 of cell k+2],
// rows 1-4 of cell k+1.  The bracketed part is not among the 13 neighbour cells, and
// neither is anything a lane reads past the end of a run (blocks S_{k+3}...): such
// points lie two or more cells away along z and can never pass the predicate --
//
//   cells k and k+2 on one axis:  the reference's index (main.c:134-136) puts the
//   points at least 0.05 - 9e-15 apart on that axis (binary64 rounding of the two
//   quotients, at most 803 ulps of 0.05); their binary32 difference is then at least
//   0.05f = 0.0500000007..., its rounded square at least 0.00250000017... > 0.0025f,
//   and adding the other two (non-negative) squares with round-to-nearest cannot
//   bring the sum below that --
//
// so they may be tested freely: no range masks, only the loop's exit condition
// looks at the run's end.  Dead lanes and the slots after the last block hold NaN
// coordinates and never match.
//
// A slot is 16 bytes (x, y, z, one unused word): the layout a 12-byte LDS-DMA load
// writes, so the copy is ONE gather instruction per 64 points, driven by a list of
// byte offsets into the sorted array, and a candidate is ONE 16-byte LDS read,
// consecutive lanes on consecutive slots (no bank conflicts).  A group of G lanes
// takes one home particle, lane j the slots j, j+G, ... of its run; G follows the
// mean run length of the window.
//
// Windows this scheme does not suit (a cell with its neighbourhood exceeds the LDS
// array, or runs of more than about 130 points) are not processed here: the rest
// of the pencil goes on the segment list of variant 3's kernel, which runs after
// this one for the planned and the deferred segments only; if the list is full the
// wave counts the rest of the pencil itself, one lane per home particle, from global
// memory.  pc_set_variant("count", 5); chosen by density in the default variant 4.
//
// Build switches for measurements (none is set in the product build): PC_M_EXP_NOTEST compiles the tests out
// (what the bookkeeping and the copy cost alone), PC_M_EXP_PAD=<words> pads the LDS (fewer waves per SIMD),
// PC_M_PLAIN writes the distance with unpacked instructions only, PC_M_NO_QUADPERM takes the lanes' home
// particles in lane order, PC_M_CAP / PC_M_WAVES / PC_M_MAX_RUN / PC_M_G4_MAX / PC_M_UNROLL resize the kernel.
#pragma once
#include "pc_count_global.h"
#include "pc_count_group.h"

#ifndef PC_L_REV
#define PC_L_REV 2                      // (direction switches for measurements: see pc_bin_lds.h)
#endif
#ifndef PC_M_CAP
#define PC_M_CAP 376                    // slots of a window's blocks (even)
#endif
#define PC_M_CAPH 160                   // home particles of a window
#define PC_M_REACH 32                   // a lane's pass reads at most this many slots past the first slot of its group
#ifndef PC_M_SLACK
#define PC_M_SLACK 40                   // never-matching slots behind the last block (more than PC_M_REACH, at most 64)
#endif
#define PC_M_LEN (PC_M_CAP + PC_M_SLACK) // 416 slots = 6656 bytes: six waves per SIMD fit the CU's 160 KB
#define PC_M_HOME0 96                   // the list of home particles lies in the unused fourth words of slots 96 ...
#define PC_M_MAX_G 16
#ifndef PC_M_MAX_RUN
#define PC_M_MAX_RUN 130u                 // longest mean run (slots) this kernel takes
#endif
#ifndef PC_M_G4_MAX
#define PC_M_G4_MAX 28u                  // longest mean run (slots) that groups of four lanes take
#endif
#ifndef PC_M_UNROLL
#define PC_M_UNROLL 3                    // points of a (cell, row) written without a loop
#endif

// A window in LDS: one 16-byte slot per point -- the layout a 12-byte LDS-DMA load writes (lane t's three
// words land at byte 16 t of the destination, the fourth is left alone).
// The gather list -- one byte offset into the sorted array per slot -- lies over the start of the same array:
// the copy runs from the last chunk of 64 slots to the first, so a chunk's 1024 bytes never reach the
// 256-byte list pieces of the chunks still to come.
typedef float pc_v4f __attribute__((ext_vector_type(4)));
#define PC_M_SPARE (4 * (PC_M_HOME0 - 1))
static_assert(PC_M_CAP <= PC_M_SPARE, "the gather list must end before the spare word and the list of home particles");
static_assert(PC_M_HOME0 + PC_M_CAPH + 32 <= PC_M_LEN, "the list of home particles must fit the slot array");
static_assert(PC_M_SLACK > PC_M_REACH && PC_M_SLACK <= 64, "slack");

struct alignas(16) pc_merge_lds {
    union {
        pc_v4f pt[PC_M_LEN];
        uint32_t src[4 * PC_M_LEN];     // words 0 .. PC_M_CAP-1: the gather list; word PC_M_SPARE: writes of lanes with nothing to write
    };
    // entry h of the list of home particles (slot | end of its run << 16; behind the last one a step's worth of
    // dead entries): the fourth word of slot PC_M_HOME0 + h, which neither the copy nor the gather list touches
    __device__ __forceinline__ uint32_t &home(uint32_t h) { return src[4 * (PC_M_HOME0 + h) + 3]; }
    __device__ __forceinline__ const uint32_t &home(uint32_t h) const { return src[4 * (PC_M_HOME0 + h) + 3]; }
#ifdef PC_M_EXP_PAD
    uint32_t pad[PC_M_EXP_PAD];
#endif
};

// one point (12 bytes) per lane from global memory straight into LDS: lane t's x,y,z land in lds_chunk[t]
__device__ __forceinline__ void pc_m_dma_point(const void *g, pc_v4f *lds_chunk)
{
    __builtin_amdgcn_global_load_lds((pc_glb_void *)g, (pc_lds_void *)lds_chunk, 12, 0, 0);
}

template <int BYTES>
__device__ __forceinline__ pc_v4f pc_m_slot(uint32_t lds_byte_addr)
{
    pc_v4f v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "n"(BYTES));
    return v;
}
__device__ __forceinline__ void pc_m_arrived(pc_v4f &a)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory");
}

// slots c = 0..NS-1 of a lane: G*c past the lane's first; all requests go out before the first value is used
template <int G, int... C>
__device__ __forceinline__ void pc_m_request(uint32_t ap, pc_v4f *a, std::integer_sequence<int, C...>)
{
    ((a[C] = pc_m_slot<16 * G * C>(ap)), ...);
}

// NS slots of every lane against its group's home particle: hits of the whole wave
template <int G, int NS>
__device__ __forceinline__ uint32_t pc_m_test(uint32_t ap, pc_v2f hxy, float hz, float d2max)
{
    pc_v4f a[NS];
    float sq[NS];
    pc_m_request<G>(ap, a, std::make_integer_sequence<int, NS>());
#pragma unroll
    for (int k = 0; k < NS; k++) {
        pc_m_arrived(a[k]);
        // (dx*dx + dy*dy) + dz*dz, each step rounded separately (no contraction: see build flags)
#ifdef PC_M_PLAIN
        const float dx = hxy.x - a[k].x, dy = hxy.y - a[k].y, e = hz - a[k].z;
        sq[k] = (dx * dx + dy * dy) + e * e;
#else
        const pc_v2f d = hxy - pc_v2f{a[k].x, a[k].y};
        const float e = hz - a[k].z;
        const pc_v2f dd = d * d;
        sq[k] = (dd.x + dd.y) + e * e;
#endif
    }
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) hits += __popcll(__builtin_amdgcn_ballot_w64(sq[k] <= d2max));
    return hits;
}

// All nh home particles of the staged window, 64/G per step.
template <int G>
__device__ __forceinline__ uint32_t pc_m_steps(const pc_merge_lds &L, uint32_t nh, uint32_t tot, int lane, float d2max)
{
    constexpr int H = 64 / G;
    constexpr int NS0 = G <= 4 ? 8 : PC_M_REACH / G;   // slots per lane before the run's end is looked at
    constexpr int NST = G == 8 ? 4 : 2;                // and per turn of the loop after that
    static_assert(G * NS0 <= PC_M_REACH && G * NST <= PC_M_REACH, "reads must stay inside the never-matching slots");
    uint32_t hits = 0;
    asm volatile("" : "+v"(lane));                      // per-lane constants are recomputed per window, not kept in registers
    const uint32_t j = lane & (G - 1);
    uint32_t grp = lane / G;
#ifndef PC_M_NO_QUADPERM
    // A 16-byte LDS read is served in four groups of 16 lanes, quads (lane / 4) {0,3,5,6} {1,2,4,7} {8,11,13,14}
    // {9,10,12,15}; two lanes of a group collide when their slots lie a multiple of 16 apart.  Neighbouring
    // home particles sit a few slots apart, so the quads of one group take neighbouring home particles.
    if (G <= 4) {
        const uint32_t quad = lane >> 2;
        const uint32_t pq = ((0x73261540u >> (4 * (quad & 7))) & 7u) | (quad & 8u);
        grp = pq * (4 / G) + ((lane / G) & (4 / G - 1));
    }
#endif
    nh = __builtin_amdgcn_readfirstlane(nh);
    tot = __builtin_amdgcn_readfirstlane(tot);
    const uint32_t pt_lds = pc_lds_addr(L.pt);
    for (uint32_t base = 0; base < nh; base += H) {
        const uint32_t e = L.home(base + grp);           // past nh: dead entries (a NaN slot, an empty run)
        const uint32_t q = e & 0xFFFFu;
        const pc_v4f h = L.pt[q];
        const pc_v2f hxy = {h.x, h.y};
        const float hz = h.z;
        uint32_t p = q + 1 + j;                          // the lane's first slot
        int rem = (int)(e >> 16) - (int)p;                // slots from there to the end of the run
        hits += pc_m_test<G, NS0>(pt_lds + 16 * p, hxy, hz, d2max);
        rem -= G * NS0;
        p += G * NS0;
        while (__builtin_amdgcn_ballot_w64(rem > 0) != 0) {
            p = min(p, tot);                              // finished lanes wait on the never-matching slots
            hits += pc_m_test<G, NST>(pt_lds + 16 * p, hxy, hz, d2max);
            rem -= NST * G;
            p += NST * G;
        }
    }
    return hits;    // wave-uniform
}

// The rest [z0, 400) of a pencil that this kernel does not process: onto the segment
// list of pc_c3_count_groups if there is room, else counted here and now.
__device__ __forceinline__ unsigned long long pc_m_defer(const float *__restrict__ pts, const uint32_t *__restrict__ T,
                                                         unsigned int *__restrict__ queue, uint32_t seg_cap,
                                                         uint32_t pencil, int z0, uint32_t home_hi, int lane, const pc_geom &geo,
                                                         uint32_t dense_min)
{
    uint32_t at = 0xFFFFFFFFu;
    if (lane == 0 && seg_cap != 0) {
        at = atomicAdd(queue + PC_G_SEG_COUNT, 1u);
        if (at < seg_cap) {
            queue[PC_G_SEGS + 2 * at] = pencil + 1;
            queue[PC_G_SEGS + 2 * at + 1] = (uint32_t)z0 | ((uint32_t)PC_NLAYERS << 16);
            if (dense_min != 0) {                          // the rest's points, to the sum of its kind (see pc_segments_go_to)
                const uint32_t pts_seg = home_hi - T[(uint64_t)pencil * PC_NBB + z0];
                atomicAdd(queue + (pc_segment_is_dense(pts_seg, PC_NLAYERS - z0, dense_min) ? PC_G_SEG_POINTS_DENSE : PC_G_SEG_POINTS_THIN), pts_seg);
            }
        }
    }
    at = __builtin_amdgcn_readfirstlane(at);
    if (at < seg_cap) return 0;
    unsigned long long acc = 0;
    for (uint64_t p = (uint64_t)T[(uint64_t)pencil * PC_NBB + z0] + lane; p < home_hi; p += 64)
        acc += pc_count_one_home(pts, T, (uint32_t)p, geo);
    return acc;
}

#ifndef PC_M_WAVES
#define PC_M_WAVES 6                    // per SIMD: what the slot array and 80 registers allow
#endif
__global__ void __launch_bounds__(64, PC_M_WAVES)
pc_c5_count_merged(const float *__restrict__ pts, const uint32_t *__restrict__ T, int x_lo, int x_hi,
                   unsigned int *__restrict__ queue, uint32_t seg_cap, int gate, uint32_t gate_limit,
                   unsigned long long *__restrict__ out, const pc_geom geo, uint32_t dense_min)
{
    if (!pc_gate_open(queue, gate, gate_limit)) return;
    // (every home pencil cut into segments by the planning kernel: nothing is left for this kernel)
    if (__builtin_amdgcn_readfirstlane(queue[PC_G_PLANNED_PENCILS]) == (uint32_t)(x_hi - x_lo) * PC_NLAYERS) return;
    __shared__ pc_merge_lds L;
    const int lane = threadIdx.x;
    unsigned long long acc = 0;          // per-lane partial (lane 0 also carries the wave-uniform hit counts)

    // Persistent waves, one work counter per XCD group (workgroups with equal
    // blockIdx%8 share an XCD): the group walks the band of 50 pencils
    // y = g*50 .. g*50+49 layer by layer, so neighbouring pencils are in flight
    // together and stay in that XCD's L2; a wave whose band is used up moves on to
    // the next group's band and leaves once all eight counters are past the end.
    int g = blockIdx.x & 7, bands_done = 0;
    const int band_pencils = (x_hi - x_lo) * PC_G_BAND;
    const unsigned char *planned = reinterpret_cast<const unsigned char *>(queue + PC_G_FLAGS);
    int m = 0;
    if (lane == 0) m = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE, 1u);
    m = __builtin_amdgcn_readfirstlane(m);
    for (;;) {
        while (m >= band_pencils && ++bands_done < 8) {
            g = (g + 1) & 7;
            if (lane == 0) m = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE, 1u);
            m = __builtin_amdgcn_readfirstlane(m);
        }
        if (m >= band_pencils) break;
        int m_next = 0;
        if (lane == 0) m_next = (int)atomicAdd(queue + g * PC_G_QUEUE_STRIDE, 1u);   // (its latency hides behind the pencil)
        const int layer = (PC_L_REV & 4) ? x_hi - 1 - m / PC_G_BAND : x_lo + m / PC_G_BAND, j = g * PC_G_BAND + m % PC_G_BAND;
        const uint32_t pencil = (uint32_t)layer * PC_NBB + j;
        const uint32_t skip = planned[pencil];            // cut into segments by the planning kernel: variant 3's kernel takes it

        // table bases of the five rows; a row outside the occupied grid is empty (its base only has to be readable at -1)
        const int ri[5] = {layer, layer, layer + 1, layer + 1, layer + 1};
        const int rj[5] = {j, j + 1, j - 1, j, j + 1};
        uint32_t tb[5];
        bool valid[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            valid[r] = ri[r] < PC_NLAYERS && rj[r] >= 0 && rj[r] < PC_NLAYERS;
            tb[r] = valid[r] ? ((uint32_t)ri[r] * PC_NBB + rj[r]) * PC_NBB : tb[0] + PC_NBB;
        }
        const uint32_t home_lo = T[tb[0]], home_hi = T[tb[0] + PC_NLAYERS];
        int z0 = 0;
        uint32_t done = skip ? home_hi : home_lo;          // first home slot not yet processed
        // The gather list holds 32-bit byte offsets from the pencil's first point; every point a window gathers lies
        // between it and the end of layer i+1.  Should those two layers hold more than 4 GB of points (an input of
        // over 3.5e8 points nearly all of which sit in two layers), the pencil goes to the segment list as a whole.
        // (the last table entry of layer i+1: its always-empty padding pencil starts where the layer's points end)
        const uint32_t reach = T[(uint32_t)(layer + 2) * PC_NBB2 - 1u] - home_lo;
        if (!skip && done != home_hi && reach > 0xFFFFFFFFu / 12u) {
            acc += pc_m_defer(pts, T, queue, seg_cap, pencil, 0, home_hi, lane, geo, dense_min);
            done = home_hi;
        }
        // Lane t of a window starting at home cell z0 owns block S_{z0+t}: row 0 at cell z0+t, rows 1-4 at
        // cell z0+t-1.  (Cell -1 of a row reads the last, always empty, padding cell of the pencil before it;
        // cells past 402 are clamped to 402: empty as well.)  a0/a1 = start of that cell and of the next one.
        uint32_t a0[5], a1[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const int c = min(lane - (r ? 1 : 0), PC_NBB - 1);
            const uint32_t *e = T + ((int64_t)tb[r] + c);
            a0[r] = e[0];
            a1[r] = e[1];
        }
        while (done != home_hi && z0 < PC_NLAYERS) {
            uint32_t s[5], n[5];
#pragma unroll
            for (int r = 0; r < 5; r++) {
                s[r] = a0[r];
                n[r] = valid[r] ? a1[r] - a0[r] : 0u;
            }
            const uint32_t n_blk = n[0] + n[1] + n[2] + n[3] + n[4];
            const uint32_t cum = pc_wave_incl_scan(n_blk), cumh = pc_wave_incl_scan(n[0]);
            // blocks that fit (a prefix); the last two only supply candidates
            const int nblk = __popcll(__builtin_amdgcn_ballot_w64(cum <= PC_M_CAP && cumh <= PC_M_CAPH));
            const bool fits = nblk >= 3;
            const int w = nblk - 2;                         // home cells z0 .. z0+w-1
            const int z_next = z0 + w;
            if (fits) {   // request the next window's table entries now: they arrive while this one is processed
#pragma unroll
                for (int r = 0; r < 5; r++) {
                    const int c = min(z_next + lane - (r ? 1 : 0), PC_NBB - 1);
                    const uint32_t *e = T + ((int64_t)tb[r] + c);
                    a0[r] = e[0];
                    a1[r] = e[1];
                }
            }
            uint32_t tot = 0, nh = 0;
            if (fits) {
                tot = pc_lane_value(cum, nblk - 1);
                nh = pc_lane_value(cumh, w - 1);
            }
            // run lengths: three blocks on average; beyond about 130 slots variant 3's kernel does better
            const bool suited = fits && 3u * tot <= PC_M_MAX_RUN * (uint32_t)nblk;
            if (!suited) {
                const unsigned long long slow = pc_m_defer(pts, T, queue, seg_cap, pencil, z0, home_hi, lane, geo, dense_min);
                acc += slow;
                break;
            }
            done = pc_lane_value(s[0], w);                 // start of the first home cell of the next window
            if (nh != 0) {
                if (lane >= nblk) { n[0] = 0; n[1] = 0; n[2] = 0; n[3] = 0; n[4] = 0; }
                const uint32_t ex = cum - n_blk, exh = cumh - n[0];
                // end of S_{t+2}: the inclusive sum two lanes up
                const uint32_t end2 = pc_from_lane_above(pc_from_lane_above(cum));
                // ---- stage 1, lane = block: byte offsets of the block's points, row after row; home entries
                // (branch-free for the first PC_M_UNROLL points of a row: lanes that have none write to a spare word)
                {
                    const uint32_t nhome = lane < w ? n[0] : 0u;
                    const uint32_t ent = ex | (end2 << 16);
                    const uint32_t spare = PC_M_SPARE;
#pragma unroll
                    for (uint32_t v = 0; v < PC_M_UNROLL; v++) {
                        L.src[v < n[0] ? ex + v : spare] = 12u * (s[0] - home_lo + v);
                        L.home(v < nhome ? exh + v : PC_M_CAPH + 31) = ent + v;
                    }
                    for (uint32_t v = PC_M_UNROLL; __builtin_amdgcn_ballot_w64(v < n[0]) != 0; v++) {
                        if (v < n[0]) L.src[ex + v] = 12u * (s[0] - home_lo + v);
                        if (v < nhome) L.home(exh + v) = ent + v;
                    }
                    uint32_t o = ex + n[0];
#pragma unroll
                    for (int r = 1; r < 5; r++) {
#pragma unroll
                        for (uint32_t v = 0; v < PC_M_UNROLL; v++) L.src[v < n[r] ? o + v : spare] = 12u * (s[r] - home_lo + v);
                        for (uint32_t v = PC_M_UNROLL; __builtin_amdgcn_ballot_w64(v < n[r]) != 0; v++)
                            if (v < n[r]) L.src[o + v] = 12u * (s[r] - home_lo + v);
                        o += n[r];
                    }
                    // one step's worth of dead home entries: a slot of the never-matching tail, an empty run
                    if (lane < 32) L.home(nh + lane) = tot | (tot << 16);
                }
                pc_wave_lds_sync();
                // ---- stage 2, lane = slot: gather the points through the list, straight into LDS (no registers)
                const char *bytes = reinterpret_cast<const char *>(pts) + 12ull * home_lo;   // (wave-uniform: scalar registers)
                // (all list reads first, then the copies, last chunk first: see pc_merge_lds)
                constexpr int NCHUNK = (PC_M_CAP + 63) / 64;
                uint32_t off[NCHUNK];
#pragma unroll
                for (int c = 0; c < NCHUNK; c++) {
                    const uint32_t q = 64u * c + lane;
                    off[c] = q < tot ? L.src[q] : 0u;
                }
#pragma unroll
                for (int c = NCHUNK - 1; c >= 0; c--)
                    if (64u * c < tot) pc_m_dma_point(bytes + off[c], L.pt + 64 * c);
                pc_dma_wait();
                const float never = __builtin_nanf("");
                if (lane < PC_M_SLACK) L.pt[tot + lane].x = never;
                pc_wave_lds_sync();
                // ---- the tests: group width from the mean run length (three blocks)
                const uint32_t m3 = 3u * tot, blocks = (uint32_t)nblk;
                uint32_t hits;
#ifdef PC_M_EXP_NOTEST
                hits = 0;
                if (m3 == 0xFFFFFFFFu)
#endif
                {
                    if (m3 <= 12u * blocks) hits = pc_m_steps<2>(L, nh, tot, lane, geo.d2max);
                    else if (m3 <= PC_M_G4_MAX * blocks) hits = pc_m_steps<4>(L, nh, tot, lane, geo.d2max);
                    else if (m3 <= 60u * blocks) hits = pc_m_steps<8>(L, nh, tot, lane, geo.d2max);
                    else hits = pc_m_steps<16>(L, nh, tot, lane, geo.d2max);
                }
                if (lane == 0) acc += hits;
                pc_wave_lds_sync();
            }
            z0 = z_next;
        }
        m = __builtin_amdgcn_readfirstlane(m_next);
    }
    pc_block_add_u64<1>(acc, out);
}
