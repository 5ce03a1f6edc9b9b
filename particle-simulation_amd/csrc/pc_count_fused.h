// Pair test, variant 7 ("fused"): level Z of the binning and the pair test in ONE
// kernel, for moderately filled grids (about 2/3 of a point to 2 points per cell).
//
// Replaces counter_thread/compare/count_chunk/dist2 (reference main.c:85-91,
// 161-234) and, for the callers that only want the count, the last level of
// insert/atomic_insert/flatten (hashmap.h:69-153): the input is the output of
// binning level Y -- points grouped by pencil (i,j), in no order along z -- and
// neither the z-sorted array nor the cell-start table is ever written.
//
// One workgroup owns one home pencil (i,j).  The five pencils a home particle
// meets -- (i,j) "row 0" and (i,j+1) (i+1,j-1) (i+1,j) (i+1,j+1) "rows 1-4", the
// offsets of main.c:195-209 -- are TWO contiguous ranges of the level-Y array
// (rows 0-1 and rows 2-4).  Every thread loads a few of their points (coalesced
// 12-byte loads) and keeps them in registers, computes the z cell, and the
// workgroup counting-sorts them into the order
//
//     S_k = [ row 0, cell k | rows 1-4, cell k-1 ]        k = 0 .. 403
//
// (two bins per block: an LDS atomic gives each point its rank, one wave scans
// the 808 counts).  As in the merged-window kernel (pc_count_merge.h, where the
// argument is written out) everything a home particle in slot q of S_k has to be
// tested against is then the single run (q, end of S_{k+2}); what the run holds
// besides the 13 neighbour cells -- row 0 of cell k+2 -- and whatever a lane reads
// past its end lies two or more cells away along z and can never pass the
// predicate, so the tests need no range masks.  All bookkeeping is lane = point:
// no table reads, no gather lists, no per-window prefix sums.
//
// The sorted order is materialised in LDS one CHUNK of home cells at a time (the
// blocks of the chunk and the two behind it, at most SLOTS points; the cuts are
// read off the scanned counts), so that several workgroups share a CU's LDS: the
// points stay in their registers from chunk to chunk.  Per chunk, the home
// particles are sorted by the length of their runs (a 64-bucket counting sort,
// longest first) and handed out sixteen at a time: four lanes per home particle,
// lane j taking slots j, j+4, ... of the run.  The sixteen runs of a step have
// (almost) equal lengths, so the step ends when all of them end and its pass
// count is wave-uniform: a pass is one 16-byte LDS read and seven vector
// instructions per lane.
//
// What does not fit: a pencil neighbourhood of more than PC_F_CAP points (the
// registers of a workgroup) -- the binning's level-Y scan records the largest one
// of the slab, and when it exceeds the capacity this kernel returns at once and
// the z sort and the classic pair-test kernels run instead (`fgate`); and three
// consecutive blocks of more than SLOTS points (many points in a few cells of an
// otherwise thin pencil) -- such a pencil goes on a list that a second
// instantiation with the whole capacity in LDS works off (pc_c7_count_fused_listed).
#pragma once
#include "pc_bin_lds.h"
#include "pc_count_merge.h"

#ifndef PC_F_TPB
#define PC_F_TPB 512
#endif
#ifndef PC_F_PPT
#define PC_F_PPT 8                                  // points a thread keeps in registers
#endif
#define PC_F_CAP (PC_F_TPB * PC_F_PPT)              // 4096 points of the five pencils
#ifndef PC_F_SLOTS
#define PC_F_SLOTS 2304                             // points of a chunk in LDS: three workgroups per CU
#endif
#define PC_F_PAD 72                                 // never-matching slots behind the last block
#define PC_F_NBIN 832                               // two per block S_0 .. S_403 and the (empty) ones a run's end may index: 13 per lane of the scanning wave
#define PC_F_TOTBIN 808                             // first bin behind the blocks: its start is the number of points kept
#define PC_F_LBINS 64                               // buckets of the sort by run length
#define PC_F_G 4                                    // lanes per home particle
#ifndef PC_F_NS
#define PC_F_NS 2                                   // slots per lane whose reads are in flight together
#endif
#define PC_F_WAVES (PC_F_TPB / 64)
#ifndef PC_F_WAVES_PER_SIMD
#define PC_F_WAVES_PER_SIMD 6                       // three workgroups per CU: 80 registers
#endif

template <int SLOTS>
struct alignas(16) pc_fused_lds {
    pc_v4f pt[SLOTS + PC_F_PAD];                    // x, y, z, and for a home particle the end of its run
    uint32_t start[PC_F_NBIN];                      // counts, then exclusive starts, of the bins (2 k: row 0 cell k; 2 k + 1: rows 1-4 cell k-1)
    uint32_t lcount[2][PC_F_LBINS], lstart[PC_F_LBINS];
    uint16_t home[SLOTS];                           // slots of the chunk's home particles, longest run first
};

// wave-wide maximum (DPP row operations; the last lane ends up with it)
__device__ __forceinline__ uint32_t pc_wave_max(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// NS slots of every lane (PC_F_G apart) against its group's home particle; hits of the lanes in `live`
template <int NS, bool CLAMP>
__device__ __forceinline__ uint32_t pc_f_test(uint32_t pt_lds, uint32_t p, uint32_t last, pc_v2f hxy, float hz, float d2max,
                                              unsigned long long live)
{
    pc_v4f a[NS];
    float sq[NS];
    if (CLAMP) {
#pragma unroll
        for (int k = 0; k < NS; k++) a[k] = pc_m_slot<0>(pt_lds + 16u * min(p + PC_F_G * k, last));
    } else {
        pc_m_request<PC_F_G>(pt_lds + 16u * p, a, std::make_integer_sequence<int, NS>());
    }
#pragma unroll
    for (int k = 0; k < NS; k++) {
        pc_m_arrived(a[k]);
        // (dx*dx + dy*dy) + dz*dz, each step rounded separately (no contraction: see build flags)
        const pc_v2f d = hxy - pc_v2f{a[k].x, a[k].y};
        const float e = hz - a[k].z;
        const pc_v2f dd = d * d;
        sq[k] = (dd.x + dd.y) + e * e;
    }
    uint32_t hits = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) hits += __popcll(__builtin_amdgcn_ballot_w64(sq[k] <= d2max) & live);
    return hits;
}

// One home pencil p = (layer, j), by the whole workgroup.  Returns false if the pencil needs more LDS than SLOTS
// (nothing has been counted then).  hits: wave-uniform; tests: per lane (wave 7 only), if asked for.
template <int SLOTS>
__device__ __forceinline__ bool pc_f_pencil(pc_fused_lds<SLOTS> &L, const float *__restrict__ pencils,
                                            const uint32_t *__restrict__ pencil_start, uint32_t p, bool want_tests,
                                            uint32_t &hits, unsigned long long &tests, bool &bad, const pc_geom &geo)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // rows 0-1 = pencils p, p+1; rows 2-4 = pencils p+402 .. p+404 (for j = 0 the first is the empty padding pencil 402 of layer i)
    const uint32_t a0 = pencil_start[p], a1 = pencil_start[p + 1], a2 = pencil_start[p + 2];
    const uint32_t b0 = pencil_start[p + PC_NBB - 1], b3 = pencil_start[p + PC_NBB + 2];
    if (a1 == a0) return true;                       // no home particles
    const uint32_t nA = a2 - a0, n = nA + (b3 - b0);  // (n <= PC_F_CAP: the gate)

    for (int b = tid; b < PC_F_NBIN; b += PC_F_TPB) L.start[b] = 0;
    if (tid < 2 * PC_F_LBINS) L.lcount[0][tid] = 0;
    // ---- the points -> registers (all loads before any use)
    const pc_pt3 *G = reinterpret_cast<const pc_pt3 *>(pencils);
    pc_pt3 pt[PC_F_PPT];
    uint32_t tag[PC_F_PPT];
#pragma unroll
    for (int q = 0; q < PC_F_PPT; q++) {
        const uint32_t v = min((uint32_t)(tid + q * PC_F_TPB), n - 1);
        const uint32_t idx = v < nA ? a0 + v : b0 + (v - nA);
        pt[q] = G[idx];
        tag[q] = idx;
    }
    __syncthreads();
    // ---- bin (2 per block) of every point, then its rank inside the bin: all atomics go out before the first answer is used
#pragma unroll
    for (int q = 0; q < PC_F_PPT; q++) {
        const uint32_t idx = tag[q];
        tag[q] = 0xFFFFFFFFu;
        if ((uint32_t)(tid + q * PC_F_TPB) < n) {
            const bool row0 = (uint32_t)(tid + q * PC_F_TPB) < nA && idx < a1;
            const int k = pc_cell_index(pt[q].z, geo);
            if ((unsigned)k < (unsigned)geo.ncell) tag[q] = row0 ? 2u * (uint32_t)k : 2u * (uint32_t)k + 3u;      // rows 1-4, cell k: block k+1
            else bad = true;
        }
    }
#ifdef PC_F_EXP_NOATOMIC
    return true;
#endif
    uint32_t rank[PC_F_PPT];
#pragma unroll
    for (int q = 0; q < PC_F_PPT; q++) {
        rank[q] = 0;
        if (tag[q] != 0xFFFFFFFFu) rank[q] = atomicAdd(&L.start[tag[q]], 1u);
    }
#pragma unroll
    for (int q = 0; q < PC_F_PPT; q++)
        if (tag[q] != 0xFFFFFFFFu) tag[q] |= rank[q] << 10;
    __syncthreads();
    // ---- exclusive scan of the bins by wave 0 (13 per lane)
    if (wave == 0) {
        uint32_t c[13], s = 0;
#pragma unroll
        for (int q = 0; q < 13; q++) {
            c[q] = L.start[lane * 13 + q];
            s += c[q];
        }
        uint32_t run = pc_wave_incl_scan(s) - s;
#pragma unroll
        for (int q = 0; q < 13; q++) {
            L.start[lane * 13 + q] = run;
            run += c[q];
        }
    }
    __syncthreads();
    if (want_tests && wave == PC_F_WAVES - 1) {
        // reference main.c:210-229 per home cell k: n(n-1)/2 + n * (row 0 cell k+1 + rows 1-4 cells k-1..k+1)
        for (int k = lane; k < PC_NLAYERS; k += 64) {
            const uint32_t s0 = L.start[2 * k], s1 = L.start[2 * k + 1], s4 = L.start[2 * k + 4], s5 = L.start[2 * k + 5],
                           s6 = L.start[2 * k + 6];
            const unsigned long long n0 = s1 - s0;
            tests += n0 * (n0 - 1) / 2 + n0 * ((s4 - s1) + (s6 - s5));
        }
    }
    const uint32_t jj = lane & (PC_F_G - 1), grp = lane / PC_F_G;
    const uint32_t pt_lds = pc_lds_addr(L.pt);
    const uint32_t kept = L.start[PC_F_TOTBIN];

    // ---- chunk after chunk of home cells [k0, k1): blocks k0 .. k1+1 in LDS
    int k0 = 0, chunk = 0;
    while (k0 < PC_NLAYERS) {
        const uint32_t s_lo = L.start[2 * k0];
        if (s_lo == kept) break;                      // nothing from here on
        // k1: the last home cell whose run still ends inside SLOTS points (every wave finds it for itself)
        int k1 = k0;
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int k = k0 + 1 + lane + 64 * t;
            const bool fits = k <= PC_NLAYERS && L.start[2 * min(k, PC_NLAYERS) + 4] - s_lo <= (uint32_t)SLOTS;
            k1 += __popcll(__builtin_amdgcn_ballot_w64(fits));
        }
        if (k1 == k0) return false;                   // blocks k0 .. k0+2 alone exceed the chunk: the listed kernel's case
#ifdef PC_F_EXP_NOSCATTER                             // measurement builds (none is set in the product build)
        k0 = k1;
        continue;
#endif
        const uint32_t s_hi = L.start[2 * k1 + 4];
        uint32_t *lcount = L.lcount[chunk & 1];
        // the chunk's points in S order -> LDS; a home particle also takes its run's end along and is counted by run length
        uint32_t hometag[PC_F_PPT];
#pragma unroll
        for (int q = 0; q < PC_F_PPT; q++) {
            hometag[q] = 0xFFFFFFFFu;
            if (tag[q] != 0xFFFFFFFFu) {
                const uint32_t bin = tag[q] & 1023u, slot = L.start[bin] + (tag[q] >> 10);
                if (slot >= s_lo && slot < s_hi) {
                    uint32_t end = 0;
                    if ((bin & 1u) == 0 && bin < 2u * (uint32_t)k1) {
                        end = L.start[bin + 6] - s_lo;                    // start of S_{k+3}
                        const uint32_t len = end - (slot - s_lo) - 1u;
                        const uint32_t bucket = (PC_F_LBINS - 1) - min(len, (uint32_t)PC_F_LBINS - 1);
                        const uint32_t r = atomicAdd(&lcount[bucket], 1u);
                        hometag[q] = (slot - s_lo) | (bucket << 13) | (r << 19);
                    }
                    L.pt[slot - s_lo] = pc_v4f{pt[q].x, pt[q].y, pt[q].z, __uint_as_float(end)};
                }
            }
        }
        const uint32_t tot = s_hi - s_lo;
        if (tid < PC_F_PAD) {
            const float never = __builtin_nanf("");
            L.pt[tot + tid] = pc_v4f{never, never, never, never};
        }
        __syncthreads();
        // every wave scans the 64 bucket counts for itself (all write the same starts); the other set of counters is cleared for the next chunk
        uint32_t nh;
        {
            const uint32_t c = lcount[lane];
            const uint32_t incl = pc_wave_incl_scan(c);
            L.lstart[lane] = incl - c;
            nh = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (wave == 0) L.lcount[(chunk & 1) ^ 1][lane] = 0;
            pc_wave_lds_sync();
#pragma unroll
            for (int q = 0; q < PC_F_PPT; q++)
                if (hometag[q] != 0xFFFFFFFFu)
                    L.home[L.lstart[(hometag[q] >> 13) & 63u] + (hometag[q] >> 19)] = (uint16_t)(hometag[q] & 8191u);
        }
        __syncthreads();
#ifdef PC_F_EXP_NOTEST
        nh = 0;
#endif
        // the tests: sixteen home particles per step and wave, four lanes each
        for (uint32_t base = wave * (64 / PC_F_G); base < nh; base += PC_F_WAVES * (64 / PC_F_G)) {
            const uint32_t h = base + grp;
            const bool live_lane = h < nh;
            const uint32_t q = L.home[min(h, nh - 1)];
            const pc_v4f hv = L.pt[q];
            const pc_v2f hxy = {hv.x, hv.y};
            const float hz = hv.z;
            const uint32_t len = live_lane ? __float_as_uint(hv.w) - q - 1u : 0u;
            const unsigned long long live = __builtin_amdgcn_ballot_w64(live_lane);
            // longest run of the step: the first one's, unless it lies in the open-ended bucket
            uint32_t lmax = __builtin_amdgcn_readfirstlane(len);
            if (lmax >= PC_F_LBINS - 1) lmax = pc_wave_max(len);
            int passes = (int)((lmax + PC_F_G - 1) / PC_F_G);
            uint32_t s = q + 1 + jj;                       // the lane's first slot
            if (lmax + 2 * PC_F_G + 2 < PC_F_PAD) {
                // lanes whose runs are shorter read on into later blocks or the never-matching tail: at most lmax + 6 slots past the end
                for (; passes >= PC_F_NS; passes -= PC_F_NS, s += PC_F_NS * PC_F_G) hits += pc_f_test<PC_F_NS, false>(pt_lds, s, tot, hxy, hz, geo.d2max, live);
                for (; passes > 0; passes -= 2, s += 2 * PC_F_G) hits += pc_f_test<2, false>(pt_lds, s, tot, hxy, hz, geo.d2max, live);
            } else {
                for (; passes > 0; passes -= 2, s += 2 * PC_F_G) hits += pc_f_test<2, true>(pt_lds, s, tot, hxy, hz, geo.d2max, live);
            }
        }
        k0 = k1;
        chunk++;
        if (k0 < PC_NLAYERS) __syncthreads();         // the next chunk overwrites the slots and the list
    }
    return true;
}

__global__ void __launch_bounds__(PC_F_TPB, PC_F_WAVES_PER_SIMD)
pc_c7_count_fused(const float *__restrict__ pencils, const uint32_t *__restrict__ pencil_start, int x_lo, int x_hi,
                  const uint32_t *__restrict__ fgate, uint32_t *__restrict__ listed_n, uint32_t *__restrict__ listed,
                  unsigned long long *__restrict__ out_pairs, unsigned long long *__restrict__ out_tests,
                  int32_t *__restrict__ status, const pc_geom geo)
{
    if (!pc_fused_serves(fgate)) return;             // some neighbourhood of the slab does not fit: the classic kernels run
    __shared__ pc_fused_lds<PC_F_SLOTS> L;
    // Workgroups with equal blockIdx % 8 share an XCD (round-robin dispatch: speed only): group g walks the band of
    // 50 pencils y = 50 g .. 50 g + 49 layer by layer, so the four neighbour pencils are hits in that XCD's L2.
    const int g = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int layer = x_lo + m / PC_G_BAND, j = g * PC_G_BAND + m % PC_G_BAND;
    if (layer >= x_hi) return;
    const uint32_t p = (uint32_t)layer * PC_NBB + j;
    uint32_t hits = 0;
    unsigned long long tests = 0;
    bool bad = false;
    if (!pc_f_pencil<PC_F_SLOTS>(L, pencils, pencil_start, p, out_tests != nullptr, hits, tests, bad, geo)) {
        if (threadIdx.x == 0) listed[atomicAdd(listed_n, 1u)] = p;
        return;
    }
    if (bad) atomicOr(status, 1);
    // hits is wave-uniform: lane 0 of every wave carries it into the workgroup's sum
    unsigned long long acc = (threadIdx.x & 63) == 0 ? (unsigned long long)hits : 0ull;
    pc_block_add_u64<PC_F_WAVES>(acc, out_pairs);
    if (out_tests != nullptr) {
        __syncthreads();                             // (the two sums share the reduction's LDS words)
        pc_block_add_u64<PC_F_WAVES>(tests, out_tests);
    }
}

// The pencils the kernel above listed: the same work with the whole capacity in LDS (one chunk always fits).
__global__ void __launch_bounds__(PC_F_TPB)
pc_c7_count_fused_listed(const float *__restrict__ pencils, const uint32_t *__restrict__ pencil_start,
                         const uint32_t *__restrict__ fgate, const uint32_t *__restrict__ listed_n,
                         const uint32_t *__restrict__ listed, unsigned long long *__restrict__ out_pairs,
                         unsigned long long *__restrict__ out_tests, int32_t *__restrict__ status, const pc_geom geo)
{
    if (!pc_fused_serves(fgate)) return;
    const uint32_t entries = *listed_n;
    if (entries == 0) return;
    __shared__ pc_fused_lds<PC_F_CAP> L;
    uint32_t hits = 0;
    unsigned long long tests = 0;
    bool bad = false;
    for (uint32_t e = blockIdx.x; e < entries; e += gridDim.x) {
        __syncthreads();                             // the previous pencil's LDS reads are done
        pc_f_pencil<PC_F_CAP>(L, pencils, pencil_start, listed[e], out_tests != nullptr, hits, tests, bad, geo);
    }
    if (bad) atomicOr(status, 1);
    unsigned long long acc = (threadIdx.x & 63) == 0 ? (unsigned long long)hits : 0ull;
    pc_block_add_u64<PC_F_WAVES>(acc, out_pairs);
    if (out_tests != nullptr) {
        __syncthreads();
        pc_block_add_u64<PC_F_WAVES>(tests, out_tests);
    }
}
