// libpaircount.so -- C ABI (include/paircount.h) over the HIP kernels.
// One translation unit; gfx950 only.
#include "../../include/paircount.h"

#include <hip/hip_runtime.h>
#include <atomic>
#include <pthread.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pc_bin_atomic.h"
#include "pc_bin_lds.h"
#include "pc_count_global.h"
#include "pc_count_tile.h"
#include "pc_count_wave.h"
#include "pc_count_group.h"
#include "pc_count_merge.h"
#include "pc_count_dense.h"
#include "pc_count_fused.h"
#include "pc_generate.h"
#include "pc_tokenize.h"

// ------------------------------------------------------------------ errors --
static thread_local char g_err[512] = "";

static int pc_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define PC_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return pc_fail(PC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                              \
    } while (0)

extern "C" const char *pc_last_error(void) { return g_err; }
extern "C" const char *pc_version(void) { return "1.0 gfx950"; }

// ---------------------------------------------------------------- geometry --
// The geometry the kernels of the calling thread's current entry point run with.  The
// classic entry points leave it at the reference's constants; the *_geometry entry
// points and contexts with pc_set_geometry install theirs for the duration of the call
// (entry points only enqueue work, so a thread-local "current" value is enough).
static pc_geom geom_default()
{
    pc_geom g;
    g.add = 10.0;            // f - (-10.)            main.c:134
    g.cell = 0.05;           // BBSIZE                main.c:30
    g.addf = 10.0f;
    g.invf = 20.0f;
    g.hi = 400.5f;
    g.ncell = PC_NLAYERS;
    g.d2max = 0.0025f;       // (double)d2 <= 0.0025  main.c:165, see pc_hit
    return g;
}
static thread_local pc_geom t_geo = geom_default();
struct geo_scope {
    pc_geom saved;
    explicit geo_scope(const pc_geom &g) : saved(t_geo) { t_geo = g; }
    ~geo_scope() { t_geo = saved; }
};

extern "C" int pc_geometry_default(pc_geometry *g)
{
    if (!g) return pc_fail(PC_ERR_ARG, "pc_geometry_default: null pointer");
    g->origin = -10.0;
    g->cell = 0.05;
    g->ncell = PC_NLAYERS;
    g->d2max = 0.0025f;
    return PC_OK;
}

// Checks a caller's geometry and derives the kernels' form of it.
static int make_geom(const pc_geometry *in, pc_geom *out)
{
    if (!in) return pc_fail(PC_ERR_ARG, "null geometry");
    const double origin = in->origin, cell = in->cell;
    if (!(cell > 0.0) || !(cell < 1e30) || !(origin > -1e30 && origin < 1e30))
        return pc_fail(PC_ERR_ARG, "geometry: origin %g / cell edge %g not usable", origin, cell);
    if (in->ncell < 1 || in->ncell > PC_NLAYERS)
        return pc_fail(PC_ERR_ARG, "geometry: %d cells per axis not in [1,%d]", in->ncell, PC_NLAYERS);
    if (!(in->d2max > 0.0f) || !(in->d2max < 1e30f))
        return pc_fail(PC_ERR_ARG, "geometry: squared cutoff %g not positive and finite", (double)in->d2max);
    pc_geom g = geom_default();
    const bool is_default = origin == -10.0 && cell == 0.05 && in->d2max == 0.0025f;
    // The stencil of 13 neighbour cells (main.c:195-209) finds every pair only if the cutoff does not exceed the
    // cell edge, and the merged-window pair test also needs points two cells apart to FAIL the predicate with
    // every rounding accounted for.  For the reference's constants that is shown bit by bit (pc_count_merge.h);
    // any other pair of values must leave a margin that dwarfs the binary32 roundings.
    if (!is_default && !((double)in->d2max <= cell * cell * (1.0 - 1e-5)))
        return pc_fail(PC_ERR_ARG, "geometry: squared cutoff %.9g exceeds cell edge squared %.9g less a 1e-5 margin",
                       (double)in->d2max, cell * cell);
    g.add = -origin;
    g.cell = cell;
    g.ncell = in->ncell;
    g.d2max = in->d2max;
    g.addf = (float)g.add;
    g.invf = (float)(1.0 / cell);
    // error budget of the binary32 shortcut in cells (see pc_cell_index): representation of add and 1/cell,
    // rounding of the sum and of the product; the shortcut is trusted 1e-3 cells away from a boundary
    const double span = fabs(g.add) + (in->ncell + 1) * cell;                 // largest |f + add| of interest
    const double err = (fabs(g.add - (double)g.addf) + ldexp(span, -24)) / cell
                     + (in->ncell + 1) * (fabs(1.0 / cell - (double)g.invf) * cell + ldexp(1.0, -23));
    g.hi = err < 2.5e-4 ? (float)in->ncell + 0.5f : 0.0f;                     // 0: every point takes the exact formula
    *out = g;
    return PC_OK;
}

// ---------------------------------------------------------------- variants --
// Process-wide measurement knobs (pc_set_variant): atomics, so that a host thread changing one while
// another thread launches is a defined read of the old or the new value, never a torn one.
static std::atomic<int> g_variant_bin{1};     // 0 = atomic counting sort, 1 = LDS-staged three-level bucket sort
// 0 = global memory, 1 = LDS-tiled strips, 2 = wave per pencil, 3 = candidate-parallel groups,
// 5 = merged single-run windows (variant 3's kernel for heavy pencils and what 5 defers),
// 7 = level Z fused into the pair test (whole-path entry points only; the z sort and variant 4's choice take
//     over, decided on the device, when a pencil neighbourhood exceeds the fused kernel's capacity),
// 4 = by density: 0 below 2/3 of a point per cell of the slab on average, 3 from 6 to 11 points per cell, else 5
static std::atomic<int> g_variant_count{4};
static std::atomic<int> g_heavy{PC_G_HEAVY_CAP};    // capacity of the list of over-full cells' home-particle chunks (0: never listed)
static std::atomic<int> g_segments{PC_G_SEG_CAP};  // capacity of the heavy-pencil segment list (0: pencils are never cut)
static std::atomic<int> g_even{1};                 // 1 = the merged-window kernel's grid is cut down to a whole number of pencils per wave
static std::atomic<int> g_dense_min{PC_DENSE_MIN_DEFAULT};   // points per cell from which a listed segment counts as well filled
static std::atomic<int> g_clustered_share{12};     // percent of the points in pencils far fuller than the average from which an input counts as clustered
static std::atomic<int> g_dense{1};                // who works the segment list off after the merged-window kernel: 1 = decided on the device from the list (dense kernel or variant 3's), 0 = variant 3's, 2 = the dense kernel

// A context may carry its own choice of the two algorithm variants (pc_context_set_variant); while its pc_finish
// runs, the calling thread sees that choice instead of the process-wide one (-1: not overridden).
static thread_local int t_variant_bin = -1, t_variant_count = -1;
static int variant_bin() { return t_variant_bin >= 0 ? t_variant_bin : g_variant_bin.load(); }
static int variant_count() { return t_variant_count >= 0 ? t_variant_count : g_variant_count.load(); }

static int check_variant(const char *who, const char *name, int value, int *which)
{
    if (!name) return pc_fail(PC_ERR_ARG, "%s: null name", who);
    if (!strcmp(name, "bin")) {
        if (value < 0 || value > 1) return pc_fail(PC_ERR_ARG, "%s: bin variant %d unknown", who, value);
        *which = 0;
        return PC_OK;
    }
    if (!strcmp(name, "count")) {
        if (value < 0 || value > 7 || value == 6) return pc_fail(PC_ERR_ARG, "%s: count variant %d unknown", who, value);
        *which = 1;
        return PC_OK;
    }
    *which = -1;
    return PC_OK;
}

extern "C" int pc_set_variant(const char *name, int value)
{
    if (!name) return pc_fail(PC_ERR_ARG, "pc_set_variant: null name");
    if (!strcmp(name, "bin")) {
        if (value < 0 || value > 1) return pc_fail(PC_ERR_ARG, "pc_set_variant: bin variant %d unknown", value);
        g_variant_bin = value;
        return PC_OK;
    }
    if (!strcmp(name, "count")) {
        if (value < 0 || value > 7 || value == 6) return pc_fail(PC_ERR_ARG, "pc_set_variant: count variant %d unknown", value);
        g_variant_count = value;
        return PC_OK;
    }
    if (!strcmp(name, "heavy")) {
        if (value < 0 || value > PC_G_HEAVY_CAP) return pc_fail(PC_ERR_ARG, "pc_set_variant: heavy-cell list capacity %d not in [0,%d]", value, PC_G_HEAVY_CAP);
        g_heavy = value;
        return PC_OK;
    }
    if (!strcmp(name, "segments")) {
        if (value < 0 || value > PC_G_SEG_CAP) return pc_fail(PC_ERR_ARG, "pc_set_variant: segment capacity %d not in [0,%d]", value, PC_G_SEG_CAP);
        g_segments = value;
        return PC_OK;
    }
    if (!strcmp(name, "even")) {
        if (value < 0 || value > 1) return pc_fail(PC_ERR_ARG, "pc_set_variant: even %d not 0 or 1", value);
        g_even = value;
        return PC_OK;
    }
    if (!strcmp(name, "clustered_share")) {
        if (value < 0 || value > 100) return pc_fail(PC_ERR_ARG, "pc_set_variant: clustered_share %d not a percentage", value);
        g_clustered_share = value;
        return PC_OK;
    }
    if (!strcmp(name, "dense_min")) {
        if (value < 1 || value > 65535) return pc_fail(PC_ERR_ARG, "pc_set_variant: dense_min %d not in [1,65535]", value);
        g_dense_min = value;
        return PC_OK;
    }
    if (!strcmp(name, "dense")) {
        if (value < 0 || value > 2) return pc_fail(PC_ERR_ARG, "pc_set_variant: dense %d not 0, 1 or 2", value);
        g_dense = value;
        return PC_OK;
    }
    return pc_fail(PC_ERR_ARG, "pc_set_variant: unknown name '%s'", name);
}

// --------------------------------------------------------------- profiling --
// Optional per-kernel timing: a pair of HIP events recorded on the launch stream
// around each kernel.  Off by default; the first PC_PROF_RING launches of each
// kernel after pc_profile_enable(1) are timed.
#define PC_PROF_SLOTS 24
#define PC_PROF_RING 512
struct pc_prof_slot {
    const char *name;
    hipEvent_t ev[PC_PROF_RING][2];
    int created, used;
    unsigned long long launches;
};
static pc_prof_slot g_prof[PC_PROF_SLOTS];
static int g_prof_n = 0;
static std::atomic<int> g_prof_on{0};
static pthread_mutex_t g_prof_lock = PTHREAD_MUTEX_INITIALIZER;   // slots and rings; taken only while profiling is on

static pc_prof_slot *prof_slot(const char *name)
{
    for (int i = 0; i < g_prof_n; i++)
        if (g_prof[i].name == name || !strcmp(g_prof[i].name, name)) return &g_prof[i];
    if (g_prof_n == PC_PROF_SLOTS) return nullptr;
    g_prof[g_prof_n].name = name;
    return &g_prof[g_prof_n++];
}

struct pc_prof_scope {
    pc_prof_slot *s = nullptr;
    hipStream_t st;
    int idx = -1;
    pc_prof_scope(const char *name, hipStream_t stream) : st(stream)
    {
        const int period = g_prof_on.load(std::memory_order_relaxed);
        if (!period) return;
        pthread_mutex_lock(&g_prof_lock);
        s = prof_slot(name);
        if (s) {
            s->launches++;
            // (period k: launches 1, k+1, 2k+1, ... of each kernel are timed -- the events themselves cost a few
            // microseconds per kernel, which a caller timing whole steps may not want on every launch)
            if (s->used < PC_PROF_RING && (period == 1 || s->launches % (unsigned)period == 1)) {
                if (s->used == s->created && hipEventCreate(&s->ev[s->created][0]) == hipSuccess) {
                    if (hipEventCreate(&s->ev[s->created][1]) == hipSuccess) s->created++;
                    else (void)hipEventDestroy(s->ev[s->created][0]);
                }
                if (s->used < s->created) idx = s->used++;
            }
        }
        pthread_mutex_unlock(&g_prof_lock);
        if (idx >= 0) (void)hipEventRecord(s->ev[idx][0], st);
    }
    ~pc_prof_scope()
    {
        if (idx >= 0) (void)hipEventRecord(s->ev[idx][1], st);
    }
};

#define PC_LAUNCH(name, kernel, grid, block, stream, ...)                          \
    do {                                                                            \
        pc_prof_scope prof_(name, stream);                                          \
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);            \
    } while (0)

extern "C" int pc_profile_enable(int on)
{
    pthread_mutex_lock(&g_prof_lock);
    if (on)
        for (int i = 0; i < g_prof_n; i++) { g_prof[i].used = 0; g_prof[i].launches = 0; }
    g_prof_on.store(on > 0 ? on : 0, std::memory_order_relaxed);
    pthread_mutex_unlock(&g_prof_lock);
    return PC_OK;
}

extern "C" int pc_profile_read(char *buf, size_t buflen)
{
    if (!buf || buflen == 0) return pc_fail(PC_ERR_ARG, "pc_profile_read: null buffer");
    size_t off = 0;
    buf[0] = 0;
    struct prof_turn {
        prof_turn() { pthread_mutex_lock(&g_prof_lock); }
        ~prof_turn() { pthread_mutex_unlock(&g_prof_lock); }
    } turn;
    for (int i = 0; i < g_prof_n; i++) {
        pc_prof_slot &s = g_prof[i];
        double total = 0;
        for (int k = 0; k < s.used; k++) {
            PC_HIP(hipEventSynchronize(s.ev[k][1]));
            float ms = 0;
            PC_HIP(hipEventElapsedTime(&ms, s.ev[k][0], s.ev[k][1]));
            total += ms;
        }
        int w = snprintf(buf + off, buflen - off, "%s %d %llu %.6f\n", s.name, s.used, s.launches, total);
        if (w < 0 || (size_t)w >= buflen - off) return pc_fail(PC_ERR_ARG, "pc_profile_read: buffer too small");
        off += (size_t)w;
    }
    return PC_OK;
}

// ------------------------------------------------------------------ stages --
static constexpr size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static const uint64_t PC_MAX_POINTS = 0xFFFFFFF0ull;   // slots are uint32

static constexpr uint32_t scan_tiles(uint64_t m) { return (uint32_t)((m + PC_SCAN_TILE - 1) / PC_SCAN_TILE); }
static inline int grid_for(uint64_t n, int tpb, int cap);

// Grids of the kernels whose workgroups loop over a work list: as many workgroups as are resident on the
// device at once (registers and LDS decide: asked of the runtime once per device, then cached -- the queries
// load the code object and are not for a stream under capture, see pc_noalloc_prepare).
struct launch_sizes {
    int ready;
    int l7b_small, l7b_medium, l7b_large;   // pc_l7b_sort_big_pencils, its instantiations
    int c3, c5, c6;                 // pc_c3_count_groups, pc_c5_count_merged, pc_c6_count_dense
};
static launch_sizes g_sizes[64];
static pthread_mutex_t g_sizes_lock = PTHREAD_MUTEX_INITIALIZER;

template <typename K>
static int resident_blocks(K kernel, int tpb, int cus, int fallback_per_cu)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, tpb, 0) != hipSuccess || per_cu < 1) per_cu = fallback_per_cu;
    return per_cu * cus;
}

static int launch_sizes_of_current_device(const launch_sizes **out);

// uint32 counters of the LDS variant, after its N*12-byte temp array
// pencils of more than PC_L7B_CAP_LARGE points there can be among n points (plus slack)
static constexpr size_t huge_rows(uint64_t n) { return (size_t)(n / PC_L7B_CAP_LARGE) + 2; }

static constexpr size_t lds_ws_counters(uint64_t n)
{
    return (size_t)PC_L_NREP * PC_L_BINS + (size_t)PC_L_NREP * PC_L_PSTRIDE + PC_L_BINS + (size_t)PC_L_NREP * PC_L_BINS + PC_L_BINS +
           ((n + PC_L_TILE - 1) / PC_L_TILE + PC_L_BINS) + (PC_L_NPENCILS + 1) + (size_t)PC_L_NREP * PC_L_PSTRIDE +
           8 + PC_L_NPENCILS + huge_rows(n) * (1 + 2 * (size_t)PC_L_BINS) + PC_L_NPENCILS;
}

static constexpr size_t workspace_bytes_for(uint64_t n)
{
    size_t keys = align_up((size_t)n * 4 + 4, 256);
    size_t sums = align_up((size_t)scan_tiles(PC_NCELLS + 2) * 4, 256);
    size_t v0 = keys + sums + 256;
    size_t v1 = align_up((size_t)n * 12 + 16, 256) + align_up(lds_ws_counters(n) * 4, 256) + align_up((size_t)n * 2 + 16, 256) + 256;
    return v0 > v1 ? v0 : v1;
}

extern "C" size_t pc_workspace_bytes(uint64_t n) { return workspace_bytes_for(n); }

static pc_lds_ws carve_lds_ws(void *ws, uint64_t n)
{
    pc_lds_ws w;
    w.tmp = (float *)ws;
    uint32_t *u = (uint32_t *)((char *)ws + align_up((size_t)n * 12 + 16, 256));
    w.layer_count = u;            u += PC_L_NREP * PC_L_BINS;
    w.pencil_count = u;           u += PC_L_NREP * PC_L_PSTRIDE;
    w.big_count = u;              u += 8;                       // (zeroed together with the two arrays before it)
    w.layer_start = u;            u += PC_L_BINS;
    w.layer_cursor = u;           u += PC_L_NREP * PC_L_BINS;
    w.layer_tile = u;             u += PC_L_BINS;
    w.tile_layer = u;             u += (n + PC_L_TILE - 1) / PC_L_TILE + PC_L_BINS;
    w.pencil_start = u;           u += PC_L_NPENCILS + 1;
    w.pencil_cursor = u;          u += PC_L_NREP * PC_L_PSTRIDE;
    w.big_list = u;               u += PC_L_NPENCILS;
    w.huge_list = u;              u += huge_rows(n);
    w.huge_count = u;             u += huge_rows(n) * PC_L_BINS;
    w.huge_cursor = u;            u += huge_rows(n) * PC_L_BINS;
    w.fused_list = u;
    w.ycell = (uint16_t *)((char *)ws + align_up((size_t)n * 12 + 16, 256) + align_up(lds_ws_counters(n) * 4, 256));
    return w;
}

// layers [layer_lo, layer_hi] get their pencils sorted and their table slices written (all: 0 .. PC_NBB-1)
// (layer_counts: the [NREP][404] counts of L1 if they were taken while the points arrived, else null)
// home layers [home_lo, home_hi) and `fused`: the call may be served by the fused pair test (level Z then runs
// only if a pencil neighbourhood of the home layers exceeds that kernel's capacity: decided on the device)
// What a whole-path call wants cleared before its pair test: the work counters of the pair test and the result.
// Level X's scan kernel does it on the way (no fill launches of their own).
struct zero_request {
    unsigned int *words = nullptr;
    uint32_t n = 0;
    unsigned long long *out = nullptr;
};

static int bin_lds(const float *d_xyz, uint64_t n, float *d_sorted, uint32_t *table, void *ws,
                   int32_t *d_status, hipStream_t s, int layer_lo, int layer_hi, const uint32_t *layer_counts,
                   int home_lo, int home_hi, bool fused, const zero_request &zero)
{
    pc_lds_ws w = carve_lds_ws(ws, n);
    uint32_t *const gate_words = w.big_count + 2;                       // pc_fused_serves: [0] largest neighbourhood, [1] capacity
    const uint32_t *const fgate = fused ? gate_words : nullptr;
    PC_HIP(hipMemsetAsync(w.layer_count, 0, (PC_L_NREP * PC_L_BINS + PC_L_NREP * PC_L_PSTRIDE + 8) * sizeof(uint32_t), s));
    const int tiles = (int)((n + PC_L_TILE - 1) / PC_L_TILE);
    if (layer_counts)
        PC_HIP(hipMemcpyAsync(w.layer_count, layer_counts, PC_L_NREP * PC_L_BINS * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    else if (n) PC_LAUNCH("pc_l1_hist_x", pc_l1_hist_x, dim3(grid_for(n, PC_L_TILE, 2048)), dim3(PC_L_TPB), s,
                          d_xyz, n, w.layer_count, d_status, t_geo);
    PC_LAUNCH("pc_l2_scan_x", pc_l2_scan_x, dim3(1), dim3(512), s, w.layer_count, w.layer_start, w.layer_cursor,
              w.layer_tile, w.tile_layer, gate_words + 1, (uint32_t)PC_F_CAP, zero.words, zero.n, zero.out);
    if (n) {
        PC_LAUNCH("pc_l3_partition_x", pc_l36_partition<0>, dim3(tiles), dim3(PC_L_XTPB), s, d_xyz, n, d_sorted,
                  w.layer_start, w.layer_tile, w.tile_layer, w.layer_cursor, w.ycell, t_geo);
        PC_LAUNCH("pc_l4_hist_y", pc_l4_hist_y, dim3(tiles + PC_NBB), dim3(PC_L_TPB), s, w.ycell, w.layer_start,
                  w.layer_tile, w.tile_layer, w.pencil_count, d_status, t_geo);
    }
    // sparse inputs (at most half the small capacity per occupied pencil on average) take the low-LDS form of L7
    const uint64_t occupied_pencils = (uint64_t)(layer_hi < PC_NLAYERS ? layer_hi - layer_lo + 1 : PC_NLAYERS - layer_lo) * PC_NLAYERS;
    const bool sparse = n * 2 <= occupied_pencils * PC_L7_CAP_SPARSE;
    PC_LAUNCH("pc_l5_scan_y", pc_l5_scan_y, dim3(PC_NBB), dim3(512), s, w.pencil_count, w.layer_start,
              w.pencil_start, w.pencil_cursor, (uint32_t)(sparse ? PC_L7_CAP_SPARSE : PC_L7_CAP), w.big_count, w.big_list,
              home_lo, home_hi, gate_words, layer_lo, layer_hi, (uint32_t)PC_L7B_CAP_LARGE, w.huge_list, w.huge_count);
    if (n)
        PC_LAUNCH("pc_l6_partition_y", pc_l36_partition<1>, dim3(tiles + PC_NBB), dim3(PC_L_PTPB), s, d_sorted, n,
                  w.tmp, w.layer_start, w.layer_tile, w.tile_layer, w.pencil_cursor, w.ycell, t_geo);
    // Level Z in four launches.  Pencils too big for a wave are listed by L5 (one workgroup each: three size classes,
    // fixed grids walk the list), those too big for a workgroup on a second list (tile-parallel histogram, scan,
    // partition).  The three steps of the latter ride at the end of three of the grids: see L7H in pc_bin_lds.h.
    const uint32_t first_pencil = (uint32_t)layer_lo * PC_NBB, end_pencil = (uint32_t)(layer_hi + 1) * PC_NBB;
    const uint32_t l7_blocks = (end_pencil - first_pencil + PC_L7_WAVES - 1) / PC_L7_WAVES;
    const dim3 l7_grid(l7_blocks + PC_L7H_GRID);
    // (large inputs stream through L7 with non-temporal loads and stores: see PC_L_STREAM_POINTS)
    auto l7 = sparse ? pc_l7_sort_pencils<PC_L7_CAP_SPARSE, false>
                     : n > PC_L_STREAM_POINTS ? pc_l7_sort_pencils<PC_L7_CAP, true> : pc_l7_sort_pencils<PC_L7_CAP, false>;
    PC_LAUNCH("pc_l7_sort_pencils", l7, l7_grid, dim3(PC_L7_WAVES * 64), s, w.tmp, d_sorted, w.pencil_start, table, first_pencil,
              end_pencil, d_status, t_geo, fgate, l7_blocks, w.big_count + 1, w.huge_list, w.huge_count);
    auto small = pc_l7b_sort_big_pencils<PC_L7B_CAP_SMALL, PC_L7B_TPB_SMALL, PC_L7B_ROLE_SCAN>;
    auto medium = pc_l7b_sort_big_pencils<PC_L7B_CAP_MEDIUM, PC_L7B_TPB_MEDIUM, PC_L7B_ROLE_NONE>;
    auto large = pc_l7b_sort_big_pencils<PC_L7B_CAP_LARGE, PC_L7B_TPB_LARGE, PC_L7B_ROLE_PARTITION>;
    const launch_sizes *sizes = nullptr;
    const int rc_sizes = launch_sizes_of_current_device(&sizes);
    if (rc_sizes) return rc_sizes;
    PC_LAUNCH("pc_l7b_sort_big_pencils", small, dim3(sizes->l7b_small + PC_L7H_SCAN_GRID), dim3(PC_L7B_TPB_SMALL), s, w.tmp, d_sorted,
              w.pencil_start, table, w.big_count, w.big_list, first_pencil, end_pencil, 0u, (uint32_t)PC_L7B_CAP_SMALL,
              (uint32_t)sizes->l7b_small, w.big_count + 1, w.huge_list, w.huge_count, w.huge_cursor, d_status, t_geo, fgate);
    PC_LAUNCH("pc_l7b_sort_medium_pencils", medium, dim3(sizes->l7b_medium), dim3(PC_L7B_TPB_MEDIUM), s, w.tmp, d_sorted,
              w.pencil_start, table, w.big_count, w.big_list, first_pencil, end_pencil, (uint32_t)PC_L7B_CAP_SMALL,
              (uint32_t)PC_L7B_CAP_MEDIUM, (uint32_t)sizes->l7b_medium, w.big_count + 1, w.huge_list, w.huge_count, w.huge_cursor,
              d_status, t_geo, fgate);
    PC_LAUNCH("pc_l7b_sort_bigger_pencils", large, dim3(sizes->l7b_large + PC_L7H_GRID), dim3(PC_L7B_TPB_LARGE), s, w.tmp, d_sorted,
              w.pencil_start, table, w.big_count, w.big_list, first_pencil, end_pencil, (uint32_t)PC_L7B_CAP_MEDIUM,
              (uint32_t)PC_L7B_CAP_LARGE, (uint32_t)sizes->l7b_large, w.big_count + 1, w.huge_list, w.huge_count, w.huge_cursor,
              d_status, t_geo, fgate);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

static inline int grid_for(uint64_t n, int tpb, int cap)
{
    uint64_t g = (n + tpb - 1) / tpb;
    if (g < 1) g = 1;
    if (g > (uint64_t)cap) g = cap;
    return (int)g;
}

extern "C" int pc_generate_device(int kind, uint64_t seed, int64_t sigma_micro, uint64_t first, uint64_t count,
                                  float *d_xyz, void *stream)
{
    if (kind < PC_GEN_UNIFORM || kind > PC_GEN_RAMP) return pc_fail(PC_ERR_ARG, "pc_generate_device: kind %d unknown", kind);
    if (count == 0) return PC_OK;
    if (!d_xyz) return pc_fail(PC_ERR_ARG, "pc_generate_device: null output");
    if (count > PC_MAX_POINTS) return pc_fail(PC_ERR_ARG, "pc_generate_device: count too large");
    PC_LAUNCH("pc_gen_points", pc_gen_points, dim3(grid_for(count, 256, 16384)), dim3(256), (hipStream_t)stream,
              kind, seed, sigma_micro, first, count, d_xyz);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_selfcheck_cell_index_device(uint64_t *d_mismatches, void *stream)
{
    if (!d_mismatches) return pc_fail(PC_ERR_ARG, "pc_selfcheck_cell_index_device: null pointer");
    PC_LAUNCH("pc_selfcheck_cell_index", pc_selfcheck_cell_index, dim3(65536), dim3(256), (hipStream_t)stream,
              (unsigned long long *)d_mismatches, t_geo);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_selfcheck_cell_index_geometry_device(const pc_geometry *g, uint64_t *d_mismatches, void *stream)
{
    pc_geom geo;
    int rc = make_geom(g, &geo);
    if (rc) return rc;
    geo_scope scope(geo);
    return pc_selfcheck_cell_index_device(d_mismatches, stream);
}

extern "C" int pc_cell_keys_geometry_device(const pc_geometry *g, const float *d_xyz, uint64_t n, uint32_t *d_keys,
                                            int32_t *d_status, void *stream)
{
    pc_geom geo;
    int rc = make_geom(g, &geo);
    if (rc) return rc;
    geo_scope scope(geo);
    return pc_cell_keys_device(d_xyz, n, d_keys, d_status, stream);
}

extern "C" int pc_bin_geometry_device(const pc_geometry *g, const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                                      uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                                      int32_t *d_status, void *stream)
{
    pc_geom geo;
    int rc = make_geom(g, &geo);
    if (rc) return rc;
    geo_scope scope(geo);
    return pc_bin_device(d_xyz, n, d_sorted_xyz, d_cell_start, d_workspace, workspace_bytes, d_status, stream);
}

extern "C" int pc_cell_keys_device(const float *d_xyz, uint64_t n, uint32_t *d_keys,
                                   int32_t *d_status, void *stream)
{
    if ((!d_xyz || !d_keys) && n) return pc_fail(PC_ERR_ARG, "pc_cell_keys_device: null pointer");
    if (!d_status) return pc_fail(PC_ERR_ARG, "pc_cell_keys_device: null status");
    if (n > PC_MAX_POINTS) return pc_fail(PC_ERR_ARG, "pc_cell_keys_device: n too large");
    if (n == 0) return PC_OK;
    PC_LAUNCH("pc_keys_only", pc_keys_only, dim3(grid_for(n, 256, 8192)), dim3(256), (hipStream_t)stream,
                       d_xyz, n, d_keys, d_status, t_geo);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

static int bin_layers(const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                      uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                      int32_t *d_status, void *stream, int layer_lo, int layer_hi, const uint32_t *layer_counts = nullptr,
                      int home_lo = 0, int home_hi = 0, bool fused = false, zero_request *zero = nullptr);
extern "C" int pc_bin_device(const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                             uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                             int32_t *d_status, void *stream);

extern "C" int pc_bin_device(const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                             uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                             int32_t *d_status, void *stream)
{
    return bin_layers(d_xyz, n, d_sorted_xyz, d_cell_start, d_workspace, workspace_bytes, d_status, stream, 0, PC_NBB - 1);
}

// Binning whose result is complete for the layers [layer_lo, layer_hi] only (the
// default variant skips the final per-pencil sort and the table slices of the
// other layers; a slab's pair test never reads them).
static int bin_layers(const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                      uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                      int32_t *d_status, void *stream, int layer_lo, int layer_hi, const uint32_t *layer_counts,
                      int home_lo, int home_hi, bool fused, zero_request *zero)
{
    if (!d_cell_start || !d_status || !d_workspace)
        return pc_fail(PC_ERR_ARG, "pc_bin_device: null pointer");
    if ((!d_xyz || !d_sorted_xyz) && n) return pc_fail(PC_ERR_ARG, "pc_bin_device: null point array");
    if (n > PC_MAX_POINTS) return pc_fail(PC_ERR_ARG, "pc_bin_device: n too large");
    if (workspace_bytes < pc_workspace_bytes(n))
        return pc_fail(PC_ERR_ARG, "pc_bin_device: workspace %zu < %zu bytes", workspace_bytes,
                       pc_workspace_bytes(n));
    hipStream_t s = (hipStream_t)stream;
    if (variant_bin() == 1) {
        const int rc = bin_lds(d_xyz, n, d_sorted_xyz, d_cell_start, d_workspace, d_status, s, layer_lo, layer_hi, layer_counts, home_lo,
                               home_hi, fused, zero ? *zero : zero_request());
        if (rc == PC_OK && zero) *zero = zero_request();     // served
        return rc;
    }
    const uint64_t m = PC_NCELLS + 2;
    uint32_t *keys = (uint32_t *)d_workspace;
    uint32_t *sums = (uint32_t *)((char *)d_workspace + align_up((size_t)n * 4 + 4, 256));
    const uint32_t ntiles = scan_tiles(m);

    PC_HIP(hipMemsetAsync(d_cell_start, 0, m * sizeof(uint32_t), s));
    if (n)
        PC_LAUNCH("pc_a1_keys_hist", pc_a1_keys_hist, dim3(grid_for(n, 256, 8192)), dim3(256), s, d_xyz, n, keys,
                           d_cell_start, d_status, t_geo);
    PC_LAUNCH("pc_a2_scan_reduce", pc_a2_scan_reduce, dim3(ntiles), dim3(PC_SCAN_TPB), s, d_cell_start, m, sums);
    PC_LAUNCH("pc_a2_scan_spine", pc_a2_scan_spine, dim3(1), dim3(PC_SCAN_TPB), s, sums, ntiles);
    PC_LAUNCH("pc_a2_scan_apply", pc_a2_scan_apply, dim3(ntiles), dim3(PC_SCAN_TPB), s, d_cell_start, m, sums);
    if (n)
        PC_LAUNCH("pc_a3_scatter", pc_a3_scatter, dim3(grid_for(n, 256, 8192)), dim3(256), s, d_xyz, keys, n,
                           d_cell_start, d_sorted_xyz);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

// Work area of the persistent pair-test kernel (layout: pc_count_group.h): work
// counters, the per-pencil "planned" flags and the segment list and the list of over-full cells, 2.8 MB in the
// code object (each device holds its own copy).  The counters are zeroed on the
// launch stream before every launch, flags and segments are rewritten by the
// planning kernel.  Two pair-test launches that overlap on one device would
// share it; the library's own entry points never overlap launches on a device.
static __device__ __align__(256) unsigned int pc_count_queue_words[PC_G_WORK_WORDS];
static unsigned int *g_queue_ptr[64];
static pthread_mutex_t g_queue_lock = PTHREAD_MUTEX_INITIALIZER;

static int count_queue(unsigned int **out)
{
    int dev = 0;
    PC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return pc_fail(PC_ERR_ARG, "device ordinal %d not supported", dev);
    pthread_mutex_lock(&g_queue_lock);
    unsigned int *p = g_queue_ptr[dev];
    if (!p) {
        hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(pc_count_queue_words));
        if (e != hipSuccess) {
            pthread_mutex_unlock(&g_queue_lock);
            return pc_fail(PC_ERR_HIP, "hipGetSymbolAddress(work counters): %s", hipGetErrorString(e));
        }
        g_queue_ptr[dev] = p;
    }
    pthread_mutex_unlock(&g_queue_lock);
    *out = p;
    return PC_OK;
}

static int launch_sizes_of_current_device(const launch_sizes **out)
{
    int dev = 0;
    PC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return pc_fail(PC_ERR_ARG, "device ordinal %d not supported", dev);
    launch_sizes *z = &g_sizes[dev];
    pthread_mutex_lock(&g_sizes_lock);
    if (!z->ready) {
        hipDeviceProp_t prop;
        const hipError_t e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) {
            pthread_mutex_unlock(&g_sizes_lock);
            return pc_fail(PC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        }
        const int cus = prop.multiProcessorCount;
        z->l7b_small = resident_blocks(pc_l7b_sort_big_pencils<PC_L7B_CAP_SMALL, PC_L7B_TPB_SMALL, PC_L7B_ROLE_SCAN>, PC_L7B_TPB_SMALL, cus, 4);
        z->l7b_medium = resident_blocks(pc_l7b_sort_big_pencils<PC_L7B_CAP_MEDIUM, PC_L7B_TPB_MEDIUM, PC_L7B_ROLE_NONE>, PC_L7B_TPB_MEDIUM, cus, 2);
        z->l7b_large = resident_blocks(pc_l7b_sort_big_pencils<PC_L7B_CAP_LARGE, PC_L7B_TPB_LARGE, PC_L7B_ROLE_PARTITION>, PC_L7B_TPB_LARGE, cus, 1);
        z->c3 = resident_blocks(pc_c3_count_groups<true>, PC_G_WAVES * 64, cus, 16);
        z->c5 = resident_blocks(pc_c5_count_merged, 64, cus, 16);
        z->c6 = resident_blocks(pc_c6_count_dense, 64, cus, 16);
        z->ready = 1;
    }
    pthread_mutex_unlock(&g_sizes_lock);
    *out = z;
    return PC_OK;
}

static int check_slab(const char *who, int x_lo, int x_hi)
{
    if (x_lo < 0 || x_hi > PC_NLAYERS || x_lo > x_hi)
        return pc_fail(PC_ERR_ARG, "%s: slab [%d,%d) not within [0,%d]", who, x_lo, x_hi, PC_NLAYERS);
    return PC_OK;
}

// fgate: the switch of a whole-path call that the fused pair test may serve (see pc_fused_serves), else null
// queue_zeroed: the work counters were cleared on this stream already (a whole-path call has level X's scan do it)
static int count_sorted(const float *d_sorted_xyz, const uint32_t *d_cell_start,
                        uint64_t n, int x_lo, int x_hi, uint64_t *d_npairs,
                        void *stream, const uint32_t *fgate, bool queue_zeroed = false)
{
    if (!d_cell_start || !d_npairs) return pc_fail(PC_ERR_ARG, "pc_count_sorted_device: null pointer");
    if (!d_sorted_xyz && n) return pc_fail(PC_ERR_ARG, "pc_count_sorted_device: null point array");
    if (n > PC_MAX_POINTS) return pc_fail(PC_ERR_ARG, "pc_count_sorted_device: n too large");
    int rc = check_slab("pc_count_sorted_device", x_lo, x_hi);
    if (rc) return rc;
    if (n == 0 || x_lo == x_hi) return PC_OK;
    int variant = variant_count();
    if (variant == 7) variant = 4;                 // (the fused kernel is part of the whole-path entry points; this is their other route)
    if (variant == 4) {
        // Sparse inputs: a window of cells holds too few particles to pay for its bookkeeping, and the
        // thread-per-particle kernel wins (measured crossover at 42 M uniform points = 0.66 per cell).
        // Between 6 and 11 points per cell the group kernel takes the whole job: the merged-window kernel's runs get
        // too long there, and the dense kernel, which works its pencils' segments off instead, pays only from about
        // 11 points per cell (measured on a slab at 3 to 16 points per cell: up to 35 % at 7.8).
        const uint64_t cells = (uint64_t)(x_hi < PC_NLAYERS ? x_hi - x_lo + 1 : x_hi - x_lo) * PC_NLAYERS * PC_NLAYERS;
        variant = n * 3 < cells * 2 ? 0 : (n >= cells * 6 && n < cells * 11) ? 3 : 5;
    }
    if (variant == 0) {
        uint32_t c_lo = (uint32_t)x_lo * PC_NBB2, c_hi = (uint32_t)x_hi * PC_NBB2;
        PC_LAUNCH("pc_c0_count_global", pc_c0_count_global, dim3(grid_for(n, 256, 4096)), dim3(256), (hipStream_t)stream, d_sorted_xyz, d_cell_start, c_lo, c_hi,
                           (unsigned long long *)d_npairs, t_geo, fgate);
    } else if (variant == 3 || variant == 5) {
        // persistent: one resident set of waves, never more waves than pencils;
        // the eight work counters are zeroed on the stream ahead of the launch
        unsigned int *queue = nullptr;
        int rc2 = count_queue(&queue);
        if (rc2) return rc2;
        const uint32_t seg_cap = (uint32_t)g_segments.load(), heavy_cap = (uint32_t)g_heavy.load();   // one read each per call
        if (!queue_zeroed) PC_HIP(hipMemsetAsync(queue, 0, PC_G_FLAGS * sizeof(unsigned int), (hipStream_t)stream));   // all counters
        const dim3 plan_grid(((x_hi - x_lo) * PC_NLAYERS + PC_G_PLAN_PENCILS - 1) / PC_G_PLAN_PENCILS);
        // a pencil holding more than four average pencils (and at least 512 points) counts as dense; an input with
        // over an eighth of its points in such pencils is clustered (measured on blobs of widths 0.5 to 1.5: at a quarter
        // the width 0.65 still took the split route, 12 % slower): the merged-window kernel then leaves the
        // whole job to the group kernel, which balances heavy pencils better
        const uint64_t slab_pencils = (uint64_t)(x_hi < PC_NLAYERS ? x_hi - x_lo + 1 : x_hi - x_lo) * PC_NLAYERS;
        const uint64_t full = 4 * n / slab_pencils;
        const uint32_t full_pencil = (uint32_t)(full < 512 ? 512 : full > 0xFFFFFFFFull ? 0xFFFFFFFFull : full);
        const uint32_t gate_limit = (uint32_t)(n * (uint64_t)g_clustered_share.load() / 100);
        // dense = 1: the listed points are added up by kind, and the kernel that suits the majority takes the list
        const int dense_mode = g_dense.load();
        const uint32_t dense_min = variant == 5 && dense_mode == 1 ? (uint32_t)g_dense_min.load() : 0u;
        PC_LAUNCH("pc_c3_plan", pc_c3_plan, plan_grid, dim3(PC_G_PLAN_TPB), (hipStream_t)stream, d_cell_start, x_lo, x_hi, queue,
                  seg_cap, full_pencil, fgate, dense_min);
        const launch_sizes *sizes = nullptr;
        rc2 = launch_sizes_of_current_device(&sizes);
        if (rc2) return rc2;
        const int resident = sizes->c3;
        const int want = resident, most = ((x_hi - x_lo) * PC_NLAYERS + PC_G_WAVES - 1) / PC_G_WAVES;
        if (variant == 5) {
            // the merged-window kernel takes every pencil the planner left whole; what it cannot process goes on the
            // segment list, which the group kernel then works off together with the planned segments
            const int r5 = sizes->c5;
            // every wave the same number of pencils: with 3.3 pencils per resident wave (a slab of an 8-GPU run) a
            // quarter of the waves would take a fourth one while the others idle; fewer waves, each a little faster
            // for the free share of its SIMD, finish the same four rounds sooner
            int grid5 = (r5 < most ? r5 : most);
            if (g_even.load()) {
                const int pencils = (x_hi - x_lo) * PC_NLAYERS, rounds = (pencils + r5 - 1) / r5;
                const int even = (pencils + rounds - 1) / rounds;
                if (even < grid5) grid5 = even;
            }
            grid5 = (grid5 + 7) / 8 * 8;
            PC_LAUNCH("pc_c5_count_merged", pc_c5_count_merged, dim3(grid5), dim3(64), (hipStream_t)stream,
                      d_sorted_xyz, d_cell_start, x_lo, x_hi, queue, seg_cap, PC_GATE_IF_EVEN, gate_limit,
                      (unsigned long long *)d_npairs, t_geo, dense_min);
        }
        const int grid = ((want < most ? want : most) + 7) / 8 * 8;
        if (variant == 5) {
            // planned segments and pencil rests: to the candidates-in-registers kernel if most of their points lie in
            // well-filled cells, else to the group kernel (one of the two launches returns at once)
            if (dense_mode != 0) {
                const int r6 = sizes->c6;
                PC_LAUNCH("pc_c6_count_dense", pc_c6_count_dense, dim3(r6), dim3(64), (hipStream_t)stream,
                          d_sorted_xyz, d_cell_start, x_lo, x_hi, queue, seg_cap, heavy_cap, PC_GATE_IF_EVEN, gate_limit,
                          (unsigned long long *)d_npairs, t_geo.d2max, dense_min);
            }
            if (dense_mode != 2)
                PC_LAUNCH("pc_c3_count_groups_segments", pc_c3_count_groups<false>, dim3(grid), dim3(PC_G_WAVES * 64), (hipStream_t)stream,
                          d_sorted_xyz, d_cell_start, x_lo, x_hi, queue, seg_cap, heavy_cap, PC_GATE_IF_EVEN, gate_limit,
                          (unsigned long long *)d_npairs, t_geo.d2max, dense_min);
            PC_LAUNCH("pc_c3_count_groups", pc_c3_count_groups<true>, dim3(grid), dim3(PC_G_WAVES * 64), (hipStream_t)stream,
                      d_sorted_xyz, d_cell_start, x_lo, x_hi, queue, seg_cap, heavy_cap, PC_GATE_IF_CLUSTERED, gate_limit,
                      (unsigned long long *)d_npairs, t_geo.d2max, 0u);
        } else
            PC_LAUNCH("pc_c3_count_groups", pc_c3_count_groups<true>, dim3(grid), dim3(PC_G_WAVES * 64), (hipStream_t)stream,
                      d_sorted_xyz, d_cell_start, x_lo, x_hi, queue, seg_cap, heavy_cap, PC_GATE_ALWAYS, 0u,
                      (unsigned long long *)d_npairs, t_geo.d2max, 0u);
        // the home particles of over-full cells that the waves listed instead of processing (normally none:
        // the workgroups then return at once)
        PC_LAUNCH("pc_c3_heavy", pc_c3_heavy, dim3(resident >= 2048 ? 2048 : resident), dim3(PC_HV_TPB), (hipStream_t)stream,
                  d_sorted_xyz, d_cell_start, queue, heavy_cap, (unsigned long long *)d_npairs, t_geo.d2max);
    } else if (variant == 2) {
        const int grid = 8 * (x_hi - x_lo) * PC_W_WG_PER_BAND;
        PC_LAUNCH("pc_c2_count_waves", pc_c2_count_waves, dim3(grid), dim3(PC_W_WAVES * 64), (hipStream_t)stream,
                  d_sorted_xyz, d_cell_start, x_lo, x_hi, (unsigned long long *)d_npairs, t_geo);
    } else {
        // 8 XCD groups x layers x (largest band of strips any group owns)
        const int band = (PC_T_NS + 7) / 8;
        const int grid = 8 * (x_hi - x_lo) * band;
        PC_LAUNCH("pc_c1_count_tiles", pc_c1_count_tiles, dim3(grid), dim3(PC_T_TPB), (hipStream_t)stream,
                           d_sorted_xyz, d_cell_start, x_lo, x_hi, (unsigned long long *)d_npairs, t_geo);
    }
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_count_sorted_device(const float *d_sorted_xyz, const uint32_t *d_cell_start,
                                      uint64_t n, int x_lo, int x_hi, uint64_t *d_npairs,
                                      void *stream)
{
    return count_sorted(d_sorted_xyz, d_cell_start, n, x_lo, x_hi, d_npairs, stream, nullptr);
}

// Whether a whole-path call tries the fused pair test (level Z inside the pair test, pc_count_fused.h): only when
// asked for by name.  Measured at 1.56 points per cell it takes 2.6-2.75 ms against 2.06 ms for the z sort plus the
// merged-window kernel, so the default choice by density does not include it.
static bool fused_wanted(uint64_t n, int x_lo, int x_hi)
{
    return variant_count() == 7 && variant_bin() == 1 && n != 0 && x_lo != x_hi;
}

static const uint32_t *fused_gate_words(void *d_workspace, uint64_t n) { return carve_lds_ws(d_workspace, n).big_count + 2; }

// The fused pair test on the level-Y output that bin_lds left in the workspace.  ADDS to *d_npairs and, if
// asked, to *d_ntests; returns at once (on the device) if a neighbourhood exceeds its capacity.
static int count_fused(void *d_workspace, uint64_t n, int x_lo, int x_hi, uint64_t *d_npairs, uint64_t *d_ntests,
                       int32_t *d_status, hipStream_t s)
{
    const pc_lds_ws w = carve_lds_ws(d_workspace, n);
    const dim3 grid((unsigned)(x_hi - x_lo) * PC_NLAYERS);
    PC_LAUNCH("pc_c7_count_fused", pc_c7_count_fused, grid, dim3(PC_F_TPB), s, w.tmp, w.pencil_start, x_lo, x_hi,
              w.big_count + 2, w.big_count + 4, w.fused_list, (unsigned long long *)d_npairs, (unsigned long long *)d_ntests,
              d_status, t_geo);
    // pencils with too many points in three consecutive blocks for a chunk of the kernel above (normally none)
    PC_LAUNCH("pc_c7_count_fused_listed", pc_c7_count_fused_listed, dim3(256), dim3(PC_F_TPB), s, w.tmp, w.pencil_start,
              w.big_count + 2, w.big_count + 4, w.fused_list, (unsigned long long *)d_npairs, (unsigned long long *)d_ntests,
              d_status, t_geo);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_count_tests_device(const uint32_t *d_cell_start, int x_lo, int x_hi,
                                     uint64_t *d_ntests, void *stream)
{
    if (!d_cell_start || !d_ntests) return pc_fail(PC_ERR_ARG, "pc_count_tests_device: null pointer");
    int rc = check_slab("pc_count_tests_device", x_lo, x_hi);
    if (rc) return rc;
    if (x_lo == x_hi) return PC_OK;
    uint32_t c_lo = (uint32_t)x_lo * PC_NBB2, c_hi = (uint32_t)x_hi * PC_NBB2;
    PC_LAUNCH("pc_count_tests", pc_count_tests, dim3(2048), dim3(256), (hipStream_t)stream, d_cell_start, c_lo,
                       c_hi, (unsigned long long *)d_ntests);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

extern "C" int pc_count_pairs_device(const float *d_xyz, uint64_t n, int x_lo, int x_hi,
                                     float *d_sorted_xyz, uint32_t *d_cell_start,
                                     void *d_workspace, size_t workspace_bytes,
                                     uint64_t *d_npairs, int32_t *d_status, void *stream)
{
    if (!d_npairs) return pc_fail(PC_ERR_ARG, "pc_count_pairs_device: null result pointer");
    int rc = check_slab("pc_count_pairs_device", x_lo, x_hi);
    if (rc) return rc;
    // home layers [x_lo, x_hi) and the halo layer x_hi are all the pair test reads
    const bool fused = fused_wanted(n, x_lo, x_hi);
    // the result and the pair test's work counters are cleared by a binning kernel on its way (LDS variant), else here
    zero_request zero;
    rc = count_queue(&zero.words);
    if (rc) return rc;
    zero.n = PC_G_FLAGS;
    zero.out = (unsigned long long *)d_npairs;
    rc = bin_layers(d_xyz, n, d_sorted_xyz, d_cell_start, d_workspace, workspace_bytes, d_status, stream, x_lo,
                    x_hi < PC_NLAYERS ? x_hi : PC_NBB - 1,         // the last slab also gets the empty padding layers
                    nullptr, x_lo, x_hi, fused, &zero);
    if (rc) return rc;
    const bool zeroed = zero.out == nullptr;
    if (!zeroed) PC_HIP(hipMemsetAsync(d_npairs, 0, sizeof(uint64_t), (hipStream_t)stream));
    if (fused) {
        rc = count_fused(d_workspace, n, x_lo, x_hi, d_npairs, nullptr, d_status, (hipStream_t)stream);
        if (rc) return rc;
    }
    return count_sorted(d_sorted_xyz, d_cell_start, n, x_lo, x_hi, d_npairs, stream,
                        fused ? fused_gate_words(d_workspace, n) : nullptr, zeroed);
}

extern "C" int pc_count_pairs_geometry_device(const pc_geometry *g, const float *d_xyz, uint64_t n, int x_lo, int x_hi,
                                              float *d_sorted_xyz, uint32_t *d_cell_start,
                                              void *d_workspace, size_t workspace_bytes,
                                              uint64_t *d_npairs, int32_t *d_status, void *stream)
{
    pc_geom geo;
    int rc = make_geom(g, &geo);
    if (rc) return rc;
    geo_scope scope(geo);
    return pc_count_pairs_device(d_xyz, n, x_lo, x_hi, d_sorted_xyz, d_cell_start, d_workspace, workspace_bytes, d_npairs,
                                 d_status, stream);
}

// ------------------------------------------------- allocation-free entry point --
// The binning workspace of pc_count_pairs_device (level-Y point array, y cells,
// layer and pencil counters, cursors) in a fixed array inside the code object, so
// that a caller who cannot allocate has nothing to bring but inputs and outputs.
// Each device holds its own copy at its own address; host code reaches it only
// through the pointer hipGetSymbolAddress returned for that device.
#define PC_STATIC_MAX_POINTS PC_NOALLOC_MAX_POINTS
static constexpr size_t PC_SCRATCH_BYTES = workspace_bytes_for(PC_STATIC_MAX_POINTS);
static __device__ __align__(256) unsigned char scratch[PC_SCRATCH_BYTES];
static unsigned char *g_scratch_ptr[64];

static int scratch_of_current_device(unsigned char **out)
{
    int dev = 0;
    PC_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return pc_fail(PC_ERR_ARG, "device ordinal %d not supported", dev);
    pthread_mutex_lock(&g_queue_lock);
    unsigned char *p = g_scratch_ptr[dev];
    if (!p) {
        hipError_t e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            pthread_mutex_unlock(&g_queue_lock);
            return pc_fail(PC_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
        }
        g_scratch_ptr[dev] = p;
    }
    pthread_mutex_unlock(&g_queue_lock);
    *out = p;
    return PC_OK;
}

// Everything pc_count_pairs_noalloc looks up once per device -- the addresses of the two built-in arrays
// (hipGetSymbolAddress also loads the code object), the device's properties and the resident grid sizes of the
// persistent kernels (occupancy queries) -- done ahead of time, so that the call itself consists of nothing but
// kernel launches and memsets on the caller's stream.
extern "C" int pc_noalloc_prepare(void)
{
    unsigned char *ws = nullptr;
    unsigned int *queue = nullptr;
    const launch_sizes *sizes = nullptr;
    int rc = scratch_of_current_device(&ws);
    if (!rc) rc = count_queue(&queue);
    if (!rc) rc = launch_sizes_of_current_device(&sizes);
    return rc;
}

extern "C" int pc_count_pairs_noalloc(const float *d_xyz, uint64_t n, int x_lo, int x_hi,
                                      float *d_sorted_xyz, uint32_t *d_cell_start,
                                      uint64_t *d_npairs, int32_t *d_status, void *stream)
{
    if (n > PC_STATIC_MAX_POINTS)
        return pc_fail(PC_ERR_ARG, "pc_count_pairs_noalloc: %llu points exceed the built-in workspace's %llu; use "
                       "pc_count_pairs_device with a workspace of pc_workspace_bytes(n)",
                       (unsigned long long)n, (unsigned long long)PC_STATIC_MAX_POINTS);
    unsigned char *ws = nullptr;
    int rc = scratch_of_current_device(&ws);
    if (rc) return rc;
    // the kernels clear the counters they need at the start of every call (bin_lds), the rest is overwritten
    return pc_count_pairs_device(d_xyz, n, x_lo, x_hi, d_sorted_xyz, d_cell_start, ws, PC_SCRATCH_BYTES, d_npairs,
                                 d_status, stream);
}

// --------------------------------------------------------------- streaming --
#define PC_TEXT_STAGES 64
// Device buffers of one pc_append_text call (text, tile numbers, record count, status); kept
// per context in a small pool so that parser threads reuse theirs from block to block.
struct text_stage {
    char *d_text;
    size_t cap;
    void *d_tiles;
    size_t tiles_cap;
    uint64_t *d_count;       // [0] records, then 4 bytes of status
};

static void text_stage_free(text_stage &t)
{
    if (t.d_text) (void)hipFree(t.d_text);
    if (t.d_tiles) (void)hipFree(t.d_tiles);
    if (t.d_count) (void)hipFree(t.d_count);
    t = text_stage{};
}

struct pc_context {
    int device;
    int x_lo, x_hi;
    pthread_rwlock_t lock;      // appends: read side while copying; growth: write side
    pthread_mutex_t reserve;    // serialises slot reservation
    float *d_xyz;
    uint64_t cap, size;
    pc_geom geo;                // geometry of the run (pc_set_geometry; the reference's by default)
    int finished;               // appends are closed (set by the first pc_finish)
    int done;                   // pc_finish has run to its end: finish_rc, npairs, ntests, ms are final
    int finish_rc;              // what that pc_finish returned; later calls return it again
    int append_rc;              // first failure of a copy into a reserved range (the range then holds no points)
    // The x-layer histogram (level X's first pass) is taken block by block as the points arrive, so that
    // it overlaps the producer's work and pc_finish starts from the scan.  layers_counted stays set while
    // every appended point has been counted under the context's current geometry.
    text_stage stages[PC_TEXT_STAGES];   // pool of pc_append_text's device buffers (under `reserve`)
    int n_stages;
    uint32_t *d_layer_count;    // [NREP][404]
    int32_t *d_status;          // bit 0: a point outside the grid (set by any kernel of the context)
    int layers_counted;
    char finish_msg[256], append_msg[256];
    int64_t npairs, ntests;
    float ms[2];
    // pc_export_layers: the points grouped by x layer (inside export_ws, which also holds the kernels' counters)
    void *export_ws;
    const float *export_points;
    int exported;
    uint32_t export_start[PC_L_BINS];
    int variant_bin, variant_count;      // this context's own algorithm variants, or -1: the process-wide ones
};

extern "C" int pc_create(pc_context **out, int device, uint64_t expected_points)
{
    if (!out) return pc_fail(PC_ERR_ARG, "pc_create: null out pointer");
    *out = NULL;
    int ndev = 0;
    PC_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return pc_fail(PC_ERR_ARG, "pc_create: device %d not in [0,%d)", device, ndev);
    if (expected_points > PC_MAX_POINTS) return pc_fail(PC_ERR_ARG, "pc_create: expected_points too large");
    PC_HIP(hipSetDevice(device));
    pc_context *c = (pc_context *)calloc(1, sizeof *c);
    if (!c) return pc_fail(PC_ERR_NOMEM, "pc_create: out of host memory");
    c->device = device;
    c->x_lo = 0;
    c->x_hi = PC_NLAYERS;
    c->geo = geom_default();
    c->variant_bin = c->variant_count = -1;
    c->cap = expected_points ? expected_points : (1u << 20);
    pthread_rwlock_init(&c->lock, NULL);
    pthread_mutex_init(&c->reserve, NULL);
    hipError_t e = hipMalloc((void **)&c->d_xyz, c->cap * 12);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_layer_count, PC_L_NREP * PC_L_BINS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_status, sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(c->d_layer_count, 0, PC_L_NREP * PC_L_BINS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(c->d_status, 0, sizeof(int32_t));
    if (e != hipSuccess) {
        if (c->d_xyz) (void)hipFree(c->d_xyz);
        if (c->d_layer_count) (void)hipFree(c->d_layer_count);
        if (c->d_status) (void)hipFree(c->d_status);
        free(c);
        return pc_fail(PC_ERR_NOMEM, "pc_create: hipMalloc(%llu points): %s", (unsigned long long)expected_points,
                       hipGetErrorString(e));
    }
    c->layers_counted = 1;
    *out = c;
    return PC_OK;
}

extern "C" int pc_set_slab(pc_context *c, int x_lo, int x_hi)
{
    if (!c) return pc_fail(PC_ERR_ARG, "pc_set_slab: null context");
    int rc = check_slab("pc_set_slab", x_lo, x_hi);
    if (rc) return rc;
    c->x_lo = x_lo;
    c->x_hi = x_hi;
    return PC_OK;
}

extern "C" int pc_context_set_variant(pc_context *c, const char *name, int value)
{
    if (!c) return pc_fail(PC_ERR_ARG, "pc_context_set_variant: null context");
    int which = -1;
    int rc = check_variant("pc_context_set_variant", name, value, &which);
    if (rc) return rc;
    if (which < 0) return pc_fail(PC_ERR_ARG, "pc_context_set_variant: '%s' is not a per-context choice (\"bin\" and \"count\" are)", name);
    pthread_rwlock_wrlock(&c->lock);
    if (c->finished) rc = pc_fail(PC_ERR_STATE, "pc_context_set_variant after pc_finish");
    else if (which == 0) c->variant_bin = value;
    else c->variant_count = value;
    pthread_rwlock_unlock(&c->lock);
    return rc;
}

extern "C" int pc_set_geometry(pc_context *c, const pc_geometry *g)
{
    if (!c) return pc_fail(PC_ERR_ARG, "pc_set_geometry: null context");
    pc_geom geo;
    int rc = make_geom(g, &geo);
    if (rc) return rc;
    pthread_rwlock_wrlock(&c->lock);
    if (c->finished) rc = pc_fail(PC_ERR_STATE, "pc_set_geometry after pc_finish");
    else {
        c->geo = geo;
        if (c->size) __atomic_store_n(&c->layers_counted, 0, __ATOMIC_RELAXED);     // points counted under the old geometry: pc_finish counts afresh
    }
    pthread_rwlock_unlock(&c->lock);
    return rc;
}

// Reserve [off, off+n) in the device array, growing it if needed; returns with
// the read lock held so the array cannot move while the caller copies.
static int reserve_slots(pc_context *c, uint64_t n, uint64_t *off)
{
    for (;;) {
        pthread_rwlock_rdlock(&c->lock);
        pthread_mutex_lock(&c->reserve);
        if (c->finished) {
            pthread_mutex_unlock(&c->reserve);
            pthread_rwlock_unlock(&c->lock);
            return pc_fail(PC_ERR_STATE, "append after pc_finish");
        }
        if (c->size + n <= c->cap) {
            *off = c->size;
            c->size += n;
            pthread_mutex_unlock(&c->reserve);
            return PC_OK;                       // read lock stays held
        }
        uint64_t need = c->size + n;
        pthread_mutex_unlock(&c->reserve);
        pthread_rwlock_unlock(&c->lock);
        if (need > PC_MAX_POINTS) return pc_fail(PC_ERR_ARG, "more than %llu points", (unsigned long long)PC_MAX_POINTS);

        pthread_rwlock_wrlock(&c->lock);        // no copy in flight now
        if (need > c->cap) {
            uint64_t ncap = c->cap * 2 > need ? c->cap * 2 : need;
            if (ncap > PC_MAX_POINTS) ncap = PC_MAX_POINTS;
            float *nd = NULL;
            hipError_t e = hipMalloc((void **)&nd, ncap * 12);
            if (e == hipSuccess && c->size)
                e = hipMemcpy(nd, c->d_xyz, c->size * 12, hipMemcpyDeviceToDevice);
            if (e != hipSuccess) {
                if (nd) (void)hipFree(nd);
                pthread_rwlock_unlock(&c->lock);
                return pc_fail(PC_ERR_NOMEM, "growing the point array to %llu: %s", (unsigned long long)ncap,
                               hipGetErrorString(e));
            }
            (void)hipFree(c->d_xyz);
            c->d_xyz = nd;
            c->cap = ncap;
        }
        pthread_rwlock_unlock(&c->lock);
    }
}

// A copy into an already reserved range failed: the range stays in the array but holds
// no points, so the context can no longer produce a count.  pc_finish returns this error.
static int append_failed(pc_context *c, const char *fmt, const char *what)
{
    const int rc = pc_fail(PC_ERR_HIP, fmt, what);
    pthread_mutex_lock(&c->reserve);
    if (!c->append_rc) {
        c->append_rc = rc;
        snprintf(c->append_msg, sizeof c->append_msg, "%s", g_err);
    }
    pthread_mutex_unlock(&c->reserve);
    return rc;
}

// Level X's histogram of the block [off, off+n) that has just been copied in (the caller holds the read
// lock: the array cannot move).  The kernel is queued behind the copy -- on the null stream after a blocking
// copy, else on the copy's stream -- and nobody waits for it before pc_finish.  Only the default binning takes
// the counts; a launch failure merely switches the short cut off.
static void count_layers_of(pc_context *c, uint64_t off, uint64_t n, hipStream_t s)
{
    if (!__atomic_load_n(&c->layers_counted, __ATOMIC_RELAXED) || n == 0) return;
    const uint64_t tiles = (off + n + PC_L_TILE - 1) / PC_L_TILE - off / PC_L_TILE;
    const unsigned grid = (unsigned)((tiles < 2048 ? tiles : 2048) + PC_L_NREP - 1) / PC_L_NREP * PC_L_NREP;
    hipLaunchKernelGGL(pc_l1_hist_x_range, dim3(grid), dim3(PC_L_TPB), 0, s, c->d_xyz, off, off + n, c->d_layer_count,
                       c->d_status, c->geo);
    if (hipGetLastError() != hipSuccess) __atomic_store_n(&c->layers_counted, 0, __ATOMIC_RELAXED);
}

extern "C" int pc_append(pc_context *c, const float *xyz, uint64_t n)
{
    if (!c) return pc_fail(PC_ERR_ARG, "pc_append: null context");
    if (n == 0) return PC_OK;
    if (!xyz) return pc_fail(PC_ERR_ARG, "pc_append: null points");
    PC_HIP(hipSetDevice(c->device));            // the device is per-thread state
    uint64_t off;
    int rc = reserve_slots(c, n, &off);
    if (rc) return rc;
    hipError_t e = hipMemcpy(c->d_xyz + 3 * off, xyz, n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) count_layers_of(c, off, n, nullptr);
    pthread_rwlock_unlock(&c->lock);
    if (e != hipSuccess) return append_failed(c, "pc_append: hipMemcpy: %s", hipGetErrorString(e));
    return PC_OK;
}

// Page-locked host memory for the blocks a producer hands to pc_append: the copy is
// then one DMA instead of a staged copy through the runtime's own pinned buffers.
extern "C" int pc_host_alloc(void **p, uint64_t bytes)
{
    if (!p) return pc_fail(PC_ERR_ARG, "pc_host_alloc: null pointer");
    *p = NULL;
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return pc_fail(PC_ERR_NOMEM, "pc_host_alloc(%llu): %s", (unsigned long long)bytes, hipGetErrorString(e));
    return PC_OK;
}

extern "C" int pc_host_free(void *p)
{
    if (!p) return PC_OK;
    PC_HIP(hipHostFree(p));
    return PC_OK;
}

extern "C" int pc_append_device(pc_context *c, const float *d_xyz, uint64_t n, void *stream)
{
    if (!c) return pc_fail(PC_ERR_ARG, "pc_append_device: null context");
    if (n == 0) return PC_OK;
    if (!d_xyz) return pc_fail(PC_ERR_ARG, "pc_append_device: null points");
    PC_HIP(hipSetDevice(c->device));
    uint64_t off;
    int rc = reserve_slots(c, n, &off);
    if (rc) return rc;
    hipError_t e = hipMemcpyAsync(c->d_xyz + 3 * off, d_xyz, n * 12, hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e == hipSuccess) count_layers_of(c, off, n, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    pthread_rwlock_unlock(&c->lock);
    if (e != hipSuccess) return append_failed(c, "pc_append_device: copy: %s", hipGetErrorString(e));
    return PC_OK;
}

// ---------------------------------------------------------------- tokenizer --
static constexpr uint64_t tk_tiles(uint64_t bytes) { return (bytes + PC_TK_TILE - 1) / PC_TK_TILE; }

extern "C" size_t pc_parse_workspace_bytes(uint64_t bytes) { return align_up((size_t)tk_tiles(bytes) * 4 + 4, 256); }

// T1 + scan: *d_npoints = records in the text; the workspace then holds every tile's first record number
static int tk_count(const char *d_text, uint64_t bytes, uint64_t *d_npoints, void *d_workspace, hipStream_t s)
{
    const uint64_t tiles = tk_tiles(bytes);
    PC_LAUNCH("pc_t1_count_records", pc_t1_count_records, dim3((unsigned)tiles), dim3(PC_TK_TPB), s,
              (const unsigned char *)d_text, bytes, (uint32_t *)d_workspace);
    PC_LAUNCH("pc_t1_scan_tiles", pc_t1_scan_tiles, dim3(1), dim3(1024), s, (uint32_t *)d_workspace, (uint32_t)tiles,
              (unsigned long long *)d_npoints);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

static int tk_parse(const char *d_text, uint64_t bytes, float *d_xyz, uint64_t capacity, int32_t *d_status,
                    const void *d_workspace, hipStream_t s)
{
    PC_LAUNCH("pc_t2_parse_records", pc_t2_parse_records, dim3((unsigned)tk_tiles(bytes)), dim3(PC_TK_TPB), s,
              (const unsigned char *)d_text, bytes, (const uint32_t *)d_workspace, d_xyz, capacity, d_status);
    PC_HIP(hipGetLastError());
    return PC_OK;
}

static int tk_check_args(const char *who, const char *d_text, uint64_t bytes)
{
    if (!d_text && bytes) return pc_fail(PC_ERR_ARG, "%s: null text", who);
    if ((uintptr_t)d_text & 15u) return pc_fail(PC_ERR_ARG, "%s: the text must be 16-byte aligned", who);
    if (tk_tiles(bytes) > 0x7FFFFFFFull) return pc_fail(PC_ERR_ARG, "%s: more than %llu bytes in one call", who,
                                                        (unsigned long long)0x7FFFFFFFull * PC_TK_TILE);
    return PC_OK;
}

extern "C" int pc_parse_text_device(const char *d_text, uint64_t bytes, float *d_xyz, uint64_t capacity,
                                    uint64_t *d_npoints, int32_t *d_status, void *d_workspace,
                                    size_t workspace_bytes, void *stream)
{
    if (!d_npoints || !d_status || !d_workspace) return pc_fail(PC_ERR_ARG, "pc_parse_text_device: null pointer");
    if (!d_xyz && capacity) return pc_fail(PC_ERR_ARG, "pc_parse_text_device: null point array");
    int rc = tk_check_args("pc_parse_text_device", d_text, bytes);
    if (rc) return rc;
    if (workspace_bytes < pc_parse_workspace_bytes(bytes))
        return pc_fail(PC_ERR_ARG, "pc_parse_text_device: workspace %zu < %zu bytes", workspace_bytes, pc_parse_workspace_bytes(bytes));
    hipStream_t s = (hipStream_t)stream;
    if (bytes == 0) {
        PC_HIP(hipMemsetAsync(d_npoints, 0, 8, s));
        return PC_OK;
    }
    rc = tk_count(d_text, bytes, d_npoints, d_workspace, s);
    if (rc) return rc;
    return tk_parse(d_text, bytes, d_xyz, capacity, d_status, d_workspace, s);
}

static int tk_status_error(int32_t st)
{
    if (st & PC_TK_ERR_NODOT) return pc_fail(PC_ERR_TEXT, "Invalid number format: expected '.' after the first digit");
    if (st & PC_TK_ERR_POSEXP) return pc_fail(PC_ERR_TEXT, "Positive decimal exponents are not supported");
    if (st & PC_TK_ERR_LONG) return pc_fail(PC_ERR_TEXT, "a record is longer than %d bytes", PC_TK_TAIL);
    return pc_fail(PC_ERR_TEXT, "tokenizer status %d", (int)st);
}

extern "C" int pc_append_text(pc_context *c, const char *text, uint64_t bytes, uint64_t *n_points)
{
    if (n_points) *n_points = 0;
    if (!c) return pc_fail(PC_ERR_ARG, "pc_append_text: null context");
    if (bytes == 0) return PC_OK;
    if (!text) return pc_fail(PC_ERR_ARG, "pc_append_text: null text");
    if (tk_tiles(bytes) > 0x7FFFFFFFull) return pc_fail(PC_ERR_ARG, "pc_append_text: block too large");
    PC_HIP(hipSetDevice(c->device));
    // a stage from the pool (or a new one), grown to this block
    text_stage t{};
    pthread_mutex_lock(&c->reserve);
    if (c->n_stages) t = c->stages[--c->n_stages];
    pthread_mutex_unlock(&c->reserve);
    struct give_back {
        pc_context *c;
        text_stage *t;
        ~give_back()
        {
            pthread_mutex_lock(&c->reserve);
            const bool room = c->n_stages < PC_TEXT_STAGES;
            if (room) c->stages[c->n_stages++] = *t;
            pthread_mutex_unlock(&c->reserve);
            if (!room) text_stage_free(*t);
        }
    } back{c, &t};
    const size_t want_tiles = pc_parse_workspace_bytes(bytes);
    if (t.cap < bytes) {
        if (t.d_text) (void)hipFree(t.d_text);
        t.d_text = nullptr;
        t.cap = 0;
        const size_t cap = align_up((size_t)bytes + bytes / 8, 1 << 20);
        if (hipMalloc((void **)&t.d_text, cap) != hipSuccess) return pc_fail(PC_ERR_NOMEM, "pc_append_text: %zu bytes of device memory", cap);
        t.cap = cap;
    }
    if (t.tiles_cap < want_tiles) {
        if (t.d_tiles) (void)hipFree(t.d_tiles);
        t.d_tiles = nullptr;
        t.tiles_cap = 0;
        if (hipMalloc(&t.d_tiles, want_tiles * 2) != hipSuccess) return pc_fail(PC_ERR_NOMEM, "pc_append_text: device memory");
        t.tiles_cap = want_tiles * 2;
    }
    if (!t.d_count && hipMalloc((void **)&t.d_count, 16) != hipSuccess) return pc_fail(PC_ERR_NOMEM, "pc_append_text: device memory");
    hipStream_t s = nullptr;
    PC_HIP(hipMemcpy(t.d_text, text, bytes, hipMemcpyHostToDevice));
    PC_HIP(hipMemsetAsync(t.d_count, 0, 16, s));
    int rc = tk_count(t.d_text, bytes, t.d_count, t.d_tiles, s);
    if (rc) return rc;
    uint64_t host_count = 0;
    PC_HIP(hipMemcpy(&host_count, t.d_count, 8, hipMemcpyDeviceToHost));
    if (host_count == 0) return PC_OK;
    uint64_t off;
    rc = reserve_slots(c, host_count, &off);           // returns with the read lock held
    if (rc) return rc;
    int32_t *d_st = reinterpret_cast<int32_t *>(t.d_count + 1);
    rc = tk_parse(t.d_text, bytes, c->d_xyz + 3 * off, host_count, d_st, t.d_tiles, s);
    int32_t st = 0;
    hipError_t e = rc ? hipSuccess : hipMemcpy(&st, d_st, 4, hipMemcpyDeviceToHost);
    if (!rc && e == hipSuccess && st == 0) count_layers_of(c, off, host_count, s);
    pthread_rwlock_unlock(&c->lock);
    // the slots are reserved: whatever went wrong, the context can no longer produce the file's count
    if (rc) return append_failed(c, "pc_append_text: %s", g_err);
    if (e != hipSuccess) return append_failed(c, "pc_append_text: hipMemcpy: %s", hipGetErrorString(e));
    if (st) {
        const int code = tk_status_error(st);
        pthread_mutex_lock(&c->reserve);
        if (!c->append_rc) {
            c->append_rc = code;
            snprintf(c->append_msg, sizeof c->append_msg, "%s", g_err);
        }
        pthread_mutex_unlock(&c->reserve);
        return code;
    }
    if (n_points) *n_points = host_count;
    return PC_OK;
}

// ------------------------------------------------------- routing on the devices --
// Level X of the binning as a service: the context's points grouped by x layer, so that a multi-GPU host hands
// each slab owner its layers as ONE contiguous range per source context (a peer copy) instead of routing point
// by point on the host.  The order inside a layer is unspecified.
extern "C" int pc_export_layers(pc_context *c, const float **d_points, uint32_t *layer_start)
{
    if (!c || !d_points || !layer_start) return pc_fail(PC_ERR_ARG, "pc_export_layers: null pointer");
    *d_points = nullptr;
    pthread_rwlock_wrlock(&c->lock);            // waits for copies in flight
    struct unlock_on_exit {
        pthread_rwlock_t *l;
        ~unlock_on_exit() { pthread_rwlock_unlock(l); }
    } guard{&c->lock};
    if (c->append_rc) return pc_fail(c->append_rc, "an earlier append failed (%s): the context holds no usable point set", c->append_msg);
    c->finished = 1;                             // closed for appends: the exported ranges stay what they are
    if (!c->exported) {
        PC_HIP(hipSetDevice(c->device));
        geo_scope scope(c->geo);
        const uint64_t n = c->size;
        if (!c->export_ws && hipMalloc(&c->export_ws, pc_workspace_bytes(n)) != hipSuccess)
            return pc_fail(PC_ERR_NOMEM, "pc_export_layers: device allocation failed for %llu points", (unsigned long long)n);
        const pc_lds_ws w = carve_lds_ws(c->export_ws, n);
        hipStream_t s = nullptr;
        const bool counted = c->layers_counted != 0;
        PC_HIP(hipMemsetAsync(w.layer_count, 0, (PC_L_NREP * PC_L_BINS + PC_L_NREP * PC_L_PSTRIDE + 8) * sizeof(uint32_t), s));
        if (counted)
            PC_HIP(hipMemcpyAsync(w.layer_count, c->d_layer_count, PC_L_NREP * PC_L_BINS * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        else {
            PC_HIP(hipMemsetAsync(c->d_status, 0, 4, s));
            if (n) PC_LAUNCH("pc_l1_hist_x", pc_l1_hist_x, dim3(grid_for(n, PC_L_TILE, 2048)), dim3(PC_L_TPB), s, c->d_xyz, n,
                             w.layer_count, c->d_status, t_geo);
        }
        PC_LAUNCH("pc_l2_scan_x", pc_l2_scan_x, dim3(1), dim3(512), s, w.layer_count, w.layer_start, w.layer_cursor,
                  w.layer_tile, w.tile_layer, w.big_count + 3, 0u, (unsigned int *)nullptr, 0u, (unsigned long long *)nullptr);
        if (n) PC_LAUNCH("pc_l3_partition_x", pc_l36_partition<0>, dim3((unsigned)((n + PC_L_TILE - 1) / PC_L_TILE)), dim3(PC_L_XTPB), s,
                         c->d_xyz, n, w.tmp, w.layer_start, w.layer_tile, w.tile_layer, w.layer_cursor, w.ycell, t_geo);
        PC_HIP(hipGetLastError());
        int32_t host_status = 0;
        PC_HIP(hipMemcpyAsync(c->export_start, w.layer_start, PC_L_BINS * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        PC_HIP(hipMemcpyAsync(&host_status, c->d_status, 4, hipMemcpyDeviceToHost, s));
        PC_HIP(hipStreamSynchronize(s));
        if (host_status & 1) return pc_fail(PC_ERR_DOMAIN, "a coordinate lies outside the reference's domain (-10,10)");
        c->export_points = w.tmp;
        c->exported = 1;
    }
    *d_points = c->export_points;
    memcpy(layer_start, c->export_start, PC_L_BINS * sizeof(uint32_t));
    return PC_OK;
}

extern "C" int pc_import_device(pc_context *c, int src_device, const float *d_xyz, uint64_t n)
{
    if (!c) return pc_fail(PC_ERR_ARG, "pc_import_device: null context");
    if (n == 0) return PC_OK;
    if (!d_xyz) return pc_fail(PC_ERR_ARG, "pc_import_device: null points");
    int ndev = 0;
    PC_HIP(hipGetDeviceCount(&ndev));
    if (src_device < 0 || src_device >= ndev) return pc_fail(PC_ERR_ARG, "pc_import_device: device %d not in [0,%d)", src_device, ndev);
    PC_HIP(hipSetDevice(c->device));
    uint64_t off;
    int rc = reserve_slots(c, n, &off);         // returns with the read lock held
    if (rc) return rc;
    hipStream_t s = nullptr;
    hipError_t e = src_device == c->device
                       ? hipMemcpyAsync(c->d_xyz + 3 * off, d_xyz, n * 12, hipMemcpyDeviceToDevice, s)
                       : hipMemcpyPeerAsync(c->d_xyz + 3 * off, c->device, d_xyz, src_device, n * 12, s);
    if (e == hipSuccess) count_layers_of(c, off, n, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pthread_rwlock_unlock(&c->lock);
    if (e != hipSuccess) return append_failed(c, "pc_import_device: copy: %s", hipGetErrorString(e));
    return PC_OK;
}

extern "C" int pc_size(pc_context *c, uint64_t *n)
{
    if (!c || !n) return pc_fail(PC_ERR_ARG, "pc_size: null pointer");
    pthread_mutex_lock(&c->reserve);
    *n = c->size;
    pthread_mutex_unlock(&c->reserve);
    return PC_OK;
}

struct finish_bufs {
    float *sorted = nullptr;
    uint32_t *table = nullptr;
    void *ws = nullptr;
    uint64_t *res = nullptr;     // [0] pairs, [1] tests
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // binning start/end, pair test start/end
    ~finish_bufs()
    {
        if (sorted) (void)hipFree(sorted);
        if (table) (void)hipFree(table);
        if (ws) (void)hipFree(ws);
        if (res) (void)hipFree(res);
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
    }
};

static pthread_mutex_t g_finish_lock[64] = {
#define PC_M4 PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER, PTHREAD_MUTEX_INITIALIZER
#define PC_M16 PC_M4, PC_M4, PC_M4, PC_M4
    PC_M16, PC_M16, PC_M16, PC_M16
#undef PC_M16
#undef PC_M4
};

static int finish_work(pc_context *c)
{
    if (c->append_rc) return pc_fail(c->append_rc, "an earlier append failed (%s): the context holds no usable point set", c->append_msg);
    PC_HIP(hipSetDevice(c->device));
    geo_scope scope(c->geo);
    struct variant_scope {
        int bin, count;
        variant_scope(int b, int k) : bin(t_variant_bin), count(t_variant_count) { t_variant_bin = b; t_variant_count = k; }
        ~variant_scope() { t_variant_bin = bin; t_variant_count = count; }
    } variants(c->variant_bin, c->variant_count);
    const uint64_t n = c->size;
    finish_bufs b;
    const size_t wsb = pc_workspace_bytes(n);
    if (hipMalloc((void **)&b.sorted, (n ? n : 1) * 12) != hipSuccess ||
        hipMalloc((void **)&b.table, (PC_NCELLS + 2) * 4) != hipSuccess ||
        hipMalloc(&b.ws, wsb) != hipSuccess || hipMalloc((void **)&b.res, 16) != hipSuccess)
        return pc_fail(PC_ERR_NOMEM, "pc_finish: device allocation failed for %llu points", (unsigned long long)n);
    for (auto &e : b.ev) PC_HIP(hipEventCreate(&e));
    hipStream_t s = nullptr;
    // one finish at a time per device: the pair test's work area is per device (see count_queue)
    struct device_turn {
        pthread_mutex_t *m;
        explicit device_turn(pthread_mutex_t *mm) : m(mm) { pthread_mutex_lock(m); }
        ~device_turn() { pthread_mutex_unlock(m); }
    } turn(&g_finish_lock[c->device & 63]);
    // whatever way this function is left, nothing it queued may still run when the lock is released and the
    // buffers are freed (declared last: destroyed first)
    struct drained {
        hipStream_t st;
        ~drained() { (void)hipStreamSynchronize(st); }
    } drain{s};
    // the histogram kernels of the appends ran on the null stream or on streams their callers have synchronised
    // (pc_append_device waits for its copy AND the kernel queued behind it): this stream is ordered behind them
    const bool counted = c->layers_counted && variant_bin() == 1;
    PC_HIP(hipMemsetAsync(b.res, 0, 16, s));
    if (!counted) PC_HIP(hipMemsetAsync(c->d_status, 0, 4, s));      // (the counts taken under another geometry may have flagged points)
    PC_HIP(hipEventRecord(b.ev[0], s));
    const bool fused = fused_wanted(n, c->x_lo, c->x_hi);
    int rc = bin_layers(c->d_xyz, n, b.sorted, b.table, b.ws, wsb, c->d_status, s, c->x_lo,
                        c->x_hi < PC_NLAYERS ? c->x_hi : PC_NBB - 1, counted ? c->d_layer_count : nullptr, c->x_lo, c->x_hi, fused);
    if (rc) return rc;
    PC_HIP(hipEventRecord(b.ev[1], s));
    // a point outside the grid leaves the sorted array and the table incomplete: stop before the pair test
    int32_t host_status = 0;
    uint32_t gate[2] = {1, 0};                       // [0] largest pencil neighbourhood, [1] the fused kernel's capacity
    PC_HIP(hipMemcpyAsync(&host_status, c->d_status, 4, hipMemcpyDeviceToHost, s));
    if (fused) PC_HIP(hipMemcpyAsync(gate, fused_gate_words(b.ws, n), 8, hipMemcpyDeviceToHost, s));
    PC_HIP(hipStreamSynchronize(s));
    if (host_status & 1)
        return pc_fail(PC_ERR_DOMAIN, "a coordinate lies outside the reference's domain (-10,10)");
    PC_HIP(hipEventRecord(b.ev[2], s));
    if (gate[0] <= gate[1]) {
        // level Z inside the pair test: no sorted array, no table; the kernel counts the reference's tests as well
        rc = count_fused(b.ws, n, c->x_lo, c->x_hi, b.res, b.res + 1, c->d_status, s);
        if (rc) return rc;
        PC_HIP(hipEventRecord(b.ev[3], s));
    } else {
        rc = count_sorted(b.sorted, b.table, n, c->x_lo, c->x_hi, b.res, s, nullptr);
        if (rc) return rc;
        PC_HIP(hipEventRecord(b.ev[3], s));
        rc = pc_count_tests_device(b.table, c->x_lo, c->x_hi, b.res + 1, s);
        if (rc) return rc;
    }
    uint64_t host_res[2];
    PC_HIP(hipMemcpyAsync(host_res, b.res, 16, hipMemcpyDeviceToHost, s));
    PC_HIP(hipMemcpyAsync(&host_status, c->d_status, 4, hipMemcpyDeviceToHost, s));
    PC_HIP(hipStreamSynchronize(s));
    if (host_status & 1)                             // (the fused kernel is the one that looks at z)
        return pc_fail(PC_ERR_DOMAIN, "a coordinate lies outside the reference's domain (-10,10)");
    PC_HIP(hipEventElapsedTime(&c->ms[0], b.ev[0], b.ev[1]));
    PC_HIP(hipEventElapsedTime(&c->ms[1], b.ev[2], b.ev[3]));
    c->npairs = (int64_t)host_res[0];
    c->ntests = (int64_t)host_res[1];
    return PC_OK;
}

// The first call closes the context for appends, does the work and records its
// outcome; every later call (and pc_tests / pc_phase_ms) reports that outcome
// again -- a failed finish stays a failure, it never turns into a count of 0.
extern "C" int pc_finish(pc_context *c, int64_t *npairs)
{
    if (!c || !npairs) return pc_fail(PC_ERR_ARG, "pc_finish: null pointer");
    pthread_rwlock_wrlock(&c->lock);            // waits for copies in flight; held to the end: one finish at a time
    c->finished = 1;
    if (!c->done) {
        c->finish_rc = finish_work(c);
        if (c->finish_rc) snprintf(c->finish_msg, sizeof c->finish_msg, "%s", g_err);
        c->done = 1;
    }
    const int rc = c->finish_rc;
    if (rc) pc_fail(rc, "%s", c->finish_msg);
    else *npairs = c->npairs;
    pthread_rwlock_unlock(&c->lock);
    return rc;
}

// outcome of the context's pc_finish for the calls that read its results
static int finished_ok(pc_context *c, const char *who)
{
    pthread_rwlock_rdlock(&c->lock);
    const int done = c->done, rc = c->finish_rc;
    if (done && rc) pc_fail(rc, "%s", c->finish_msg);
    pthread_rwlock_unlock(&c->lock);
    if (!done) return pc_fail(PC_ERR_STATE, "%s before pc_finish", who);
    return rc;
}

extern "C" int pc_tests(pc_context *c, int64_t *ntests)
{
    if (!c || !ntests) return pc_fail(PC_ERR_ARG, "pc_tests: null pointer");
    const int rc = finished_ok(c, "pc_tests");
    if (rc) return rc;
    *ntests = c->ntests;
    return PC_OK;
}

extern "C" int pc_phase_ms(pc_context *c, float ms[2])
{
    if (!c || !ms) return pc_fail(PC_ERR_ARG, "pc_phase_ms: null pointer");
    const int rc = finished_ok(c, "pc_phase_ms");
    if (rc) return rc;
    ms[0] = c->ms[0];
    ms[1] = c->ms[1];
    return PC_OK;
}

extern "C" void pc_destroy(pc_context *c)
{
    if (!c) return;
    if (hipSetDevice(c->device) == hipSuccess) {
        if (c->d_xyz) (void)hipFree(c->d_xyz);
        if (c->d_layer_count) (void)hipFree(c->d_layer_count);
        if (c->d_status) (void)hipFree(c->d_status);
        if (c->export_ws) (void)hipFree(c->export_ws);
        for (int i = 0; i < c->n_stages; i++) text_stage_free(c->stages[i]);
    }
    pthread_rwlock_destroy(&c->lock);
    pthread_mutex_destroy(&c->reserve);
    free(c);
}

#ifdef PC_L_TRACE
// Measurement build only: reads and clears the partition kernels' phase counters (see pc_bin_lds.h).
extern "C" int pc_trace_read(unsigned long long out[16])
{
    static unsigned long long rows[2][PC_L_TRACE_ROWS][8];
    PC_HIP(hipDeviceSynchronize());
    PC_HIP(hipMemcpyFromSymbol(rows, HIP_SYMBOL(pc_l_trace), sizeof rows));
    for (int l = 0; l < 2; l++)
        for (int i = 0; i < 8; i++) {
            out[8 * l + i] = 0;
            for (int r = 0; r < PC_L_TRACE_ROWS; r++) out[8 * l + i] += rows[l][r][i];
        }
    memset(rows, 0, sizeof rows);
    PC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pc_l_trace), rows, sizeof rows));
    return PC_OK;
}
#endif

#ifdef PC_G_TRACE
// Measurement build only: reads and clears the group kernel's per-part cycle and window counters (see pc_count_group.h).
extern "C" int pc_group_trace_read(unsigned long long out[32])
{
    static unsigned long long rows[64][32];
    PC_HIP(hipDeviceSynchronize());
    PC_HIP(hipMemcpyFromSymbol(rows, HIP_SYMBOL(pc_g_trace), sizeof rows));
    for (int i = 0; i < 32; i++) {
        out[i] = 0;
        for (int r = 0; r < 64; r++) out[i] += rows[r][i];
    }
    memset(rows, 0, sizeof rows);
    PC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(pc_g_trace), rows, sizeof rows));
    return PC_OK;
}
#endif
