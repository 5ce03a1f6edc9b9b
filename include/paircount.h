/*
 * paircount.h -- C ABI of the MI355X pair-counting library (libpaircount.so).
 *
 * The library replaces one path of Remmarg0rPA/particle-simulation: the
 * spatial hash in src/hashmap.h and the counter-thread pool in src/main.c.
 * The reference has no plugin or FFI layer; these entry points sit exactly on
 * its two internal seams (the parser's per-point sink and the
 * flatten()+count() call made by main()), so a C host binds them directly.
 *
 * Conventions: plain C types only; every function returns PC_OK (0) or a
 * negative PC_ERR_* code and never throws or exits; pc_last_error() returns a
 * thread-local message for the last failing call.  "device" pointers are HIP
 * device addresses on the context's / current GPU; `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  No function here falls back to
 * the CPU: without a usable GPU they fail with PC_ERR_HIP.
 */
#ifndef PAIRCOUNT_H
#define PAIRCOUNT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_OK 0
#define PC_ERR_ARG (-1)     /* bad argument (null pointer, size, slab range)        */
#define PC_ERR_HIP (-2)     /* a HIP runtime call failed / no GPU                   */
#define PC_ERR_DOMAIN (-3)  /* a coordinate maps outside cells [0,400) on some axis */
#define PC_ERR_NOMEM (-4)   /* host or device allocation failed                     */
#define PC_ERR_STATE (-5)   /* call order violated (append after finish, ...)       */
#define PC_ERR_TEXT (-6)    /* text the reference's parser exits on (parse.h:35-39,119-122), or a record too long */

/* Grid geometry fixed by the reference: main.c:30-32
   (BBSIZE 0.05, NBB = 403 cells per axis of which 0..399 can be occupied,
   linear key = i*NBB*NBB + j*NBB + k). */
#define PC_NBB 403
#define PC_NCELLS 65450827ull   /* 403^3 */
#define PC_NLAYERS 400          /* occupied x layers */

typedef struct pc_context pc_context;

/* Geometry of a run.  The reference compiles it in (main.c:30-31 BBSIZE 0.05 and the
   domain from -10, main.c:134-136 the cell formula, main.c:165 the threshold 0.0025);
   pc_geometry_default() is exactly that, and every entry point without a geometry
   argument uses it.  Other values go through the same arithmetic:
       cell(f) = (int)(((double)f - origin) / cell)      per axis, valid cells 0 .. ncell-1
       a pair counts if (dx*dx + dy*dy) + dz*dz <= d2max   (unfused binary32)
   Requirements: 1 <= ncell <= PC_NLAYERS (the table keeps 403 cells per axis; a point
   that maps to a cell >= ncell is outside the domain), and the cutoff must not exceed
   the cell edge: d2max <= cell*cell*(1 - 1e-5) unless all three values are the
   reference's.  There is no reference program for other geometries: their results
   are checked against a brute-force count (tests), not pinned by the reference. */
typedef struct pc_geometry {
    double origin;   /* lower bound of the domain on every axis   (reference: -10.0)  */
    double cell;     /* cell edge                                 (reference: 0.05)   */
    int32_t ncell;   /* occupied cells per axis                   (reference: 400)    */
    float d2max;     /* squared cutoff                            (reference: 0.0025f)*/
} pc_geometry;
int pc_geometry_default(pc_geometry *g);

const char *pc_last_error(void);
/* "major.minor" of this ABI and the gfx target the kernels were built for. */
const char *pc_version(void);

/* ---- streaming interface: the reference's seams ------------------------- */

/* Replaces init_hashmap(hm, capacity) (hashmap.h:31,45; called at
   main.c:316-317 with file_size/128).  `device` is the HIP device ordinal;
   `expected_points` is a capacity hint (0 = unknown; storage grows on demand). */
int pc_create(pc_context **ctx, int device, uint64_t expected_points);

/* Restricts the context to one x-slab of a sharded run: only cells whose x
   index is in [x_lo, x_hi) are home cells.  The caller appends the slab's
   points plus those of layer x_hi (the +x halo: every neighbour offset at
   main.c:195-209 has di in {0,+1}).  Default is the whole grid, [0, 400).
   No counterpart in the reference (single process). */
int pc_set_slab(pc_context *ctx, int x_lo, int x_hi);

/* Gives the context another geometry than the reference's (any time before
   pc_finish; best before the first append: blocks appended earlier were
   pre-counted under the old geometry and pc_finish then counts them again).
   PC_ERR_ARG for values outside the requirements above. */
int pc_set_geometry(pc_context *ctx, const pc_geometry *g);

/* Replaces the per-point sink insert(hm, key, val) (hashmap.h:33,94; called
   from every parser thread at main.c:137).  Appends n points given as packed
   float x,y,z triples in HOST memory.  Thread-safe: parser threads call it
   concurrently, each with its own buffer; the data is copied before return.
   The cell key (main.c:134-136) is computed by the library on the GPU; the
   first pass of the binning (the histogram of x layers) is queued on the GPU
   for each block as it arrives, behind its copy, so that it runs while the
   producers work on their next blocks and pc_finish starts from its result. */
int pc_append(pc_context *ctx, const float *xyz, uint64_t n);

/* Page-locked host memory for the blocks handed to pc_append: the copy is then
   a single DMA (about half the time of a copy from pageable memory).  Allocate
   once and reuse -- page-locking is slow.  No counterpart in the reference. */
int pc_host_alloc(void **p, uint64_t bytes);
int pc_host_free(void *p);

/* Same, for points that already live in device memory (copied device to
   device on `stream`, which must be complete before pc_finish). */
int pc_append_device(pc_context *ctx, const float *d_xyz, uint64_t n, void *stream);

/* The parser threads' whole inner loop -- parse_line + insert, main.c:132-138 --
   for a block of TEXT in host memory: the block is copied to the GPU and
   tokenized there with the reference's parse_float / parse_line semantics
   (parse.h:24-163: truncating conversion, its leading-zero scan, "e-NN"
   exponents, CRLF), bit for bit, and the points are appended.  The block must
   start at the first byte of a record and end behind the last byte of one (the
   split the reference's parser threads make at main.c:109-129); a record is
   three tokens, ends with bytes in 1..31 and holds none, and is at most 256
   bytes long.  Thread-safe like pc_append.  *n_points (may be NULL) receives
   the number of points appended.  PC_ERR_TEXT where the reference would print
   and exit (no '.' after the first digit, positive decimal exponent). */
int pc_append_text(pc_context *ctx, const char *text, uint64_t bytes, uint64_t *n_points);

/* The same tokenizer as a stateless stage on device memory: d_text (16-byte
   aligned, `bytes` long) -> packed float triples in d_xyz (room for `capacity`
   points), in file order; *d_npoints = records found (also when more than
   capacity: then nothing past capacity is written and bit 4 of *d_status is set).
   *d_status is OR-ed with 2 (no '.'), 4 (positive exponent), 8 (record longer
   than 256 bytes), 16 (capacity); the caller zeroes it.  d_workspace:
   pc_parse_workspace_bytes(bytes). */
size_t pc_parse_workspace_bytes(uint64_t bytes);
int pc_parse_text_device(const char *d_text, uint64_t bytes, float *d_xyz, uint64_t capacity,
                         uint64_t *d_npoints, int32_t *d_status, void *d_workspace,
                         size_t workspace_bytes, void *stream);

/* Multi-GPU runs without per-point routing on the host.  The parser threads append
   whole, unrouted blocks to an "arrival" context on any GPU; pc_export_layers then
   closes that context for appends, groups its points by x layer ON ITS DEVICE (the
   first level of the binning) and returns the device array and, in
   layer_start[0 .. PC_NBB] (404 entries, host memory), the first point of every
   layer: layer l is points [layer_start[l], layer_start[l+1]).  The array belongs
   to the context and lives until pc_destroy.  The owner of an x slab [lo, hi) takes
   layers lo .. hi (its halo layer included: every neighbour offset at main.c:195-209
   has di in {0,+1}) from every arrival context with one pc_import_device each -- a
   peer copy from device `src_device` (a plain device copy when it is the context's
   own) into a context that pc_set_slab has restricted to the slab -- and pc_finish
   counts it.  Both calls block until done; thread-safe like pc_append.
   PC_ERR_DOMAIN from pc_export_layers if a coordinate lies outside the grid.
   No counterpart in the reference (one process, one address space): this is the
   "insert" of main.c:132-138 for points that already sit on another GPU. */
int pc_export_layers(pc_context *ctx, const float **d_points, uint32_t *layer_start);
int pc_import_device(pc_context *ctx, int src_device, const float *d_xyz, uint64_t n);

/* Replaces flatten(hm,&keys,&size) + count(hm,keys,len) (hashmap.h:34,137;
   main.c:236,331-335): bins every appended point and counts the pairs.
   Called once, from one thread, after all appends returned.  *npairs is the
   value the reference prints with "%ld\n" (main.c:336).  Blocks until done. */
int pc_finish(pc_context *ctx, int64_t *npairs);

/* Number of dist2() evaluations the reference would make on the appended
   points (sum over home cells of n(n-1)/2 + n*sum of 13 neighbour counts,
   main.c:210-229); valid after pc_finish.  The unit of "pairs-tested/s". */
int pc_tests(pc_context *ctx, int64_t *ntests);

/* Points appended so far. */
int pc_size(pc_context *ctx, uint64_t *n);

/* Millisecond device timings of the last pc_finish: [0] binning, [1] count. */
int pc_phase_ms(pc_context *ctx, float ms[2]);

/* The reference never frees its table (process exit); a library must. */
void pc_destroy(pc_context *ctx);

/* ---- stateless device-pointer stages (no allocation, no host sync) ------- */
/* These launch on `stream` and return immediately; all pointers are device
   memory of the CURRENT device: the caller makes the data's device current
   (hipSetDevice) before the call -- the per-device work areas and launch sizes
   are looked up by the current device, not by the pointers.  They are what the streaming interface runs internally and what the
   parity tests and the benchmark drive directly. */

/* Bytes of scratch pc_bin_device / pc_count_pairs_device need for n points. */
size_t pc_workspace_bytes(uint64_t n);

/* Cell keys, main.c:134-137: keys[p] = INDEX(cell(x), cell(y), cell(z)) with
   cell(f) = (int)(((double)f + 10.0) / 0.05).  d_status (one int32, caller
   zeroes it) gets bit 0 set if any point falls outside cells [0,400). */
int pc_cell_keys_device(const float *d_xyz, uint64_t n, uint32_t *d_keys,
                        int32_t *d_status, void *stream);

/* Binning, replaces insert/atomic_insert/flatten (hashmap.h:69-153): writes
   the points grouped by ascending key to d_sorted_xyz (3n floats) and the
   table d_cell_start (PC_NCELLS+2 uint32): points of key c occupy
   [cell_start[c], cell_start[c+1]).  Order inside a cell is unspecified (as
   in the reference, whose chunk lists fill in thread arrival order). */
int pc_bin_device(const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                  uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                  int32_t *d_status, void *stream);

/* Pair test, replaces counter_thread/compare/count_chunk/dist2
   (main.c:85-91,161-234): ADDS to *d_npairs (uint64, caller zeroes) the number
   of pairs (a,b), a in a home cell with x index in [x_lo,x_hi), b later in the
   same cell or anywhere in its 13 forward neighbour cells (main.c:195-209),
   with (dx*dx+dy*dy)+dz*dz <= 0.0025f in unfused binary32.
   The kernels keep their work counters and the list of heavy-pencil segments
   in one per-device area inside the library: calls of this function (and of
   pc_count_pairs_device / pc_finish, which use it) that could overlap on one
   device -- different streams, different host threads -- must be serialised by
   the caller. */
int pc_count_sorted_device(const float *d_sorted_xyz, const uint32_t *d_cell_start,
                           uint64_t n, int x_lo, int x_hi, uint64_t *d_npairs,
                           void *stream);

/* ADDS to *d_ntests the reference's dist2() call count for home layers
   [x_lo,x_hi), from the cell table alone. */
int pc_count_tests_device(const uint32_t *d_cell_start, int x_lo, int x_hi,
                          uint64_t *d_ntests, void *stream);

/* Whole hot path on device-resident input: keys, binning, pair test;
   *d_npairs is overwritten with the count.  d_sorted_xyz / d_cell_start as
   above (outputs, kept for inspection), except that they are complete only for
   the x layers x_lo .. x_hi (the slab and its halo layer, all the pair test
   reads; with x_hi = 400 also the two empty padding layers, i.e. to the end of
   the table).  Under pc_set_variant("count", 7) they are work areas only: the
   last binning level then runs inside the pair test's LDS wherever every pencil
   neighbourhood of the slab fits it (decided on the device), and neither the
   z-sorted array nor the table is written. */
int pc_count_pairs_device(const float *d_xyz, uint64_t n, int x_lo, int x_hi,
                          float *d_sorted_xyz, uint32_t *d_cell_start,
                          void *d_workspace, size_t workspace_bytes,
                          uint64_t *d_npairs, int32_t *d_status, void *stream);

/* pc_cell_keys_device, pc_bin_device, pc_count_pairs_device and pc_selfcheck_cell_index_device with
   an explicit geometry (see pc_geometry); with pc_geometry_default() they are the
   calls above, bit for bit.  No counterpart in the reference (main.c:30-31,165
   are compile-time constants there). */
int pc_cell_keys_geometry_device(const pc_geometry *g, const float *d_xyz, uint64_t n, uint32_t *d_keys,
                                 int32_t *d_status, void *stream);
int pc_bin_geometry_device(const pc_geometry *g, const float *d_xyz, uint64_t n, float *d_sorted_xyz,
                           uint32_t *d_cell_start, void *d_workspace, size_t workspace_bytes,
                           int32_t *d_status, void *stream);
int pc_count_pairs_geometry_device(const pc_geometry *g, const float *d_xyz, uint64_t n, int x_lo, int x_hi,
                                   float *d_sorted_xyz, uint32_t *d_cell_start,
                                   void *d_workspace, size_t workspace_bytes,
                                   uint64_t *d_npairs, int32_t *d_status, void *stream);
int pc_selfcheck_cell_index_geometry_device(const pc_geometry *g, uint64_t *d_mismatches, void *stream);

/* The same call for callers that cannot allocate device memory (code under HIP
   graph capture, tight serving loops, harnesses that forbid hipMalloc): no
   workspace argument -- the scratch memory lives in a fixed array inside the
   library, one copy per GPU, held until the process ends (about 27 MB) -- and no
   allocating call is made, by the library or through it.  n is limited to
   PC_NOALLOC_MAX_POINTS; more is PC_ERR_ARG.  Every call on a GPU uses that GPU's
   one array: calls that could overlap on one device (streams, host threads) race
   and must be ordered by the caller -- that includes every other pair-test entry
   point (pc_finish, pc_count_pairs_device, pc_count_sorted_device), which share
   the pair test's per-device work area with it; consecutive calls need no
   clean-up in between.  The data's device must be current.
   The first use on a device looks a few things up (the arrays' addresses, which
   loads the code object; device properties; occupancy of the persistent kernels):
   call pc_noalloc_prepare() once per device, with that device current, BEFORE a
   stream capture or a section that must not touch the runtime -- after it, a call
   is kernel launches and memsets on `stream` only and can be captured into a HIP
   graph (with pc_profile_enable off: profiling records events).  Without the
   prepare call the first pc_count_pairs_noalloc does the look-ups itself.
   No counterpart in the reference (its hash table is malloc'ed,
   hashmap.h:45-49,73-80). */
#define PC_NOALLOC_MAX_POINTS 1048576ull
int pc_noalloc_prepare(void);
int pc_count_pairs_noalloc(const float *d_xyz, uint64_t n, int x_lo, int x_hi,
                           float *d_sorted_xyz, uint32_t *d_cell_start,
                           uint64_t *d_npairs, int32_t *d_status, void *stream);

/* Selects kernel variants (for A/B measurement and cross-checks); values are
   documented next to the kernels.  name: "bin" (0, 1) | "count" (0-5, 7; 4 = the
   default choice by density, 7 = level Z fused into the pair test wherever a
   pencil neighbourhood fits its LDS), or "segments": the
   capacity of the pair test's list of heavy-pencil segments (default and
   maximum 262144; 0 = never cut a pencil), or "heavy": the capacity of its list
   of over-full cells' home-particle chunks (default and maximum 65536; 0 = such
   cells are processed by the wave that meets them), or "dense": which kernel works
   the segment list off after the merged-window pair test (1, the default: decided
   per call on the device -- the candidates-in-registers kernel if most of the
   listed points lie in segments with at least "dense_min" (default 11) points per
   cell, else the group kernel; 0: the group kernel; 2: the candidates-in-registers
   kernel), or "clustered_share": the percentage of the points that must sit in
   pencils of more than four average pencils for an input to count as clustered
   (default 12; the group kernel then takes the whole pair test), or "even": whether the
   merged-window pair test's grid is cut down to a whole number of pencils per wave
   (1, the default; 0: as many waves as are resident).  Returns PC_ERR_ARG for an
   unknown name or value. */
int pc_set_variant(const char *name, int value);

/* The same choice of "bin" or "count" for ONE context: its pc_finish then runs that
   variant whatever pc_set_variant says at the time (two contexts of a process may
   differ).  Before pc_finish only.  The remaining names of pc_set_variant size
   process-wide work areas and stay process-wide. */
int pc_context_set_variant(pc_context *ctx, const char *name, int value);

/* Per-kernel device timing for the stage functions above: while enabled, a
   pair of HIP events is recorded on the launch stream around each kernel.
   pc_profile_enable(k), k > 1, times only launches 1, k+1, 2k+1, ... of each
   kernel (a pair of events costs a few microseconds of stream time per
   kernel).  Enabling also clears earlier samples.  pc_profile_read waits for
   the recorded events and writes one text line per kernel:
   "<kernel> <timed launches> <launches> <total ms of the timed launches>".
   A measurement aid, not thread-safe: enable it only while a single host
   thread issues stage calls.  No counterpart in the reference (its -DTIMER
   phases, main.c:19-28, are the closest; pc_phase_ms reports those). */
int pc_profile_enable(int on);
int pc_profile_read(char *buf, size_t buflen);

/* Self-check of the cell-index arithmetic: the kernels compute the index of
   main.c:134-136 by a binary32 shortcut that falls back to the reference's
   binary64 expression near cell boundaries.  This runs both over all 2^32
   float bit patterns on the current device and adds the number of patterns on
   which they differ to *d_mismatches (device memory, zeroed by the caller);
   anything but 0 is a bug.  No counterpart in the reference; used by the GPU
   test suite. */
int pc_selfcheck_cell_index_device(uint64_t *d_mismatches, void *stream);

/* Synthetic input on the device: points [first, first+count) of the stream
   (kind, seed, sigma_micro) of this repository's host generator
   (particle-simulation_amd/host/xyzhost.c: 0 = uniform, 1 = 64 Gaussian
   blobs, 2 = a ramp through every six-decimal value), bit for bit what the host
   generator writes -- i.e. what the reference tokenizer (parse.h:24-163)
   returns for the stream's text -- as packed float triples in d_xyz.  For
   benchmarks whose input would otherwise be generated on the host and copied.
   No counterpart in the reference (it has no generator). */
int pc_generate_device(int kind, uint64_t seed, int64_t sigma_micro, uint64_t first, uint64_t count,
                       float *d_xyz, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PAIRCOUNT_H */
