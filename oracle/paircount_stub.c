/*
 * paircount_stub.c -- a CPU stand-in for libpaircount.so's streaming interface.
 *
 * THIS FILE IS TEST INFRASTRUCTURE, like the oracle it calls.  It exists so that the
 * product's C host program (particle-simulation_amd/host/main.c: parser threads,
 * slab routing, the slab queue, page-locked block hand-over) can be built and run
 * under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a machine
 * without a GPU -- the counterpart of the reference's `make debug` (Makefile:12-17,
 * DEBUG=1: -fsanitize=address,undefined).  Nothing in the shipped library or host
 * program links it; oracle/Makefile (target `san`) builds oracle/_san/main_asan and
 * oracle/_san/main_tsan from it, and only tests/test_host_sanitizers.py runs them.
 *
 * Every entry point the host program calls is here with the semantics of
 * include/paircount.h; the count itself is orc_count_pairs (oracle/pairs_oracle.c).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/paircount.h"

int64_t orc_count_pairs(const float *xyz, uint64_t n, int nthreads, int x_lo, int x_hi, int64_t *tests_out,
                        float *sorted_out, uint32_t *start_out);
int64_t orc_parse_buffer(const char *buf, uint64_t len, float *xyz, uint64_t cap);
int orc_cell_index(float f);
float orc_dist2(const float *a, const float *b);
int orc_within(float d2);

/* Small inputs (the golden files): every pair looked at once -- no 403^3 table per context, which the binned
   count allocates and sweeps.  A pair belongs to the slab whose layers hold the lower-key cell of the two (the
   home cell: all 13 offsets of main.c:195-209 raise the key), counts as a test if the cells are equal or
   adjacent, and as a pair if the predicate holds as well. */
#define STUB_SMALL 4000
static int64_t count_small(const float *xyz, uint64_t n, int x_lo, int x_hi, int64_t *tests_out)
{
    int (*cell)[3] = (int (*)[3])malloc((n ? n : 1) * sizeof *cell);
    if (!cell) return -2;
    for (uint64_t p = 0; p < n; p++)
        for (int a = 0; a < 3; a++) {
            cell[p][a] = orc_cell_index(xyz[3 * p + a]);
            if (cell[p][a] < 0 || cell[p][a] >= PC_NLAYERS) { free(cell); return -1; }
        }
    int64_t pairs = 0, tests = 0;
    for (uint64_t a = 0; a < n; a++)
        for (uint64_t b = a + 1; b < n; b++) {
            int adjacent = 1, order = 0;               /* order: sign of key(a) - key(b) */
            for (int k = 0; k < 3; k++) {
                const int d = cell[a][k] - cell[b][k];
                if (d < -1 || d > 1) adjacent = 0;
                if (!order) order = d;
            }
            if (!adjacent) continue;
            const int home_x = order <= 0 ? cell[a][0] : cell[b][0];
            if (home_x < x_lo || home_x >= x_hi) continue;
            tests++;
            pairs += orc_within(orc_dist2(xyz + 3 * a, xyz + 3 * b));
        }
    free(cell);
    *tests_out = tests;
    return pairs;
}

struct pc_context {
    pthread_mutex_t lock;
    float *xyz;
    uint64_t size, cap;
    int x_lo, x_hi, finished;
    int64_t npairs, tests;
    float *by_layer;                 /* pc_export_layers: the points grouped by x layer */
    uint32_t layer_start[PC_NBB + 1];
};

static __thread char g_err[256];
const char *pc_last_error(void) { return g_err; }
const char *pc_version(void) { return "stub"; }

static int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int pc_create(pc_context **out, int device, uint64_t expected_points)
{
    if (!out || device < 0) return fail(PC_ERR_ARG, "pc_create: bad argument");
    pc_context *c = (pc_context *)calloc(1, sizeof *c);
    if (!c) return fail(PC_ERR_NOMEM, "pc_create: out of memory");
    pthread_mutex_init(&c->lock, NULL);
    c->cap = expected_points ? expected_points : 1024;
    c->xyz = (float *)malloc(c->cap * 12);
    c->x_hi = PC_NLAYERS;
    if (!c->xyz) { free(c); return fail(PC_ERR_NOMEM, "pc_create: out of memory"); }
    *out = c;
    return PC_OK;
}

int pc_set_slab(pc_context *c, int x_lo, int x_hi)
{
    if (!c || x_lo < 0 || x_hi > PC_NLAYERS || x_lo > x_hi) return fail(PC_ERR_ARG, "pc_set_slab: bad slab");
    c->x_lo = x_lo;
    c->x_hi = x_hi;
    return PC_OK;
}

int pc_append(pc_context *c, const float *xyz, uint64_t n)
{
    if (!c || (!xyz && n)) return fail(PC_ERR_ARG, "pc_append: bad argument");
    pthread_mutex_lock(&c->lock);
    if (c->finished) { pthread_mutex_unlock(&c->lock); return fail(PC_ERR_STATE, "append after pc_finish"); }
    if (c->size + n > c->cap) {
        uint64_t ncap = c->cap * 2 > c->size + n ? c->cap * 2 : c->size + n;
        float *nx = (float *)realloc(c->xyz, ncap * 12);
        if (!nx) { pthread_mutex_unlock(&c->lock); return fail(PC_ERR_NOMEM, "pc_append: out of memory"); }
        c->xyz = nx;
        c->cap = ncap;
    }
    memcpy(c->xyz + 3 * c->size, xyz, n * 12);       /* (reads every byte of the caller's block: what the sanitizers watch) */
    c->size += n;
    pthread_mutex_unlock(&c->lock);
    return PC_OK;
}

int pc_append_text(pc_context *c, const char *text, uint64_t bytes, uint64_t *n_points)
{
    if (n_points) *n_points = 0;
    if (!c || (!text && bytes)) return fail(PC_ERR_ARG, "pc_append_text: bad argument");
    char *copy = (char *)calloc(bytes + 2, 1);        /* the tokenizer looks past the end: the separator after the last token, and one more (parse.h:150-163) */
    float *pts = (float *)malloc((bytes / 6 + 1) * 12);
    if (!copy || !pts) { free(copy); free(pts); return fail(PC_ERR_NOMEM, "pc_append_text: out of memory"); }
    memcpy(copy, text, bytes);
    const int64_t n = orc_parse_buffer(copy, bytes, pts, bytes / 6 + 1);
    int rc = n < 0 ? fail(PC_ERR_TEXT, n == -2 ? "Positive decimal exponents are not supported"
                                               : "Invalid number format: expected '.' after the first digit")
                   : pc_append(c, pts, (uint64_t)n);
    if (rc == PC_OK && n_points) *n_points = (uint64_t)n;
    free(copy);
    free(pts);
    return rc;
}

/* the points grouped by x layer (a counting sort on the reference's cell index, main.c:134) */
int pc_export_layers(pc_context *c, const float **d_points, uint32_t *layer_start)
{
    if (!c || !d_points || !layer_start) return fail(PC_ERR_ARG, "pc_export_layers: null pointer");
    pthread_mutex_lock(&c->lock);
    c->finished = 1;
    if (!c->by_layer) {
        uint32_t count[PC_NBB + 1] = {0};
        for (uint64_t p = 0; p < c->size; p++) {
            const int l = orc_cell_index(c->xyz[3 * p]);
            if (l < 0 || l >= PC_NLAYERS) { pthread_mutex_unlock(&c->lock); return fail(PC_ERR_DOMAIN, "a coordinate lies outside the reference's domain (-10,10)"); }
            count[l]++;
        }
        uint32_t run = 0, cursor[PC_NBB + 1];
        for (int l = 0; l <= PC_NBB; l++) { c->layer_start[l] = cursor[l] = run; run += count[l]; }
        c->by_layer = (float *)malloc((c->size ? c->size : 1) * 12);
        if (!c->by_layer) { pthread_mutex_unlock(&c->lock); return fail(PC_ERR_NOMEM, "pc_export_layers: out of memory"); }
        for (uint64_t p = 0; p < c->size; p++)
            memcpy(c->by_layer + 3 * (size_t)cursor[orc_cell_index(c->xyz[3 * p])]++, c->xyz + 3 * p, 12);
    }
    *d_points = c->by_layer;
    memcpy(layer_start, c->layer_start, sizeof c->layer_start);
    pthread_mutex_unlock(&c->lock);
    return PC_OK;
}

int pc_import_device(pc_context *c, int src_device, const float *d_xyz, uint64_t n)
{
    if (src_device < 0) return fail(PC_ERR_ARG, "pc_import_device: bad device");
    return pc_append(c, d_xyz, n);
}

int pc_host_alloc(void **p, uint64_t bytes)
{
    if (!p) return fail(PC_ERR_ARG, "pc_host_alloc: null pointer");
    *p = malloc(bytes ? bytes : 1);                   /* exact size: an overrun of a "page-locked" block is caught */
    return *p ? PC_OK : fail(PC_ERR_NOMEM, "pc_host_alloc: out of memory");
}

int pc_host_free(void *p) { free(p); return PC_OK; }

int pc_finish(pc_context *c, int64_t *npairs)
{
    if (!c || !npairs) return fail(PC_ERR_ARG, "pc_finish: null pointer");
    pthread_mutex_lock(&c->lock);
    if (!c->finished) {
        c->finished = 1;
        c->npairs = c->size <= STUB_SMALL ? count_small(c->xyz, c->size, c->x_lo, c->x_hi, &c->tests)
                                          : orc_count_pairs(c->xyz, c->size, 2, c->x_lo, c->x_hi, &c->tests, NULL, NULL);
    }
    const int64_t r = c->npairs;
    pthread_mutex_unlock(&c->lock);
    if (r == -1) return fail(PC_ERR_DOMAIN, "a coordinate lies outside the reference's domain (-10,10)");
    if (r < 0) return fail(PC_ERR_NOMEM, "pc_finish: out of memory");
    *npairs = r;
    return PC_OK;
}

int pc_tests(pc_context *c, int64_t *ntests) { *ntests = c->tests; return PC_OK; }
int pc_size(pc_context *c, uint64_t *n) { pthread_mutex_lock(&c->lock); *n = c->size; pthread_mutex_unlock(&c->lock); return PC_OK; }
int pc_phase_ms(pc_context *c, float ms[2]) { (void)c; ms[0] = ms[1] = 0.0f; return PC_OK; }

void pc_destroy(pc_context *c)
{
    if (!c) return;
    pthread_mutex_destroy(&c->lock);
    free(c->by_layer);
    free(c->xyz);
    free(c);
}
