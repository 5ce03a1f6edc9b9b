/*
 * xyzhost.c -- host-side helpers of the product, built as libxyzhost.so:
 *   - multithreaded parsing of a .xyz text buffer with the reference's thread
 *     split (main.c:109-129) and tokenizer semantics (xyz_parse.h)
 *   - a bit-reproducible synthetic input generator (uniform / Gaussian blobs)
 *     that uses integer arithmetic only: positions are drawn in micro-units,
 *     written as "[-]d.dddddd" text, and their binary32 value is by definition
 *     what the tokenizer returns for that text
 *   - x-slab routing of points for sharded runs
 * Nothing here touches the GPU.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xyz_parse.h"
#include "xyz_route.h"

#define XYZ_MAX_THREADS 256

/* ---- cell routing (host copy of the key formula, reference main.c:134) ---- */
int xyz_cell_index(float f) { return (int)(((double)f + 10.0) / 0.05); }

/* Number of binary32 bit patterns in [first, first + count) for which the routing shortcut
   (xyz_route.h) and the reference's expression name different layers. */
uint64_t xyz_selfcheck_route_layers(uint64_t first, uint64_t count)
{
    uint64_t bad = 0;
    for (uint64_t b = first; b < first + count; b++) {
        const uint32_t bits = (uint32_t)b;
        float f;
        memcpy(&f, &bits, 4);
        bad += xyz_route_layer(f) != xyz_route_layer_exact(f);
    }
    return bad;
}

/* ---- tokenizer exports (tests bind these) --------------------------------- */
uint32_t xyz_token_bits(const char *tok, int *consumed, int *err)
{
    const char *e = tok;
    int er = 0;
    uint32_t b = xyz_parse_float_bits(tok, &e, &er);
    if (consumed) *consumed = (int)(e - tok);
    if (err) *err = er;
    return b;
}

/* Compares the short-cut tokenizer with the general one on every token of the
   shape it handles, "[-]d.dddddd" followed by a blank or control character
   (2 * 10^7 tokens), and on the same digits followed by characters that must
   send it to the general function.  Returns the number of disagreements. */
uint64_t xyz_selfcheck_fast_tokens(void)
{
    static const char follow[8] = {' ', '\n', '\r', '\t', '7', 'e', 'x', 0};
    uint64_t bad = 0;
    char buf[24];
    for (int neg = 0; neg < 2; neg++) {
        for (uint32_t v = 0; v < 10000000u; v++) {
            int k = 0;
            if (neg) buf[k++] = '-';
            char dg[7];
            uint32_t d = v;
            for (int i = 6; i >= 0; i--) { dg[i] = (char)('0' + d % 10u); d /= 10u; }
            buf[k++] = dg[0];
            buf[k++] = '.';
            for (int i = 1; i < 7; i++) buf[k++] = dg[i];
            /* mostly the four blanks; every 64th token gets one of the other followers */
            const char f = follow[(v & 63u) == 63u ? 4 + ((v >> 6) & 3u) : (v & 3u)];
            buf[k] = f;
            memcpy(buf + k + 1, "-3 \n", 5);               /* something readable after it (an exponent if f is 'e') */
            const char *e1, *e2;
            int r1 = 0, r2 = 0;
            const uint32_t a = xyz_parse_float_bits(buf, &e1, &r1);
            const uint32_t b = xyz_parse_float_bits_fast(buf, &e2, &r2);
            bad += (a != b) | (e1 != e2) | (r1 != r2);
        }
    }
    return bad;
}

/* Random lines over an alphabet of digits, '.', '-', 'e', blanks, control and
   other bytes (well-formed six-decimal tokens mixed in): the bounded line
   parser, which tries the short cut, must return the same position, error and
   floats as the general one.  Returns the number of disagreements. */
uint64_t xyz_selfcheck_line_fuzz(uint64_t seed, uint64_t lines)
{
    static const char alphabet[] = "0123456789012345678901234567890123456789....---ee  \n\r\t+x\x80\xff";
    uint64_t bad = 0, state = seed * 2 + 1;
    char buf[160];
    for (uint64_t it = 0; it < lines; it++) {
        memset(buf, 0, sizeof buf);
        int k = 0;
        for (int tok = 0; tok < 3; tok++) {
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            uint64_t r = state >> 20;
            if (r & 3) {                                  /* mostly well formed, then damaged below */
                if (r & 4) buf[k++] = '-';
                r >>= 3;
                buf[k++] = (char)('0' + r % 10); r /= 10;
                buf[k++] = '.';
                for (int d = 0; d < 6; d++) { buf[k++] = (char)('0' + r % 10); r /= 10; }
            } else {
                const int len = 1 + (int)((r >> 2) % 12);
                r >>= 6;
                for (int d = 0; d < len; d++) { buf[k++] = alphabet[r % (sizeof alphabet - 1)]; r = r * 31 + (r >> 7) + 1; }
            }
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            r = state >> 24;
            if ((r & 15) == 0) buf[(r >> 4) % (uint64_t)k] = alphabet[(r >> 12) % (sizeof alphabet - 1)];   /* damage one byte */
            buf[k++] = tok < 2 ? ' ' : '\n';
        }
        float a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
        int ea = 0, eb = 0;
        const char *pa = xyz_parse_line(buf, a, &ea);
        const char *pb = xyz_parse_line_bounded(buf, buf + 100, b, &eb);
        bad += (pa != pb) | (ea != eb) | (!ea && memcmp(a, b, 12) != 0);
    }
    return bad;
}

typedef struct {
    const char *file;
    size_t size;
    int t, nt;
    float *out;          /* thread-private growing buffer */
    uint64_t n, cap;
    int err;
} parse_job;

static void *parse_worker(void *arg)
{
    parse_job *j = (parse_job *)arg;
    const char *p, *stop;
    xyz_thread_range(j->file, j->size, j->t, j->nt, &p, &stop);
    while (p < stop) {
        if (j->n == j->cap) {
            uint64_t nc = j->cap ? j->cap * 2 : 4096;
            float *nb = (float *)realloc(j->out, nc * 12);
            if (!nb) { j->err = -1; return NULL; }
            j->out = nb;
            j->cap = nc;
        }
        int e = 0;
        p = xyz_parse_line_bounded(p, j->file + j->size, j->out + 3 * j->n, &e);
        if (e) { j->err = e; return NULL; }
        j->n++;
    }
    return NULL;
}

/* Parses buf[0..len) (buf[len] must be readable) with nthreads threads.
   Points land in out in file order.  Returns the number of points, -1 on a
   tokenizer error, -2 if cap is too small, -3 on OOM. */
int64_t xyz_parse_buffer_mt(const char *buf, uint64_t len, int nthreads, float *out, uint64_t cap)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > XYZ_MAX_THREADS) nthreads = XYZ_MAX_THREADS;
    parse_job jobs[XYZ_MAX_THREADS];
    pthread_t th[XYZ_MAX_THREADS];
    memset(jobs, 0, sizeof(parse_job) * (size_t)nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t].file = buf; jobs[t].size = len; jobs[t].t = t; jobs[t].nt = nthreads;
    }
    int threaded[XYZ_MAX_THREADS];
    for (int t = 1; t < nthreads; t++) {
        threaded[t] = pthread_create(&th[t], NULL, parse_worker, &jobs[t]) == 0;
        if (!threaded[t]) parse_worker(&jobs[t]);
    }
    parse_worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) if (threaded[t]) pthread_join(th[t], NULL);
    int64_t total = 0, rc = 0;
    for (int t = 0; t < nthreads; t++) {
        if (jobs[t].err) rc = jobs[t].err < 0 ? -3 : -1;
        total += (int64_t)jobs[t].n;
    }
    if (!rc && (uint64_t)total > cap) rc = -2;
    uint64_t off = 0;
    for (int t = 0; t < nthreads; t++) {
        if (!rc && jobs[t].n) memcpy(out + 3 * off, jobs[t].out, jobs[t].n * 12);
        off += jobs[t].n;
        free(jobs[t].out);
    }
    return rc ? rc : total;
}

/* ---- generator -------------------------------------------------------------- */
#define XYZ_KIND_UNIFORM 0
#define XYZ_KIND_BLOBS 1
#define XYZ_KIND_RAMP 2           /* every micro-unit value in turn (checks of the conversion, not a workload) */
#define XYZ_SPAN 9999000          /* coordinates in [-9.999000, 9.999000] */
#define XYZ_NBLOBS 64

static inline uint64_t xyz_mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint64_t xyz_draw(uint64_t seed, uint64_t i, unsigned j)
{
    return xyz_mix(xyz_mix(seed) + i * 16u + j);
}

/* micro-unit coordinates of point i of stream (kind, seed, sigma_micro) */
void xyz_point_micro(int kind, uint64_t seed, uint64_t i, int64_t sigma_micro, int64_t m[3])
{
    if (kind == XYZ_KIND_UNIFORM) {
        for (unsigned a = 0; a < 3; a++)
            m[a] = (int64_t)(xyz_draw(seed, i, a) % (2u * XYZ_SPAN + 1u)) - XYZ_SPAN;
        return;
    }
    if (kind == XYZ_KIND_RAMP) {
        for (unsigned a = 0; a < 3; a++)
            m[a] = (int64_t)((i * 3u + a) % (2u * XYZ_SPAN + 1u)) - XYZ_SPAN;
        return;
    }
    /* Gaussian blobs: 64 centres uniform in (-8,8)^3; each offset is a sum of 12
       16-bit uniforms (Irwin-Hall, std = 65536 counts) scaled to sigma */
    const uint64_t blob = xyz_draw(seed, i, 3) % XYZ_NBLOBS;
    uint64_t w[9];
    for (unsigned q = 0; q < 9; q++) w[q] = xyz_draw(seed, i, 4 + q);
    for (unsigned a = 0; a < 3; a++) {
        const int64_t centre = (int64_t)(xyz_mix(seed ^ (0xC0FFEEull + blob * 4u + a)) % 16000001u) - 8000000;
        int64_t s = 0;
        for (unsigned q = 0; q < 12; q++) {
            const unsigned idx = a * 12 + q;
            s += (int64_t)((w[idx >> 2] >> (16 * (idx & 3))) & 0xFFFF);
        }
        s -= 393210;                                   /* 12 * 32767.5 */
        int64_t v = centre + s * sigma_micro / 65536;
        if (v > XYZ_SPAN) v = XYZ_SPAN;
        if (v < -XYZ_SPAN) v = -XYZ_SPAN;
        m[a] = v;
    }
}

/* "[-]d.dddddd" ; returns length (8 or 9) */
static inline int xyz_format_micro(int64_t m, char *dst)
{
    char *p = dst;
    uint64_t u = (uint64_t)(m < 0 ? -m : m);
    if (m < 0) *p++ = '-';
    *p++ = (char)('0' + u / 1000000u);
    *p++ = '.';
    uint32_t frac = (uint32_t)(u % 1000000u);
    for (int d = 5; d >= 0; d--) { p[d] = (char)('0' + frac % 10u); frac /= 10u; }
    p += 6;
    return (int)(p - dst);
}

static inline float xyz_micro_to_float(int64_t m)
{
    char tok[16];
    int n = xyz_format_micro(m, tok);
    tok[n] = '\n';
    tok[n + 1] = 0;
    const char *e;
    int err = 0;
    uint32_t b = xyz_parse_float_bits(tok, &e, &err);
    float f;
    memcpy(&f, &b, 4);
    return f;
}

typedef struct {
    int kind; uint64_t seed; int64_t sigma;
    uint64_t first, count;
    int x_lo, x_hi;            /* keep points with x cell in [x_lo, x_hi] (inclusive) */
    float *out; uint64_t n, cap;     /* f32 mode: private buffer when filtering, shared slice otherwise */
    int filter;
    char *text; size_t tlen, tcap;   /* text mode */
    int want_text;
    int err;
} gen_job;

static void *gen_worker(void *arg)
{
    gen_job *j = (gen_job *)arg;
    for (uint64_t i = j->first; i < j->first + j->count; i++) {
        int64_t m[3];
        xyz_point_micro(j->kind, j->seed, i, j->sigma, m);
        float x = xyz_micro_to_float(m[0]);
        if (j->filter) {
            int cx = xyz_cell_index(x);
            if (cx < j->x_lo || cx > j->x_hi) continue;
        }
        if (j->want_text) {
            if (j->tlen + 40 > j->tcap) {
                size_t nc = j->tcap ? j->tcap * 2 : (1u << 20);
                char *nb = (char *)realloc(j->text, nc);
                if (!nb) { j->err = -3; return NULL; }
                j->text = nb; j->tcap = nc;
            }
            char *p = j->text + j->tlen;
            p += xyz_format_micro(m[0], p); *p++ = ' ';
            p += xyz_format_micro(m[1], p); *p++ = ' ';
            p += xyz_format_micro(m[2], p); *p++ = '\n';
            j->tlen = (size_t)(p - j->text);
            j->n++;
        } else {
            if (j->n == j->cap) {
                if (!j->filter) { j->err = -2; return NULL; }
                uint64_t nc = j->cap ? j->cap * 2 : 65536;
                float *nb = (float *)realloc(j->out, nc * 12);
                if (!nb) { j->err = -3; return NULL; }
                j->out = nb; j->cap = nc;
            }
            float *o = j->out + 3 * j->n;
            o[0] = x;
            o[1] = xyz_micro_to_float(m[1]);
            o[2] = xyz_micro_to_float(m[2]);
            j->n++;
        }
    }
    return NULL;
}

static int64_t gen_run(int kind, uint64_t seed, int64_t sigma, uint64_t first, uint64_t count, int x_lo,
                       int x_hi, float *out, uint64_t cap, const char *path, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > XYZ_MAX_THREADS) nthreads = XYZ_MAX_THREADS;
    const int filter = !(x_lo <= 0 && x_hi >= 399);
    gen_job jobs[XYZ_MAX_THREADS];
    pthread_t th[XYZ_MAX_THREADS];
    int threaded[XYZ_MAX_THREADS];
    memset(jobs, 0, sizeof(gen_job) * (size_t)nthreads);
    const uint64_t per = (count + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        gen_job *j = &jobs[t];
        j->kind = kind; j->seed = seed; j->sigma = sigma;
        uint64_t b = (uint64_t)t * per;
        if (b > count) b = count;
        uint64_t e = b + per > count ? count : b + per;
        j->first = first + b; j->count = e - b;
        j->x_lo = x_lo; j->x_hi = x_hi; j->filter = filter;
        j->want_text = path != NULL;
        if (!path && !filter) {
            if (e > cap) return -2;
            j->out = out + 3 * b; j->cap = e - b;
        }
        threaded[t] = 0;
    }
    for (int t = 1; t < nthreads; t++) {
        threaded[t] = pthread_create(&th[t], NULL, gen_worker, &jobs[t]) == 0;
        if (!threaded[t]) gen_worker(&jobs[t]);
    }
    gen_worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) if (threaded[t]) pthread_join(th[t], NULL);
    int64_t total = 0, rc = 0;
    for (int t = 0; t < nthreads; t++) { if (jobs[t].err) rc = jobs[t].err; total += (int64_t)jobs[t].n; }
    if (path) {
        FILE *f = rc ? NULL : fopen(path, "wb");
        if (!rc && !f) rc = -4;
        for (int t = 0; t < nthreads; t++) {
            if (f && jobs[t].tlen && fwrite(jobs[t].text, 1, jobs[t].tlen, f) != jobs[t].tlen) rc = -4;
            free(jobs[t].text);
        }
        if (f && fclose(f) != 0) rc = -4;
    } else if (filter) {
        if (!rc && (uint64_t)total > cap) rc = -2;
        uint64_t off = 0;
        for (int t = 0; t < nthreads; t++) {
            if (!rc && jobs[t].n) memcpy(out + 3 * off, jobs[t].out, jobs[t].n * 12);
            off += jobs[t].n;
            free(jobs[t].out);
        }
    }
    return rc ? rc : total;
}

/* Points [first, first+count) of the stream whose x cell lies in [x_lo, x_hi]
   (inclusive; 0..399 = no filter) as float triples.  Returns points written,
   -2 if cap is too small, -3 on OOM. */
int64_t xyz_gen_f32(int kind, uint64_t seed, int64_t sigma_micro, uint64_t first, uint64_t count,
                    int x_lo, int x_hi, float *out, uint64_t cap, int nthreads)
{
    return gen_run(kind, seed, sigma_micro, first, count, x_lo, x_hi, out, cap, NULL, nthreads);
}

/* Same selection written as a .xyz text file.  Returns lines written, -4 on I/O error. */
int64_t xyz_gen_text(int kind, uint64_t seed, int64_t sigma_micro, uint64_t first, uint64_t count,
                     int x_lo, int x_hi, const char *path, int nthreads)
{
    return gen_run(kind, seed, sigma_micro, first, count, x_lo, x_hi, NULL, 0, path, nthreads);
}

/* Copies to out the points of in whose x cell lies in [x_lo, x_hi] (inclusive):
   what the rank owning home layers [x_lo, x_hi) must receive. */
int64_t xyz_filter_slab(const float *in, uint64_t n, int x_lo, int x_hi, float *out)
{
    uint64_t k = 0;
    for (uint64_t i = 0; i < n; i++) {
        int cx = xyz_cell_index(in[3 * i]);
        if (cx >= x_lo && cx <= x_hi) { memcpy(out + 3 * k, in + 3 * i, 12); k++; }
    }
    return (int64_t)k;
}

/* Population of each occupied x layer (400 counters) of stream points
   [first, first+count): what a host needs to place cost-balanced slab cuts
   before any point is routed. */
typedef struct { int kind; uint64_t seed; int64_t sigma; uint64_t first, count; uint64_t hist[400]; } hist_job;

static void *hist_worker(void *arg)
{
    hist_job *j = (hist_job *)arg;
    for (uint64_t i = j->first; i < j->first + j->count; i++) {
        int64_t m[3];
        xyz_point_micro(j->kind, j->seed, i, j->sigma, m);
        int cx = xyz_cell_index(xyz_micro_to_float(m[0]));
        if (cx >= 0 && cx < 400) j->hist[cx]++;
    }
    return NULL;
}

int xyz_layer_histogram(int kind, uint64_t seed, int64_t sigma_micro, uint64_t first, uint64_t count,
                        uint64_t *hist400, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    hist_job *jobs = (hist_job *)calloc((size_t)nthreads, sizeof(hist_job));
    pthread_t th[64];
    int threaded[64];
    if (!jobs) return -3;
    const uint64_t per = (count + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        uint64_t b = (uint64_t)t * per; if (b > count) b = count;
        uint64_t e = b + per > count ? count : b + per;
        jobs[t].kind = kind; jobs[t].seed = seed; jobs[t].sigma = sigma_micro;
        jobs[t].first = first + b; jobs[t].count = e - b;
        threaded[t] = 0;
    }
    for (int t = 1; t < nthreads; t++) {
        threaded[t] = pthread_create(&th[t], NULL, hist_worker, &jobs[t]) == 0;
        if (!threaded[t]) hist_worker(&jobs[t]);
    }
    hist_worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) if (threaded[t]) pthread_join(th[t], NULL);
    memset(hist400, 0, 400 * sizeof(uint64_t));
    for (int t = 0; t < nthreads; t++)
        for (int l = 0; l < 400; l++) hist400[l] += jobs[t].hist[l];
    free(jobs);
    return 0;
}

/* Estimated work per occupied x layer (400 numbers) of stream points [first, first+count), for placing slab
   cuts that balance TIME rather than points: the reference's test count for a home cell with n points amid
   equally full neighbours is about 13.5 n^2 (n(n-1)/2 + n * 13 n, main.c:210-229), estimated here from a
   histogram over blocks of 8x8 cells per layer (population^2 / 64 per block), plus a fixed number of
   test-equivalents per point for the binning and the pair test's per-point work. */
#define XYZ_COST_BLOCK 8
#define XYZ_COST_NB (400 / XYZ_COST_BLOCK)
typedef struct { int kind; uint64_t seed; int64_t sigma; uint64_t first, count; uint32_t *blocks; } cost_job;

static void *cost_worker(void *arg)
{
    cost_job *j = (cost_job *)arg;
    for (uint64_t i = j->first; i < j->first + j->count; i++) {
        int64_t m[3];
        xyz_point_micro(j->kind, j->seed, i, j->sigma, m);
        const int cx = xyz_cell_index(xyz_micro_to_float(m[0])), cy = xyz_cell_index(xyz_micro_to_float(m[1])),
                  cz = xyz_cell_index(xyz_micro_to_float(m[2]));
        if (cx >= 0 && cx < 400 && cy >= 0 && cy < 400 && cz >= 0 && cz < 400)
            j->blocks[((size_t)cx * XYZ_COST_NB + cy / XYZ_COST_BLOCK) * XYZ_COST_NB + cz / XYZ_COST_BLOCK]++;
    }
    return NULL;
}

int xyz_layer_cost(int kind, uint64_t seed, int64_t sigma_micro, uint64_t first, uint64_t count,
                   double per_point, double *cost400, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    const size_t nb = (size_t)400 * XYZ_COST_NB * XYZ_COST_NB;
    cost_job jobs[64];
    pthread_t th[64];
    int threaded[64];
    const uint64_t per = (count + (uint64_t)nthreads - 1) / (uint64_t)nthreads;
    for (int t = 0; t < nthreads; t++) {
        uint64_t b = (uint64_t)t * per; if (b > count) b = count;
        uint64_t e = b + per > count ? count : b + per;
        jobs[t] = (cost_job){kind, seed, sigma_micro, first + b, e - b, (uint32_t *)calloc(nb, sizeof(uint32_t))};
        if (!jobs[t].blocks) { for (int u = 0; u < t; u++) free(jobs[u].blocks); return -3; }
        threaded[t] = 0;
    }
    for (int t = 1; t < nthreads; t++) {
        threaded[t] = pthread_create(&th[t], NULL, cost_worker, &jobs[t]) == 0;
        if (!threaded[t]) cost_worker(&jobs[t]);
    }
    cost_worker(&jobs[0]);
    for (int t = 1; t < nthreads; t++) if (threaded[t]) pthread_join(th[t], NULL);
    for (int l = 0; l < 400; l++) {
        double tests = 0, pop = 0;
        for (size_t b = (size_t)l * XYZ_COST_NB * XYZ_COST_NB; b < (size_t)(l + 1) * XYZ_COST_NB * XYZ_COST_NB; b++) {
            double p = 0;
            for (int t = 0; t < nthreads; t++) p += jobs[t].blocks[b];
            pop += p;
            tests += 13.5 * p * p / (XYZ_COST_BLOCK * XYZ_COST_BLOCK);
        }
        cost400[l] = tests + per_point * pop;
    }
    for (int t = 0; t < nthreads; t++) free(jobs[t].blocks);
    return 0;
}
