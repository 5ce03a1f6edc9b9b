/*
 * pairs_oracle.c -- CPU oracle for the pair-counting hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is a plain-C restatement of the
 * reference algorithm and exists only to check the HIP path.  Nothing in the
 * shipped library or host program may include, link or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  Every function below is compared, in
 * tests/golden/make_golden.py (vectors) and tests/test_oracle_golden.py (check), against the
 * unmodified reference compiled from /root/reference/src (oracle/ref_shim.c,
 * built into oracle/_ref/), and the resulting vectors are committed under
 * tests/golden/.  The reference itself ships no tests or fixtures.
 * (Only with the reference's own geometry: see orc_set_geometry.)
 *
 * Each function cites the reference file:line it restates.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
 * (x86-64: float arithmetic is SSE scalar IEEE binary32, no excess precision).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* Grid constants: reference main.c:30-32 (BBSIZE 0.05, NBB = 400+1+2 = 403). */
#define ORC_NBB 403
#define ORC_NCELLS ((int64_t)ORC_NBB * ORC_NBB * ORC_NBB)

/* ------------------------------------------------------------------------- */
/* Tokenizer: reference parse.h:24-143 (parse_float).                          */
/* Error codes replace the reference's perror+exit(-1):                        */
/*   1 = second character is not '.' (parse.h:35-39)                           */
/*   2 = positive decimal exponent   (parse.h:119-122)                         */
/* ------------------------------------------------------------------------- */
static inline int orc_isdig(char c) { return c >= '0' && c <= '9'; }

static inline int orc_clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }

uint32_t orc_parse_float_bits(const char *tok, const char **endp, int *err)
{
    const char *p = tok;
    int neg = 0;
    if (err) *err = 0;
    if (*p == '-') { neg = 1; p++; }
    /* parse.h:32  first digit seeds the integer accumulator */
    uint64_t acc = (uint64_t)(int64_t)(*p - '0');
    p++;
    if (*p != '.') {                      /* parse.h:35-39 */
        if (err) *err = 1;
        if (endp) *endp = p;
        return 0;
    }
    p++;
    /* parse.h:42-52  at most 17 decimals enter the accumulator */
    int ndec = 0;
    while (ndec < 17 && orc_isdig(*p)) {
        acc = acc * 10u + (uint64_t)(*p - '0');
        ndec++;
        p++;
    }
    while (orc_isdig(*p)) p++;            /* parse.h:54-56 surplus decimals dropped */
    int dexp = 0;
    if (*p == 'e') {                      /* parse.h:58-68 (lower-case only) */
        p++;
        dexp = atoi(p);
        p++;                              /* one char: the sign or first digit */
        while (orc_isdig(*p)) p++;
    }
    if (endp) *endp = p;
    if (acc == 0) return 0;               /* parse.h:72-74  +0.0f even for "-0.0" */

    /* parse.h:77-90  leading zeros and first non-zero digit.  The scan covers
       source characters [neg, ndec] only, so it never sees the last decimal of
       an unsigned token nor the last two of a signed one (reference quirk that
       has to be reproduced). */
    int lead0 = 0, first = 0;
    for (int i = neg; i <= ndec; i++) {
        char c = tok[i];
        if (c > '0') { first = c - '0'; break; }
        if (c == '0') lead0++;
    }
    dexp -= lead0;
    ndec -= lead0;

    /* parse.h:92-104  divide by 5 per decimal, renormalising every 16 steps */
    int e2 = 0;                           /* reference keeps this in an int8_t; only
                                             (e2+127)&0xFF is used, so mod-256
                                             arithmetic in an int is equivalent */
    uint64_t m = acc;
    for (int i = 0; i < ndec; i++) {
        if ((i & 15) == 0) m <<= orc_clz64(m);
        m /= 5u;
    }
    if (dexp < 0) {                       /* parse.h:106-118 */
        for (int i = 0; i < -dexp; i++) {
            if ((i & 15) == 0) {
                int s = orc_clz64(m);
                m <<= s;
                if (i != 0) e2 -= s;
            }
            m /= 5u;
        }
        e2 += -orc_clz64(m) + dexp;
    } else if (dexp > 0) {                /* parse.h:119-122 */
        if (err) *err = 2;
        return 0;
    }
    /* parse.h:128-133  top 24 bits -> mantissa, exponent from the first digit */
    int s = orc_clz64(m);
    int sh = 40 - s;
    uint32_t mant = (sh >= 0 && sh < 64) ? (uint32_t)(m >> sh) : 0u;
    static const int exp_of_digit[10] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3};
    e2 += exp_of_digit[first > 9 ? 9 : first];   /* (a byte above '9' in the scan: the reference reads its table out of range) */
    /* parse.h:135-140 */
    return ((uint32_t)neg << 31) | ((uint32_t)((e2 + 127) & 0xFF) << 23) | (mant & 0x7FFFFFu);
}

/* reference parse.h:150-163 (parse_line): three floats, one separator char
   after each, then skip control characters (0 < c < ' '). */
const char *orc_parse_line(const char *s, uint32_t bits[3], int *err)
{
    const char *end = s;
    for (int a = 0; a < 3; a++) {
        int e = 0;
        bits[a] = orc_parse_float_bits(s, &end, &e);
        if (e) { if (err) *err = e; return end; }
        s = end + 1;
    }
    while (*s < ' ' && *s > '\0') s++;
    return s;
}

/* Whole buffer, sequential: the reference's thread split (main.c:109-129)
   assigns every line to exactly one parser thread, so the multiset of parsed
   points equals a sequential sweep.  buf must be readable up to buf[len]
   (NUL- or newline-terminated), as the mmap'd file is in the reference.
   Returns number of points written (at most cap), or -1 on a tokenizer error. */
int64_t orc_parse_buffer(const char *buf, uint64_t len, float *xyz, uint64_t cap)
{
    const char *p = buf, *stop = buf + len;
    uint64_t n = 0;
    while (p < stop) {
        uint32_t b[3];
        int e = 0;
        p = orc_parse_line(p, b, &e);
        if (e) return -1;
        if (n < cap) memcpy(xyz + 3 * n, b, 12);
        n++;
    }
    return (int64_t)n;
}

/* ------------------------------------------------------------------------- */
/* Cell index and key: reference main.c:30-32, 134-137.                        */
/* ------------------------------------------------------------------------- */
/* The reference compiles the geometry in (main.c:30-31, 134-136, 165).  The checker can be given another one --
   same expressions, other constants -- for the product's configurable geometry; nothing of the reference
   pins those results (it has no such option): they are checked against orc_count_pairs_anycell below. */
static double orc_origin = -10., orc_cell = 0.05, orc_d2max = 0.0025;
static int orc_ncell = 400;

void orc_set_geometry(double origin, double cell, int ncell, float d2max)
{
    orc_origin = origin;
    orc_cell = cell;
    orc_ncell = ncell;
    /* the reference compares (double)d2 <= 0.0025 (the double literal); for its own threshold keep exactly that */
    orc_d2max = d2max == 0.0025f ? 0.0025 : (double)d2max;
}

int orc_cell_index(float f)
{
    return (int)((f - orc_origin) / orc_cell);    /* float promoted to double, IEEE divide; defaults: (f - (-10.)) / 0.05 */
}

int64_t orc_cell_key(float x, float y, float z)
{
    int i = orc_cell_index(x), j = orc_cell_index(y), k = orc_cell_index(z);
    return (int64_t)(i * ORC_NBB * ORC_NBB + j * ORC_NBB + k);
}

void orc_cell_keys(const float *xyz, uint64_t n, int64_t *keys)
{
    for (uint64_t p = 0; p < n; p++)
        keys[p] = orc_cell_key(xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2]);
}

/* ------------------------------------------------------------------------- */
/* Squared distance and predicate: reference main.c:85-91, 165, 216.           */
/* dpps imm 0x71: products of lanes 0..2, lane 3 replaced by +0.0, summed as   */
/* (p0+p1)+(p2+p3); every step a separately rounded binary32 operation.        */
/* ------------------------------------------------------------------------- */
float orc_dist2(const float *a, const float *b)
{
    volatile float dx = a[0] - b[0];
    volatile float dy = a[1] - b[1];
    volatile float dz = a[2] - b[2];
    volatile float p0 = dx * dx, p1 = dy * dy, p2 = dz * dz;
    volatile float lo = p0 + p1;
    volatile float hi = p2 + 0.0f;
    return lo + hi;
}

int orc_within(float d2) { return d2 <= orc_d2max; }   /* double compare, main.c:165 */

static inline int pair_hit(const float *a, const float *b)
{
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float s = dx * dx + dy * dy;          /* -ffp-contract=off: no fusion */
    s = s + dz * dz;
    return (double)s <= orc_d2max;
}

/* ------------------------------------------------------------------------- */
/* Bin + count: restates hashmap.h:94-153 (insert/flatten: group points by     */
/* key) and main.c:181-252 (counter_thread/count: for every occupied cell, all */
/* in-cell pairs plus all pairs with the 13 forward neighbour cells).          */
/* The grouping is done with a dense table over all 403^3 keys instead of a    */
/* hash map; the set of points per key, which is all the count depends on, is  */
/* the same.  Neighbour keys are formed by adding the linear offsets exactly   */
/* as main.c:195-209 does (no per-axis bounds checks; layers 400..402 are the  */
/* reference's padding and stay empty).                                        */
/* ------------------------------------------------------------------------- */
static const int32_t NB_DELTA[13] = {
    /* (1,0,0) (1,1,0) (1,1,1) (1,1,-1) (1,-1,0) (1,-1,1) (1,-1,-1) (1,0,-1) (1,0,1) */
    162409, 162812, 162813, 162811, 162006, 162007, 162005, 162408, 162410,
    /* (0,1,0) (0,1,1) (0,1,-1) (0,0,1) */
    403, 404, 402, 1};

typedef struct {
    const float *pts;          /* sorted by key, 3 floats each */
    const uint32_t *start;     /* start[key] .. start[key+1]   */
    int64_t key_lo, key_hi;    /* home keys [key_lo, key_hi)   */
    int64_t next;              /* work cursor (atomic)         */
    int64_t pairs, tests;      /* folded totals (atomic)       */
} count_job;

#define ORC_KEY_CHUNK 8192

static void *count_worker(void *arg)
{
    count_job *job = (count_job *)arg;
    const float *pts = job->pts;
    const uint32_t *start = job->start;
    int64_t pairs = 0, tests = 0;
    for (;;) {
        int64_t k0 = __atomic_fetch_add(&job->next, ORC_KEY_CHUNK, __ATOMIC_RELAXED);
        if (k0 >= job->key_hi) break;
        int64_t k1 = k0 + ORC_KEY_CHUNK < job->key_hi ? k0 + ORC_KEY_CHUNK : job->key_hi;
        for (int64_t key = k0; key < k1; key++) {
            uint32_t b = start[key], e = start[key + 1];
            if (b == e) continue;
            for (uint32_t i = b; i < e; i++) {
                const float *pi = pts + 3 * (size_t)i;
                for (uint32_t j = i + 1; j < e; j++)      /* main.c:215-221 */
                    pairs += pair_hit(pi, pts + 3 * (size_t)j);
            }
            tests += (int64_t)(e - b) * (e - b - 1) / 2;
            for (int d = 0; d < 13; d++) {                /* main.c:224-226 */
                int64_t nk = key + NB_DELTA[d];
                uint32_t nb = start[nk], ne = start[nk + 1];
                if (nb == ne) continue;
                tests += (int64_t)(e - b) * (ne - nb);
                for (uint32_t i = b; i < e; i++) {
                    const float *pi = pts + 3 * (size_t)i;
                    for (uint32_t j = nb; j < ne; j++)
                        pairs += pair_hit(pi, pts + 3 * (size_t)j);
                }
            }
        }
    }
    __atomic_fetch_add(&job->pairs, pairs, __ATOMIC_RELAXED);   /* main.c:231 */
    __atomic_fetch_add(&job->tests, tests, __ATOMIC_RELAXED);
    return NULL;
}

/*
 * Count pairs among n points.  Only cells whose x index lies in [x_lo, x_hi)
 * act as home cells (x_lo=0, x_hi=403 is the whole reference computation; a
 * narrower range restates one x-slab of a sharded run: the caller passes the
 * slab's points plus the points of layer x_hi).
 * sorted_out (3n floats) and start_out (403^3+1 entries) are optional outputs.
 * Returns the pair count, or -1 if a point lies outside cells [0,400) on some
 * axis (outside the reference's documented input domain), or -2 on OOM.
 */
int64_t orc_count_pairs(const float *xyz, uint64_t n, int nthreads, int x_lo, int x_hi,
                        int64_t *tests_out, float *sorted_out, uint32_t *start_out)
{
    const int64_t C = ORC_NCELLS;
    uint32_t *start = (uint32_t *)calloc((size_t)C + 2, sizeof(uint32_t));
    uint32_t *keys = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    float *sorted = sorted_out ? sorted_out : (float *)malloc((n ? n : 1) * 12);
    if (!start || !keys || !sorted) return -2;
    for (uint64_t p = 0; p < n; p++) {
        int i = orc_cell_index(xyz[3 * p]), j = orc_cell_index(xyz[3 * p + 1]),
            k = orc_cell_index(xyz[3 * p + 2]);
        if (i < 0 || i >= orc_ncell || j < 0 || j >= orc_ncell || k < 0 || k >= orc_ncell) {
            free(start); free(keys); if (!sorted_out) free(sorted);
            return -1;
        }
        uint32_t key = (uint32_t)(i * ORC_NBB * ORC_NBB + j * ORC_NBB + k);
        keys[p] = key;
        start[key + 1]++;
    }
    for (int64_t c = 0; c < C; c++) start[c + 1] += start[c];   /* start[c] = first slot */
    /* scatter using a moving cursor kept in place: afterwards start[c] is the END of c */
    for (uint64_t p = 0; p < n; p++) {
        uint32_t slot = start[keys[p]]++;
        memcpy(sorted + 3 * (size_t)slot, xyz + 3 * p, 12);
    }
    for (int64_t c = C; c > 0; c--) start[c] = start[c - 1];    /* shift back */
    start[0] = 0;
    start[C + 1] = (uint32_t)n;
    free(keys);

    if (x_lo < 0) x_lo = 0;
    if (x_hi > 400) x_hi = 400;           /* occupied x indices are 0..399 */
    count_job job;
    job.pts = sorted;
    job.start = start;
    job.key_lo = (int64_t)x_lo * ORC_NBB * ORC_NBB;
    job.key_hi = (int64_t)x_hi * ORC_NBB * ORC_NBB;
    job.next = job.key_lo;
    job.pairs = 0;
    job.tests = 0;
    if (job.key_hi > job.key_lo) {
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 256) nthreads = 256;
        pthread_t th[256];
        int started = 0;
        for (int t = 1; t < nthreads; t++)
            if (pthread_create(&th[started], NULL, count_worker, &job) == 0) started++;
        count_worker(&job);
        for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    }
    if (tests_out) *tests_out = job.tests;
    if (start_out) memcpy(start_out, start, ((size_t)C + 1) * sizeof(uint32_t));
    free(start);
    if (!sorted_out) free(sorted);
    return job.pairs;
}

/* Brute force over all pairs with the derived invariant
 *   count = #{a<b : max-axis |cell(a)-cell(b)| <= 1  and  dist2(a,b) <= 0.0025}
 * Independent of any binning; O(n^2), small n only. */
int64_t orc_count_pairs_bruteforce(const float *xyz, uint64_t n)
{
    int64_t cnt = 0;
    for (uint64_t a = 0; a < n; a++) {
        int ai = orc_cell_index(xyz[3 * a]), aj = orc_cell_index(xyz[3 * a + 1]),
            ak = orc_cell_index(xyz[3 * a + 2]);
        for (uint64_t b = a + 1; b < n; b++) {
            int di = ai - orc_cell_index(xyz[3 * b]);
            int dj = aj - orc_cell_index(xyz[3 * b + 1]);
            int dk = ak - orc_cell_index(xyz[3 * b + 2]);
            if (di < -1 || di > 1 || dj < -1 || dj > 1 || dk < -1 || dk > 1) continue;
            cnt += orc_within(orc_dist2(xyz + 3 * a, xyz + 3 * b));
        }
    }
    return cnt;
}

/* Brute force over ALL pairs, no cell condition: what "pairs within the cutoff" means when no grid is given.
 * Equal to the stencil count whenever the cutoff does not exceed the cell edge (then points two cells apart
 * cannot be within it). */
int64_t orc_count_pairs_anycell(const float *xyz, uint64_t n)
{
    int64_t cnt = 0;
    for (uint64_t a = 0; a < n; a++)
        for (uint64_t b = a + 1; b < n; b++) cnt += orc_within(orc_dist2(xyz + 3 * a, xyz + 3 * b));
    return cnt;
}

#ifdef ORC_MAIN
/* Stand-alone checker with the reference's command line (main.c:78-83,
   258-295): ./pairs_oracle -f <file> [-p N] [-c N]; prints the count. */
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>
static double now_s(void) { struct timeval t; gettimeofday(&t, NULL); return t.tv_sec + 1e-6 * t.tv_usec; }
int main(int argc, char **argv)
{
    const char *path = NULL;
    int cthreads = 4, timer = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "-f") && i + 1 < argc) path = argv[++i];
        else if (!strcmp(argv[i], "-c") && i + 1 < argc) cthreads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-p") && i + 1 < argc) ++i;   /* parsing here is sequential */
        else if (!strcmp(argv[i], "-t")) timer = 1;
        else { fprintf(stderr, "usage: %s -f <file> [-p N] [-c N] [-t]\n", argv[0]); return 255; }
    }
    int fd = path ? open(path, O_RDONLY) : -1;
    if (fd < 0) { perror("open"); return 255; }
    struct stat sb;
    if (fstat(fd, &sb) < 0) { perror("fstat"); return 255; }
    char *buf = (char *)malloc((size_t)sb.st_size + 1);
    if (!buf) { perror("malloc"); return 255; }
    size_t got = 0;
    while (got < (size_t)sb.st_size) {
        ssize_t r = read(fd, buf + got, (size_t)sb.st_size - got);
        if (r <= 0) { perror("read"); return 255; }
        got += (size_t)r;
    }
    buf[got] = 0;
    double t0 = now_s();
    uint64_t cap = got / 6 + 16;
    float *xyz = (float *)malloc(cap * 12);
    int64_t n = orc_parse_buffer(buf, got, xyz, cap);
    if (n < 0) { fprintf(stderr, "parse error\n"); return 255; }
    double t1 = now_s();
    int64_t tests = 0;
    int64_t c = orc_count_pairs(xyz, (uint64_t)n, cthreads, 0, 403, &tests, NULL, NULL);
    double t2 = now_s();
    if (c < 0) { fprintf(stderr, "count error %ld\n", (long)c); return 255; }
    printf("%ld\n", (long)c);
    if (timer) fprintf(stderr, "points %ld tests %ld parse_s %.6f bin_count_s %.6f\n", (long)n, (long)tests, t1 - t0, t2 - t1);
    return 0;
}
#endif
