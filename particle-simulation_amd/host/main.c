/*
 * main.c -- C host program of the product.  Same command line, output and exit
 * statuses as the reference program (reference src/main.c:78-83, 254-340):
 *
 *     ./main -f <data file> [-p NUM] [-c NUM] [-h]
 *
 * prints the pair count as "%ld\n".  The file is mmap'ed and tokenized by -p
 * parser threads exactly as in the reference (thread split main.c:109-129,
 * tokenizer semantics parse.h); where the reference's parser threads call
 * insert() per point (main.c:137), these append blocks of parsed points to the
 * GPU library (pc_append from a page-locked block), and where main() calls flatten()+count()
 * (main.c:331-335) this calls pc_finish().  -c is accepted for compatibility:
 * the counting threads it sized are replaced by GPU kernels.
 * A data file whose name ends in ".f32" is taken as the parsed form of such a
 * file -- packed binary32 x,y,z triples -- and appended without tokenizing
 * (kernel-side runs at sizes where the text would be tens of gigabytes).
 * One addition to the reference's command line: -g NUM (or PAIRCOUNT_GPUS=NUM)
 * cuts the x range into NUM slabs, one library context per slab, slab s on GPU s
 * (PAIRCOUNT_DEVICES="0,0,1" names the GPU of each slab instead;
 * PAIRCOUNT_CUTS="0,170,400" the first x layer of each slab and the end).  The parser
 * threads then route every point to the slab that owns its x layer -- and the first
 * layer of a slab also to the slab below it, whose pairs reach one layer up
 * (every neighbour offset of the reference has dx in {0,+1}, main.c:195-209) -- the
 * slabs are finished concurrently and their counts added: the fold of the
 * counter threads' totals at main.c:231.  Without -g nothing changes.
 * By default the routing itself happens on the GPUs: parser thread t appends whole,
 * unrouted blocks to an arrival context on GPU t mod NUM, each arrival context groups
 * its points by x layer (pc_export_layers), and every slab owner pulls its layers and
 * its halo layer from every arrival context with one peer copy each
 * (pc_import_device).  The exact number of points per x layer is known at that moment,
 * so unless PAIRCOUNT_CUTS names the cuts they are placed where an estimate of the work
 * -- per layer, 40 units per point for its binning plus the reference's 13.5 n^2 tests per
 * cell at the layer's mean filling -- balances; for an evenly filled grid these are the
 * equal widths.  PAIRCOUNT_ROUTE=host keeps the routing in the parser threads, point by
 * point, with equal widths.
 * PAIRCOUNT_QUEUE=K (K >= 2, with or without -g) cuts the x range into K slabs that
 * are not tied to a GPU: the parser threads keep each slab's points in host memory,
 * and afterwards one thread per GPU takes the next slab from a shared counter -- most
 * points first -- uploads it, counts it and comes back for more, until none is left.
 * That is the reference's hand-out of work to its counter threads (an atomic cursor,
 * main.c:192) one level up: a GPU that got sparse slabs takes more of them, so a
 * clustered input needs no cost model to stay balanced.
 * PAIRCOUNT_GPU_TOKENIZER=K (one slab only) shares the tokenizing with the GPU: the file
 * is cut into blocks of whole lines that the parser threads draw from a counter; up to K
 * threads at a time hand theirs to the library as text (pc_append_text: copy, tokenize
 * with the reference's semantics, append), the others tokenize theirs on the host.  K at
 * least the thread count leaves all of it to the GPU.
 * With PAIRCOUNT_TIMER=1 in the environment the four phase times the
 * reference's -DTIMER build prints are written to stderr.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "../../include/paircount.h"
#include "xyz_parse.h"
#include "xyz_route.h"

#define BLOCK_POINTS 65536
#define TEXT_BLOCK_BYTES (8u << 20)    /* GPU tokenizer: bytes per pc_append_text call (plus the rest of the last line) ... */
#define TEXT_LINE_MAX 4096             /* ... which fits the thread's page-locked block if no line is longer than this */
#define MAX_SLABS 64
#define LAYERS PC_NLAYERS
#define CHUNK_POINTS_MIN 4096         /* host-memory chunk of a queued slab: each is one upload, and an */
#define CHUNK_POINTS_MAX 262144       /* upload costs ~30 us before its first byte moves */

/* points of one (parser thread, slab) held in host memory until a GPU takes the slab */
typedef struct chunk {
    struct chunk *next;
    uint32_t n;
    float xyz[];                 /* 3 * g_chunk_points */
} chunk;

/* Queued runs take their chunks from one page-locked arena (uploads from it run at the link's rate, from
   pageable memory at a fraction of it); the library maps it lazily, so the parser threads' first touches
   populate it in parallel.  When it runs out -- lines much shorter than the size estimate assumes -- chunks
   come from malloc. */
static char *g_arena;
static size_t g_arena_chunks, g_chunk_bytes;
static uint32_t g_chunk_points;
static atomic_size_t g_arena_next;

static chunk *new_chunk(void)
{
    if (g_arena) {
        const size_t i = atomic_fetch_add(&g_arena_next, 1);
        if (i < g_arena_chunks) return (chunk *)(g_arena + i * g_chunk_bytes);
    }
    return (chunk *)malloc(g_chunk_bytes);
}

/* the x slabs of a run (one slab, the whole grid, unless -g asks for more) */
typedef struct {
    int n;                           /* slabs */
    int cut[MAX_SLABS + 1];          /* slab s owns x layers [cut[s], cut[s+1]) */
    int device[MAX_SLABS];
    pc_context *ctx[MAX_SLABS];
    unsigned char owner[LAYERS];     /* layer -> slab */
    uint32_t block_points;           /* capacity of one (thread, slab) block */
    int queued;                      /* slabs wait in host memory for the next free GPU */
    int gpus;                        /* queued: GPUs (taker threads), device[g] of taker g */
    int device_routing;              /* ctx[] are arrival contexts: the slabs are cut on the GPUs afterwards */
} slab_set;

typedef struct {
    const char *file;
    size_t size;
    int t, nt;
    const slab_set *slabs;
    int status;              /* 0 ok, 1 tokenizer error, 2 library error */
    double append_us;        /* time this thread spent inside pc_append */
    float *block;            /* this thread's page-locked blocks: one of block_points points per slab */
    uint32_t fill[MAX_SLABS];
    chunk *held[MAX_SLABS];  /* queued slabs: this thread's chunks, newest first */
    uint64_t held_points[MAX_SLABS];
    int home;                /* the context this thread appends whole blocks to (0 unless the GPUs do the routing) */
    int binary;              /* the file holds float triples, not text */
    int gpu_text;            /* the GPU tokenizes: this thread only cuts its range into blocks of whole lines */
    char msg[256];
} worker;

static double now_us(void)
{
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec * 1e6 + t.tv_usec;
}

static int append_block(worker *w, int slab, const float *block, uint64_t n)
{
    const double t = now_us();
    const int rc = pc_append(w->slabs->ctx[slab], block, n);
    w->append_us += now_us() - t;
    if (rc != PC_OK) { w->status = 2; snprintf(w->msg, sizeof w->msg, "%s", pc_last_error()); }
    return rc;
}

/* one point into this thread's block of slab s; a full block is appended */
static inline int put_point(worker *w, int s, const float *pt)
{
    if (w->slabs->queued) {
        chunk *c = w->held[s];
        if (!c || c->n == g_chunk_points) {
            c = new_chunk();
            if (!c) { w->status = 2; snprintf(w->msg, sizeof w->msg, "out of host memory for slab %d", s); return -1; }
            c->next = w->held[s];
            c->n = 0;
            w->held[s] = c;
        }
        memcpy(c->xyz + 3 * (size_t)c->n++, pt, 12);
        w->held_points[s]++;
        return PC_OK;
    }
    float *blk = w->block + (size_t)s * w->slabs->block_points * 3;
    memcpy(blk + 3 * (size_t)w->fill[s], pt, 12);
    if (++w->fill[s] == w->slabs->block_points) {
        w->fill[s] = 0;
        return append_block(w, s, blk, w->slabs->block_points);
    }
    return PC_OK;
}

/* Several slabs: the point goes to the owner of its x layer (the reference's cell
   formula, main.c:134) and, when that layer is the first of a slab, also to the slab
   below as its +x halo.  A point outside the grid goes to slab 0, whose pc_finish
   reports it. */
static inline int route_point(worker *w, const float *pt)
{
    const slab_set *S = w->slabs;
    const int cx = xyz_route_layer(pt[0]);
    if (cx < 0) return put_point(w, 0, pt);
    const int s = S->owner[cx];
    if (put_point(w, s, pt) != PC_OK) return -1;
    if (s > 0 && cx == S->cut[s]) return put_point(w, s - 1, pt);
    return PC_OK;
}

static void flush_blocks(worker *w)
{
    if (w->slabs->queued) return;
    for (int s = 0; s < w->slabs->n && !w->status; s++)
        if (w->fill[s]) {
            append_block(w, s, w->block + (size_t)s * w->slabs->block_points * 3, w->fill[s]);
            w->fill[s] = 0;
        }
}

/* ".f32" input: thread t takes its share of the triples straight from the mapping */
static void append_binary_share(worker *w)
{
    const uint64_t total = w->size / 12;
    const uint64_t first = total * (uint64_t)w->t / (uint64_t)w->nt, last = total * (uint64_t)(w->t + 1) / (uint64_t)w->nt;
    const float *pts = (const float *)w->file;
    if ((w->slabs->n == 1 || w->slabs->device_routing) && !w->slabs->queued) {
        for (uint64_t i = first; i < last; i += 16 * BLOCK_POINTS) {
            const uint64_t m = last - i < 16 * BLOCK_POINTS ? last - i : 16 * BLOCK_POINTS;
            if (append_block(w, w->home, pts + 3 * i, m) != PC_OK) return;
        }
        return;
    }
    for (uint64_t i = first; i < last; i++)
        if (route_point(w, pts + 3 * i) != PC_OK) return;
    flush_blocks(w);
}

/* one slab: the lines of [p, stop) tokenized here, appended in blocks of BLOCK_POINTS points */
static void tokenize_range(worker *w, const char *p, const char *stop)
{
    float *block = w->block;
    uint64_t n = 0;
    while (p < stop) {
        int err = 0;
        p = xyz_parse_line_bounded(p, w->file + w->size, block + 3 * n, &err);
        if (err) {
            w->status = 1;
            snprintf(w->msg, sizeof w->msg, err == XYZ_ERR_NODOT ? "*str != '.'" : "sci_exp > 0");
            return;
        }
        if (++n == BLOCK_POINTS) {
            if (append_block(w, w->home, block, n) != PC_OK) return;
            n = 0;
        }
    }
    if (n) append_block(w, w->home, block, n);
}

/* PAIRCOUNT_GPU_TOKENIZER=K: the file is cut into blocks of whole lines (the reference's thread split, main.c:109-129,
   applied with one "thread" per block), the parser threads draw the blocks from a shared counter, and up to K of them
   at a time send theirs through the GPU tokenizer -- copy into the thread's page-locked block, pc_append_text --
   while the others tokenize theirs here: the PCIe link and the host cores work side by side. */
static atomic_size_t g_text_next;
static size_t g_text_blocks;
static atomic_int g_gpu_lanes_busy;
static int g_gpu_lanes;

/* Whether some line of [p, e) is longer than the GPU tokenizer takes (256 bytes: the reference skips any number of
   redundant decimals, parse.h:54-56, so such a line is legal).  A line that long holds a whole 128-byte-aligned
   window without a line end; only then the exact length is looked at. */
#define GPU_RECORD_MAX 256
static int has_long_line(const char *p, const char *e)
{
    for (const char *q = p; q + 128 <= e; q += 128) {
        if (memchr(q, '\n', 128)) continue;
        const char *b = q, *a = q + 128;
        while (b > p && b[-1] != '\n') b--;
        while (a < e && *a != '\n') a++;
        if (a - b >= GPU_RECORD_MAX) return 1;
    }
    return 0;
}

static void share_text_blocks(worker *w)
{
    for (;;) {
        const size_t b = atomic_fetch_add(&g_text_next, 1);
        if (b >= g_text_blocks || w->status) return;
        const char *p, *e;
        xyz_thread_range(w->file, w->size, (int)b, (int)g_text_blocks, &p, &e);
        if (p >= e) continue;
        int lane = 0;
        if ((size_t)(e - p) <= TEXT_BLOCK_BYTES + TEXT_LINE_MAX && !has_long_line(p, e)) {
            if (atomic_fetch_add(&g_gpu_lanes_busy, 1) < g_gpu_lanes) lane = 1;
            else atomic_fetch_sub(&g_gpu_lanes_busy, 1);
        }
        if (!lane) { tokenize_range(w, p, e); continue; }
        memcpy(w->block, p, (size_t)(e - p));               /* page-locked: the upload is one DMA at the link's rate */
        const double t = now_us();
        const int rc = pc_append_text(w->slabs->ctx[0], (const char *)w->block, (uint64_t)(e - p), NULL);
        w->append_us += now_us() - t;
        atomic_fetch_sub(&g_gpu_lanes_busy, 1);
        if (rc != PC_OK) {
            w->status = rc == PC_ERR_TEXT ? 1 : 2;
            if (rc == PC_ERR_TEXT) snprintf(w->msg, sizeof w->msg, strstr(pc_last_error(), "exponent") ? "sci_exp > 0" : "*str != '.'");
            else snprintf(w->msg, sizeof w->msg, "%s", pc_last_error());
            return;
        }
    }
}

static void *parser_thread(void *arg)
{
    worker *w = (worker *)arg;
    if (w->binary) { append_binary_share(w); return NULL; }
    if (w->gpu_text) { share_text_blocks(w); return NULL; }
    const char *p, *stop;
    xyz_thread_range(w->file, w->size, w->t, w->nt, &p, &stop);
    if (w->slabs->n > 1 && !w->slabs->device_routing) {
        while (p < stop) {
            int err = 0;
            float pt[3];
            p = xyz_parse_line_bounded(p, w->file + w->size, pt, &err);
            if (err) {
                w->status = 1;
                snprintf(w->msg, sizeof w->msg, err == XYZ_ERR_NODOT ? "*str != '.'" : "sci_exp > 0");
                return NULL;
            }
            if (route_point(w, pt) != PC_OK) return NULL;
        }
        flush_blocks(w);
        return NULL;
    }
    tokenize_range(w, p, stop);
    return NULL;
}

/* "a,b,c" -> out[0..]; returns how many numbers were read (at most cap) */
static int parse_int_list(const char *s, int *out, int cap)
{
    int n = 0;
    while (s && *s && n < cap) {
        char *e;
        const long v = strtol(s, &e, 10);
        if (e == s) break;
        out[n++] = (int)v;
        s = *e == ',' ? e + 1 : e;
    }
    return n;
}

typedef struct {
    pc_context *ctx;
    int64_t npairs, tests;
    float ms[2];
    int rc;
    char msg[256];
} finisher;

static void *finish_thread(void *arg)
{
    finisher *f = (finisher *)arg;
    f->rc = pc_finish(f->ctx, &f->npairs);
    if (f->rc != PC_OK) { snprintf(f->msg, sizeof f->msg, "%s", pc_last_error()); return NULL; }
    pc_phase_ms(f->ctx, f->ms);
    pc_tests(f->ctx, &f->tests);
    return NULL;
}

/* Routing on the GPUs, step 1: an arrival context groups its points by x layer. */
typedef struct {
    pc_context *ctx;
    int device, rc;
    const float *points;                 /* device array, layer l = points [start[l], start[l+1]) */
    uint32_t start[PC_NBB + 1];
    char msg[256];
} exporter;

static void *export_thread(void *arg)
{
    exporter *e = (exporter *)arg;
    e->rc = pc_export_layers(e->ctx, &e->points, e->start);
    if (e->rc != PC_OK) snprintf(e->msg, sizeof e->msg, "%s", pc_last_error());
    return NULL;
}

/* Step 2: the owner of slab o pulls layers cut[o] .. cut[o+1] (its halo layer included; the last slab ends at
   layer 399) out of every arrival context -- one copy per source -- and counts. */
typedef struct {
    const slab_set *slabs;
    const exporter *sources;
    int o;
    finisher *fin;
    pc_context *ctx;
} owner;

static void *owner_thread(void *arg)
{
    owner *w = (owner *)arg;
    const slab_set *S = w->slabs;
    finisher *f = w->fin;
    const int lo = S->cut[w->o], hi = w->o == S->n - 1 ? LAYERS : S->cut[w->o + 1] + 1;      /* layers [lo, hi) */
    uint64_t total = 0;
    for (int s = 0; s < S->n; s++) total += w->sources[s].start[hi] - w->sources[s].start[lo];
    f->rc = pc_create(&w->ctx, S->device[w->o], total + 1);
    if (f->rc == PC_OK) f->rc = pc_set_slab(w->ctx, S->cut[w->o], S->cut[w->o + 1]);
    for (int s = 0; s < S->n && f->rc == PC_OK; s++) {
        const exporter *e = &w->sources[s];
        f->rc = pc_import_device(w->ctx, e->device, e->points + 3 * (size_t)e->start[lo], e->start[hi] - e->start[lo]);
    }
    if (f->rc != PC_OK) { snprintf(f->msg, sizeof f->msg, "%s", pc_last_error()); return NULL; }
    f->ctx = w->ctx;
    finish_thread(f);
    return NULL;
}

/* one GPU of a queued run: takes slabs from the shared cursor until none is left */
typedef struct {
    const slab_set *slabs;
    worker *w;
    int nt;
    const int *order;            /* slabs, most points first */
    const uint64_t *points;      /* points held per slab */
    atomic_int *next, *failed;
    int g;                       /* taker g runs on slabs->device[g] */
    finisher *fin;               /* results, per slab */
    int *taken_by;               /* per slab: the taker that counted it */
    double *busy_us;             /* per taker */
    double part_us[4];           /* this taker's time in create, append, finish, destroy */
} taker;

static void *taker_thread(void *arg)
{
    taker *k = (taker *)arg;
    const slab_set *S = k->slabs;
    const double t0 = now_us();
    for (;;) {
        const int i = atomic_fetch_add(k->next, 1);
        if (i >= S->n || atomic_load(k->failed)) break;
        const int s = k->order[i];
        finisher *f = &k->fin[s];
        k->taken_by[s] = k->g;
        if (k->points[s] == 0) continue;                 /* nothing in it: no pairs */
        pc_context *ctx = NULL;
        double t = now_us(), u;
        f->rc = pc_create(&ctx, S->device[k->g], k->points[s]);
        if (f->rc == PC_OK) f->rc = pc_set_slab(ctx, S->cut[s], S->cut[s + 1]);
        k->part_us[0] += (u = now_us()) - t;
        for (int t = 0; t < k->nt && f->rc == PC_OK; t++)
            for (const chunk *c = k->w[t].held[s]; c && f->rc == PC_OK; c = c->next)
                f->rc = pc_append(ctx, c->xyz, c->n);
        k->part_us[1] += (t = now_us()) - u;
        if (f->rc == PC_OK) {
            f->ctx = ctx;
            finish_thread(f);
        } else {
            snprintf(f->msg, sizeof f->msg, "%s", pc_last_error());
        }
        k->part_us[2] += (u = now_us()) - t;
        if (ctx) pc_destroy(ctx);
        k->part_us[3] += now_us() - u;
        f->ctx = NULL;
        if (f->rc != PC_OK) atomic_store(k->failed, 1);
    }
    k->busy_us[k->g] = now_us() - t0;
    return NULL;
}

static void usage(char **argv)
{
    printf("Usage: %s -f <data file> [-p NUM] [-c NUM] [-g NUM]\n", argv[0]);
    puts("\t-c Num\tAccepted for compatibility (pairs are counted on the GPU).");
    puts("\t-g Num\tNumber of GPUs (x slabs) to count on. Default is 1.");
    puts("\t-p Num\tNumber of threads to use while parsing. Default is 4.");
    puts("\t-h\tPrint this message and exit.");
}

int main(int argc, char **argv)
{
    const int timer = getenv("PAIRCOUNT_TIMER") != NULL;
    double t0 = now_us();
    const char *file_name = NULL;
    int pthreads = 4, cthreads = 4;
    int gpus = 1;
    if (getenv("PAIRCOUNT_GPUS") && (gpus = atoi(getenv("PAIRCOUNT_GPUS"))) <= 0) {
        fprintf(stderr, "PAIRCOUNT_GPUS: need a positive count\n");
        return -1;
    }
    for (int i = 1; i < argc; i++) {
        if (argv[i][0] != '-') {
            printf("Unrecognized option \"%s\"\n", argv[i]);
            usage(argv);
            return -1;
        }
        switch (argv[i][1]) {
        case 'c':
            if (++i >= argc || (cthreads = atoi(argv[i])) <= 0) { perror("atoi"); return -1; }
            break;
        case 'p':
            if (++i >= argc || (pthreads = atoi(argv[i])) <= 0) { perror("atoi"); return -1; }
            break;
        case 'g':
            if (++i >= argc || (gpus = atoi(argv[i])) <= 0) { perror("atoi"); return -1; }
            break;
        case 'f':
            file_name = ++i < argc ? argv[i] : NULL;
            break;
        case 'h':
            usage(argv);
            return 0;
        default:
            printf("Unrecognized option \"-%s\"\n", &argv[i][1]);
            usage(argv);
            return -1;
        }
    }
    (void)cthreads;
    if (pthreads > 256) pthreads = 256;
    /* no -f: the reference passes its NULL file name to open(), which fails with EFAULT (main.c:297-300) */
    if (!file_name) errno = EFAULT;
    int fd = file_name ? open(file_name, O_RDONLY) : -1;
    if (fd == -1) { perror("open"); return -1; }
    struct stat sb;
    if (fstat(fd, &sb) == -1) { perror("fstat"); return -1; }
    size_t size = (size_t)sb.st_size;
    /* The tokenizer looks one character past the last line (as the reference
       does).  The zero fill of the last mapped page provides it, except when the
       size is an exact page multiple: then read the file into a terminated buffer. */
    const size_t name_len = strlen(file_name);
    const int binary = name_len > 4 && strcmp(file_name + name_len - 4, ".f32") == 0;
    if (binary && size % 12 != 0) {
        fprintf(stderr, "%s: %zu bytes is not a whole number of float triples\n", file_name, size);
        return -1;
    }
    /* an empty text file: the reference's mmap of zero bytes fails (main.c:311-315) */
    if (size == 0 && !binary) { errno = EINVAL; perror("mmap"); return -1; }
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const int mapped = size && (binary || (size % page) != 0);
    char *file = NULL;
    if (mapped) {
        file = mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (file == MAP_FAILED) { perror("mmap"); return -1; }
    } else if (size) {
        file = (char *)malloc(size + 1);
        if (!file) { perror("malloc"); return -1; }
        size_t got = 0;
        while (got < size) {
            ssize_t r = read(fd, file + got, size - got);
            if (r <= 0) { perror("read"); return -1; }
            got += (size_t)r;
        }
        file[size] = 0;
    }

    /* the slabs: equal widths unless PAIRCOUNT_CUTS names them; slab s on GPU s unless PAIRCOUNT_DEVICES does */
    static slab_set S;
    if (gpus > MAX_SLABS) gpus = MAX_SLABS;
    S.n = S.gpus = gpus;
    if (getenv("PAIRCOUNT_QUEUE")) {
        const int k = atoi(getenv("PAIRCOUNT_QUEUE"));
        if (k < 2 || k > MAX_SLABS || k < gpus) {
            fprintf(stderr, "PAIRCOUNT_QUEUE: need a slab count from %d to %d\n", gpus > 2 ? gpus : 2, MAX_SLABS);
            return -1;
        }
        S.n = k;
        S.queued = 1;
    }
    for (int s = 0; s <= S.n; s++) S.cut[s] = LAYERS * s / S.n;
    if (getenv("PAIRCOUNT_CUTS")) {
        int c[MAX_SLABS + 1];
        const int k = parse_int_list(getenv("PAIRCOUNT_CUTS"), c, MAX_SLABS + 1);
        int ok = k == S.n + 1 && c[0] == 0 && c[S.n] == LAYERS;
        for (int s = 0; ok && s < S.n; s++) ok = c[s] < c[s + 1];
        if (!ok) { fprintf(stderr, "PAIRCOUNT_CUTS: need %d increasing layers from 0 to %d\n", S.n + 1, LAYERS); return -1; }
        memcpy(S.cut, c, sizeof(int) * (size_t)(S.n + 1));
    }
    /* device[] is per slab, or per GPU of a queued run */
    const int ndev = S.queued ? S.gpus : S.n;
    for (int s = 0; s < S.n; s++) {
        S.device[s] = ndev == 1 && getenv("PAIRCOUNT_DEVICE") ? atoi(getenv("PAIRCOUNT_DEVICE")) : s;
        for (int l = S.cut[s]; l < S.cut[s + 1]; l++) S.owner[l] = (unsigned char)s;
    }
    if (getenv("PAIRCOUNT_DEVICES") &&
        parse_int_list(getenv("PAIRCOUNT_DEVICES"), S.device, ndev) != ndev) {
        fprintf(stderr, "PAIRCOUNT_DEVICES: need %d device ordinals\n", ndev);
        return -1;
    }
    /* every (thread, slab) block holds BLOCK_POINTS points up to 16 slabs (16 threads x 8 slabs: 100 MB page-locked);
       smaller blocks mean more copies, and a copy costs ~20 us before its first byte moves: with 8 192-point
       blocks 8 slabs took 0.45 s for 100 M points and 4.1 s for 1 G, with 65 536-point blocks 0.24 s and 1.6 s */
    S.block_points = S.n <= 16 ? BLOCK_POINTS : (BLOCK_POINTS * 16 / (uint32_t)S.n < 4096 ? 4096 : BLOCK_POINTS * 16 / (uint32_t)S.n);
    const uint64_t expect = binary ? size / 12 + 1 : size / 24 + 1024;
    S.device_routing = S.n > 1 && !S.queued && !(getenv("PAIRCOUNT_ROUTE") && strcmp(getenv("PAIRCOUNT_ROUTE"), "host") == 0);
    for (int s = 0; s < S.n && !S.queued; s++) {
        /* capacity hint: the slab's share of the points, with room for its halo layer and some imbalance -- or,
           when the GPUs do the routing, an arrival context's share of the parser threads' output */
        const uint64_t hint = S.n == 1 ? expect : S.device_routing ? expect / (uint64_t)S.n + expect / 16 + 4096
                                       : expect * (uint64_t)(S.cut[s + 1] - S.cut[s] + 2) / LAYERS + 4096;
        if (pc_create(&S.ctx[s], S.device[s], hint) != PC_OK ||
            (S.n > 1 && !S.device_routing && pc_set_slab(S.ctx[s], S.cut[s], S.cut[s + 1]) != PC_OK)) {
            fprintf(stderr, "pc_create: %s\n", pc_last_error());
            return -1;
        }
    }
    if (S.queued && size) {
        /* every point once, a halo layer per slab, and a part-filled chunk per (thread, slab) */
        /* chunk size: about two per (thread, slab) of evenly spread points, within the limits above */
        const size_t lists = (size_t)pthreads * (size_t)S.n;
        g_chunk_points = CHUNK_POINTS_MIN;
        while (g_chunk_points < CHUNK_POINTS_MAX && (uint64_t)g_chunk_points * 2 * lists < expect) g_chunk_points *= 2;
        g_chunk_bytes = (sizeof(chunk) + 12 * (size_t)g_chunk_points + 63) & ~(size_t)63;
        g_arena_chunks = (size_t)(expect + expect * (uint64_t)S.n / LAYERS) / g_chunk_points + lists + 1;
        if (pc_host_alloc((void **)&g_arena, (uint64_t)g_arena_chunks * g_chunk_bytes) != PC_OK) g_arena = NULL;
    }
    if (timer) fprintf(stderr, "%16s:  %10.0f us\n", "init", now_us() - t0);

    t0 = now_us();
    pthread_t th[256];
    worker w[256];
    /* one page-locked allocation for all threads' blocks (page-locking is slow per call) */
    float *pinned = NULL;
    const int gpu_text = !binary && S.n == 1 && !S.queued && getenv("PAIRCOUNT_GPU_TOKENIZER") != NULL &&
                         atoi(getenv("PAIRCOUNT_GPU_TOKENIZER")) > 0;
    if (gpu_text) {
        g_gpu_lanes = atoi(getenv("PAIRCOUNT_GPU_TOKENIZER"));
        g_text_blocks = (size + TEXT_BLOCK_BYTES - 1) / TEXT_BLOCK_BYTES;
        if (g_text_blocks > 0x7FFFFFFF) g_text_blocks = 0x7FFFFFFF;
    }
    const size_t thread_floats = gpu_text ? (TEXT_BLOCK_BYTES + TEXT_LINE_MAX) / 4
                                          : S.device_routing ? (size_t)BLOCK_POINTS * 3 : (size_t)S.n * S.block_points * 3;
    if (size && !S.queued && pc_host_alloc((void **)&pinned, (uint64_t)pthreads * thread_floats * 4) != PC_OK) {
        fprintf(stderr, "pc_host_alloc: %s\n", pc_last_error());
        return -1;
    }
    for (int t = 0; t < pthreads; t++) {
        w[t] = (worker){.file = file, .size = size, .t = t, .nt = pthreads, .slabs = &S, .status = 0,
                        .block = pinned + (size_t)t * thread_floats, .binary = binary,
                        .gpu_text = gpu_text, .home = S.device_routing ? t % S.n : 0};
        if (size == 0) continue;
        if (pthread_create(&th[t], NULL, parser_thread, &w[t]) != 0) { perror("pthread_create"); exit(-1); }
    }
    for (int t = 0; t < pthreads && size; t++)
        if (pthread_join(th[t], NULL) != 0) { perror("pthread_join"); exit(-1); }
    for (int t = 0; t < pthreads; t++) {
        if (w[t].status == 1) { perror(w[t].msg); exit(-1); }
        if (w[t].status == 2) { fprintf(stderr, "pc_append: %s\n", w[t].msg); exit(-1); }
    }
    if (pinned) pc_host_free(pinned);
    close(fd);
    if (mapped) munmap(file, size);
    else free(file);
    if (timer) {
        double worst = 0;
        for (int t = 0; t < pthreads; t++) if (w[t].append_us > worst) worst = w[t].append_us;
        fprintf(stderr, "%16s:  %10.0f us\n", "parsing", now_us() - t0);
        fprintf(stderr, "%16s:  %10.0f us\n", "append(max)", worst);        /* part of "parsing": the slowest thread's copies */
    }

    t0 = now_us();
    /* flatten()+count() of the reference (main.c:331-335), one per slab, concurrently; the sum is its
       fold of the counter threads' totals (main.c:231) */
    static finisher fin[MAX_SLABS];
    pthread_t fth[MAX_SLABS];
    static uint64_t held[MAX_SLABS];
    static int order[MAX_SLABS], taken_by[MAX_SLABS];
    static double busy_us[MAX_SLABS];
    static taker tk[MAX_SLABS];
    atomic_int next = 0, failed = 0;
    if (S.queued) {
        for (int s = 0; s < S.n; s++) {
            order[s] = s;
            for (int t = 0; t < pthreads; t++) held[s] += w[t].held_points[s];
        }
        for (int a = 1; a < S.n; a++)                  /* most points first: the long jobs start early */
            for (int b = a; b > 0 && held[order[b]] > held[order[b - 1]]; b--) {
                const int o = order[b]; order[b] = order[b - 1]; order[b - 1] = o;
            }
        for (int g = 0; g < S.gpus; g++) {
            tk[g] = (taker){.slabs = &S, .w = w, .nt = pthreads, .order = order, .points = held, .next = &next,
                            .failed = &failed, .g = g, .fin = fin, .taken_by = taken_by, .busy_us = busy_us};
            if (pthread_create(&fth[g], NULL, taker_thread, &tk[g]) != 0) { perror("pthread_create"); exit(-1); }
        }
        for (int g = 0; g < S.gpus; g++)
            if (pthread_join(fth[g], NULL) != 0) { perror("pthread_join"); exit(-1); }
    }
    static exporter ex[MAX_SLABS];
    static owner own[MAX_SLABS];
    if (S.device_routing) {
        /* every arrival context groups its points by x layer, concurrently; then the slab owners pull and count */
        double t1 = now_us();
        for (int s = 0; s < S.n; s++) {
            ex[s] = (exporter){.ctx = S.ctx[s], .device = S.device[s]};
            if (pthread_create(&fth[s], NULL, export_thread, &ex[s]) != 0) { perror("pthread_create"); exit(-1); }
        }
        for (int s = 0; s < S.n; s++)
            if (pthread_join(fth[s], NULL) != 0) { perror("pthread_join"); exit(-1); }
        for (int s = 0; s < S.n; s++)
            if (ex[s].rc != PC_OK) { fprintf(stderr, "pc_finish: %s\n", ex[s].msg); return -1; }
        if (timer) fprintf(stderr, "%16s:  %10.0f us\n", "export(wall)", now_us() - t1);
        if (!getenv("PAIRCOUNT_CUTS")) {
            /* cuts where the estimated work balances: layer l holds n points, its cells n / 160000 each on
               average, and a cell with m points costs the reference 13.5 m^2 tests (main.c:210-229: m^2/2 in the
               cell, 13 m^2 with its neighbours); binning and per-point work count as 40 tests per point */
            double below[LAYERS + 1];                     /* below[l] = estimated work of layers < l */
            below[0] = 0;
            for (int l = 0; l < LAYERS; l++) {
                double n = 0;
                for (int s = 0; s < S.n; s++) n += (double)(ex[s].start[l + 1] - ex[s].start[l]);
                below[l + 1] = below[l] + n * (40.0 + 13.5 * n / ((double)LAYERS * LAYERS));
            }
            for (int s = 1, l = 0; s < S.n; s++) {
                while (l < LAYERS && below[l] * S.n < below[LAYERS] * s) l++;       /* first cut with s/n of the work below it */
                int c = l;
                if (c <= S.cut[s - 1]) c = S.cut[s - 1] + 1;                        /* every slab keeps at least one layer */
                if (c > LAYERS - (S.n - s)) c = LAYERS - (S.n - s);
                S.cut[s] = c;
            }
        }
        for (int s = 0; s < S.n; s++) {
            own[s] = (owner){.slabs = &S, .sources = ex, .o = s, .fin = &fin[s]};
            if (pthread_create(&fth[s], NULL, owner_thread, &own[s]) != 0) { perror("pthread_create"); exit(-1); }
        }
    }
    for (int s = 0; s < S.n && !S.queued && !S.device_routing; s++) {
        fin[s].ctx = S.ctx[s];
        if (S.n == 1) finish_thread(&fin[s]);
        else if (pthread_create(&fth[s], NULL, finish_thread, &fin[s]) != 0) { perror("pthread_create"); exit(-1); }
    }
    int64_t npairs = 0, tests = 0;
    float ms[2] = {0, 0};
    for (int s = 0; s < S.n && S.n > 1 && !S.queued; s++)   /* all of them, before any result is judged: no thread may */
        if (pthread_join(fth[s], NULL) != 0) { perror("pthread_join"); exit(-1); }   /* still be inside HIP at exit */
    if (S.device_routing)
        for (int s = 0; s < S.n; s++) {                 /* the arrival contexts are done with; the owners' take their place */
            pc_destroy(S.ctx[s]);
            S.ctx[s] = own[s].ctx;
        }
    for (int s = 0; s < S.n; s++) {
        if (fin[s].rc != PC_OK) {
            fprintf(stderr, "pc_finish: %s\n", fin[s].msg);
            return -1;
        }
        npairs += fin[s].npairs;
        tests += fin[s].tests;
        for (int k = 0; k < 2; k++) if (fin[s].ms[k] > ms[k]) ms[k] = fin[s].ms[k];
    }
    printf("%ld\n", (long)npairs);
    if (timer) {
        fprintf(stderr, "%16s:  %10.0f us\n", "binning", ms[0] * 1e3);     /* reference phase "flatten"; the slowest slab's */
        fprintf(stderr, "%16s:  %10.0f us\n", "count", ms[1] * 1e3);
        fprintf(stderr, "%16s:  %10.0f us\n", "finish(wall)", now_us() - t0);
        fprintf(stderr, "%16s:  %10ld\n", "tests", (long)tests);
        for (int s = 0; s < S.n && S.n > 1; s++) {
            uint64_t in_slab = held[s];
            if (!S.queued && S.ctx[s]) pc_size(S.ctx[s], &in_slab);
            fprintf(stderr, "   slab %2d gpu %2d layers [%3d,%3d): %10lu points, binning %8.0f us, count %8.0f us\n", s,
                    S.device[S.queued ? taken_by[s] : s], S.cut[s], S.cut[s + 1], (unsigned long)in_slab,
                    fin[s].ms[0] * 1e3, fin[s].ms[1] * 1e3);
        }
        for (int g = 0; g < S.gpus && S.queued; g++)
            fprintf(stderr, "   taker %2d gpu %2d: busy %10.0f us (create %.0f, append %.0f, finish %.0f, destroy %.0f)\n", g,
                    S.device[g], busy_us[g], tk[g].part_us[0], tk[g].part_us[1], tk[g].part_us[2], tk[g].part_us[3]);
    }
    for (int s = 0; s < S.n && !S.queued; s++) pc_destroy(S.ctx[s]);      /* (pc_destroy(NULL) is a no-op) */
    return 0;
}
