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
 * With PAIRCOUNT_TIMER=1 in the environment the four phase times the
 * reference's -DTIMER build prints are written to stderr.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "../../include/paircount.h"
#include "xyz_parse.h"

#define BLOCK_POINTS 65536

typedef struct {
    const char *file;
    size_t size;
    int t, nt;
    pc_context *ctx;
    int status;              /* 0 ok, 1 tokenizer error, 2 library error */
    double append_us;        /* time this thread spent inside pc_append */
    float *block;            /* this thread's page-locked block of BLOCK_POINTS points */
    int binary;              /* the file holds float triples, not text */
    char msg[256];
} worker;

static double now_us(void)
{
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec * 1e6 + t.tv_usec;
}

static int append_block(worker *w, const float *block, uint64_t n)
{
    const double t = now_us();
    const int rc = pc_append(w->ctx, block, n);
    w->append_us += now_us() - t;
    if (rc != PC_OK) { w->status = 2; snprintf(w->msg, sizeof w->msg, "%s", pc_last_error()); }
    return rc;
}

/* ".f32" input: thread t appends its share of the triples straight from the mapping */
static void append_binary_share(worker *w)
{
    const uint64_t total = w->size / 12;
    const uint64_t first = total * (uint64_t)w->t / (uint64_t)w->nt, last = total * (uint64_t)(w->t + 1) / (uint64_t)w->nt;
    const float *pts = (const float *)w->file;
    for (uint64_t i = first; i < last; i += 16 * BLOCK_POINTS) {
        const uint64_t m = last - i < 16 * BLOCK_POINTS ? last - i : 16 * BLOCK_POINTS;
        if (append_block(w, pts + 3 * i, m) != PC_OK) return;
    }
}

static void *parser_thread(void *arg)
{
    worker *w = (worker *)arg;
    if (w->binary) { append_binary_share(w); return NULL; }
    float *block = w->block;
    const char *p, *stop;
    xyz_thread_range(w->file, w->size, w->t, w->nt, &p, &stop);
    uint64_t n = 0;
    while (p < stop) {
        int err = 0;
        p = xyz_parse_line_bounded(p, w->file + w->size, block + 3 * n, &err);
        if (err) {
            w->status = 1;
            snprintf(w->msg, sizeof w->msg, err == XYZ_ERR_NODOT ? "*str != '.'" : "sci_exp > 0");
            break;
        }
        if (++n == BLOCK_POINTS) {
            if (append_block(w, block, n) != PC_OK) break;
            n = 0;
        }
    }
    if (!w->status && n) append_block(w, block, n);
    return NULL;
}

static void usage(char **argv)
{
    printf("Usage: %s -f <data file> [-p NUM] [-c NUM]\n", argv[0]);
    puts("\t-c Num\tAccepted for compatibility (pairs are counted on the GPU).");
    puts("\t-p Num\tNumber of threads to use while parsing. Default is 4.");
    puts("\t-h\tPrint this message and exit.");
}

int main(int argc, char **argv)
{
    const int timer = getenv("PAIRCOUNT_TIMER") != NULL;
    double t0 = now_us();
    const char *file_name = NULL;
    int pthreads = 4, cthreads = 4;
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

    int device = getenv("PAIRCOUNT_DEVICE") ? atoi(getenv("PAIRCOUNT_DEVICE")) : 0;
    pc_context *ctx = NULL;
    if (pc_create(&ctx, device, binary ? size / 12 + 1 : size / 24 + 1024) != PC_OK) {
        fprintf(stderr, "pc_create: %s\n", pc_last_error());
        return -1;
    }
    if (timer) fprintf(stderr, "%16s:  %10.0f us\n", "init", now_us() - t0);

    t0 = now_us();
    if (pthreads > 256) pthreads = 256;
    pthread_t th[256];
    worker w[256];
    /* one page-locked allocation for all threads' blocks (page-locking is slow per call) */
    float *pinned = NULL;
    if (size && pc_host_alloc((void **)&pinned, (uint64_t)pthreads * BLOCK_POINTS * 12) != PC_OK) {
        fprintf(stderr, "pc_host_alloc: %s\n", pc_last_error());
        return -1;
    }
    for (int t = 0; t < pthreads; t++) {
        w[t] = (worker){.file = file, .size = size, .t = t, .nt = pthreads, .ctx = ctx, .status = 0,
                        .block = pinned + (size_t)t * BLOCK_POINTS * 3, .binary = binary};
        if (size == 0) continue;
        if (pthread_create(&th[t], NULL, parser_thread, &w[t]) != 0) { perror("pthread_create"); exit(-1); }
    }
    for (int t = 0; t < pthreads && size; t++)
        if (pthread_join(th[t], NULL) != 0) { perror("pthread_join"); exit(-1); }
    for (int t = 0; t < pthreads; t++) {
        if (w[t].status == 1) { perror(w[t].msg); exit(-1); }
        if (w[t].status == 2) { fprintf(stderr, "pc_append: %s\n", w[t].msg); exit(-1); }
    }
    pc_host_free(pinned);
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
    int64_t npairs = 0;
    if (pc_finish(ctx, &npairs) != PC_OK) {
        fprintf(stderr, "pc_finish: %s\n", pc_last_error());
        return -1;
    }
    printf("%ld\n", (long)npairs);
    if (timer) {
        float ms[2] = {0, 0};
        int64_t tests = 0;
        pc_phase_ms(ctx, ms);
        pc_tests(ctx, &tests);
        fprintf(stderr, "%16s:  %10.0f us\n", "binning", ms[0] * 1e3);     /* reference phase "flatten" */
        fprintf(stderr, "%16s:  %10.0f us\n", "count", ms[1] * 1e3);
        fprintf(stderr, "%16s:  %10.0f us\n", "finish(wall)", now_us() - t0);
        fprintf(stderr, "%16s:  %10ld\n", "tests", (long)tests);
    }
    pc_destroy(ctx);
    return 0;
}
