/*
 * ref_shim.c -- thin exported wrappers around the UNMODIFIED reference sources.
 *
 * TEST INFRASTRUCTURE, built only where /root/reference exists (this
 * container).  The reference files are compiled from where they lie
 * (-I/root/reference/src); nothing of them is copied into this repository.
 * Output goes to oracle/_ref/ (git-ignored).  The wrappers give the tests
 * function-level access to the reference's tokenizer, cell-key expression,
 * dist2 kernel and hash-map + counter pool so that oracle/pairs_oracle.c and
 * the golden vectors can be pinned against the real thing.
 */
#define main reference_main            /* keep the reference's main() out of the way */
#include "main.c"                      /* resolved through -I/root/reference/src */
#undef main

#include <string.h>

/* reference parse.h:24 */
uint32_t ref_parse_float_bits(const char *tok, long *consumed)
{
    char *end = NULL;
    float f = parse_float((char *)tok, &end);
    if (consumed) *consumed = (long)(end - tok);
    uint32_t b;
    memcpy(&b, &f, 4);
    return b;
}

/* reference parse.h:150; returns characters consumed */
long ref_parse_line(const char *line, float out[3])
{
    char *next = parse_line((char *)line, out);
    return (long)(next - line);
}

/* reference main.c:134-137, using the reference's own BBSIZE/INDEX macros */
int64_t ref_cell_key(const float p[3])
{
    int x = (int)((p[0] - (-10.)) / BBSIZE);
    int y = (int)((p[1] - (-10.)) / BBSIZE);
    int z = (int)((p[2] - (-10.)) / BBSIZE);
    return (int64_t)INDEX(x, y, z);
}

int ref_nbb(void) { return NBB; }

/* reference main.c:85 and the compare at main.c:165 */
uint32_t ref_dist2_bits(const float a[4], const float b[4], int *within)
{
    float d = dist2((float *)a, (float *)b);
    if (within) *within = (d <= 0.0025);
    uint32_t u;
    memcpy(&u, &d, 4);
    return u;
}

/* Feeds n points through the reference's own insert/flatten/count
   (hashmap.h:94,137; main.c:236) with `threads` counter threads.
   The reference keeps its state in globals and never frees; reset what a
   second call needs. */
long ref_count_points(const float *xyz, size_t n, int threads)
{
    Hashmap *hm = malloc(sizeof(Hashmap));
    size_t cap = n / 4 + 16;           /* main.c:317 uses file_size/128; any non-zero capacity is equivalent */
    init_hashmap(hm, cap);
    for (size_t i = 0; i < n; i++) {
        float buf[4] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.0f};
        insert(hm, (uint64_t)ref_cell_key(buf), buf);
    }
    uint64_t *keys = NULL;
    size_t size = 0;
    flatten(hm, &keys, &size);
    cstate.nthreads = threads > 0 ? threads : 1;
    cstate.keys_next_index = 0;
    cstate.npairs = 0;
    long r = count(hm, keys, hm->used);
    free(keys);
    return r;
}
