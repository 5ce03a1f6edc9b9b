/*
 * xyz_parse.h -- host tokenizer of the product: text "x y z" lines -> binary32.
 *
 * Reproduces, bit for bit, what the reference's parse_float/parse_line
 * (reference src/parse.h:24-163) return for every token they accept,
 * including their truncating conversion and the leading-zero scan that skips
 * the last decimal(s) (parse.h:80-88), because the pair count is defined on
 * those float bits.  The arithmetic is organised differently from the
 * reference: k successive truncating divisions by 5 equal one truncating
 * division by 5^k, so each run of up to 16 steps is a single divide by a
 * constant from a table.
 *
 * Where the reference prints and exits (no '.' after the first digit,
 * positive decimal exponent) these functions return an error code instead;
 * the host program turns it into the same exit status.
 */
#ifndef XYZ_PARSE_H
#define XYZ_PARSE_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XYZ_ERR_NODOT 1
#define XYZ_ERR_POSEXP 2

static const uint64_t xyz_pow5[17] = {
    1ull, 5ull, 25ull, 125ull, 625ull, 3125ull, 15625ull, 78125ull, 390625ull, 1953125ull,
    9765625ull, 48828125ull, 244140625ull, 1220703125ull, 6103515625ull, 30517578125ull,
    152587890625ull};

static inline int xyz_digit(char c) { return (unsigned)(c - '0') <= 9u; }

/* bits of the parsed float; *end = first unconsumed char; *err = 0 or XYZ_ERR_* */
static inline uint32_t xyz_parse_float_bits(const char *tok, const char **end, int *err)
{
    const char *p = tok;
    const int neg = (*p == '-');
    p += neg;
    uint64_t digits = (uint64_t)(int64_t)(p[0] - '0');
    if (p[1] != '.') { *err = XYZ_ERR_NODOT; *end = p + 1; return 0; }
    p += 2;
    int nd = 0;
    for (; nd < 17 && xyz_digit(*p); nd++, p++) digits = digits * 10u + (uint64_t)(*p - '0');
    while (xyz_digit(*p)) p++;
    int e10 = 0;
    if (*p == 'e') {
        e10 = atoi(p + 1);
        p += 2;
        while (xyz_digit(*p)) p++;
    }
    *end = p;
    *err = 0;
    if (digits == 0) return 0;

    /* zeros before the first non-zero digit, looking only at source chars [neg, nd] */
    int zeros = 0, lead = 0;
    for (int i = neg; i <= nd; i++) {
        const char c = tok[i];
        if (c > '0') { lead = c - '0'; break; }
        zeros += (c == '0');
    }
    e10 -= zeros;
    nd -= zeros;
    if (e10 > 0) { *err = XYZ_ERR_POSEXP; return 0; }

    uint64_t m = digits;
    int e2 = 0;
    /* value = digits / 10^nd: the 2^-nd part is implied by the leading digit,
       the 5^-nd part is applied here in runs of <= 16 on a left-justified word */
    for (int left = nd; left > 0; left -= 16) {
        m <<= __builtin_clzll(m);
        m /= xyz_pow5[left < 16 ? left : 16];
    }
    if (e10 < 0) {
        int first = 1;
        for (int left = -e10; left > 0; left -= 16) {
            const int s = __builtin_clzll(m);
            m <<= s;
            if (!first) e2 -= s;
            first = 0;
            m /= xyz_pow5[left < 16 ? left : 16];
        }
        e2 += e10 - __builtin_clzll(m);
    }
    const int sh = 40 - __builtin_clzll(m);
    const uint32_t mant = (sh >= 0) ? (uint32_t)(m >> sh) : 0u;
    static const int8_t log2_digit[10] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3};
    e2 += log2_digit[lead];
    return ((uint32_t)neg << 31) | ((uint32_t)((e2 + 127) & 0xFF) << 23) | (mant & 0x7FFFFFu);
}

/* one line: three tokens, each followed by exactly one separator char, then any
   run of control characters (1..31) -- reference parse.h:150-163 */
static inline const char *xyz_parse_line(const char *s, float out[3], int *err)
{
    uint32_t b[3];
    for (int a = 0; a < 3; a++) {
        const char *e;
        b[a] = xyz_parse_float_bits(s, &e, err);
        if (*err) return e;
        s = e + 1;
    }
    memcpy(out, b, 12);
    while ((unsigned char)(*s - 1) < 31u) s++;
    return s;
}

/* byte range of parser thread t of nt: the reference's split, main.c:109-129 */
static inline void xyz_thread_range(const char *file, size_t size, int t, int nt,
                                    const char **begin, const char **stop)
{
    const size_t chunk = size / (size_t)nt;
    const char *b = file + (size_t)t * chunk;
    if (t != 0) {
        while (*b > '\n') b++;
        b++;
    }
    const char *e;
    if (t == nt - 1) {
        e = file + size;
    } else {
        e = file + (size_t)(t + 1) * chunk;
        while (*e > '\n') e++;
    }
    *begin = b;
    *stop = e;
}
#endif
