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
    /* (a byte above '9' inside the scanned characters -- malformed text -- makes the reference read its table,
       parse.h:132, out of range; here, as in the device tokenizer, it counts as a 9) */
    e2 += log2_digit[lead > 9 ? 9 : lead];
    return ((uint32_t)neg << 31) | ((uint32_t)((e2 + 127) & 0xFF) << 23) | (mant & 0x7FFFFFu);
}

/* ---- short cut for the common token shape "[-]d.dddddd" ------------------------
 * Same result as xyz_parse_float_bits, computed without per-character loops: the
 * eight characters are checked and converted as one 64-bit word, the leading-zero
 * scan is a count of trailing zero bytes, and the truncating divisions by 5^k use
 * a reciprocal with one correction step (exact).  Anything else -- other digit
 * counts, an exponent, a missing '.', fewer than XYZ_FAST_BYTES readable bytes --
 * goes to the general function.  The caller guarantees tok[0..XYZ_FAST_BYTES) is
 * readable. */
#define XYZ_FAST_BYTES 10           /* sign + 8 characters + the terminator */

/* floor(2^64 / 5^k), k = 0..6 (entry 0 unused) */
static const uint64_t xyz_recip5[7] = {
    0ull, 3689348814741910323ull, 737869762948382064ull, 147573952589676412ull,
    29514790517935282ull, 5902958103587056ull, 1180591620717411ull};

/* floor(m / 5^k) for 1 <= k <= 6 */
static inline uint64_t xyz_div_pow5(uint64_t m, int k)
{
    const uint64_t d = xyz_pow5[k];
    uint64_t q = (uint64_t)(((unsigned __int128)m * xyz_recip5[k]) >> 64);   /* floor(m/d) or one less */
    q += (m - q * d >= d);
    return q;
}

static inline uint32_t xyz_parse_float_bits_fast(const char *tok, const char **end, int *err)
{
    static const int8_t log2_digit[10] = {0, 0, 1, 1, 2, 2, 2, 2, 3, 3};
    const int neg = (*tok == '-');
    const char *p = tok + neg;
    uint64_t w;
    memcpy(&w, p, 8);
    /* little endian: byte 0 = first character.  x has digit values where digits are expected, 0 for the '.' */
    const uint64_t x = w ^ 0x3030303030302E30ull;
    const int shape_ok = ((((x + 0x7676767676767676ull) | x) & 0x8080808080808080ull) == 0) &   /* every byte <= 9 */
                         ((x & 0xFF00u) == 0) &                                              /* '.' in place */
                         ((unsigned char)p[8] <= ' ');                                       /* blank or control character next */
    if (__builtin_expect(!shape_ok, 0)) return xyz_parse_float_bits(tok, end, err);
    *end = p + 8;
    *err = 0;
    /* "0" d0 d1..d6 as eight digit bytes -> the integer d0d1d2d3d4d5d6 */
    const uint64_t d0 = x & 0xFFu;
    uint64_t v = (x & ~0xFFFFull) | (d0 << 8);
    v = (v * 10 + (v >> 8)) & 0x00FF00FF00FF00FFull;
    v = (v * 100 + (v >> 16)) & 0x0000FFFF0000FFFFull;
    const uint64_t digits = (v * 10000 + (v >> 32)) & 0xFFFFFFFFull;
    if (__builtin_expect(d0 != 0, 1)) {
        /* 1 <= |value| < 10: no leading zeros, one truncating division by 5^6.  The general function
           keeps the top 24 bits of floor(digits * 2^clz / 5^6); a floor of a floor is the floor of the
           whole quotient, so they are floor(digits * 2^t / 5^6) with t putting the result in [2^23, 2^24). */
        const int lg = log2_digit[d0];
        const uint32_t mant = (uint32_t)((digits << (17 - lg)) / 15625u);
        return ((uint32_t)neg << 31) | ((uint32_t)(127 + lg) << 23) | (mant & 0x7FFFFFu);
    }
    if (digits == 0) return 0;
    /* the zero scan looks at source characters [neg, 6]: d0 and the first 5 (4 after a sign) decimals.
       A marker bit one byte past them ends the count when they are all zero (no branch on the sign). */
    const uint64_t seen = x & (0x00FFFFFFFFFFFFFFull >> (8 * neg));
    const int byte = __builtin_ctzll(seen | (1ull << (8 * (7 - neg)))) >> 3;
    const int zeros = byte - 1;                           /* d0 is '0' here; the '.' byte is not a zero digit */
    const int lead = (int)((seen >> (8 * byte)) & 0xFF);  /* 0 if every scanned digit is zero */
    const int nd = 6 - zeros;
    uint64_t m = digits;
    if (nd > 0) m = xyz_div_pow5(m << __builtin_clzll(m), nd);
    m = xyz_div_pow5(m << __builtin_clzll(m), zeros);
    const int e2 = -zeros - __builtin_clzll(m) + log2_digit[lead];
    const int sh = 40 - __builtin_clzll(m);
    const uint32_t mant = (sh >= 0) ? (uint32_t)(m >> sh) : 0u;
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

/* the same line when s[0..limit) is known to be readable: a line that starts far
   enough before limit for three short-cut tokens tries the short cut on each */
static inline const char *xyz_parse_line_bounded(const char *s, const char *limit, float out[3], int *err)
{
    if (limit - s < 3 * XYZ_FAST_BYTES + 2) return xyz_parse_line(s, out, err);
    for (int a = 0; a < 3; a++) {
        const char *e;
        const uint32_t b = xyz_parse_float_bits_fast(s, &e, err);
        if (*err) return e;
        memcpy(out + a, &b, 4);
        s = e + 1;
    }
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
