// pc_tokenize.h -- the reference's text tokenizer on the GPU: "x y z" lines -> binary32 triples.
//
// Restates parse_float / parse_line (reference src/parse.h:24-163) for device code, bit for bit
// for every token they accept: sign, one integer digit, '.', up to 17 decimals into a 64-bit
// integer (further decimals skipped), optional "e<int>"; the value is the TRUNCATED quotient,
// formed by repeated division by 5 on a left-justified word, and the leading-zero scan looks
// at the source characters [sign, number of decimals] only (parse.h:80-88) -- the result for
// "0.000004" or "-0.000045" depends on it.  Where the reference prints and exits (no '.' after
// the first digit: parse.h:35-39; positive decimal exponent: parse.h:119-122) a status bit is
// set instead.  k successive truncating divisions by 5 equal one by 5^k, so each run of up to
// 16 steps is one 64-bit division.
//
// Records.  The reference parses a file front to back: three tokens, each followed by exactly
// one separator character, then any run of control characters 1..31 (parse.h:150-163); its
// parser threads enter the file behind a byte <= '\n' (main.c:113-118), i.e. it relies on a
// record ending with such bytes and holding none.  Here a record starts at byte 0 and at
// every byte outside 1..31 that follows a byte inside 1..31: the same records for every file
// the reference's own threads parse correctly.
//
// Two passes over tiles of PC_TK_TILE bytes: T1 counts the record starts of each tile, a scan
// turns the counts into first record numbers, T2 stages its tile (plus PC_TK_TAIL bytes, the
// longest record taken) in LDS and lets the thread that owns a record's first byte parse it and
// write point number (first + rank) -- records keep their file order.
#pragma once
#include "pc_device.h"

#define PC_TK_TPB 256
#define PC_TK_BPT 32                               // bytes per thread
#define PC_TK_TILE (PC_TK_TPB * PC_TK_BPT)          // 8192 bytes
#define PC_TK_TAIL 256                             // a record may reach this far past its tile
#define PC_TK_ERR_NODOT 2                          // status bits (bit 0 stays "point outside the grid")
#define PC_TK_ERR_POSEXP 4
#define PC_TK_ERR_LONG 8                           // record longer than PC_TK_TAIL bytes
#define PC_TK_ERR_ROOM 16                          // more records than the output array holds

__device__ __forceinline__ bool pc_tk_control(uint32_t c) { return c - 1u < 31u; }
__device__ __forceinline__ bool pc_tk_digit(uint32_t c) { return c - 48u <= 9u; }

__device__ __forceinline__ uint64_t pc_tk_pow5(int k)           // 5^k, 0 <= k <= 16
{
    uint64_t r = 1;
    for (int i = 0; i < k; i++) r *= 5u;
    return r;
}

// One token starting at tok[0] (a window of `room` readable bytes; bytes past it read as 0).
// Returns the float's bits; *used = characters consumed; *err |= PC_TK_ERR_*.
__device__ __forceinline__ uint32_t pc_tk_float_bits(const unsigned char *tok, int room, int *used, int *err)
{
    auto at = [&](int i) -> uint32_t { return i < room ? tok[i] : 0u; };
    const int neg = at(0) == '-';
    int p = neg;
    uint64_t digits = (uint64_t)(int64_t)((int)at(p) - '0');           // (any byte: as the reference, no check)
    if (at(p + 1) != '.') { *err |= PC_TK_ERR_NODOT; *used = p + 1; return 0; }
    p += 2;
    int nd = 0;
    for (; nd < 17 && pc_tk_digit(at(p)); nd++, p++) digits = digits * 10u + (at(p) - '0');
    while (pc_tk_digit(at(p))) p++;
    int e10 = 0;
    if (at(p) == 'e') {
        // atoi(p + 1): optional blanks are not expected here; sign, digits
        int q = p + 1, sign = 1, v = 0;
        if (at(q) == '-') { sign = -1; q++; }
        else if (at(q) == '+') q++;
        while (pc_tk_digit(at(q))) { v = v * 10 + (int)(at(q) - '0'); q++; }
        e10 = sign * v;
        p += 2;                                                     // 'e' and one character (the sign), then digits
        while (pc_tk_digit(at(p))) p++;
    }
    *used = p;
    if (digits == 0) return 0;
    // zeros before the first non-zero digit, looking at source characters [neg, nd] only
    int zeros = 0, lead = 0;
    for (int i = neg; i <= nd; i++) {
        const uint32_t c = at(i);
        if (c > '0') { lead = (int)c - '0'; break; }
        zeros += c == '0';
    }
    e10 -= zeros;
    nd -= zeros;
    if (e10 > 0) { *err |= PC_TK_ERR_POSEXP; return 0; }
    uint64_t m = digits;
    int e2 = 0;
    for (int left = nd; left > 0; left -= 16) {
        m <<= __clzll((long long)m);
        m /= pc_tk_pow5(left < 16 ? left : 16);
    }
    if (e10 < 0) {
        bool first = true;
        for (int left = -e10; left > 0; left -= 16) {
            const int s = __clzll((long long)m);
            m <<= s;
            if (!first) e2 -= s;
            first = false;
            m /= pc_tk_pow5(left < 16 ? left : 16);
        }
        e2 += e10 - __clzll((long long)m);
    }
    const int sh = 40 - __clzll((long long)m);
    const uint32_t mant = sh >= 0 ? (uint32_t)(m >> sh) : 0u;
    const int lead_c = lead > 9 ? 9 : lead;                          // (a byte above '9' in the scan: the reference reads its table out of range)
    e2 += lead_c > 0 ? 31 - __clz(lead_c) : 0;                        // floor(log2(lead)): 0 0 1 1 2 2 2 2 3 3
    return ((uint32_t)neg << 31) | ((uint32_t)((e2 + 127) & 0xFF) << 23) | (mant & 0x7FFFFFu);
}

// record starts among the bytes [p0, p0 + PC_TK_BPT) of the text: bit b set = byte p0 + b starts a record
__device__ __forceinline__ uint32_t pc_tk_starts(const unsigned char *__restrict__ text, uint64_t size, uint64_t p0)
{
    uint32_t bits = 0;
    if (p0 >= size) return 0;
    uint32_t prev = p0 ? text[p0 - 1] : 1u;                           // byte 0 starts a record whatever it is
    const uint32_t cnt = (uint32_t)min((uint64_t)PC_TK_BPT, size - p0);
    for (uint32_t b = 0; b < cnt; b++) {
        const uint32_t c = text[p0 + b];
        if ((pc_tk_control(prev) && !pc_tk_control(c)) || (p0 + b == 0)) bits |= 1u << b;
        prev = c;
    }
    return bits;
}

// ---- T1: record starts per tile ------------------------------------------------------
__global__ void __launch_bounds__(PC_TK_TPB)
pc_t1_count_records(const unsigned char *__restrict__ text, uint64_t size, uint32_t *__restrict__ tile_count)
{
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    const uint64_t p0 = (uint64_t)blockIdx.x * PC_TK_TILE + (uint64_t)threadIdx.x * PC_TK_BPT;
    const uint32_t n = (uint32_t)__popc(pc_tk_starts(text, size, p0));
    const uint32_t wave_n = pc_wave_sum(n);
    if ((threadIdx.x & 63) == 0 && wave_n) atomicAdd(&total, wave_n);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

// ---- T1b: exclusive scan of the tile counts (one workgroup), total -> *n_records ------
__global__ void __launch_bounds__(1024)
pc_t1_scan_tiles(uint32_t *__restrict__ tile_count, uint32_t ntiles, unsigned long long *__restrict__ n_records)
{
    __shared__ uint32_t part[1024];
    const uint32_t per = (ntiles + 1023u) / 1024u;
    const uint32_t lo = min(ntiles, threadIdx.x * per), hi = min(ntiles, lo + per);
    uint32_t s = 0;
    for (uint32_t t = lo; t < hi; t++) s += tile_count[t];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t t = lo; t < hi; t++) {
        const uint32_t c = tile_count[t];
        tile_count[t] = run;
        run += c;
    }
    if (threadIdx.x == 1023) *n_records = part[1023];
}

// ---- T2: parse ----------------------------------------------------------------------
__global__ void __launch_bounds__(PC_TK_TPB)
pc_t2_parse_records(const unsigned char *__restrict__ text, uint64_t size, const uint32_t *__restrict__ tile_first,
                    float *__restrict__ xyz, uint64_t capacity, int32_t *__restrict__ status)
{
    __shared__ __attribute__((aligned(16))) unsigned char win[PC_TK_TILE + PC_TK_TAIL];
    __shared__ uint32_t wave_base[PC_TK_TPB / 64 + 1];
    const int tid = threadIdx.x;
    const uint64_t tile0 = (uint64_t)blockIdx.x * PC_TK_TILE;
    const uint32_t have = (uint32_t)min((uint64_t)(PC_TK_TILE + PC_TK_TAIL), size - tile0);
    // the tile and its tail -> LDS (16 bytes per thread and turn; the text buffer's start is 16-byte aligned)
    for (uint32_t o = 16u * tid; o < PC_TK_TILE + PC_TK_TAIL; o += 16u * PC_TK_TPB) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o + 16u <= have) v = *reinterpret_cast<const uint4 *>(text + tile0 + o);
        else
            for (uint32_t b = 0; b < 16u; b++)
                if (o + b < have) reinterpret_cast<unsigned char *>(&v)[b] = text[tile0 + o + b];
        *reinterpret_cast<uint4 *>(win + o) = v;
    }
    __syncthreads();
    // record starts of this thread's bytes, and their numbers
    const uint32_t l0 = (uint32_t)tid * PC_TK_BPT;
    uint32_t bits = 0;
    if (l0 < have && l0 < PC_TK_TILE) {
        uint32_t prev = l0 ? win[l0 - 1] : (tile0 ? text[tile0 - 1] : 1u);
        const uint32_t cnt = min((uint32_t)PC_TK_BPT, have - l0);
        for (uint32_t b = 0; b < cnt; b++) {
            const uint32_t c = win[l0 + b];
            if ((pc_tk_control(prev) && !pc_tk_control(c)) || (tile0 + l0 + b == 0)) bits |= 1u << b;
            prev = c;
        }
    }
    const uint32_t n = (uint32_t)__popc(bits);
    const uint32_t incl = pc_wave_incl_scan(n);
    if ((tid & 63) == 63) wave_base[(tid >> 6) + 1] = incl;
    if (tid == 0) wave_base[0] = 0;
    __syncthreads();
    uint32_t base = tile_first[blockIdx.x] + incl - n;
    for (int w = 1; w <= (tid >> 6); w++) base += wave_base[w];
    int err = 0;
    while (bits) {
        const uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        const uint32_t at = l0 + b;                                  // first byte of the record in the window
        const int room = (int)(have - at);                          // (bytes past the text read as 0)
        const unsigned char *s = win + at;
        uint32_t v[3] = {0, 0, 0};
        int pos = 0, rec_err = 0;
        for (int a = 0; a < 3; a++) {
            int used = 0;
            v[a] = pc_tk_float_bits(s + pos, room - pos, &used, &rec_err);
            pos += used + 1;                                        // exactly one separator character
            if (rec_err) break;
        }
        // a record that reaches the window's end inside the text was cut short: whatever it parsed as, it is too long
        if ((uint32_t)pos + at >= PC_TK_TILE + PC_TK_TAIL && tile0 + PC_TK_TILE + PC_TK_TAIL < size) rec_err = PC_TK_ERR_LONG;
        err |= rec_err;
        if ((uint64_t)base < capacity) {
            xyz[3 * (uint64_t)base] = __uint_as_float(v[0]);
            xyz[3 * (uint64_t)base + 1] = __uint_as_float(v[1]);
            xyz[3 * (uint64_t)base + 2] = __uint_as_float(v[2]);
        } else err |= PC_TK_ERR_ROOM;
        base++;
    }
    if (err) atomicOr(status, err);
}
