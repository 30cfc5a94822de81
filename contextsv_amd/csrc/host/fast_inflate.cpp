#include "fast_inflate.h"

#include <cstring>
#include <mutex>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace fastz {

namespace {

constexpr int LITLEN_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
constexpr int LITLEN_CAP = 4096, DIST_CAP = 1024;

// table entry: bits 0-7 codeword bits to consume | 8-12 extra bits (or sub-table index bits) | 13-15 kind | 16-31 value
enum : uint32_t { K_INVALID = 0, K_LITERAL = 1, K_LENGTH = 2, K_END = 3, K_SUB = 4, K_DIST = 5 };
inline uint32_t mk(uint32_t kind, uint32_t value, uint32_t extra) { return (value << 16) | (kind << 13) | (extra << 8); }
inline uint32_t e_bits(uint32_t e) { return e & 0xff; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 0x1f; }
inline uint32_t e_kind(uint32_t e) { return (e >> 13) & 7; }
inline uint32_t e_value(uint32_t e) { return e >> 16; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t litlen_entry(int sym)
{
    if (sym < 256) return mk(K_LITERAL, (uint32_t)sym, 0);
    if (sym == 256) return mk(K_END, 0, 0);
    if (sym <= 285) return mk(K_LENGTH, kLenBase[sym - 257], kLenExtra[sym - 257]);
    return mk(K_INVALID, 0, 0);                       // 286, 287 take part in the code but may not occur
}
inline uint32_t dist_entry(int sym) { return sym < 30 ? mk(K_DIST, kDistBase[sym], kDistExtra[sym]) : mk(K_INVALID, 0, 0); }
inline uint32_t pre_entry(int sym) { return mk(K_LITERAL, (uint32_t)sym, 0); }

inline uint32_t reverse_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// Canonical Huffman code -> look-up table indexed by the next `tb` stream bits (LSB first), with sub-tables for longer codes.
// A complete code is required, except for the one-symbol code RFC 1951 allows (a single distance code of one bit).
template <class EntryOf>
bool build_table(const uint8_t *lens, int n_sym, int tb, uint32_t *table, int cap, EntryOf entry_of)
{
    int count[16] = {0};
    for (int s = 0; s < n_sym; s++) count[lens[s]]++;
    const int used = n_sym - count[0];
    if (used == 0) return false;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = left * 2 - count[l]; if (left < 0) return false; }
    if (left > 0 && !(used == 1 && count[1] == 1)) return false;
    int offs[17];
    offs[1] = 0;
    for (int l = 1; l <= 15; l++) offs[l + 1] = offs[l] + count[l];
    uint16_t sorted[320];
    {
        int at[17];
        memcpy(at, offs, sizeof at);
        for (int s = 0; s < n_sym; s++) if (lens[s]) sorted[at[lens[s]]++] = (uint16_t)s;
    }
    const uint32_t primary = 1u << tb;
    for (uint32_t i = 0; i < primary; i++) table[i] = mk(K_INVALID, 0, 0);
    // longest code behind every primary slot that needs a sub-table
    uint8_t sub_max[1 << LITLEN_BITS];
    memset(sub_max, 0, primary);
    uint32_t code = 0;
    for (int l = 1; l <= 15; l++) {
        for (int k = offs[l]; k < offs[l + 1]; k++, code++) {
            if (l > tb) {
                const uint32_t slot = reverse_bits(code, l) & (primary - 1);
                if (l > sub_max[slot]) sub_max[slot] = (uint8_t)l;
            }
        }
        code <<= 1;
    }
    uint32_t next_free = primary;
    for (uint32_t slot = 0; slot < primary; slot++) {
        if (!sub_max[slot]) continue;
        const uint32_t sb = (uint32_t)sub_max[slot] - (uint32_t)tb;
        if (next_free + (1u << sb) > (uint32_t)cap) return false;
        table[slot] = mk(K_SUB, next_free, sb) | (uint32_t)tb;
        for (uint32_t i = 0; i < (1u << sb); i++) table[next_free + i] = mk(K_INVALID, 0, 0);
        next_free += 1u << sb;
    }
    code = 0;
    for (int l = 1; l <= 15; l++) {
        for (int k = offs[l]; k < offs[l + 1]; k++, code++) {
            const uint32_t rev = reverse_bits(code, l);
            const uint32_t e = entry_of(sorted[k]);
            if (l <= tb) {
                for (uint32_t i = rev; i < primary; i += 1u << l) table[i] = e | (uint32_t)l;
            } else {
                const uint32_t p = table[rev & (primary - 1)];
                const uint32_t base = e_value(p), sb = e_extra(p);
                for (uint32_t i = rev >> tb; i < (1u << sb); i += 1u << (l - tb)) table[base + i] = e | (uint32_t)(l - tb);
            }
        }
        code <<= 1;
    }
    return true;
}

struct Tables {
    uint32_t litlen[LITLEN_CAP];
    uint32_t dist[DIST_CAP];
};

const Tables &fixed_tables()
{
    static Tables t;
    static std::once_flag once;
    std::call_once(once, [] {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        build_table(l, 288, LITLEN_BITS, t.litlen, LITLEN_CAP, litlen_entry);
        uint8_t d[32];
        for (int i = 0; i < 32; i++) d[i] = 5;
        build_table(d, 32, DIST_BITS, t.dist, DIST_CAP, dist_entry);
    });
    return t;
}

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

struct BitReader {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    uint32_t cnt = 0;            // valid bits in buf
    bool over = false;           // asked for more bits than the input holds
    inline void refill()
    {
        if (end - p >= 8) {      // eight bytes at once; the bytes that do not fit stay unconsumed
            buf |= load64(p) << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
        } else {
            while (cnt <= 56 && p < end) { buf |= (uint64_t)*p++ << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(uint32_t n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }
    inline void drop(uint32_t n)
    {
        if (n > cnt) { over = true; n = cnt; }
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

}  // namespace

bool inflate(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len)
{
    BitReader br;
    br.p = in;
    br.end = in + in_len;
    uint8_t *op = out, *const oend = out + out_len;
    Tables dyn;
    for (;;) {
        br.refill();
        const uint32_t bfinal = br.take(1), btype = br.take(2);
        if (br.over) return false;
        if (btype == 0) {                               // stored: back to a byte boundary, LEN, ~LEN, bytes
            br.drop(br.cnt & 7);
            br.refill();
            if (br.cnt < 32) return false;
            const uint32_t len = br.take(16), nlen = br.take(16);
            if ((len ^ nlen) != 0xffffu) return false;
            // hand the whole bytes still in the bit buffer back to the input
            br.p -= br.cnt >> 3;
            br.buf = 0; br.cnt = 0;
            if ((size_t)(br.end - br.p) < len || (size_t)(oend - op) < len) return false;
            memcpy(op, br.p, len);
            op += len;
            br.p += len;
        } else if (btype == 1 || btype == 2) {
            const Tables *t = &fixed_tables();
            if (btype == 2) {
                br.refill();
                const uint32_t hlit = br.take(5) + 257, hdist = br.take(5) + 1, hclen = br.take(4) + 4;
                if (hlit > 286 || hdist > 30) return false;
                uint8_t pre_lens[19] = {0};
                for (uint32_t i = 0; i < hclen; i++) {
                    if (br.cnt < 3) br.refill();
                    pre_lens[kPreOrder[i]] = (uint8_t)br.take(3);
                }
                if (br.over) return false;
                uint32_t pre[1 << PRE_BITS];
                if (!build_table(pre_lens, 19, PRE_BITS, pre, 1 << PRE_BITS, pre_entry)) return false;
                uint8_t lens[288 + 32];
                uint32_t n = 0;
                const uint32_t total = hlit + hdist;
                while (n < total) {
                    br.refill();
                    const uint32_t e = pre[br.peek(PRE_BITS)];
                    if (e_kind(e) != K_LITERAL) return false;
                    br.drop(e_bits(e));
                    const uint32_t sym = e_value(e);
                    if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                    uint32_t rep, val = 0;
                    if (sym == 16) { if (n == 0) return false; val = lens[n - 1]; rep = 3 + br.take(2); }
                    else if (sym == 17) rep = 3 + br.take(3);
                    else rep = 11 + br.take(7);
                    if (n + rep > total) return false;
                    memset(lens + n, (int)val, rep);
                    n += rep;
                }
                if (br.over || lens[256] == 0) return false;      // no end-of-block code: not decodable
                uint8_t ll[288] = {0}, dl[32] = {0};
                memcpy(ll, lens, hlit);
                memcpy(dl, lens + hlit, hdist);
                if (!build_table(ll, 288, LITLEN_BITS, dyn.litlen, LITLEN_CAP, litlen_entry)) return false;
                if (!build_table(dl, 32, DIST_BITS, dyn.dist, DIST_CAP, dist_entry)) {
                    // a block without matches may carry no usable distance code at all: decodable as long as no length code occurs
                    bool any = false;
                    for (uint32_t i = 0; i < hdist; i++) any |= dl[i] != 0;
                    if (any) return false;
                    for (int i = 0; i < (1 << DIST_BITS); i++) dyn.dist[i] = mk(K_INVALID, 0, 0);
                }
                t = &dyn;
            }
            for (;;) {
                // ---- fast section: at least 16 input bytes and 300 output bytes ahead, so no per-symbol bounds or underflow checks
                // (a symbol takes at most 48 bits and writes at most 258 + 7 bytes) ----
                bool block_done = false;
                while ((size_t)(br.end - br.p) >= 16 && (size_t)(oend - op) >= 300) {
                    br.buf |= load64(br.p) << br.cnt; br.p += (63 - br.cnt) >> 3; br.cnt |= 56;
                    uint32_t e = t->litlen[br.buf & ((1u << LITLEN_BITS) - 1)];
                    if (e_kind(e) == K_LITERAL) {                       // up to three literals from one refill (<= 33 bits)
                        br.buf >>= e_bits(e); br.cnt -= e_bits(e); *op++ = (uint8_t)(e >> 16);
                        e = t->litlen[br.buf & ((1u << LITLEN_BITS) - 1)];
                        if (e_kind(e) == K_LITERAL) {
                            br.buf >>= e_bits(e); br.cnt -= e_bits(e); *op++ = (uint8_t)(e >> 16);
                            e = t->litlen[br.buf & ((1u << LITLEN_BITS) - 1)];
                            if (e_kind(e) == K_LITERAL) { br.buf >>= e_bits(e); br.cnt -= e_bits(e); *op++ = (uint8_t)(e >> 16); continue; }
                        }
                        br.buf |= load64(br.p) << br.cnt; br.p += (63 - br.cnt) >> 3; br.cnt |= 56;
                    }
                    if (e_kind(e) == K_SUB) {
                        br.buf >>= e_bits(e); br.cnt -= e_bits(e);
                        e = t->litlen[e_value(e) + (uint32_t)(br.buf & ((1u << e_extra(e)) - 1))];
                    }
                    br.buf >>= e_bits(e); br.cnt -= e_bits(e);
                    const uint32_t kind = e_kind(e);
                    if (kind == K_LITERAL) { *op++ = (uint8_t)(e >> 16); continue; }
                    if (kind == K_END) { block_done = true; break; }
                    if (kind != K_LENGTH) return false;
                    const uint32_t xb = e_extra(e);
                    const uint32_t len = e_value(e) + (uint32_t)(br.buf & ((1u << xb) - 1));
                    br.buf >>= xb; br.cnt -= xb;
                    uint32_t d = t->dist[br.buf & ((1u << DIST_BITS) - 1)];
                    if (e_kind(d) == K_SUB) {
                        br.buf >>= e_bits(d); br.cnt -= e_bits(d);
                        d = t->dist[e_value(d) + (uint32_t)(br.buf & ((1u << e_extra(d)) - 1))];
                    }
                    if (e_kind(d) != K_DIST) return false;
                    br.buf >>= e_bits(d); br.cnt -= e_bits(d);
                    const uint32_t db = e_extra(d);
                    const uint32_t dist = e_value(d) + (uint32_t)(br.buf & ((1u << db) - 1));
                    br.buf >>= db; br.cnt -= db;
                    if (dist > (size_t)(op - out)) return false;
                    const uint8_t *src = op - dist;
                    uint8_t *dst = op, *const stop = op + len;
                    if (dist >= 8) {
                        do { store64(dst, load64(src)); dst += 8; src += 8; } while (dst < stop);
                    } else if (dist == 1) {
                        memset(dst, *src, len);
                    } else {
                        do { *dst++ = *src++; } while (dst < stop);
                    }
                    op = stop;
                }
                if (block_done) break;
                // ---- careful section: one symbol with every check, near the end of the input or of the output ----
                br.refill();
                uint32_t e = t->litlen[br.peek(LITLEN_BITS)];
                if (e_kind(e) == K_SUB) {
                    br.drop(e_bits(e));
                    e = t->litlen[e_value(e) + br.peek(e_extra(e))];
                }
                br.drop(e_bits(e));
                const uint32_t kind = e_kind(e);
                if (kind == K_LITERAL) {
                    if (op >= oend) return false;
                    *op++ = (uint8_t)e_value(e);
                    continue;
                }
                if (kind == K_END) break;
                if (kind != K_LENGTH) return false;
                const uint32_t len = e_value(e) + br.take(e_extra(e));
                uint32_t d = t->dist[br.peek(DIST_BITS)];
                if (e_kind(d) == K_SUB) {
                    br.drop(e_bits(d));
                    d = t->dist[e_value(d) + br.peek(e_extra(d))];
                }
                if (e_kind(d) != K_DIST) return false;
                br.drop(e_bits(d));
                const uint32_t dist = e_value(d) + br.take(e_extra(d));
                if (br.over) return false;
                if (dist > (size_t)(op - out) || len > (size_t)(oend - op)) return false;
                const uint8_t *src = op - dist;
                for (uint32_t i = 0; i < len; i++) op[i] = src[i];
                op += len;
            }
            if (br.over) return false;
        } else {
            return false;
        }
        if (bfinal) break;
    }
    // exactly the announced output, and nothing but the padding of the last byte left in the input
    if (op != oend || br.over) return false;
    const size_t unread = (size_t)(br.end - br.p) + (br.cnt >> 3);
    return unread == 0;
}

// ---- CRC-32 -----------------------------------------------------------------------------------------------------------
namespace {

struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};
const CrcTables kCrc;

// slicing-by-8; `c` is the running (pre-inverted) register
uint32_t crc_table(uint32_t c, const uint8_t *p, size_t n)
{
    while (n && ((uintptr_t)p & 7)) { c = kCrc.t[0][(c ^ *p++) & 0xff] ^ (c >> 8); n--; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = kCrc.t[7][v & 0xff] ^ kCrc.t[6][(v >> 8) & 0xff] ^ kCrc.t[5][(v >> 16) & 0xff] ^ kCrc.t[4][(v >> 24) & 0xff] ^
            kCrc.t[3][(v >> 32) & 0xff] ^ kCrc.t[2][(v >> 40) & 0xff] ^ kCrc.t[1][(v >> 48) & 0xff] ^ kCrc.t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = kCrc.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
// Folding with carry-less multiplication (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ"):
// four 128-bit lanes folded 64 bytes at a time, reduced to one lane, then Barrett reduction. `c` = running register,
// n >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc_clmul(uint32_t c, const uint8_t *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596ll, 0x0154442bd4ll);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009ell, 0x01751997d0ll);
    const __m128i k5k0 = _mm_set_epi64x(0x0000000000ll, 0x0163cd6124ll);
    const __m128i poly = _mm_set_epi64x(0x01f7011641ll, 0x01db710641ll);
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i *)(p + 0x00));
    x2 = _mm_loadu_si128((const __m128i *)(p + 0x10));
    x3 = _mm_loadu_si128((const __m128i *)(p + 0x20));
    x4 = _mm_loadu_si128((const __m128i *)(p + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
    x0 = k1k2;
    p += 64; n -= 64;
    while (n >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i *)(p + 0x00));
        y6 = _mm_loadu_si128((const __m128i *)(p + 0x10));
        y7 = _mm_loadu_si128((const __m128i *)(p + 0x20));
        y8 = _mm_loadu_si128((const __m128i *)(p + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        p += 64; n -= 64;
    }
    // four lanes -> one
    x0 = k3k4;
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    // remaining 16-byte pieces
    while (n >= 16) {
        x2 = _mm_loadu_si128((const __m128i *)p);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        p += 16; n -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = k5k0;
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    // Barrett reduction to 32 bits
    x0 = poly;
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

bool clmul_usable()
{
    static const bool ok = [] {
        if (!__builtin_cpu_supports("pclmul") || !__builtin_cpu_supports("sse4.1")) return false;
        uint8_t probe[64 * 3 + 48];                      // the folded path must agree with the tables before it is trusted
        for (size_t i = 0; i < sizeof probe; i++) probe[i] = (uint8_t)(i * 131 + 7);
        return crc_clmul(0xffffffffu, probe, sizeof probe) == crc_table(0xffffffffu, probe, sizeof probe);
    }();
    return ok;
}
#endif

}  // namespace

uint32_t crc32(const uint8_t *buf, size_t len)
{
    uint32_t c = 0xffffffffu;
#if defined(__x86_64__)
    if (len >= 128 && clmul_usable()) {
        const size_t body = len & ~(size_t)15;
        c = crc_clmul(c, buf, body);
        buf += body;
        len -= body;
    }
#endif
    return ~crc_table(c, buf, len);
}

}  // namespace fastz
