// cv_inflate.cpp -- raw DEFLATE (RFC 1951) decoder for whole in-memory blocks, host only.
//
// The native BAM reader (cv_bam.cpp) spends three quarters of its time in zlib's inflate() on 64 KiB BGZF blocks.
// This is the usual table-driven decoder for that case -- the whole compressed block and the exact output size
// are known, so there is no streaming state: a 64-bit bit buffer refilled with one unaligned load, an 11-bit
// first-level table for the literal/length code and an 8-bit one for the distance code (longer codes go through
// 16- / 128-entry second-level tables), literals and matches written straight into the caller's buffer with
// 8-byte copies where the distance allows.  Every BGZF block carries a CRC-32 of its inflated bytes, which the
// caller checks; a block this decoder rejects (or gets wrong) is simply handed to zlib.
//
// No reference counterpart: the reference shells out to `samtools view` (CreateTensor.py:128-130).
#include <cstdint>
#include <cstring>
#include <type_traits>

namespace {

constexpr int LBITS = 11, DBITS = 8, MAXBITS = 15;
constexpr int LSUB = 1 << (MAXBITS - LBITS), DSUB = 1 << (MAXBITS - DBITS);

// table entry: bits 0..7 total code length (0 = invalid), bits 8..12 extra bits, bits 13..15 kind,
// bits 16..31 value (literal byte / base length / base distance / second-level table number)
enum : uint32_t { K_LIT = 0u << 13, K_LEN = 1u << 13, K_EOB = 2u << 13, K_SUB = 3u << 13, K_MASK = 7u << 13 };

inline uint32_t entry(uint32_t value, uint32_t kind, uint32_t extra, uint32_t len) { return (value << 16) | kind | (extra << 8) | len; }

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct tables {
    uint32_t lit[1 << LBITS];
    uint32_t dist[1 << DBITS];
    uint32_t lsub[288 * LSUB];      // at most one second-level table per long code
    uint32_t dsub[32 * DSUB];
};

inline uint32_t reverse_bits(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; i++) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// canonical Huffman code of `lens` -> decode tables.  make(sym, len) gives the entry of a symbol.
// Returns false for an over-subscribed code; an incomplete code leaves invalid (zero) entries behind, which
// the decoder reports when it meets one (a single distance code of one bit is legal and ends up that way).
template <typename MAKE>
bool build(const uint8_t *lens, int n, uint32_t *first, int fbits, uint32_t *sub, int subsize, int max_sub, MAKE make)
{
    int count[MAXBITS + 1] = {0};
    for (int i = 0; i < n; i++) count[lens[i]]++;
    count[0] = 0;
    uint32_t next[MAXBITS + 2];
    uint32_t code = 0;
    int64_t left = 1;
    for (int l = 1; l <= MAXBITS; l++) {
        left = left * 2 - count[l];
        if (left < 0) return false;
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    memset(first, 0, sizeof(uint32_t) << fbits);
    int nsub = 0;
    for (int s = 0; s < n; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = reverse_bits(next[l]++, l);
        if (l <= fbits) {
            const uint32_t e = make(s, l);
            for (uint32_t i = rev; i < (1u << fbits); i += 1u << l) first[i] = e;
        } else {
            const uint32_t prefix = rev & ((1u << fbits) - 1);
            uint32_t t;
            if ((first[prefix] & K_MASK) == K_SUB && (first[prefix] & 0xff)) {
                t = first[prefix] >> 16;
            } else {
                if (nsub >= max_sub) return false;
                t = (uint32_t)nsub++;
                memset(sub + (size_t)t * subsize, 0, sizeof(uint32_t) * (size_t)subsize);
                first[prefix] = entry(t, K_SUB, 0, (uint32_t)fbits);
            }
            const uint32_t e = make(s, l);
            uint32_t *tab = sub + (size_t)t * subsize;
            for (uint32_t i = rev >> fbits; i < (uint32_t)subsize; i += 1u << (l - fbits)) tab[i] = e;
        }
    }
    return true;
}

inline uint32_t make_litlen(int s, int l)
{
    if (s < 256) return entry((uint32_t)s, K_LIT, 0, (uint32_t)l);
    if (s == 256) return entry(0, K_EOB, 0, (uint32_t)l);
    if (s > 285) return 0;                                          // 286, 287: never valid in a stream
    return entry(LEN_BASE[s - 257], K_LEN, LEN_EXTRA[s - 257], (uint32_t)l);
}

inline uint32_t make_dist(int s, int l)
{
    if (s > 29) return 0;
    return entry(DIST_BASE[s], K_LEN, DIST_EXTRA[s], (uint32_t)l);
}

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

}  // namespace

// src[0, n): one complete raw-DEFLATE stream, followed by at least 8 readable bytes (BGZF: the CRC-32 / ISIZE
// trailer).  dst[0, cap): the output; the stream must produce exactly `cap` bytes.  Returns cap, or -1 if the
// stream is malformed, too long or too short for `cap` (the caller falls back to zlib then).
// The decoder.  Whole-stream mode (stream == nullptr): as described above.  Streaming mode (cv_inflate_stream): starts at
// bit `stream->bitpos` of src, appends to dst behind `have` bytes of history (matches may reach back into them), and
// returns at the first BLOCK boundary at which at least `want` bytes have been produced, or at the end of the stream.
namespace {
struct stream_io { int64_t bitpos; int64_t have; int64_t want; int final; };
}
static int64_t inflate_core(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, stream_io *stream)
{
    if (!src || n < 0 || cap < 0 || (cap > 0 && !dst)) return -1;
    static thread_local tables T;
    const uint8_t *p = src, *const pend = src + n;
    uint8_t *out = dst, *const oend = dst + cap;
    uint64_t buf = 0;
    int cnt = 0;
    if (stream) {
        if (stream->bitpos < 0 || (stream->bitpos >> 3) > n || stream->have < 0 || stream->have > cap) return -1;
        p = src + (stream->bitpos >> 3);
        out = dst + stream->have;
        stream->final = 0;
    }
    uint8_t *const out0 = out;
    // top the bit buffer up to 56..63 bits while there is input left (the load reads at most 7 bytes past pend:
    // allowed, see above).  Once every byte of the stream is in the buffer nothing is added; a valid stream never
    // asks for more bits than it has, an invalid one runs the count negative and is rejected.
#define REFILL() do { if (cnt < 0) return -1; if (p <= pend) { buf |= load64(p) << cnt; p += (63 - cnt) >> 3; cnt |= 56; } } while (0)
#define DROP(k) do { buf >>= (k); cnt -= (int)(k); } while (0)
    bool last = false;
    if (stream && (stream->bitpos & 7)) { REFILL(); DROP(stream->bitpos & 7); }
    while (!last) {
        if (stream && out - out0 >= stream->want) {             // a block boundary with enough output: hand over
            stream->bitpos = (int64_t)(p - src) * 8 - cnt;
            return out - out0;
        }
        REFILL();
        last = buf & 1;
        const int type = (int)((buf >> 1) & 3);
        DROP(3);
        if (type == 0) {                                             // stored
            if (cnt < 0) return -1;
            DROP(cnt & 7);
            const uint8_t *q = p - (cnt >> 3);                       // first byte not yet consumed
            if (q + 4 > pend) return -1;
            const uint32_t len = (uint32_t)q[0] | ((uint32_t)q[1] << 8), nlen = (uint32_t)q[2] | ((uint32_t)q[3] << 8);
            if ((len ^ nlen) != 0xffff) return -1;
            q += 4;
            if (q + len > pend || out + len > oend) return -1;
            memcpy(out, q, len);
            out += len;
            p = q + len; buf = 0; cnt = 0;
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {                                             // fixed code
            uint8_t lens[288 + 32];
            int i = 0;
            for (; i < 144; i++) lens[i] = 8;
            for (; i < 256; i++) lens[i] = 9;
            for (; i < 280; i++) lens[i] = 7;
            for (; i < 288; i++) lens[i] = 8;
            for (i = 0; i < 32; i++) lens[288 + i] = 5;
            if (!build(lens, 288, T.lit, LBITS, T.lsub, LSUB, 288, make_litlen)) return -1;
            if (!build(lens + 288, 32, T.dist, DBITS, T.dsub, DSUB, 32, make_dist)) return -1;
        } else {                                                     // dynamic code
            const int hlit = (int)(buf & 31) + 257, hdist = (int)((buf >> 5) & 31) + 1, hclen = (int)((buf >> 10) & 15) + 4;
            DROP(14);
            if (hlit > 286 || hdist > 30) return -1;
            static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            REFILL();
            for (int i = 0; i < hclen; i++) {                        // 19 x 3 = 57 bits: one refill in between
                if (i == 12) REFILL();
                cl[ORDER[i]] = (uint8_t)(buf & 7);
                DROP(3);
            }
            uint32_t cltab[128];
            uint32_t nosub[1];
            if (!build(cl, 19, cltab, 7, nosub, 1, 0, [](int s, int l) { return entry((uint32_t)s, K_LIT, 0, (uint32_t)l); }))
                return -1;
            uint8_t lens[288 + 32];
            memset(lens, 0, sizeof(lens));
            uint8_t all[286 + 30];
            const int total = hlit + hdist;
            int i = 0;
            while (i < total) {
                REFILL();
                const uint32_t e = cltab[buf & 127];
                if (!(e & 0xff)) return -1;
                DROP(e & 0xff);
                const int sym = (int)(e >> 16);
                if (sym < 16) { all[i++] = (uint8_t)sym; continue; }
                int rep; uint8_t v = 0;
                if (sym == 16) {
                    if (i == 0) return -1;
                    v = all[i - 1]; rep = 3 + (int)(buf & 3); DROP(2);
                } else if (sym == 17) { rep = 3 + (int)(buf & 7); DROP(3); }
                else { rep = 11 + (int)(buf & 127); DROP(7); }
                if (i + rep > total) return -1;
                while (rep--) all[i++] = v;
            }
            if (all[256] == 0) return -1;                            // no end-of-block code
            memcpy(lens, all, (size_t)hlit);
            memcpy(lens + 288, all + hlit, (size_t)hdist);
            if (!build(lens, 288, T.lit, LBITS, T.lsub, LSUB, 288, make_litlen)) return -1;
            if (!build(lens + 288, 32, T.dist, DBITS, T.dsub, DSUB, 32, make_dist)) return -1;
        }
        // ---- the symbols of this block.  FAST: input left to load and room for three literals plus the longest
        // match with its copy slop -- no per-symbol bounds checks on the output, no bit-count checks.
        // -> 0 go on, 1 end of block, -1 malformed
        auto symbols = [&](auto fast_tag) -> int {
            constexpr bool FAST = decltype(fast_tag)::value;
            for (;;) {
                if constexpr (FAST) {
                    if (!(p <= pend && oend - out >= 3 + 258 + 8)) return 0;
                    buf |= load64(p) << cnt; p += (63 - cnt) >> 3; cnt |= 56;
                } else {
                    REFILL();
                }
                uint32_t e = T.lit[buf & ((1u << LBITS) - 1)];
                if ((e & K_MASK) == K_SUB) e = T.lsub[(size_t)(e >> 16) * LSUB + ((buf >> LBITS) & (LSUB - 1))];
                if (!(e & 0xff)) return -1;
                DROP(e & 0xff);
                if ((e & K_MASK) == K_LIT) {
                    if (!FAST && out >= oend) return -1;
                    *out++ = (uint8_t)(e >> 16);
                    // second and third literal out of the same refill (15 + 10 + 10 bits < 56)
                    e = T.lit[buf & ((1u << LBITS) - 1)];
                    if ((e & (K_MASK | 0xff00)) == K_LIT && (e & 0xff) && (FAST || out < oend)) {
                        DROP(e & 0xff);
                        *out++ = (uint8_t)(e >> 16);
                        e = T.lit[buf & ((1u << LBITS) - 1)];
                        if ((e & (K_MASK | 0xff00)) == K_LIT && (e & 0xff) && (FAST || out < oend)) {
                            DROP(e & 0xff);
                            *out++ = (uint8_t)(e >> 16);
                        }
                    }
                    continue;
                }
                if ((e & K_MASK) == K_EOB) return 1;
                // length / distance pair: up to 15 + 5 + 15 + 13 = 48 bits, all in the buffer already
                const uint32_t lx = (e >> 8) & 31;
                const uint32_t len = (e >> 16) + (uint32_t)(buf & ((1u << lx) - 1));
                DROP(lx);
                uint32_t d = T.dist[buf & ((1u << DBITS) - 1)];
                if ((d & K_MASK) == K_SUB) d = T.dsub[(size_t)(d >> 16) * DSUB + ((buf >> DBITS) & (DSUB - 1))];
                if (!(d & 0xff)) return -1;
                DROP(d & 0xff);
                const uint32_t dx = (d >> 8) & 31;
                const uint32_t dist = (d >> 16) + (uint32_t)(buf & ((1u << dx) - 1));
                DROP(dx);
                if (!FAST && cnt < 0) return -1;                     // ran past the end of the input
                if (dist > (uint32_t)(out - dst)) return -1;
                if (!FAST && len > (uint32_t)(oend - out)) return -1;
                const uint8_t *from = out - dist;
                if (dist >= 8 && (FAST || (uint32_t)(oend - out) >= len + 8)) {   // wide copy, up to 7 bytes of slop
                    uint8_t *o = out, *const oe = out + len;
                    do { memcpy(o, from, 8); o += 8; from += 8; } while (o < oe);
                } else if (dist == 1) {
                    memset(out, *from, len);
                } else {
                    for (uint32_t k = 0; k < len; k++) out[k] = from[k];
                }
                out += len;
            }
        };
        int st = symbols(std::true_type());
        if (st == 0) st = symbols(std::false_type());
        if (st < 0) return -1;
    }
#undef REFILL
#undef DROP
    if (stream) {
        if (cnt < 0) return -1;
        stream->bitpos = (int64_t)(p - src) * 8 - cnt;
        stream->final = 1;
        return out - out0;
    }
    return (out == oend && cnt >= 0) ? cap : -1;
}

extern "C" int64_t cv_inflate_raw(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap)
{
    return inflate_core(src, n, dst, cap, nullptr);
}

// Streaming form for a gzip member that does not fit in memory at once (the text-tensor files of utils_v2.GetTensor):
// src[0, n) = the raw-DEFLATE data from its first byte on (the caller has skipped the gzip header), followed by at
// least 8 readable bytes (the member's CRC-32 / ISIZE trailer); *bitpos = where to go on (0 at the start).  dst[0, have)
// holds the output produced last (at least the last 32 768 bytes of it, or all of it if less); new output is appended at
// dst + have, at most up to dst + cap.  Returns when a block ends with >= want new bytes, or at the end of the stream
// (*final = 1; *bitpos then points behind the last block: the trailer starts at the next byte boundary).  Returns the
// number of new bytes, or -1 (malformed, or a block that does not fit: cap - have must leave room for want plus the
// largest block -- the caller falls back to an external gzip).
extern "C" int64_t cv_inflate_stream(const uint8_t *src, int64_t n, int64_t *bitpos, uint8_t *dst, int64_t have, int64_t cap,
                                     int64_t want, int32_t *final)
{
    if (!bitpos || !final) return -1;
    stream_io io{*bitpos, have, want, 0};
    const int64_t got = inflate_core(src, n, dst, cap, &io);
    if (got >= 0) { *bitpos = io.bitpos; *final = io.final; }
    return got;
}

// CRC-32 of the gzip / BGZF trailer (polynomial 0xEDB88320, reflected), sixteen bytes per step through sixteen
// tables ("slicing"): ~3x zlib 1.2.11's byte-wise tables, which would otherwise cost as much as the decoder above.
namespace {
struct crc_tables {
    uint32_t t[16][256];
    crc_tables()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (int k = 1; k < 16; k++)
            for (uint32_t i = 0; i < 256; i++) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
};
const crc_tables CRC;
inline uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
}  // namespace

extern "C" uint32_t cv_crc32_ieee(uint32_t crc, const uint8_t *p, int64_t n)
{
    crc = ~crc;
    while (n >= 16) {
        const uint32_t a = load32(p) ^ crc, b = load32(p + 4), c = load32(p + 8), d = load32(p + 12);
        crc = CRC.t[15][a & 0xff] ^ CRC.t[14][(a >> 8) & 0xff] ^ CRC.t[13][(a >> 16) & 0xff] ^ CRC.t[12][a >> 24] ^
              CRC.t[11][b & 0xff] ^ CRC.t[10][(b >> 8) & 0xff] ^ CRC.t[9][(b >> 16) & 0xff] ^ CRC.t[8][b >> 24] ^
              CRC.t[7][c & 0xff] ^ CRC.t[6][(c >> 8) & 0xff] ^ CRC.t[5][(c >> 16) & 0xff] ^ CRC.t[4][c >> 24] ^
              CRC.t[3][d & 0xff] ^ CRC.t[2][(d >> 8) & 0xff] ^ CRC.t[1][(d >> 16) & 0xff] ^ CRC.t[0][d >> 24];
        p += 16; n -= 16;
    }
    while (n-- > 0) crc = CRC.t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
    return ~crc;
}
