// cv_hostio.cpp -- host-side data plane of the hot path (no GPU code):
//   * cv_parse_tensor_text: the text-tensor reader of utils_v2.GetTensor
//     (/root/reference/clairvoyante/utils_v2.py:20-59; producer format
//     /root/reference/dataPrepScripts/CreateTensor.py:24,56):  one candidate per line,
//     "<ctg> <pos> <refSeq33> " + 528 numbers, value index = 16*offset + 4*base + matrix.
//     The reference tokenises each row in CPython; this is a single pass over the bytes.
//   * cv_blosc_decompress: c-blosc 1.x chunk decoder (LZ4 / LZ4HC streams, byte shuffle,
//     split blocks, memcpy'd chunks) for the 500-item blocks of the .bin training file
//     (utils_v2.py:159-207, tensor2Bin.py:24-28).  c-blosc is a third-party dependency of
//     the reference (python-blosc, requirements.txt:3, unpinned, not vendored); the decoder
//     follows its published chunk format (16-byte header, bstarts table, per-split
//     length-prefixed streams) and the LZ4 block format.
//   * cv_blosc_compress_lz4: writer of the same container (greedy LZ4, byte shuffle).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "../../include/clairvoyante_amd.h"

void cv_set_error(const char *fmt, ...);

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

// Fast path for the "%0.1f"-style decimals CreateTensor writes; anything else goes to strtod.
inline bool parse_number(const char *&p, const char *end, float &out)
{
    const char *s = p;
    bool neg = false;
    if (s < end && (*s == '-' || *s == '+')) { neg = *s == '-'; s++; }
    if (s >= end || ((*s < '0' || *s > '9') && *s != '.')) return false;
    uint64_t ip = 0; int nd = 0;
    while (s < end && *s >= '0' && *s <= '9' && nd < 15) { ip = ip * 10 + (uint64_t)(*s - '0'); s++; nd++; }
    uint64_t fp = 0; int fd = 0;
    bool simple = true;
    if (s < end && *s == '.') {
        s++;
        while (s < end && *s >= '0' && *s <= '9' && fd < 6) { fp = fp * 10 + (uint64_t)(*s - '0'); s++; fd++; }
    }
    if (s < end && !is_space(*s) && *s != '\n') simple = false;   // exponent, long mantissa, inf/nan ...
    if (nd == 0 && fd == 0) simple = false;
    if (!simple) {
        char tmp[64]; size_t n = 0;
        const char *q = p;
        while (q < end && !is_space(*q) && *q != '\n' && n + 1 < sizeof(tmp)) tmp[n++] = *q++;
        tmp[n] = 0;
        char *ep = nullptr;
        double v = strtod(tmp, &ep);
        if (ep == tmp || *ep != 0) return false;
        out = (float)v; p = q;
        return true;
    }
    static const double p10[7] = {1, 10, 100, 1000, 10000, 100000, 1000000};
    double v = (double)ip;
    if (fd) v += (double)fp / p10[fd];      // exact for the one-decimal values of the format
    out = (float)(neg ? -v : v);
    p = s;
    return true;
}

}  // namespace

// Parses the complete lines of [p, end): for every accepted row (centre base of the upper-cased refSeq in ACGT,
// utils_v2.py:38-40) 528 floats to x_out (matrices 1..3 minus matrix 0, utils_v2.py:45-46) and 6 int64 to
// meta_out: byte offsets / lengths of ctg, pos, seq relative to `buf`.  Stops after max_rows rows.
static const char *parse_lines(const char *buf, const char *p, const char *end, int64_t max_rows, float *x_out,
                               int64_t *meta_out, int64_t *nrows, int64_t *nbad)
{
    const int NV = CV_INPUT_H * CV_INPUT_W * CV_INPUT_C;
    int64_t rows = 0, bad = 0;
    while (rows < max_rows) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!nl) break;
        const char *q = p;
        p = nl + 1;
        // three string tokens
        const char *tok[3]; int64_t tl[3]; int nt = 0;
        while (nt < 3) {
            while (q < nl && is_space(*q)) q++;
            if (q >= nl) break;
            tok[nt] = q;
            while (q < nl && !is_space(*q)) q++;
            tl[nt] = q - tok[nt];
            nt++;
        }
        if (nt == 0) continue;                       // blank line
        float *xr = x_out + (size_t)rows * NV;
        bool ok = nt == 3;
        int nv = 0;
        while (ok) {
            while (q < nl && is_space(*q)) q++;
            if (q >= nl) break;
            if (nv >= NV) { ok = false; break; }
            if (!parse_number(q, nl, xr[nv])) { ok = false; break; }
            nv++;
        }
        if (!ok || nv != NV) { bad++; continue; }     // the reference prints "UnpackATensorRecord Failure"
        if (tl[2] <= CV_INPUT_H / 2) continue;
        char c = tok[2][CV_INPUT_H / 2];
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') continue;
        for (int e = 0; e < NV; e += 4) { float m0 = xr[e]; xr[e + 1] -= m0; xr[e + 2] -= m0; xr[e + 3] -= m0; }
        int64_t *mr = meta_out + rows * 6;
        for (int k = 0; k < 3; k++) { mr[2 * k] = tok[k] - buf; mr[2 * k + 1] = tl[k]; }
        rows++;
    }
    *nrows = rows;
    *nbad = bad;
    return p;
}

static int g_host_threads = 1;

// ---- host thread pool -------------------------------------------------------------------------------------------
// The data plane is called once per batch (20 blosc blocks of ~1 MB, a few text slices): spawning its threads per
// call cost as much as the work of a block, and a static split of 20 blocks over 16 threads leaves 12 of them idle
// for half of the call.  Workers are created once (detached, at most 63), a job is a task count + a function; tasks
// are handed out through an atomic counter, the calling thread works too.  One job at a time: a second caller (the
// trainer decompresses X and Y of the next batch from two producer threads) runs its tasks on threads of its own.
namespace {
struct HostPool {
    std::mutex job_mu;                          // owner of the pool for the duration of a job
    std::mutex mu;
    std::condition_variable cv_start, cv_done;
    int nworkers = 0;
    pid_t pid = 0;                              // workers do not survive a fork
    const std::function<void(int64_t)> *fn = nullptr;
    std::atomic<int64_t> next{0};
    int64_t ntasks = 0;
    int active = 0, want = 0;
    uint64_t gen = 0;

    void drain()
    {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= ntasks) break;
            (*fn)(i);
        }
    }
    void worker(int id, uint64_t seen)
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_start.wait(lk, [&] { return gen != seen; });
            seen = gen;
            const bool part = id < want;
            lk.unlock();
            if (part) drain();
            lk.lock();
            if (--active == 0) cv_done.notify_one();
        }
    }
    void run(int64_t n, int T, const std::function<void(int64_t)> &f)
    {
        if (T > n) T = (int)n;
        if (T <= 1) { for (int64_t i = 0; i < n; i++) f(i); return; }
        std::unique_lock<std::mutex> job(job_mu, std::try_to_lock);
        if (!job.owns_lock()) {                 // pool busy: threads of this call, same dynamic hand-out
            std::atomic<int64_t> nx{0};
            auto body = [&]() { for (;;) { const int64_t i = nx.fetch_add(1); if (i >= n) break; f(i); } };
            std::vector<std::thread> th;
            for (int t = 1; t < T; t++) th.emplace_back(body);
            body();
            for (auto &x : th) x.join();
            return;
        }
        if (pid != getpid()) { pid = getpid(); nworkers = 0; gen = 0; }
        {
            std::lock_guard<std::mutex> lk(mu);
            while (nworkers < T - 1 && nworkers < 63) {
                std::thread(&HostPool::worker, this, nworkers, gen).detach();
                nworkers++;
            }
            fn = &f; ntasks = n; next.store(0); want = T - 1; active = nworkers; gen++;
        }
        cv_start.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });
        fn = nullptr;
    }
};
HostPool &host_pool() { static HostPool *p = new HostPool(); return *p; }   // never destroyed: workers are detached
}  // namespace

extern "C" int cv_set_host_threads(int n)
{
    g_host_threads = n < 1 ? 1 : n > 64 ? 64 : n;
    return 0;
}

// Parses complete lines from buf[0..len): see parse_lines.  Stops after max_rows rows or at the last complete
// line.  *consumed = bytes eaten (whole lines only), *nrows = accepted rows, *nbad = malformed rows.
// With cv_set_host_threads(T > 1) the first max_rows lines are cut into T slices parsed concurrently, each into
// the row range its line count reserves; ranges are closed up afterwards if a slice dropped rows.  The result is
// the single-threaded one.
extern "C" int cv_parse_tensor_text(const char *buf, int64_t len, int64_t max_rows, float *x_out,
                                    int64_t *meta_out, int64_t *consumed, int64_t *nrows, int64_t *nbad)
{
    if (!buf || !x_out || !meta_out || !consumed || !nrows) { cv_set_error("cv_parse_tensor_text: null argument"); return 1; }
    const int NV = CV_INPUT_H * CV_INPUT_W * CV_INPUT_C;
    const char *end = buf + len;
    int T = g_host_threads;
    if (T > 1 && (len < (1 << 20) || max_rows < 4 * T)) T = 1;
    if (T == 1) {
        int64_t r = 0, bd = 0;
        const char *p = parse_lines(buf, buf, end, max_rows, x_out, meta_out, &r, &bd);
        *consumed = p - buf; *nrows = r;
        if (nbad) *nbad = bd;
        return 0;
    }
    // the byte range of the first max_rows lines, cut into T slices of equal line counts
    std::vector<const char *> cut;
    std::vector<int64_t> first_line;
    {
        std::vector<const char *> starts;
        starts.reserve((size_t)(max_rows < (1 << 20) ? max_rows + 1 : (1 << 20)));
        const char *p = buf;
        int64_t lines = 0;
        while (lines < max_rows) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!nl) break;
            starts.push_back(p);
            p = nl + 1;
            lines++;
        }
        starts.push_back(p);                                  // end of the last whole line taken
        if (lines < 4 * T) T = 1;
        for (int t = 0; t <= T; t++) { first_line.push_back(lines * t / T); cut.push_back(starts[(size_t)(lines * t / T)]); }
    }
    std::vector<int64_t> got((size_t)T, 0), bads((size_t)T, 0);
    host_pool().run(T, T, [&](int64_t t) {
        parse_lines(buf, cut[(size_t)t], cut[(size_t)t + 1], first_line[(size_t)t + 1] - first_line[(size_t)t],
                    x_out + (size_t)first_line[(size_t)t] * NV, meta_out + first_line[(size_t)t] * 6, &got[(size_t)t],
                    &bads[(size_t)t]);
    });
    int64_t rows = 0, bd = 0;
    for (int t = 0; t < T; t++) {                              // close the gaps left by dropped rows
        if (rows != first_line[(size_t)t] && got[(size_t)t]) {
            memmove(x_out + (size_t)rows * NV, x_out + (size_t)first_line[(size_t)t] * NV, (size_t)got[(size_t)t] * NV * sizeof(float));
            memmove(meta_out + rows * 6, meta_out + first_line[(size_t)t] * 6, (size_t)got[(size_t)t] * 6 * sizeof(int64_t));
        }
        rows += got[(size_t)t];
        bd += bads[(size_t)t];
    }
    *consumed = cut[(size_t)T] - buf;
    *nrows = rows;
    if (nbad) *nbad = bd;
    return 0;
}

// ---- c-blosc 1.x container ------------------------------------------------------------
namespace {

int lz4_decompress(const uint8_t *src, int srclen, uint8_t *dst, int dstcap)
{
    const uint8_t *ip = src, *iend = src + srclen;
    uint8_t *op = dst, *oend = dst + dstcap;
    while (ip < iend) {
        unsigned token = *ip++;
        size_t lit = token >> 4;
        if (lit == 15) { unsigned b; do { if (ip >= iend) return -1; b = *ip++; lit += b; } while (b == 255); }
        if ((size_t)(iend - ip) < lit || (size_t)(oend - op) < lit) return -1;
        if (lit <= 16 && (size_t)(iend - ip) >= 16 && (size_t)(oend - op) >= 16) {
            memcpy(op, ip, 8); memcpy(op + 8, ip + 8, 8);           // fixed-size copies inline to two moves
        } else {
            memcpy(op, ip, lit);
        }
        op += lit; ip += lit;
        if (ip >= iend) break;                       // last sequence has no match
        if (iend - ip < 2) return -1;
        size_t off = ip[0] | ((size_t)ip[1] << 8); ip += 2;
        if (off == 0 || (size_t)(op - dst) < off) return -1;
        size_t ml = token & 15;
        if (ml == 15) { unsigned b; do { if (ip >= iend) return -1; b = *ip++; ml += b; } while (b == 255); }
        ml += 4;
        if ((size_t)(oend - op) < ml) return -1;
        const uint8_t *m = op - off;
        if (off >= 16 && (size_t)(oend - op) >= ml + 16) {          // 16 bytes at a time (may write <= 15 bytes past ml,
            for (size_t i = 0; i < ml; i += 16) memcpy(op + i, m + i, 16); // still inside the block: overwritten next)
        } else if (off >= 8 && (size_t)(oend - op) >= ml + 8) {
            for (size_t i = 0; i < ml; i += 8) memcpy(op + i, m + i, 8);
        } else if (off == 1) {
            memset(op, m[0], ml);                                    // a run of one byte (the usual case in byte planes)
        } else if (ml >= 32 && (size_t)(oend - op) >= ml + 8) {     // short period: 16 bytes one by one, then 8 at a time
            size_t i = 0;                                            // from a distance that is a multiple of the period
            for (; i < 16; i++) op[i] = m[i];
            const size_t o2 = off * ((7 + off) / off);               // 8 <= o2 <= 14
            for (; i < ml; i += 8) memcpy(op + i, op + i - o2, 8);
        } else {
            for (size_t i = 0; i < ml; i++) op[i] = m[i];           // overlapping copies are the point
        }
        op += ml;
    }
    return (int)(op - dst);
}

// greedy single-pass LZ4 block compressor (hash of 4 bytes); returns size or -1 if dst too small
int lz4_compress(const uint8_t *src, int n, uint8_t *dst, int cap)
{
    const int HB = 13;
    int table[1 << HB];
    for (int i = 0; i < (1 << HB); i++) table[i] = -1;
    int ip = 0, anchor = 0, op = 0;
    auto emit = [&](int litlen, int mlen, int off) -> bool {
        int need = 1 + litlen + litlen / 255 + 1 + (mlen >= 0 ? 2 + (mlen - 4) / 255 + 1 : 0);
        if (op + need > cap) return false;
        int tok = op++;
        int l = litlen;
        if (l >= 15) { dst[tok] = 15 << 4; l -= 15; while (l >= 255) { dst[op++] = 255; l -= 255; } dst[op++] = (uint8_t)l; }
        else dst[tok] = (uint8_t)(l << 4);
        memcpy(dst + op, src + anchor, (size_t)litlen); op += litlen;
        if (mlen >= 0) {
            dst[op++] = (uint8_t)(off & 255); dst[op++] = (uint8_t)(off >> 8);
            int m = mlen - 4;
            if (m >= 15) { dst[tok] |= 15; m -= 15; while (m >= 255) { dst[op++] = 255; m -= 255; } dst[op++] = (uint8_t)m; }
            else dst[tok] |= (uint8_t)m;
        }
        return true;
    };
    const int mflimit = n - 12;      // LZ4 end-of-block rules: last 5 bytes literals, last match >= 12 from end
    while (ip < mflimit) {
        uint32_t v; memcpy(&v, src + ip, 4);
        uint32_t h = (v * 2654435761u) >> (32 - HB);
        int cand = table[h];
        table[h] = ip;
        uint32_t cv = 0;
        if (cand >= 0) memcpy(&cv, src + cand, 4);
        if (cand >= 0 && ip - cand < 65536 && cv == v) {
            int ml = 4;
            const int maxml = n - 5 - ip;
            while (ml < maxml && src[cand + ml] == src[ip + ml]) ml++;
            if (!emit(ip - anchor, ml, ip - cand)) return -1;
            ip += ml; anchor = ip;
        } else ip++;
    }
    if (!emit(n - anchor, -1, 0)) return -1;
    return op;
}

void shuffle_bytes(const uint8_t *src, uint8_t *dst, int n, int ts)
{
    int ne = n / ts;
    for (int j = 0; j < ts; j++)
        for (int i = 0; i < ne; i++) dst[j * ne + i] = src[i * ts + j];
    memcpy(dst + (size_t)ne * ts, src + (size_t)ne * ts, (size_t)(n - ne * ts));
}

void unshuffle_bytes(const uint8_t *src, uint8_t *dst, int n, int ts)
{
    int ne = n / ts;
    int i = 0;
#if defined(__SSE2__)
    if (ts == 4) {          // fp32 blocks: 16 elements per step, byte planes interleaved with two unpack levels
        const uint8_t *p0 = src, *p1 = src + ne, *p2 = src + 2 * (size_t)ne, *p3 = src + 3 * (size_t)ne;
        for (; i + 16 <= ne; i += 16) {
            const __m128i a = _mm_loadu_si128((const __m128i *)(p0 + i)), b = _mm_loadu_si128((const __m128i *)(p1 + i));
            const __m128i c = _mm_loadu_si128((const __m128i *)(p2 + i)), d = _mm_loadu_si128((const __m128i *)(p3 + i));
            const __m128i ab0 = _mm_unpacklo_epi8(a, b), ab1 = _mm_unpackhi_epi8(a, b);
            const __m128i cd0 = _mm_unpacklo_epi8(c, d), cd1 = _mm_unpackhi_epi8(c, d);
            _mm_storeu_si128((__m128i *)(dst + (size_t)i * 4), _mm_unpacklo_epi16(ab0, cd0));
            _mm_storeu_si128((__m128i *)(dst + (size_t)i * 4 + 16), _mm_unpackhi_epi16(ab0, cd0));
            _mm_storeu_si128((__m128i *)(dst + (size_t)i * 4 + 32), _mm_unpacklo_epi16(ab1, cd1));
            _mm_storeu_si128((__m128i *)(dst + (size_t)i * 4 + 48), _mm_unpackhi_epi16(ab1, cd1));
        }
    }
#endif
    for (; i < ne; i++)
        for (int j = 0; j < ts; j++) dst[(size_t)i * ts + j] = src[(size_t)j * ne + i];
    memcpy(dst + (size_t)ne * ts, src + (size_t)ne * ts, (size_t)(n - ne * ts));
}

inline int32_t rd32(const uint8_t *p) { return (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)); }
inline void wr32(uint8_t *p, int32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

}  // namespace

// uncompressed size recorded in a blosc chunk header (or -1)
extern "C" int64_t cv_blosc_nbytes(const uint8_t *chunk, int64_t clen)
{
    if (!chunk || clen < 16) return -1;
    return (int64_t)(uint32_t)rd32(chunk + 4);
}

extern "C" int cv_blosc_decompress(const uint8_t *chunk, int64_t clen, uint8_t *dst, int64_t dstcap)
{
    if (!chunk || !dst || clen < 16) { cv_set_error("blosc: truncated chunk"); return 1; }
    const int flags = chunk[2], typesize = chunk[3] ? chunk[3] : 1;
    const int32_t nbytes = rd32(chunk + 4), blocksize = rd32(chunk + 8), cbytes = rd32(chunk + 12);
    if (chunk[0] != 2) { cv_set_error("blosc: unsupported format version %d", chunk[0]); return 1; }
    if (nbytes < 0 || dstcap < nbytes || cbytes > clen || blocksize <= 0) { cv_set_error("blosc: bad header"); return 1; }
    if (nbytes == 0) return 0;
    if (flags & 0x2) {                                  // memcpy'd
        if (clen < 16 + (int64_t)nbytes) { cv_set_error("blosc: truncated memcpy chunk"); return 1; }
        memcpy(dst, chunk + 16, (size_t)nbytes);
        return 0;
    }
    if (flags & 0x4) { cv_set_error("blosc: bit-shuffle is not supported"); return 1; }
    const int codec = (flags & 0xe0) >> 5;
    if (codec != 1) { cv_set_error("blosc: compressor format %d not supported (the .bin files use lz4hc)", codec); return 1; }
    const bool doshuffle = (flags & 0x1) && typesize > 1;
    const bool dont_split = (flags & 0x10) != 0;
    const int nblocks = (nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * (int64_t)nblocks > clen) { cv_set_error("blosc: truncated bstarts"); return 1; }
    uint8_t *tmp = doshuffle ? (uint8_t *)malloc((size_t)blocksize) : nullptr;
    int rc = 0;
    for (int b = 0; b < nblocks && !rc; b++) {
        int bsize = blocksize;
        bool leftover = false;
        if (b == nblocks - 1 && nbytes % blocksize) { bsize = nbytes % blocksize; leftover = true; }
        int nsplits = 1;
        if (!dont_split && typesize <= 16 && blocksize / typesize >= 128 && !leftover) nsplits = typesize;
        const int neblock = bsize / nsplits;
        int64_t ip = rd32(chunk + 16 + 4 * b);
        uint8_t *out = doshuffle ? tmp : dst + (size_t)b * blocksize;
        for (int s = 0; s < nsplits; s++) {
            if (ip + 4 > clen) { rc = 1; break; }
            int32_t cb = rd32(chunk + ip); ip += 4;
            if (cb < 0 || ip + cb > clen) { rc = 1; break; }
            if (cb == neblock) memcpy(out + (size_t)s * neblock, chunk + ip, (size_t)cb);
            else if (lz4_decompress(chunk + ip, cb, out + (size_t)s * neblock, neblock) != neblock) { rc = 1; break; }
            ip += cb;
        }
        if (!rc && doshuffle) unshuffle_bytes(tmp, dst + (size_t)b * blocksize, bsize, typesize);
    }
    free(tmp);
    if (rc) cv_set_error("blosc: corrupt LZ4 stream");
    return rc;
}

// Several chunks at once, on the host threads of cv_set_host_threads (DecompressArray unpacks the 20 blocks of a
// 10 000-item batch per call, utils_v2.py:196-203).  status[i] = 0 / 1 per chunk.
extern "C" int cv_blosc_decompress_many(const uint8_t *const *chunks, const int64_t *clens, uint8_t *const *dsts,
                                        const int64_t *dstcaps, int64_t n, int32_t *status)
{
    if ((!chunks || !clens || !dsts || !dstcaps || !status) && n > 0) { cv_set_error("blosc: null argument"); return 1; }
    host_pool().run(n, g_host_threads, [=](int64_t i) { status[i] = cv_blosc_decompress(chunks[i], clens[i], dsts[i], dstcaps[i]); });
    for (int64_t i = 0; i < n; i++)
        if (status[i]) { cv_set_error("blosc: chunk %lld is corrupt or unsupported", (long long)i); return 1; }
    return 0;
}

// The raw data of ONE pickled ndarray inside a decompressed block: python-blosc's pack_array pickles the array with
// the highest protocol (utils_v2.py:167-181), so the stream is a short header, one bytes object holding the data,
// and a short trailer (protocol 5 puts dtype / shape after the data: < 100 bytes).  Finds that object by its opcode + length: BINBYTES 'B' (u32) / BINBYTES8 0x8e /
// BYTEARRAY8 0x96 (u64) of protocols 3-5, BINSTRING 'T' (i32) of Python 2's protocol 2.  Returns the offset of the
// data and its length, or false.
static bool find_array_payload(const uint8_t *st, int64_t n, int64_t *off, int64_t *len)
{
    const int64_t head = n < 1024 ? n : 1024;
    for (int64_t i = 0; i + 9 < head; i++) {
        int64_t L = -1, h = 0;
        if (st[i] == 'B' || st[i] == 'T') { L = (int64_t)(uint32_t)rd32(st + i + 1); h = 5; }
        else if (st[i] == 0x8e || st[i] == 0x96) {
            uint64_t v = 0;
            for (int k = 7; k >= 0; k--) v = (v << 8) | st[i + 1 + k];
            if (v < (1ull << 40)) L = (int64_t)v;
            h = 9;
        }
        if (L < 0) continue;
        const int64_t endp = i + h + L;
        if (endp <= n && n - endp < 256 && L >= 16) { *off = i + h; *len = L; return true; }
    }
    return false;
}

// Blocks of one DecompressArray call straight into ONE destination array: chunk i is decompressed (host threads),
// its ndarray payload located and copied to dst + i*block_bytes.  Every chunk but the last must hold exactly
// block_bytes of data; lens[i] receives the payload bytes of chunk i.  status[i]: 0 ok, 1 corrupt chunk, 2 payload not
// recognised / unexpected size (the caller falls back to un-pickling).  Returns 0 when every status is 0.
extern "C" int cv_blosc_unpack_blocks(const uint8_t *const *chunks, const int64_t *clens, int64_t n, uint8_t *dst,
                                      int64_t block_bytes, int64_t *lens, int32_t *status)
{
    if ((!chunks || !clens || !dst || !lens || !status) && n > 0) { cv_set_error("blosc: null argument"); return 1; }
    host_pool().run(n, g_host_threads, [=](int64_t i) {
        static thread_local std::vector<uint8_t> scratch;       // per worker, kept between calls
        const int64_t nb = cv_blosc_nbytes(chunks[i], clens[i]);
        status[i] = 1; lens[i] = 0;
        if (nb < 0) return;
        if ((int64_t)scratch.size() < nb + 16) scratch.resize((size_t)nb + 16);
        if (cv_blosc_decompress(chunks[i], clens[i], scratch.data(), nb)) return;
        int64_t off = 0, L = 0;
        status[i] = 2;
        if (!find_array_payload(scratch.data(), nb, &off, &L)) {
            // an empty trailing block pickles an array without a data object worth finding: accept a tiny stream
            if (i == n - 1 && nb < 512) { status[i] = 0; lens[i] = 0; }
            return;
        }
        if ((i < n - 1 && L != block_bytes) || L > block_bytes) return;
        memcpy(dst + (size_t)i * (size_t)block_bytes, scratch.data() + off, (size_t)L);
        lens[i] = L;
        status[i] = 0;
    });
    for (int64_t i = 0; i < n; i++)
        if (status[i]) { cv_set_error("blosc: block %lld: status %d", (long long)i, status[i]); return 1; }
    return 0;
}

// Writes one chunk (single block, no split, byte shuffle when typesize > 1, LZ4 stream,
// compressor tag lz4).  dstcap >= n + 32.  Returns the chunk size in *clen.
extern "C" int cv_blosc_compress_lz4(const uint8_t *src, int64_t n, int typesize, uint8_t *dst, int64_t dstcap,
                                     int64_t *clen)
{
    if (!src || !dst || !clen || n < 0 || n > 0x7fffff00) { cv_set_error("blosc: bad compress arguments"); return 1; }
    if (dstcap < n + 32) { cv_set_error("blosc: destination too small"); return 1; }
    if (typesize < 1 || typesize > 255) typesize = 1;
    const bool sh = typesize > 1 && n >= typesize;
    dst[0] = 2; dst[1] = 1; dst[3] = (uint8_t)typesize;
    wr32(dst + 4, (int32_t)n); wr32(dst + 8, (int32_t)(n > 0 ? n : 1));
    int flags = (1 << 5) | 0x10 | (sh ? 1 : 0);
    int64_t total = -1;
    if (n >= 64) {
        uint8_t *tmp = sh ? (uint8_t *)malloc((size_t)n) : nullptr;
        if (sh) shuffle_bytes(src, tmp, (int)n, typesize);
        wr32(dst + 16, 20);
        int c = lz4_compress(sh ? tmp : src, (int)n, dst + 24, (int)(n - 1));
        if (c > 0) { wr32(dst + 20, c); total = 24 + c; }
        free(tmp);
    }
    if (total < 0) {                                   // incompressible / tiny: memcpy'd chunk
        flags = (1 << 5) | 0x10 | 0x2;
        memcpy(dst + 16, src, (size_t)n);
        total = 16 + n;
    }
    dst[2] = (uint8_t)flags;
    wr32(dst + 12, (int32_t)total);
    *clen = total;
    return 0;
}

// ---- CRC32C (Castagnoli) for the TensorFlow checkpoint bundle -------------------------------
namespace {
uint32_t crc_table[8][256];
bool crc_init_done = false;
void crc_init()
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82f63b78u & (0u - (c & 1)));
        crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++) crc_table[t][i] = (crc_table[t - 1][i] >> 8) ^ crc_table[0][crc_table[t - 1][i] & 255];
    crc_init_done = true;
}
__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t crc, const uint8_t *p, int64_t n)
{
    uint64_t c = crc;
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); c = __builtin_ia32_crc32di(c, v); p += 8; n -= 8; }
    while (n-- > 0) c = __builtin_ia32_crc32qi((uint32_t)c, *p++);
    return (uint32_t)c;
}
uint32_t crc_sw(uint32_t c, const uint8_t *p, int64_t n)
{
    if (!crc_init_done) crc_init();
    while (n >= 8) {
        uint32_t lo, hi; memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = crc_table[7][lo & 255] ^ crc_table[6][(lo >> 8) & 255] ^ crc_table[5][(lo >> 16) & 255] ^ crc_table[4][lo >> 24] ^
            crc_table[3][hi & 255] ^ crc_table[2][(hi >> 8) & 255] ^ crc_table[1][(hi >> 16) & 255] ^ crc_table[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n-- > 0) c = (c >> 8) ^ crc_table[0][(c ^ *p++) & 255];
    return c;
}
}  // namespace

// CRC32C of data, continuing from `crc` (0 to start); unmasked.
extern "C" uint32_t cv_crc32c(uint32_t crc, const void *data, int64_t n)
{
    if (!data || n <= 0) return crc;
    const uint8_t *p = (const uint8_t *)data;
    uint32_t c = ~crc;
    c = __builtin_cpu_supports("sse4.2") ? crc_hw(c, p, n) : crc_sw(c, p, n);
    return ~c;
}
