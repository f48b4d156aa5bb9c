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
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include <unistd.h>
#include <pthread.h>
#include <stdio.h>
#include <string>
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

// The 528 numbers of a row in the producer's own format (CreateTensor.py:24,56: "%0.1f" tokens, one blank between
// them): [-]digits[.digit] each followed by ' ' or the line's '\n' (the newline is the sentinel -- no bounds checks).
// Returns false at the first token that is anything else (the caller then re-parses the row with parse_number, which
// takes every format strtod does); true with q at the newline when exactly nv_want numbers fill the line.  The value
// is formed exactly as parse_number forms it: (double)integer + digit / 10.0, rounded to float once.
inline bool parse_row_fast(const char *q, const char *nl, float *xr, int nv_want)
{
    static const double tenth[10] = {0.0 / 10.0, 1.0 / 10.0, 2.0 / 10.0, 3.0 / 10.0, 4.0 / 10.0,
                                     5.0 / 10.0, 6.0 / 10.0, 7.0 / 10.0, 8.0 / 10.0, 9.0 / 10.0};
    for (int nv = 0; nv < nv_want; nv++) {
        if (*q != ' ') return false;
        q++;
        const bool neg = *q == '-';
        q += neg;
        unsigned d = (unsigned)(*q - '0');
        if (d > 9) return false;
        uint32_t ip = d;
        const char *s = q + 1;
        while ((d = (unsigned)(*s - '0')) <= 9) { ip = ip * 10 + d; s++; }
        if (s - q > 9) return false;                      // would not fit 32 bits: the general path
        double v = (double)ip;
        if (*s == '.') {
            const unsigned f = (unsigned)(s[1] - '0');
            if (f > 9) return false;
            v += tenth[f];
            s += 2;
        }
        if (*s != ' ' && *s != '\n') return false;
        xr[nv] = (float)(neg ? -v : v);
        q = s;
    }
    return q == nl;
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
        if (ok && parse_row_fast(q, nl, xr, NV)) { nv = NV; q = nl; }
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
    std::condition_variable cv_done;
    // one condition variable per worker: a job wakes only the `want` workers that take part in it (waking all 63
    // for a 4-task job made every small call pay for 63 wake-ups and check-ins)
    struct Slot { std::condition_variable cv; uint64_t go = 0; };
    Slot slot[63];
    int nworkers = 0;
    const std::function<void(int64_t)> *fn = nullptr;
    std::atomic<int64_t> next{0};
    int64_t ntasks = 0;
    int active = 0;

    void drain()
    {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= ntasks) break;
            (*fn)(i);
        }
    }
    void worker(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            slot[id].cv.wait(lk, [&] { return slot[id].go != seen; });
            seen = slot[id].go;
            lk.unlock();
            drain();
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
        int part = T - 1 < 63 ? T - 1 : 63;
        {
            std::lock_guard<std::mutex> lk(mu);
            while (nworkers < part) {
                std::thread(&HostPool::worker, this, nworkers).detach();
                nworkers++;
            }
            fn = &f; ntasks = n; next.store(0); active = part;
            for (int w = 0; w < part; w++) slot[w].go++;
        }
        for (int w = 0; w < part; w++) slot[w].cv.notify_one();
        drain();
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });
        fn = nullptr;
    }
};
// The pool object is never destroyed (its workers are detached).  A forked child inherits the memory but none of the
// threads, and possibly a mutex locked by a thread that does not exist there: the child gets a fresh pool
// (pthread_atfork), the inherited one is abandoned.
std::atomic<HostPool *> g_pool{nullptr};
void pool_after_fork_child() { g_pool.store(new HostPool()); }
HostPool &host_pool()
{
    HostPool *p = g_pool.load();
    if (!p) {
        static std::once_flag once;
        std::call_once(once, [] { g_pool.store(new HostPool()); pthread_atfork(nullptr, nullptr, pool_after_fork_child); });
        p = g_pool.load();
    }
    return *p;
}
}  // namespace

extern "C" int cv_set_host_threads(int n)
{
    g_host_threads = n < 1 ? 1 : n > 64 ? 64 : n;
    return 0;
}

namespace {
inline int64_t count_newlines(const char *p, const char *e)
{
    int64_t n = 0;
    while (p < e) {
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        n++; p = q + 1;
    }
    return n;
}
// the byte after the k-th newline of [p, e) (k >= 1), or nullptr when there are fewer
inline const char *after_kth_newline(const char *p, const char *e, int64_t k)
{
    while (p < e) {
        const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!q) return nullptr;
        p = q + 1;
        if (--k == 0) return p;
    }
    return nullptr;
}
}  // namespace

// Parses complete lines from buf[0..len): see parse_lines.  Stops after max_rows LINES or at the last complete
// line.  *consumed = bytes eaten (whole lines only), *nrows = accepted rows, *nbad = malformed rows.
// With cv_set_host_threads(T > 1) nothing walks the text serially: the threads count the newlines of equal byte
// slices of a window sized from the first line (so a caller may hand over a whole memory-mapped file), the end of the
// max_rows-th line follows from the counts, the lines up to it are cut into T byte slices moved to line starts, every
// slice is parsed into the row range its line count reserves, and the ranges are closed up if a slice dropped rows.
// When the window holds fewer than max_rows lines the call returns what it holds (the caller asks again).  The rows
// are the single-threaded ones.
extern "C" int cv_parse_tensor_text(const char *buf, int64_t len, int64_t max_rows, float *x_out,
                                    int64_t *meta_out, int64_t *consumed, int64_t *nrows, int64_t *nbad)
{
    if (!buf || !x_out || !meta_out || !consumed || !nrows) { cv_set_error("cv_parse_tensor_text: null argument"); return 1; }
    const int NV = CV_INPUT_H * CV_INPUT_W * CV_INPUT_C;
    const char *end = buf + len;
    int T = g_host_threads;
    if (T > 1 && (len < (1 << 20) || max_rows < 4 * T)) T = 1;
    const char *nl0 = T > 1 ? (const char *)memchr(buf, '\n', (size_t)len) : nullptr;
    if (T > 1 && !nl0) T = 1;
    const char *lim = nullptr;          // end of the last line taken
    int64_t lines = 0;
    if (T > 1) {
        // window: 1.25 x (first line + its newline) x max_rows, whole buffer if that is shorter
        const int64_t first = nl0 - buf + 1;
        int64_t W = len;
        if (max_rows < len / first) { W = first * max_rows; W += W / 4 + 4096; if (W > len) W = len; }
        const char *wend = buf + W;
        std::vector<int64_t> cnt((size_t)T, 0);
        auto piece = [&](int t) { return buf + (int64_t)((__int128)W * t / T); };
        host_pool().run(T, T, [&](int64_t t) { cnt[(size_t)t] = count_newlines(piece((int)t), piece((int)t + 1)); });
        int64_t total = 0;
        for (int t = 0; t < T; t++) total += cnt[(size_t)t];
        if (total >= max_rows) {
            int64_t before = 0; int t = 0;
            while (before + cnt[(size_t)t] < max_rows) before += cnt[(size_t)t++];
            lim = after_kth_newline(piece(t), piece(t + 1), max_rows - before);
            lines = max_rows;
        } else if (total > 0) {
            const char *q = wend;
            while (q > buf && q[-1] != '\n') q--;             // (at most one line back)
            lim = q; lines = total;
        } else {
            lim = buf;
        }
        (void)wend;
        if (lines < 4 * T) T = 1;
    }
    if (T == 1) {
        int64_t r = 0, bd = 0;
        const char *stop = lim ? lim : end;
        const char *p = parse_lines(buf, buf, stop, max_rows, x_out, meta_out, &r, &bd);
        *consumed = p - buf; *nrows = r;
        if (nbad) *nbad = bd;
        return 0;
    }
    // [buf, lim) = `lines` whole lines: T byte slices, each moved forward to the next line start
    std::vector<const char *> cut((size_t)T + 1);
    const int64_t span = lim - buf;
    cut[0] = buf; cut[(size_t)T] = lim;
    for (int t = 1; t < T; t++) {
        const char *c = buf + (int64_t)((__int128)span * t / T);
        if (c < cut[(size_t)t - 1]) c = cut[(size_t)t - 1];
        const char *q = c > buf ? (const char *)memchr(c - 1, '\n', (size_t)(lim - (c - 1))) : nullptr;
        cut[(size_t)t] = c == buf ? buf : (q ? q + 1 : lim);
    }
    std::vector<int64_t> nline((size_t)T, 0), first_line((size_t)T + 1, 0);
    host_pool().run(T, T, [&](int64_t t) { nline[(size_t)t] = count_newlines(cut[(size_t)t], cut[(size_t)t + 1]); });
    for (int t = 0; t < T; t++) first_line[(size_t)t + 1] = first_line[(size_t)t] + nline[(size_t)t];
    if (first_line[(size_t)T] != lines) { cv_set_error("cv_parse_tensor_text: line count mismatch (internal)"); return 1; }
    std::vector<int64_t> got((size_t)T, 0), bads((size_t)T, 0);
    host_pool().run(T, T, [&](int64_t t) {
        parse_lines(buf, cut[(size_t)t], cut[(size_t)t + 1], nline[(size_t)t],
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
    *consumed = lim - buf;
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
    const int64_t nblocks64 = ((int64_t)nbytes + blocksize - 1) / blocksize;
    if (16 + 4 * nblocks64 > clen) { cv_set_error("blosc: truncated bstarts"); return 1; }
    const int nblocks = (int)nblocks64;
    const int64_t data0 = 16 + 4 * nblocks64;           // the first byte a block may start at
    uint8_t *tmp = doshuffle ? (uint8_t *)malloc((size_t)(blocksize < nbytes ? blocksize : nbytes)) : nullptr;
    if (doshuffle && !tmp) { cv_set_error("blosc: out of memory"); return 1; }
    int rc = 0;
    for (int b = 0; b < nblocks && !rc; b++) {
        int bsize = blocksize;
        bool leftover = false;
        if (b == nblocks - 1 && nbytes % blocksize) { bsize = nbytes % blocksize; leftover = true; }
        int nsplits = 1;
        if (!dont_split && typesize <= 16 && blocksize / typesize >= 128 && !leftover) nsplits = typesize;
        const int neblock = bsize / nsplits;
        int64_t ip = rd32(chunk + 16 + 4 * b);
        if (ip < data0) { rc = 1; break; }             // (a corrupt start offset: negative, or inside the header)
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

// ---- VCF records (host half of callVar.Output) -------------------------------------------------------------------
// One record per candidate from the decisions the device made (cv_call_postproc): allele / indel-length inference
// and the text of /root/reference/clairvoyante/callVar.py:72-153, reproduced field for field:
//   qual  = int(-4.343 * log((p2 + 1e-300) / (p1 + 1e-300)))   -- fp32 products widened to double, truncation (:72)
//   SNP / REF allele (:91-95), insertion bases = arg-max of matrix 1 per position, first maximum (:101,:107),
//   length guess while sum(matrix) >= 0.125 * sum(matrix 0) (:104-110, :122-127), <INS> / <DEL> + SVTYPE at >= 16
//   inferred bases (:111-113, :129-131), LENGUESS (:139-140), GT / FILTER / "%.4f" allele fraction (:143-153).
// Records are independent: the host threads format contiguous ranges into private buffers that are joined in order.
namespace {

struct vcf_job {
    const int32_t *call; const float *qual; const float *x; const int64_t *xrow;
    const char *pos_buf; const int64_t *pos_meta; const int64_t *pos_row;
    int show_ref, has_qual, qual_min;
};

inline float sum4(const float *p, int stride) { return ((p[0] + p[stride]) + p[2 * stride]) + p[3 * stride]; }
inline int argmax4(const float *p, int stride)
{
    int b = 0; float m = p[0];
    for (int k = 1; k < 4; k++) if (p[k * stride] > m) { m = p[k * stride]; b = k; }      // first maximum (np.argmax)
    return b;
}

inline char *put_str(char *p, const char *s) { while (*s) *p++ = *s++; return p; }
inline char *put_int(char *p, long long v)
{
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    if (v < 0) *p++ = '-';
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}
// "%.4f" of a float: the value times 10^4 is exact in double (24 + 14 bits), so nearbyint() -- round to nearest,
// ties to even, the default mode -- gives the correctly rounded digits printf derives from the exact expansion
inline char *put_f4(char *p, float v)
{
    const double a = fabs((double)v);
    if (!(a < 1e9)) return p + snprintf(p, 48, "%.4f", (double)v);        // inf, nan, absurd magnitudes: the library's way
    const unsigned long long r = (unsigned long long)nearbyint(a * 10000.0);
    if (signbit(v)) *p++ = '-';
    p = put_int(p, (long long)(r / 10000));
    unsigned f = (unsigned)(r % 10000);
    *p++ = '.';
    p[3] = (char)('0' + f % 10); f /= 10; p[2] = (char)('0' + f % 10); f /= 10; p[1] = (char)('0' + f % 10); f /= 10; p[0] = (char)('0' + f);
    return p + 4;
}

// appends one record; returns 0 ok / skipped, 1 error (message set)
int vcf_record(const vcf_job &J, int64_t i, std::string &out, int64_t *nrec)
{
    const int F = CV_INPUT_H / 2;                      // flankingBaseNum = 16
    const int32_t *c = J.call + i * 8;
    int varType = c[0];
    // (the decisions are cv_call_postproc's argmax indices; a caller's own array is checked before it sizes a copy)
    if ((unsigned)varType > 3u || (unsigned)c[1] > 1u || (unsigned)c[2] > 5u) {
        cv_set_error("cv_format_vcf: record %lld: decision (%d, %d, %d) outside varType 0..3 / zygosity 0..1 / length 0..5",
                     (long long)i, c[0], c[1], c[2]);
        return 1;
    }
    if (varType == 0 && !J.show_ref) return 0;
    const float *q = J.qual + i * 4;
    const float dp = q[2];
    if (dp == 0.0f) return 0;
    if (!(fabsf(dp) < 2.0e9f)) { cv_set_error("cv_format_vcf: record %lld: depth is not a finite count", (long long)i); return 1; }
    const int64_t *pm = J.pos_meta + (J.pos_row ? J.pos_row[i] : i) * 6;
    const char *chrom = J.pos_buf + pm[0]; const int64_t chrom_len = pm[1];
    const char *cs = J.pos_buf + pm[2]; int64_t cl = pm[3];
    const char *seq = J.pos_buf + pm[4]; const int64_t seq_len = pm[5];
    // int(coordination): optional surrounding blanks and sign, decimal digits
    while (cl > 0 && (*cs == ' ' || *cs == '\t')) { cs++; cl--; }
    while (cl > 0 && (cs[cl - 1] == ' ' || cs[cl - 1] == '\t' || cs[cl - 1] == '\r')) cl--;
    bool neg = false;
    if (cl > 0 && (*cs == '+' || *cs == '-')) { neg = *cs == '-'; cs++; cl--; }
    if (cl <= 0 || cl > 18) { cv_set_error("cv_format_vcf: record %lld: position is not an integer", (long long)i); return 1; }
    long long coord = 0;
    for (int64_t k = 0; k < cl; k++) {
        if (cs[k] < '0' || cs[k] > '9') { cv_set_error("cv_format_vcf: record %lld: position is not an integer", (long long)i); return 1; }
        coord = coord * 10 + (cs[k] - '0');
    }
    if (neg) coord = -coord;
    if (seq_len <= F) { cv_set_error("cv_format_vcf: record %lld: reference sequence shorter than %d", (long long)i, F + 1); return 1; }
    char sq[CV_INPUT_H + 1];
    const int sl = seq_len < CV_INPUT_H ? (int)seq_len : CV_INPUT_H;
    for (int k = 0; k < sl; k++) { char ch = seq[k]; sq[k] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 32) : ch; }
    const float *x = J.x + (size_t)(J.xrow ? J.xrow[i] : i) * (CV_INPUT_H * 16);       // [33][4 bases][4 matrices]
    static const char N2B[4] = {'A', 'C', 'G', 'T'};
    char ref[CV_INPUT_H + 1]; int ref_len = 1; ref[0] = sq[F];
    char alt[CV_INPUT_H + 8]; int alt_len = 0;
    const char *svtype = nullptr;
    int inferred = 0;
    int varLength = c[2];
    float af;
    if (varType == 0 || varType == 1) {
        char ab;
        if (varType == 0) ab = sq[F];
        else { const char b1 = N2B[c[3] & 3], b2 = N2B[c[4] & 3]; ab = b1 != sq[F] ? b1 : b2; }
        int bi = ab == 'A' ? 0 : ab == 'C' ? 1 : ab == 'G' ? 2 : ab == 'T' ? 3 : -1;
        if (bi < 0) { cv_set_error("cv_format_vcf: record %lld: centre base '%c' is not one of ACGT", (long long)i, ab); return 1; }
        af = x[(F * 4 + bi) * 4 + 3] / dp;
        alt[0] = ab; alt_len = 1;
    } else if (varType == 2) {
        if (varLength == 0) varLength = 1;
        af = sum4(x + (F + 1) * 16 + 1, 4) / dp;
        char ins[CV_INPUT_H]; int nins = 0;
        if (varLength != 5) {
            for (int k = F + 1; k < F + varLength + 1; k++) ins[nins++] = N2B[argmax4(x + k * 16 + 1, 4)];
        } else {
            for (int k = F + 1; k < 2 * F + 1; k++) {
                if (k < F + 5 || sum4(x + k * 16 + 1, 4) >= 0.125f * sum4(x + k * 16 + 0, 4)) {
                    inferred++;
                    ins[nins++] = N2B[argmax4(x + k * 16 + 1, 4)];
                } else break;
            }
        }
        if (inferred >= F) { memcpy(alt, "<INS>", 5); alt_len = 5; svtype = "SVTYPE=INS"; }
        else { alt[0] = sq[F]; memcpy(alt + 1, ins, (size_t)nins); alt_len = 1 + nins; }
    } else {
        if (varLength == 0) varLength = 1;
        af = sum4(x + (F + 1) * 16 + 2, 4) / dp;
        if (varLength == 5) {
            for (int k = F + 1; k < 2 * F + 1; k++) {
                if (k < F + 5 || sum4(x + k * 16 + 2, 4) >= 0.125f * sum4(x + k * 16 + 0, 4)) inferred++;
                else break;
            }
        }
        if (inferred >= F) { memcpy(alt, "<DEL>", 5); alt_len = 5; svtype = "SVTYPE=DEL"; }
        else {
            const int want = (varLength != 5 ? varLength : inferred) + 1;          // refSeq[F : F + want], clipped like a slice
            ref_len = F + want <= sl ? want : sl - F;
            memcpy(ref, sq + F, (size_t)ref_len);
            alt[0] = sq[F]; alt_len = 1;
        }
    }
    const double ratio = ((double)q[1] + 1e-300) / ((double)q[0] + 1e-300);
    const int qv = (int)(-4.343 * log(ratio));                                      // int(): truncation toward zero
    const char *gt = varType == 0 ? "0/0" : (c[1] == 0 ? "0/1" : "1/1");
    const char *filt = !J.has_qual ? "." : (qv >= J.qual_min ? "PASS" : "LowQual");
    // the line, assembled by hand (snprintf costs more than everything above): "%s\t%d\t.\t%s\t%s\t%d\t%s\t%s\tGT:GQ:DP:AF\t%s:%d:%d:%.4f"
    char line[256]; char *p = line;
    p = put_int(p, coord); *p++ = '\t'; *p++ = '.'; *p++ = '\t';
    memcpy(p, ref, (size_t)ref_len); p += ref_len; *p++ = '\t';
    memcpy(p, alt, (size_t)alt_len); p += alt_len; *p++ = '\t';
    p = put_int(p, qv); *p++ = '\t';
    p = put_str(p, filt); *p++ = '\t';
    if (svtype) p = put_str(p, svtype);
    if (inferred > 0 && inferred < F) { if (svtype) *p++ = ';'; p = put_str(p, "LENGUESS="); p = put_int(p, inferred); }
    else if (!svtype) *p++ = '.';
    p = put_str(p, "\tGT:GQ:DP:AF\t");
    p = put_str(p, gt); *p++ = ':';
    p = put_int(p, qv); *p++ = ':';
    p = put_int(p, (long long)(int)dp); *p++ = ':';
    p = put_f4(p, af);
    *p++ = '\n';
    out.append(chrom, (size_t)chrom_len);
    out.push_back('\t');
    out.append(line, (size_t)(p - line));
    (*nrec)++;
    return 0;
}

}  // namespace

extern "C" int cv_format_vcf(const int32_t *call, const float *qual, int64_t n, const float *x, const int64_t *xrow,
                             const char *pos_buf, const int64_t *pos_meta, const int64_t *pos_row, int show_ref,
                             int has_qual, int qual_min, char *out, int64_t out_cap, int64_t *out_len, int64_t *nrecords)
{
    if (n < 0 || !out_len) { cv_set_error("cv_format_vcf: bad argument"); return 1; }
    *out_len = 0;
    if (nrecords) *nrecords = 0;
    if (n == 0) return 0;
    if (!call || !qual || !x || !pos_buf || !pos_meta) { cv_set_error("cv_format_vcf: null argument"); return 1; }
    vcf_job J{call, qual, x, xrow, pos_buf, pos_meta, pos_row, show_ref, has_qual, qual_min};
    int T = g_host_threads;
    if (n < 2048) T = 1;
    if (T > 64) T = 64;
    // the scratch strings are kept between calls (a free list): after the first batches no call allocates or
    // page-faults
    static std::mutex scratch_mu;
    static std::vector<std::unique_ptr<std::string>> scratch_free;      // (owned: released at process exit)
    std::vector<std::string *> part((size_t)T, nullptr);
    {
        std::lock_guard<std::mutex> lk(scratch_mu);
        for (int t = 0; t < T; t++) {
            if (!scratch_free.empty()) { part[(size_t)t] = scratch_free.back().release(); scratch_free.pop_back(); }
            else part[(size_t)t] = new std::string();
            part[(size_t)t]->clear();
        }
    }
    struct give_back {
        std::vector<std::string *> &v; std::mutex &m; std::vector<std::unique_ptr<std::string>> &f;
        ~give_back() { std::lock_guard<std::mutex> lk(m); for (auto *x : v) f.emplace_back(x); }
    } gb{part, scratch_mu, scratch_free};
    std::vector<int64_t> cnt((size_t)T, 0);
    std::vector<int> bad((size_t)T, 0);
    std::vector<std::string> msg((size_t)T);
    auto body = [&](int64_t t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        std::string &o = *part[(size_t)t];
        for (int64_t i = lo; i < hi; i++)
            if (vcf_record(J, i, o, &cnt[(size_t)t])) { bad[(size_t)t] = 1; msg[(size_t)t] = cv_last_error(); return; }
    };
    if (T == 1) body(0); else host_pool().run(T, T, body);
    int64_t total = 0, recs = 0;
    for (int t = 0; t < T; t++) {
        if (bad[(size_t)t]) { cv_set_error("%s", msg[(size_t)t].c_str()); return 1; }     // the error of a worker thread is thread-local there
        total += (int64_t)part[(size_t)t]->size(); recs += cnt[(size_t)t];
    }
    *out_len = total;
    if (nrecords) *nrecords = recs;
    if (total > out_cap || !out) return total > 0 ? 2 : 0;        // 2: the caller's buffer is too small, *out_len = needed
    char *p = out;
    for (int t = 0; t < T; t++) { memcpy(p, part[(size_t)t]->data(), part[(size_t)t]->size()); p += part[(size_t)t]->size(); }
    return 0;
}
