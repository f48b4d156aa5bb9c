// Pileup front end: SAM text -> [n,33,4,4] count tensors in HBM (include/clairvoyante_amd.h,
// "pileup front end").  Follows the behaviour of /root/reference/dataPrepScripts/CreateTensor.py
// (OutputAlnTensor :93-246, GenerateTensor :23-54) with a different decomposition:
//
//   host   : one pass over the SAM text; every CIGAR run becomes alignment SEGMENTS of <= 64 columns
//            (20 bytes each) + the read's SEQ bytes.  No per-candidate buffering.
//   scatter: one wave per segment, one lane per alignment column.  Whether a column counts for a
//            candidate is a LOCAL rule (derived from the reference's activation state machine,
//            :175-229, and checked against it in tests/): with c the 1-based centre, r the 0-based
//            column position, p = r - c + 17 the window offset and POS the read's first position,
//              match  column: counts iff 0 <= p <= 32
//              delete column: counts iff 1 <= p <= 32 and r > POS  (the column that activates a read is
//                             appended BEFORE the activation, :216-226)
//              insert column: counts iff 1 <= p <= 32 and r > POS, at offset min(p + k, 32) (:41)
//            and without --considerleftedge additionally POS <= c - 17 (:63-64: a read is activated
//            only at the window start).  A candidate gets a row iff a match/delete column fell on
//            [c-17, c+16] (left-edge mode) or exactly on c-17 (otherwise).
//            Counters: 9 int32 per (candidate, offset): inserted A,C,G,T | deleted | matched query
//            A,C,G,T.  Matrix 0 / 2 rows are implied: all match columns at one position share the
//            reference base, so matrix0[ref] = sum of matched, matrix2[ref] = that + deleted.
//   final  : one thread per (candidate, offset): 9 counters -> 16 floats (64 B), optional
//            "matrices 1..3 minus matrix 0" (utils_v2.py:46), centre depth for --minCoverage.
//
// HBM-bound integer work: per column 1 SEQ byte + 1 reference byte + one L2 atomic per covering
// candidate; per candidate 1 188 B of counters read once and 2 112 B of tensor written once.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/clairvoyante_amd.h"

void cv_set_error(const char *fmt, ...);

#define PL_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                             \
        }                                                                                         \
    } while (0)

namespace {

constexpr int FLANK = 16;                 // dataPrepScripts/param.py:1
constexpr int WIDTH = 2 * FLANK + 1;      // 33
constexpr int NCNT = 9;                   // counters per (candidate, offset)
constexpr int SEG_MAX = 64;               // columns per segment = lanes per wave
constexpr int BUCKET_SHIFT = 4;           // candidate lookup table: one entry per 16 positions

enum { T_MATCH = 0, T_INS = 1, T_DEL = 2 };

struct seg_t {
    int32_t r0;      // 0-based reference position of the first column (insert: the position it precedes)
    uint32_t q0;     // offset of the first query base in the SEQ byte buffer (unused for deletes)
    int32_t info;    // columns (1..64) | type << 8
    int32_t adv0;    // insert: index of the first column inside its insertion (queryAdv, :203-211)
    int32_t pos;     // the read's POS (0-based)
};

__device__ __forceinline__ int base_code(uint8_t ch)
{
    // upper-case ACGT only (CreateTensor.py:28-31 tests `in "ACGT-"`)
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
}

__global__ void bucket_build(const int32_t *__restrict__ cands, int n, int32_t lo, int64_t nb,
                             int32_t *__restrict__ first)
{
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    int64_t target = (int64_t)lo + (b << BUCKET_SHIFT);
    int l = 0, h = n;
    while (l < h) {
        int m = (l + h) >> 1;
        if ((int64_t)cands[m] < target) l = m + 1; else h = m;
    }
    first[b] = l;
}

__global__ void __launch_bounds__(256)
pileup_scatter(const seg_t *__restrict__ segs, int64_t nseg, const uint8_t *__restrict__ seq,
               const uint8_t *__restrict__ ref, int64_t ref_first, int64_t ref_len,
               const int32_t *__restrict__ cands, int n, const int32_t *__restrict__ bucket_first,
               int32_t bucket_lo, int64_t nb, int32_t *__restrict__ cnt, uint8_t *__restrict__ touched, int left)
{
    int64_t s = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= nseg) return;
    const seg_t sg = segs[s];
    const int lane = threadIdx.x & 63;
    const int len = sg.info & 0xff;
    const int type = sg.info >> 8;
    if (lane >= len) return;
    const int32_t r = type == T_INS ? sg.r0 : sg.r0 + lane;
    const int q = type == T_DEL ? -1 : base_code(seq[(size_t)sg.q0 + lane]);
    int rb = -1;
    if (type != T_INS) {
        int64_t ri = (int64_t)r - ref_first;
        if (ri >= 0 && ri < ref_len) rb = base_code(ref[ri]);
    }
    // candidates c with r - 16 <= c <= r + 17
    const int64_t lo_c = (int64_t)r - FLANK, hi_c = (int64_t)r + FLANK + 1;
    int64_t b = (lo_c - bucket_lo) >> BUCKET_SHIFT;
    if (b < 0) b = 0;
    if (b >= nb) return;
    int i = bucket_first[b];
    while (i < n && (int64_t)cands[i] < lo_c) ++i;
    for (; i < n; ++i) {
        const int64_t c = cands[i];
        if (c > hi_c) break;
        const int p = (int)(r - c) + FLANK + 1;
        if (type != T_INS && (left || p == 0)) touched[i] = 1;
        if (!left && (int64_t)sg.pos > c - (FLANK + 1)) continue;
        if (p > WIDTH - 1) continue;
        int32_t *row = cnt + ((size_t)i * WIDTH) * NCNT;
        if (type == T_MATCH) {
            if (rb >= 0 && q >= 0) atomicAdd(row + p * NCNT + 5 + q, 1);
        } else if (p >= 1 && r > sg.pos) {
            if (type == T_DEL) {
                if (rb >= 0) atomicAdd(row + p * NCNT + 4, 1);
            } else if (q >= 0) {
                int idx = p + sg.adv0 + lane;
                if (idx > WIDTH - 1 || idx < 0) idx = WIDTH - 1;
                atomicAdd(row + idx * NCNT + q, 1);
            }
        }
    }
}

__global__ void __launch_bounds__(256)
pileup_finalize(const int32_t *__restrict__ cnt, const int32_t *__restrict__ cands, int64_t n,
                const uint8_t *__restrict__ ref, int64_t ref_first, int64_t ref_len, float *__restrict__ out,
                int32_t *__restrict__ depth, int subtract)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * WIDTH) return;
    const int64_t i = t / WIDTH;
    const int p = (int)(t - i * WIDTH);
    const int32_t *c9 = cnt + t * NCNT;
    int ins[4], mq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { ins[k] = c9[k]; mq[k] = c9[5 + k]; }
    const int del = c9[4];
    const int sum = mq[0] + mq[1] + mq[2] + mq[3];
    const int64_t ri = (int64_t)cands[i] - (FLANK + 1) + p - ref_first;
    const int rb = (ri >= 0 && ri < ref_len) ? base_code(ref[ri]) : -1;
    if (p == FLANK && depth) depth[i] = sum;
    if (!out) return;
    float4 *o = reinterpret_cast<float4 *>(out + t * 16);
#pragma unroll
    for (int bse = 0; bse < 4; ++bse) {
        float m0 = bse == rb ? (float)sum : 0.f;
        float m1 = (float)(mq[bse] + ins[bse]);
        float m2 = bse == rb ? (float)(sum + del) : 0.f;
        float m3 = (float)mq[bse];
        if (subtract) { m1 -= m0; m2 -= m0; m3 -= m0; }
        o[bse] = make_float4(m0, m1, m2, m3);
    }
}

}  // namespace

struct cv_pileup {
    int device = 0, min_mq = 0, dcov = 250, left = 1;
    uint8_t *ref_dev = nullptr;
    int64_t ref_len = 0, ref_first = 0;
    int32_t *cand_dev = nullptr;
    int64_t n = 0;
    int32_t *bucket_dev = nullptr;
    int32_t bucket_lo = 0;
    int64_t nb = 0;
    int32_t *cnt_dev = nullptr;
    uint8_t *touched_dev = nullptr;
    std::vector<seg_t> segs;
    std::vector<uint8_t> seq;
    int64_t pending_cols = 0;
    seg_t *segs_dev = nullptr;
    size_t segs_cap = 0;
    uint8_t *seq_dev = nullptr;
    size_t seq_cap = 0;
    int64_t prev_pos = 0, depth_cap = 0;   // CreateTensor.py:139,165-172
    std::vector<hipEvent_t> ev;            // pairs, scatter launches not yet accumulated
    std::vector<hipEvent_t> evf;           // pairs, finalize launches
    float ms_scatter = 0.f, ms_final = 0.f;
    int64_t cols = 0, nsegs = 0, launches = 0;
};

static int drain(std::vector<hipEvent_t> &ev, float &acc)
{
    for (size_t k = 0; k + 1 < ev.size(); k += 2) {
        float ms = 0.f;
        PL_HIP(hipEventSynchronize(ev[k + 1]));
        PL_HIP(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
        acc += ms;
        hipEventDestroy(ev[k]);
        hipEventDestroy(ev[k + 1]);
    }
    ev.clear();
    return 0;
}

extern "C" int cv_pileup_create(int device, int min_mq, int dcov, int consider_left_edge, cv_pileup **out)
{
    if (!out) { cv_set_error("cv_pileup_create: null argument"); return 1; }
    int ndev = 0;
    PL_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        cv_set_error("cv_pileup_create: device %d not present (%d visible)", device, ndev);
        return 1;
    }
    cv_pileup *p = new (std::nothrow) cv_pileup();
    if (!p) { cv_set_error("cv_pileup_create: out of host memory"); return 1; }
    p->device = device; p->min_mq = min_mq; p->dcov = dcov; p->left = consider_left_edge ? 1 : 0;
    *out = p;
    return 0;
}

extern "C" void cv_pileup_destroy(cv_pileup *p)
{
    if (!p) return;
    hipSetDevice(p->device);
    float sink = 0.f;
    drain(p->ev, sink); drain(p->evf, sink);
    hipFree(p->ref_dev); hipFree(p->cand_dev); hipFree(p->bucket_dev); hipFree(p->cnt_dev);
    hipFree(p->touched_dev); hipFree(p->segs_dev); hipFree(p->seq_dev);
    delete p;
}

extern "C" int cv_pileup_set_reference(cv_pileup *p, const char *seq, int64_t len, int64_t first_pos0)
{
    if (!p || (!seq && len > 0) || len < 0) { cv_set_error("cv_pileup_set_reference: bad argument"); return 1; }
    PL_HIP(hipSetDevice(p->device));
    if (p->ref_dev) { PL_HIP(hipFree(p->ref_dev)); p->ref_dev = nullptr; }
    PL_HIP(hipMalloc(&p->ref_dev, (size_t)(len > 0 ? len : 1)));
    if (len > 0) PL_HIP(hipMemcpy(p->ref_dev, seq, (size_t)len, hipMemcpyHostToDevice));
    p->ref_len = len; p->ref_first = first_pos0;
    return 0;
}

extern "C" int cv_pileup_set_candidates(cv_pileup *p, const int64_t *centers, int64_t n)
{
    if (!p || (!centers && n > 0) || n < 0) { cv_set_error("cv_pileup_set_candidates: bad argument"); return 1; }
    if (!p->segs.empty()) { cv_set_error("cv_pileup_set_candidates: reads are queued; flush first"); return 1; }
    PL_HIP(hipSetDevice(p->device));
    std::vector<int32_t> c32((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (centers[i] < -(1LL << 30) || centers[i] > (1LL << 31) - 64 || (i && centers[i] <= centers[i - 1])) {
            cv_set_error("cv_pileup_set_candidates: centres must be strictly ascending 1-based positions below 2^31 "
                         "(index %lld: %lld)", (long long)i, (long long)centers[i]);
            return 1;
        }
        c32[(size_t)i] = (int32_t)centers[i];
    }
    hipFree(p->cand_dev); hipFree(p->bucket_dev); hipFree(p->cnt_dev); hipFree(p->touched_dev);
    p->cand_dev = nullptr; p->bucket_dev = nullptr; p->cnt_dev = nullptr; p->touched_dev = nullptr;
    p->n = n;
    size_t n1 = (size_t)(n > 0 ? n : 1);
    PL_HIP(hipMalloc(&p->cand_dev, n1 * sizeof(int32_t)));
    PL_HIP(hipMalloc(&p->cnt_dev, n1 * WIDTH * NCNT * sizeof(int32_t)));
    PL_HIP(hipMalloc(&p->touched_dev, n1));
    PL_HIP(hipMemset(p->cnt_dev, 0, n1 * WIDTH * NCNT * sizeof(int32_t)));
    PL_HIP(hipMemset(p->touched_dev, 0, n1));
    if (n > 0) {
        PL_HIP(hipMemcpy(p->cand_dev, c32.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
        p->bucket_lo = c32[0] - (FLANK + 1);
        p->nb = (((int64_t)c32[(size_t)n - 1] + FLANK + 2 - p->bucket_lo) >> BUCKET_SHIFT) + 1;
    } else {
        p->bucket_lo = 0; p->nb = 1;
    }
    PL_HIP(hipMalloc(&p->bucket_dev, (size_t)p->nb * sizeof(int32_t)));
    int threads = 256;
    bucket_build<<<(unsigned)((p->nb + threads - 1) / threads), threads>>>(p->cand_dev, (int)n, p->bucket_lo, p->nb,
                                                                             p->bucket_dev);
    PL_HIP(hipGetLastError());
    PL_HIP(hipDeviceSynchronize());
    return 0;
}

// ---- SAM text -> segments ---------------------------------------------------------------------

static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

static void emit(cv_pileup *p, int type, int64_t r0, uint64_t q0, int64_t n, int64_t pos, bool ref_advances)
{
    int64_t done = 0;
    while (done < n) {
        int64_t len = n - done < SEG_MAX ? n - done : SEG_MAX;
        seg_t s;
        s.r0 = (int32_t)(ref_advances ? r0 + done : r0);
        s.q0 = (uint32_t)(q0 + (type == T_DEL ? 0 : (uint64_t)done));
        s.info = (int32_t)len | (type << 8);
        s.adv0 = type == T_INS ? (int32_t)done : 0;
        s.pos = (int32_t)pos;
        p->segs.push_back(s);
        done += len;
    }
    p->pending_cols += n;
}

// one SAM record (fields split on white space like `l.split()`, CreateTensor.py:141-152)
static int parse_record(cv_pileup *p, const char *line, const char *end, int64_t *kept)
{
    const char *f[10];
    const char *fe[10];
    int nf = 0;
    const char *c = line;
    while (c < end && nf < 10) {
        while (c < end && is_ws(*c)) ++c;
        if (c >= end) break;
        f[nf] = c;
        while (c < end && !is_ws(*c)) ++c;
        fe[nf] = c;
        ++nf;
    }
    if (nf == 0) return 0;                 // blank line
    if (f[0][0] == '@') return 0;          // header (:142)
    if (nf < 10) {
        cv_set_error("cv_pileup_add_sam: record with %d fields (need 10): %.60s", nf, line);
        return 1;
    }
    char *endp = nullptr;
    const int64_t pos = strtoll(f[3], &endp, 10) - 1;      // 0-based (:147)
    const int64_t mq = strtoll(f[4], &endp, 10);
    if (mq < p->min_mq) return 0;                           // :155
    if (p->prev_pos != pos) { p->prev_pos = pos; p->depth_cap = 0; }
    else if (++p->depth_cap >= p->dcov) return 0;           // :165-172
    if (pos < -(1LL << 30) || pos > (1LL << 31) - (1 << 24)) {
        cv_set_error("cv_pileup_add_sam: POS %lld out of range", (long long)pos + 1);
        return 1;
    }
    const char *cg = f[5], *cge = fe[5];
    const int64_t seqlen = fe[9] - f[9];
    // query bases the CIGAR asks for (a short or absent SEQ is padded with 'not ACGT')
    int64_t need = 0;
    for (const char *q = cg; q < cge;) {
        if (*q < '0' || *q > '9') { ++q; continue; }
        int64_t v = 0;
        while (q < cge && *q >= '0' && *q <= '9') v = v * 10 + (*q++ - '0');
        if (q < cge && (*q == 'M' || *q == 'I' || *q == 'S' || *q == '=' || *q == 'X')) need += v;
    }
    if ((uint64_t)p->seq.size() + (uint64_t)(need > seqlen ? need : seqlen) >= 0xffffffffull) {
        cv_set_error("cv_pileup_add_sam: more than 4 Gi query bases queued; call cv_pileup_flush more often");
        return 1;
    }
    const uint64_t base = p->seq.size();
    p->seq.insert(p->seq.end(), (const uint8_t *)f[9], (const uint8_t *)fe[9]);
    if (need > seqlen) p->seq.insert(p->seq.end(), (size_t)(need - seqlen), (uint8_t)'?');
    int64_t r = pos, q = 0;
    for (const char *s = cg; s < cge;) {                    // re.finditer(r"(\d+)([MIDNSHP=X])") (:174)
        if (*s < '0' || *s > '9') { ++s; continue; }
        int64_t v = 0;
        while (s < cge && *s >= '0' && *s <= '9') v = v * 10 + (*s++ - '0');
        if (s >= cge) break;
        const char op = *s;
        if (op == 'S') { q += v; ++s; }
        else if (op == 'M' || op == '=' || op == 'X') { emit(p, T_MATCH, r, base + q, v, pos, true); r += v; q += v; ++s; }
        else if (op == 'I') { emit(p, T_INS, r, base + q, v, pos, false); q += v; ++s; }
        else if (op == 'D') { emit(p, T_DEL, r, 0, v, pos, true); r += v; ++s; }
        else if (op == 'N' || op == 'H' || op == 'P') { ++s; }   // no branch in the reference: nothing moves
        // any other character: the regex does not match at these digits; rescan from the next character
    }
    ++*kept;
    return 0;
}

extern "C" int cv_pileup_add_sam(cv_pileup *p, const char *text, int64_t nbytes, int final, int64_t *consumed,
                                 int64_t *kept)
{
    if (!p || (!text && nbytes > 0) || nbytes < 0) { cv_set_error("cv_pileup_add_sam: bad argument"); return 1; }
    int64_t k = 0, done = 0;
    const char *cur = text, *end = text + nbytes;
    while (cur < end) {
        const char *nl = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        if (!nl && !final) break;
        const char *le = nl ? nl : end;
        if (parse_record(p, cur, le, &k)) return 1;
        cur = nl ? nl + 1 : end;
        done = cur - text;
    }
    if (consumed) *consumed = done;
    if (kept) *kept = k;
    return 0;
}

extern "C" int64_t cv_pileup_pending(const cv_pileup *p) { return p ? p->pending_cols : 0; }

extern "C" int cv_pileup_flush(cv_pileup *p, void *stream)
{
    if (!p) { cv_set_error("cv_pileup_flush: null handle"); return 1; }
    if (p->segs.empty()) { p->seq.clear(); return 0; }
    if (!p->cand_dev || !p->ref_dev) {
        cv_set_error("cv_pileup_flush: set the reference and the candidates first");
        return 1;
    }
    PL_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t ns = p->segs.size(), nq = p->seq.size() + 64;
    if (ns > p->segs_cap) {
        PL_HIP(hipStreamSynchronize(st));
        hipFree(p->segs_dev); p->segs_dev = nullptr;
        p->segs_cap = ns + ns / 4;
        PL_HIP(hipMalloc(&p->segs_dev, p->segs_cap * sizeof(seg_t)));
    }
    if (nq > p->seq_cap) {
        PL_HIP(hipStreamSynchronize(st));
        hipFree(p->seq_dev); p->seq_dev = nullptr;
        p->seq_cap = nq + nq / 4;
        PL_HIP(hipMalloc(&p->seq_dev, p->seq_cap));
    }
    p->seq.resize(nq, (uint8_t)'?');       // a 64-byte tail keeps every lane's read in range
    PL_HIP(hipMemcpyAsync(p->segs_dev, p->segs.data(), ns * sizeof(seg_t), hipMemcpyHostToDevice, st));
    PL_HIP(hipMemcpyAsync(p->seq_dev, p->seq.data(), nq, hipMemcpyHostToDevice, st));
    if (p->n > 0) {
        if (p->ev.size() >= 128 && drain(p->ev, p->ms_scatter)) return 1;
        hipEvent_t e0, e1;
        PL_HIP(hipEventCreate(&e0)); PL_HIP(hipEventCreate(&e1));
        PL_HIP(hipEventRecord(e0, st));
        const int waves = 4;
        pileup_scatter<<<(unsigned)((ns + waves - 1) / waves), waves * 64, 0, st>>>(
            p->segs_dev, (int64_t)ns, p->seq_dev, p->ref_dev, p->ref_first, p->ref_len, p->cand_dev, (int)p->n,
            p->bucket_dev, p->bucket_lo, p->nb, p->cnt_dev, p->touched_dev, p->left);
        PL_HIP(hipGetLastError());
        PL_HIP(hipEventRecord(e1, st));
        p->ev.push_back(e0); p->ev.push_back(e1);
        p->launches += 1;
    }
    // pageable copies are staged before hipMemcpyAsync returns only for small sizes: wait for them
    PL_HIP(hipStreamSynchronize(st));
    p->cols += p->pending_cols; p->nsegs += (int64_t)ns;
    p->segs.clear(); p->seq.clear(); p->pending_cols = 0;
    return 0;
}

extern "C" int cv_pileup_finish(cv_pileup *p, float *tensors_dev, int32_t *depth_dev, uint8_t *touched_dev,
                                int subtract, void *stream)
{
    if (!p) { cv_set_error("cv_pileup_finish: null handle"); return 1; }
    if (!p->cand_dev || !p->ref_dev) {
        cv_set_error("cv_pileup_finish: set the reference and the candidates first");
        return 1;
    }
    if (cv_pileup_flush(p, stream)) return 1;
    hipStream_t st = (hipStream_t)stream;
    if (p->n == 0) return 0;
    if (tensors_dev || depth_dev) {
        hipEvent_t e0, e1;
        PL_HIP(hipEventCreate(&e0)); PL_HIP(hipEventCreate(&e1));
        PL_HIP(hipEventRecord(e0, st));
        const int64_t total = p->n * WIDTH;
        pileup_finalize<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(p->cnt_dev, p->cand_dev, p->n, p->ref_dev,
                                                                           p->ref_first, p->ref_len, tensors_dev,
                                                                           depth_dev, subtract);
        PL_HIP(hipGetLastError());
        PL_HIP(hipEventRecord(e1, st));
        p->evf.push_back(e0); p->evf.push_back(e1);
    }
    if (touched_dev)
        PL_HIP(hipMemcpyAsync(touched_dev, p->touched_dev, (size_t)p->n, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int cv_pileup_stats(cv_pileup *p, float ms[2], int64_t counts[3])
{
    if (!p) { cv_set_error("cv_pileup_stats: null handle"); return 1; }
    PL_HIP(hipSetDevice(p->device));
    if (drain(p->ev, p->ms_scatter) || drain(p->evf, p->ms_final)) return 1;
    if (ms) { ms[0] = p->ms_scatter; ms[1] = p->ms_final; }
    if (counts) { counts[0] = p->cols; counts[1] = p->nsegs; counts[2] = p->launches; }
    return 0;
}

extern "C" int64_t cv_format_tensor_row(const char *ctg, int64_t center, const char *seq, int64_t seqlen,
                                        const float *counts, char *dst, int64_t cap)
{
    if (!ctg || !counts || !dst || (!seq && seqlen > 0)) return -1;
    const int64_t nvals = (int64_t)WIDTH * 16;
    int64_t w = snprintf(dst, (size_t)cap, "%s %lld ", ctg, (long long)center);
    if (w < 0 || w + seqlen + nvals * 16 + 2 > cap) return -1;
    memcpy(dst + w, seq, (size_t)seqlen);
    w += seqlen;
    for (int64_t k = 0; k < nvals; ++k) {
        dst[w++] = ' ';
        const float v = counts[k];
        if (v >= 0.f && v < 16777216.f && v == (float)(int32_t)v) {   // "%0.1f" of a small whole number
            char tmp[12];
            int t = 0;
            int32_t u = (int32_t)v;
            do { tmp[t++] = (char)('0' + u % 10); u /= 10; } while (u);
            while (t) dst[w++] = tmp[--t];
            dst[w++] = '.'; dst[w++] = '0';
        } else {
            w += snprintf(dst + w, (size_t)(cap - w), "%0.1f", (double)v);
        }
    }
    dst[w] = 0;
    return w;
}
