// Pileup front end: SAM text -> [n,33,4,4] count tensors in HBM (include/clairvoyante_amd.h,
// "pileup front end").  Follows the behaviour of /root/reference/dataPrepScripts/CreateTensor.py
// (OutputAlnTensor :93-246, GenerateTensor :23-54) with a different decomposition:
//
//   host   : one pass over the SAM text (several threads); every CIGAR run becomes alignment SEGMENTS of
//            <= 64 columns (20 bytes each) + the read's SEQ bytes.  No per-candidate buffering.
//   scatter: a workgroup per 512 consecutive segments, a lane per alignment column, counters of the
//            candidates in reach privatised in LDS.  Whether a column counts for a
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
//   candidates (optional, ExtractVariantCandidates.py :118-246): with option "evc" the same segments
//            first feed per-position counters A,C,G,T,I,D,N (+ the "late" I/D pair, below); a select pass
//            (hipCUB DeviceSelect over positions) applies OutputCandidate (:22-42) and the region / BED
//            tests; with option "retain" the uploaded segments stay in HBM, so the tensor scatter for
//            the selected candidates re-reads them there -- the SAM text is parsed and uploaded once.
//            The reference sweeps finished positions while it reads (:176-212); an insertion / deletion
//            that opens a read is booked at POS-1, which an earlier read with the same POS has already
//            swept: those "late" events form a second entry for that position (reported at the end,
//            :215-241).  The parser flags them, the counters keep them apart (slots 7, 8).
//
// Integer / byte work: per column 1 SEQ byte + 1 reference byte + one LDS atomic per covering candidate
// (flushed once per workgroup); per candidate 1 188 B of counters read once and 2 112 B of tensor written once.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <string>
#include <thread>
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

constexpr int NPOS = 9;                   // per-position counters: A C G T I D N | late I, late D

enum { T_MATCH = 0, T_INS = 1, T_DEL = 2 };
enum { F_CT = 1 << 10, F_EVC = 1 << 11, F_LATE = 1 << 12, F_FIRST = 1 << 13 };   // in seg_t::info

struct seg_t {
    int32_t r0;      // 0-based reference position of the first column (insert: the position it precedes)
    uint32_t q0;     // offset of the first query base in the SEQ byte buffer (unused for deletes)
    int32_t info;    // columns (1..64) | type << 8 | F_* (which consumer the read passed, first segment of its run)
    int32_t adv0;    // insert: index of the first column inside its insertion (queryAdv, :203-211)
    int32_t pos;     // the read's POS (0-based)
};

__device__ __forceinline__ int base_code(uint8_t ch)
{
    // upper-case ACGT only (CreateTensor.py:28-31 tests `in "ACGT-"`)
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
}

__global__ void rebase_q0(seg_t *__restrict__ segs, int64_t n, uint32_t base)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) segs[t].q0 += base;
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

// A workgroup takes SC_SEGS consecutive segments.  The alignments are position-sorted, so their columns
// fall on a short stretch of the contig and on a handful of candidates: the counters of the first SC_CANDS
// candidates at or after (first read's POS - 16) live in LDS and are flushed once (non-zero entries only);
// columns that reach later candidates go to the global counters directly.  Any input order is handled.
constexpr int SC_SEGS = 512;
constexpr int SC_CANDS = 40;              // 40 x 33 x 9 x 4 B = 47.5 KB of LDS

__global__ void __launch_bounds__(1024)
pileup_scatter(const seg_t *__restrict__ segs, int64_t nseg, const uint8_t *__restrict__ seq,
               const uint8_t *__restrict__ ref, int64_t ref_first, int64_t ref_len,
               const int32_t *__restrict__ cands, int n, const int32_t *__restrict__ bucket_first,
               int32_t bucket_lo, int64_t nb, int32_t *__restrict__ cnt, uint8_t *__restrict__ touched, int left)
{
    __shared__ int32_t tab[SC_CANDS * WIDTH * NCNT];
    const int64_t s0 = (int64_t)blockIdx.x * SC_SEGS;
    const int64_t s1 = s0 + SC_SEGS < nseg ? s0 + SC_SEGS : nseg;
    for (int k = threadIdx.x; k < SC_CANDS * WIDTH * NCNT; k += blockDim.x) tab[k] = 0;
    // first candidate any column of this workgroup can reach
    int i0;
    {
        const int64_t lo_c = (int64_t)segs[s0].pos - FLANK;
        int64_t b = (lo_c - bucket_lo) >> BUCKET_SHIFT;
        if (b < 0) b = 0;
        if (b >= nb) b = nb - 1;
        i0 = bucket_first[b];
        while (i0 < n && (int64_t)cands[i0] < lo_c) ++i0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    constexpr int U = 4;                  // segments in flight per wave: the loop is a chain of dependent loads
    for (int64_t sb = s0 + (threadIdx.x >> 6); sb < s1; sb += (int64_t)nw * U) {
        seg_t sg[U];
        bool on[U];
        int32_t r[U];
        int q[U], rb[U], ci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t s = sb + (int64_t)u * nw;
            on[u] = s < s1;
            sg[u] = segs[on[u] ? s : s0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int len = sg[u].info & 0xff;
            const int type = (sg[u].info >> 8) & 3;
            on[u] = on[u] && lane < len && (sg[u].info & F_CT);
            r[u] = type == T_INS ? sg[u].r0 : sg[u].r0 + lane;
            uint8_t qc = '?', rc = '?';
            if (on[u] && type != T_DEL) qc = seq[(size_t)sg[u].q0 + lane];
            const int64_t ri = (int64_t)r[u] - ref_first;
            if (on[u] && type != T_INS && ri >= 0 && ri < ref_len) rc = ref[ri];
            q[u] = base_code(qc);
            rb[u] = base_code(rc);
            int64_t b = ((int64_t)r[u] - FLANK - bucket_lo) >> BUCKET_SHIFT;
            if (b < 0) b = 0;
            if (b >= nb) on[u] = false;
            ci[u] = on[u] ? bucket_first[b] : n;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const int type = (sg[u].info >> 8) & 3;
            // candidates c with r - 16 <= c <= r + 17
            const int64_t lo_c = (int64_t)r[u] - FLANK, hi_c = (int64_t)r[u] + FLANK + 1;
            int i = ci[u];
            while (i < n && (int64_t)cands[i] < lo_c) ++i;
            for (; i < n; ++i) {
                const int64_t c = cands[i];
                if (c > hi_c) break;
                const int p = (int)(r[u] - c) + FLANK + 1;
                if (type != T_INS && (left || p == 0)) touched[i] = 1;
                if (!left && (int64_t)sg[u].pos > c - (FLANK + 1)) continue;
                if (p > WIDTH - 1) continue;
                int slot = -1;
                if (type == T_MATCH) {
                    if (rb[u] >= 0 && q[u] >= 0) slot = p * NCNT + 5 + q[u];
                } else if (p >= 1 && r[u] > sg[u].pos) {
                    if (type == T_DEL) {
                        if (rb[u] >= 0) slot = p * NCNT + 4;
                    } else if (q[u] >= 0) {
                        int idx = p + sg[u].adv0 + lane;
                        if (idx > WIDTH - 1 || idx < 0) idx = WIDTH - 1;
                        slot = idx * NCNT + q[u];
                    }
                }
                if (slot < 0) continue;
                const int j = i - i0;
                if (j >= 0 && j < SC_CANDS) atomicAdd(&tab[j * WIDTH * NCNT + slot], 1);
                else atomicAdd(cnt + ((size_t)i * WIDTH) * NCNT + slot, 1);
            }
        }
    }
    __syncthreads();
    const int live = n - i0 < SC_CANDS ? n - i0 : SC_CANDS;
    for (int k = threadIdx.x; k < live * WIDTH * NCNT; k += blockDim.x) {
        const int v = tab[k];
        if (v) atomicAdd(cnt + (size_t)i0 * WIDTH * NCNT + k, v);
    }
}

// ExtractVariantCandidates.py:152-174: per-position symbol counts
// Same tiling: EVC_SEGS consecutive segments per workgroup, the counters of the EVC_WIN positions from
// (first read's POS - 1) on live in LDS, non-zero entries are flushed once; columns outside go to HBM.
constexpr int EVC_SEGS = 512;
constexpr int EVC_WIN = 1536;             // 1536 x 9 x 4 B = 54 KB of LDS

__global__ void __launch_bounds__(1024)
evc_count(const seg_t *__restrict__ segs, int64_t nseg, const uint8_t *__restrict__ seq, int64_t ref_first,
          int64_t ref_len, int32_t *__restrict__ pc)
{
    __shared__ int32_t tab[EVC_WIN * NPOS];
    const int64_t s0 = (int64_t)blockIdx.x * EVC_SEGS;
    const int64_t s1 = s0 + EVC_SEGS < nseg ? s0 + EVC_SEGS : nseg;
    for (int k = threadIdx.x; k < EVC_WIN * NPOS; k += blockDim.x) tab[k] = 0;
    const int64_t base = (int64_t)segs[s0].pos - 1;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int64_t s = s0 + (threadIdx.x >> 6); s < s1; s += (blockDim.x >> 6)) {
        const seg_t sg = segs[s];
        if (!(sg.info & F_EVC)) continue;
        const int len = sg.info & 0xff;
        const int type = (sg.info >> 8) & 3;
        int64_t r;
        int k;
        if (type == T_MATCH) {
            if (lane >= len) continue;
            const uint8_t ch = seq[(size_t)sg.q0 + lane];
            k = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : ch == 'N' ? 6 : -1;
            r = (int64_t)sg.r0 + lane;
        } else {
            if (lane != 0 || !(sg.info & F_FIRST)) continue;       // one count per insertion / deletion run, at r-1
            k = (sg.info & F_LATE) ? (type == T_INS ? 7 : 8) : (type == T_INS ? 4 : 5);
            r = (int64_t)sg.r0 - 1;
        }
        if (k < 0) continue;
        const int64_t off = r - base;
        if (off >= 0 && off < EVC_WIN) atomicAdd(&tab[off * NPOS + k], 1);
        else {
            const int64_t ri = r - ref_first;
            if (ri >= 0 && ri < ref_len) atomicAdd(pc + ri * NPOS + k, 1);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < EVC_WIN * NPOS; k += blockDim.x) {
        const int v = tab[k];
        if (!v) continue;
        const int64_t ri = base + k / NPOS - ref_first;
        if (ri >= 0 && ri < ref_len) atomicAdd(pc + ri * NPOS + k % NPOS, v);
    }
}

// OutputCandidate (:22-42) + the region / BED tests (:181-196) for entry j: position j>>1, kind j&1
// (0 = the regular entry, 1 = the late insertion/deletion entry)
struct evc_pred {
    const int32_t *pc;
    const uint8_t *ref;
    int64_t ref_first;
    double thr, mincov;
    int has_region;
    int64_t cs, ce;
    const int64_t *bb, *be;
    int nbed;                      // -1: no BED file

    __device__ bool operator()(const int64_t &j) const
    {
        const int64_t ri = j >> 1;
        const int32_t *c9 = pc + ri * NPOS;
        int c[7];
        if (j & 1) {
#pragma unroll
            for (int k = 0; k < 7; ++k) c[k] = 0;
            c[4] = c9[7]; c[5] = c9[8];
        } else {
#pragma unroll
            for (int k = 0; k < 7; ++k) c[k] = c9[k];
        }
        int total = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) total += c[k];
        if (total == 0) return false;                      // no pileup entry at all
        const int64_t p = ref_first + ri;
        if (has_region && (p < cs || p > ce)) return false;
        if (nbed >= 0) {
            int l = 0, h = nbed;                           // disjoint, sorted: last interval with begin <= p
            while (l < h) { int m = (l + h) >> 1; if (bb[m] <= p) l = m + 1; else h = m; }
            if (l == 0 || p >= be[l - 1]) return false;
        }
        if ((double)total < mincov) return false;
        int i0 = 0;
#pragma unroll
        for (int k = 1; k < 7; ++k) if (c[k] > c[i0]) i0 = k;       // stable descending sort: first maximum
        int i1 = i0 == 0 ? 1 : 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) if (k != i0 && k != i1 && c[k] > c[i1]) i1 = k;
        // among equal seconds the stable sort keeps the lowest index: the scan above only replaces on '>'
        // but started from index 0/1, which is the lowest candidate
        const double p0 = (double)c[i0] / (double)total, p1 = (double)c[i1] / (double)total;
        const char sym[8] = "ACGTIDN";
        return (p0 <= 1.0 - thr && p1 >= thr) || (uint8_t)sym[i0] != ref[ri];
    }
};

__global__ void evc_gather(const int64_t *__restrict__ sel, int64_t n, const int32_t *__restrict__ pc,
                           int32_t *__restrict__ out7)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t j = sel[t];
    const int32_t *c9 = pc + (j >> 1) * NPOS;
    for (int k = 0; k < 7; ++k) out7[t * 7 + k] = (j & 1) ? (k == 4 ? c9[7] : k == 5 ? c9[8] : 0) : c9[k];
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

struct read_rec {                 // what step (2) needs of a read
    int64_t pos;
    uint32_t seg0, nseg;          // its segments inside the part
    uint8_t ct, evc, leading;     // passed the stateless tensor / candidate filters; has a leading indel run
};

struct sam_part {                 // output of one parser thread
    std::vector<seg_t> segs;
    std::vector<uint8_t> seq;
    std::vector<read_rec> reads;
    int64_t cols = 0;
    std::string err;
};

struct dev_batch {                // one uploaded batch of segments
    seg_t *segs = nullptr;
    uint8_t *seq = nullptr;
    size_t nseg = 0, segs_cap = 0, seq_cap = 0;
};

struct cv_pileup {
    int device = 0, min_mq = 0, dcov = 250, left = 1;
    int retain = 0, evc = 0, evc_min_mq = 0, threads = 1;
    std::string contig;                 // RNAME test of the candidate pass (ExtractVariantCandidates.py:137-139)
    uint8_t *ref_dev = nullptr;
    int64_t ref_len = 0, ref_first = 0;
    int32_t *cand_dev = nullptr;
    int64_t n = 0;
    int32_t *bucket_dev = nullptr;
    int32_t bucket_lo = 0;
    int64_t nb = 0;
    int32_t *cnt_dev = nullptr;
    uint8_t *touched_dev = nullptr;
    int32_t *pos_cnt = nullptr;         // [ref_len, NPOS], candidate pass
    std::vector<sam_part> queue;        // parsed, not yet uploaded (one entry per parser slice)
    int64_t pending_cols = 0;
    dev_batch work;                     // reused staging batch (retain == 0)
    std::vector<dev_batch> kept;        // resident batches / views (retain == 1)
    int64_t prev_pos = 0, depth_cap = 0;   // CreateTensor.py:139,165-172
    int64_t evc_prev_pos = INT64_MIN;      // POS of the last read the candidate pass took
    int64_t evc_reads = 0;                 // processedReads (:150)
    std::vector<int64_t> sel_host;         // extracted entries: position << 1 | kind
    std::vector<int32_t> sel_counts;       // [n,7]
    std::vector<hipEvent_t> ev;            // pairs, scatter launches not yet accumulated
    std::vector<hipEvent_t> evf;           // pairs, finalize launches
    std::vector<hipEvent_t> eve;           // pairs, candidate-pass launches
    float ms_scatter = 0.f, ms_final = 0.f, ms_evc = 0.f;
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

struct timed {                     // brackets launches with an event pair kept for cv_pileup_stats
    std::vector<hipEvent_t> &v;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = true;
    timed(std::vector<hipEvent_t> &vec, hipStream_t s) : v(vec), st(s)
    {
        ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess &&
             hipEventRecord(e0, st) == hipSuccess;
    }
    ~timed()
    {
        if (ok && hipEventRecord(e1, st) == hipSuccess) { v.push_back(e0); v.push_back(e1); }
    }
};

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

static void free_batch(dev_batch &b)
{
    hipFree(b.segs); hipFree(b.seq);
    b = dev_batch();
}

extern "C" void cv_pileup_destroy(cv_pileup *p)
{
    if (!p) return;
    hipSetDevice(p->device);
    float sink = 0.f;
    drain(p->ev, sink); drain(p->evf, sink); drain(p->eve, sink);
    hipFree(p->ref_dev); hipFree(p->cand_dev); hipFree(p->bucket_dev); hipFree(p->cnt_dev);
    hipFree(p->touched_dev); hipFree(p->pos_cnt);
    free_batch(p->work);
    for (auto &b : p->kept) free_batch(b);
    delete p;
}

extern "C" int cv_pileup_set_option(cv_pileup *p, const char *key, int64_t value)
{
    if (!p || !key) { cv_set_error("cv_pileup_set_option: null argument"); return 1; }
    if (!p->queue.empty() || p->cols) {
        cv_set_error("cv_pileup_set_option(%s): set options before the first read is added", key);
        return 1;
    }
    if (!strcmp(key, "retain")) p->retain = value != 0;
    else if (!strcmp(key, "evc")) p->evc = value != 0;
    else if (!strcmp(key, "evc_min_mq")) p->evc_min_mq = (int)value;
    else if (!strcmp(key, "threads")) p->threads = value < 1 ? 1 : value > 64 ? 64 : (int)value;
    else { cv_set_error("cv_pileup_set_option: unknown key '%s'", key); return 1; }
    return 0;
}

extern "C" int cv_pileup_set_contig(cv_pileup *p, const char *name)
{
    if (!p || !name) { cv_set_error("cv_pileup_set_contig: null argument"); return 1; }
    p->contig = name;
    return 0;
}

extern "C" int cv_pileup_set_reference(cv_pileup *p, const char *seq, int64_t len, int64_t first_pos0)
{
    if (!p || (!seq && len > 0) || len < 0) { cv_set_error("cv_pileup_set_reference: bad argument"); return 1; }
    PL_HIP(hipSetDevice(p->device));
    if (p->ref_dev) { PL_HIP(hipFree(p->ref_dev)); p->ref_dev = nullptr; }
    if (p->pos_cnt) { PL_HIP(hipFree(p->pos_cnt)); p->pos_cnt = nullptr; }
    PL_HIP(hipMalloc(&p->ref_dev, (size_t)(len > 0 ? len : 1)));
    if (len > 0) PL_HIP(hipMemcpy(p->ref_dev, seq, (size_t)len, hipMemcpyHostToDevice));
    p->ref_len = len; p->ref_first = first_pos0;
    return 0;
}

static int install_candidates(cv_pileup *p, const std::vector<int32_t> &c32)
{
    const int64_t n = (int64_t)c32.size();
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

extern "C" int cv_pileup_set_candidates(cv_pileup *p, const int64_t *centers, int64_t n)
{
    if (!p || (!centers && n > 0) || n < 0) { cv_set_error("cv_pileup_set_candidates: bad argument"); return 1; }
    if (!p->queue.empty()) { cv_set_error("cv_pileup_set_candidates: reads are queued; flush first"); return 1; }
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
    return install_candidates(p, c32);
}

// ---- SAM text -> segments ---------------------------------------------------------------------
// Two steps, so that the text can be parsed by several threads: (1) every record is parsed on its own
// (fields, CIGAR runs -> segments, the filters that need nothing but the record); (2) one cheap sequential
// pass over the READS applies the two pieces of running state the reference keeps -- the per-POS depth cap
// of the tensor pass (CreateTensor.py:165-172) and "an earlier read with this POS has swept POS-1" of the
// candidate pass (ExtractVariantCandidates.py:176) -- by clearing flag bits in the segments of the reads
// concerned (a segment with no flag left is skipped by both kernels).

static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; }

static void emit(sam_part &out, int type, int flags, int64_t r0, uint64_t q0, int64_t n, int64_t pos, bool ref_advances)
{
    int64_t done = 0;
    while (done < n) {
        int64_t len = n - done < SEG_MAX ? n - done : SEG_MAX;
        seg_t s;
        s.r0 = (int32_t)(ref_advances ? r0 + done : r0);
        s.q0 = (uint32_t)(q0 + (type == T_DEL ? 0 : (uint64_t)done));
        s.info = (int32_t)len | (type << 8) | flags | (done == 0 ? F_FIRST : 0);
        s.adv0 = type == T_INS ? (int32_t)done : 0;
        s.pos = (int32_t)pos;
        out.segs.push_back(s);
        done += len;
    }
    out.cols += n;
}

// one SAM record (fields split on white space like `l.split()`, CreateTensor.py:141-152); false = error in out.err
static bool parse_record(const cv_pileup *p, sam_part &out, const char *line, const char *end)
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
    if (nf == 0) return true;              // blank line
    if (f[0][0] == '@') return true;       // header (:142)
    char msg[160];
    if (nf < 10) {
        snprintf(msg, sizeof(msg), "cv_pileup_add_sam: record with %d fields (need 10): %.*s", nf,
                 (int)(end - line < 60 ? end - line : 60), line);          // (the text is not NUL-terminated)
        out.err = msg;
        return false;
    }
    char *endp = nullptr;
    const int64_t pos = strtoll(f[3], &endp, 10) - 1;      // 0-based (:147)
    const int64_t mq = strtoll(f[4], &endp, 10);
    const char *cg = f[5], *cge = fe[5];
    const int64_t seqlen = fe[9] - f[9];
    // one scan of the CIGAR: query bases it asks for, all run lengths, soft-clipped bases
    // (a run is at most 2^28 - 1 in BAM; a longer one in SAM text is damage, and it would size the segment list)
    constexpr int64_t RUN_MAX = (1LL << 28) - 1, NEED_MAX = 1LL << 31;
    int64_t need = 0, total = 0, clipped = 0;
    for (const char *q = cg; q < cge;) {
        if (*q < '0' || *q > '9') { ++q; continue; }
        int64_t v = 0;
        while (q < cge && *q >= '0' && *q <= '9') { if (v <= RUN_MAX) v = v * 10 + (*q - '0'); ++q; }
        if (q >= cge) break;
        const char op = *q;
        if (v > RUN_MAX && (op == 'M' || op == 'I' || op == 'D' || op == 'N' || op == 'S' || op == 'H' || op == 'P' || op == '=' || op == 'X')) {
            snprintf(msg, sizeof(msg), "cv_pileup_add_sam: CIGAR run longer than %lld at POS %lld", (long long)RUN_MAX, (long long)pos + 1);
            out.err = msg;
            return false;
        }
        if (op == 'M' || op == 'I' || op == 'S' || op == '=' || op == 'X') need += v;
        if (op == 'M' || op == 'I' || op == 'D' || op == 'N' || op == 'S' || op == 'H' || op == 'P' || op == '=' ||
            op == 'X') { total += v; if (op == 'S') clipped += v; }
    }
    // tensor pass: --minMQ (CreateTensor.py:155).  Candidate pass: contig, --minMQ, at least 55 % of the read
    // aligned (ExtractVariantCandidates.py:137-160)
    const bool ct_ok = mq >= p->min_mq;
    bool evc_ok = p->evc != 0 && mq >= p->evc_min_mq;
    if (evc_ok && !p->contig.empty()) {
        const size_t ln = (size_t)(fe[2] - f[2]);
        evc_ok = ln == p->contig.size() && !memcmp(f[2], p->contig.data(), ln);
    }
    if (evc_ok && 1.0 - (double)clipped / (double)(total + 1) < 0.55) evc_ok = false;
    if (!ct_ok && !evc_ok) return true;
    if (pos < -(1LL << 30) || pos > (1LL << 31) - (1 << 24)) {
        snprintf(msg, sizeof(msg), "cv_pileup_add_sam: POS %lld out of range", (long long)pos + 1);
        out.err = msg;
        return false;
    }
    if (need > NEED_MAX || total > (1LL << 40)) {
        snprintf(msg, sizeof(msg), "cv_pileup_add_sam: CIGAR at POS %lld asks for %lld query bases", (long long)pos + 1, (long long)need);
        out.err = msg;
        return false;
    }
    const int rf = (ct_ok ? F_CT : 0) | (evc_ok ? F_EVC : 0);
    const uint64_t base = out.seq.size();
    out.seq.insert(out.seq.end(), (const uint8_t *)f[9], (const uint8_t *)fe[9]);
    if (need > seqlen) out.seq.insert(out.seq.end(), (size_t)(need - seqlen), (uint8_t)'?');   // short / absent SEQ
    read_rec rr;
    rr.pos = pos; rr.seg0 = (uint32_t)out.segs.size(); rr.ct = ct_ok; rr.evc = evc_ok; rr.leading = 0;
    int64_t r = pos, q = 0;
    for (const char *s = cg; s < cge;) {                    // re.finditer(r"(\d+)([MIDNSHP=X])") (:174)
        if (*s < '0' || *s > '9') { ++s; continue; }
        int64_t v = 0;
        while (s < cge && *s >= '0' && *s <= '9') { if (v <= RUN_MAX) v = v * 10 + (*s - '0'); ++s; }     // (checked above)
        if (s >= cge) break;
        const char op = *s;
        // an insertion / deletion run that opens the read (r == POS) is provisionally "late"; step (2) keeps the
        // mark only for reads that follow another candidate-pass read with the same POS
        const int lf = rf | ((evc_ok && r == pos) ? F_LATE : 0);
        if (op == 'S') { q += v; ++s; }
        else if (op == 'M' || op == '=' || op == 'X') { emit(out, T_MATCH, rf, r, base + q, v, pos, true); r += v; q += v; ++s; }
        else if (op == 'I') { if (lf & F_LATE) rr.leading = 1; emit(out, T_INS, lf, r, base + q, v, pos, false); q += v; ++s; }
        else if (op == 'D') { if (lf & F_LATE) rr.leading = 1; emit(out, T_DEL, lf, r, 0, v, pos, true); r += v; ++s; }
        else if (op == 'N' || op == 'H' || op == 'P') { ++s; }   // no branch in the reference: nothing moves
        // any other character: the regex does not match at these digits; rescan from the next character
    }
    rr.nseg = (uint32_t)out.segs.size() - rr.seg0;
    out.reads.push_back(rr);
    return true;
}

static void parse_range(const cv_pileup *p, sam_part *out, const char *cur, const char *end)
{
    out->segs.reserve((size_t)(end - cur) / 48);
    out->seq.reserve((size_t)(end - cur) * 3 / 4);
    while (cur < end) {
        const char *nl = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        const char *le = nl ? nl : end;
        if (!parse_record(p, *out, cur, le)) return;
        cur = nl ? nl + 1 : end;
    }
}

// ---- the same reads straight from BAM records (SAM/BAM specification 4.2; no text in between) -------------------
// rec points at refID; fields: refID pos | l_read_name mapq bin | n_cigar_op flag | l_seq | next_refID next_pos tlen
// | read_name | cigar (uint32: len << 4 | op, ops "MIDNSHP=X") | seq (4 bits per base, "=ACMGRSVTWYHKDBN") | qual.
// Everything follows parse_record on the line `samtools view` would print for the record: SEQ "*" for l_seq 0,
// CIGAR "*" for no operations, RNAME = the contig of the view.
static inline int32_t le32(const uint8_t *q) { int32_t v; memcpy(&v, q, 4); return v; }
extern "C" int cv_bam_record_cigar(const uint8_t *rec, const uint8_t **ops, int64_t *n);      // cv_bam.cpp

static bool parse_bam_record(const cv_pileup *p, sam_part &out, const uint8_t *rec, bool contig_ok)
{
    static const char NT[] = "=ACMGRSVTWYHKDBN";
    const int64_t pos = le32(rec + 4);
    const int l_name = rec[8];
    const int64_t mq = rec[9];
    const int n_inline = (int)(rec[12] | (rec[13] << 8));
    const int64_t l_seq = le32(rec + 16);
    const uint8_t *sq = rec + 32 + l_name + 4 * (size_t)n_inline;
    const uint8_t *cg = rec + 32 + l_name;
    int64_t n_cig = n_inline;
    if (cv_bam_record_cigar(rec, &cg, &n_cig)) { out.err = "cv_pileup_add_bam: placeholder CIGAR without a CG:B,I tag"; return false; }
    const int64_t seqlen = l_seq > 0 ? l_seq : 1;          // "*"
    int64_t need = 0, total = 0, clipped = 0;
    for (int64_t k = 0; k < n_cig; k++) {
        const uint32_t c = (uint32_t)le32(cg + 4 * k);
        const int op = (int)(c & 15);
        const int64_t v = c >> 4;
        if (op > 8) continue;                               // printed as '?': not an operation for the parser
        if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) need += v;
        total += v;
        if (op == 4) clipped += v;
    }
    const bool ct_ok = mq >= p->min_mq;
    bool evc_ok = p->evc != 0 && mq >= p->evc_min_mq && (p->contig.empty() || contig_ok);
    if (evc_ok && 1.0 - (double)clipped / (double)(total + 1) < 0.55) evc_ok = false;
    if (!ct_ok && !evc_ok) return true;
    if (pos < -(1LL << 30) || pos > (1LL << 31) - (1 << 24)) {
        char msg[96];
        snprintf(msg, sizeof(msg), "cv_pileup_add_bam: POS %lld out of range", (long long)pos + 1);
        out.err = msg;
        return false;
    }
    if (need > (1LL << 31) || total > (1LL << 40)) {
        char msg[112];
        snprintf(msg, sizeof(msg), "cv_pileup_add_bam: CIGAR at POS %lld asks for %lld query bases", (long long)pos + 1, (long long)need);
        out.err = msg;
        return false;
    }
    const int rf = (ct_ok ? F_CT : 0) | (evc_ok ? F_EVC : 0);
    const uint64_t base = out.seq.size();
    if (l_seq > 0) {
        out.seq.resize(base + (size_t)l_seq);
        uint8_t *w = out.seq.data() + base;
        int64_t k = 0;
        for (; k + 1 < l_seq; k += 2) { const uint8_t b = sq[k >> 1]; w[k] = (uint8_t)NT[b >> 4]; w[k + 1] = (uint8_t)NT[b & 15]; }
        if (k < l_seq) w[k] = (uint8_t)NT[sq[k >> 1] >> 4];
    } else {
        out.seq.push_back((uint8_t)'*');
    }
    if (need > seqlen) out.seq.insert(out.seq.end(), (size_t)(need - seqlen), (uint8_t)'?');
    read_rec rr;
    rr.pos = pos; rr.seg0 = (uint32_t)out.segs.size(); rr.ct = ct_ok; rr.evc = evc_ok; rr.leading = 0;
    int64_t r = pos, q = 0;
    for (int64_t k = 0; k < n_cig; k++) {
        const uint32_t c = (uint32_t)le32(cg + 4 * k);
        const int op = (int)(c & 15);
        const int64_t v = c >> 4;
        const int lf = rf | ((evc_ok && r == pos) ? F_LATE : 0);
        if (op == 4) q += v;
        else if (op == 0 || op == 7 || op == 8) { emit(out, T_MATCH, rf, r, base + q, v, pos, true); r += v; q += v; }
        else if (op == 1) { if (lf & F_LATE) rr.leading = 1; emit(out, T_INS, lf, r, base + q, v, pos, false); q += v; }
        else if (op == 2) { if (lf & F_LATE) rr.leading = 1; emit(out, T_DEL, lf, r, 0, v, pos, true); r += v; }
    }
    rr.nseg = (uint32_t)out.segs.size() - rr.seg0;
    out.reads.push_back(rr);
    return true;
}

static void parse_bam_range(const cv_pileup *p, sam_part *out, const uint8_t *base, const uint32_t *offs, int64_t i0,
                            int64_t i1, bool contig_ok)
{
    if (i1 > i0) {
        const size_t bytes = (size_t)(offs[i1 - 1] - offs[i0]) + 512;
        out->segs.reserve(bytes / 40);
        out->seq.reserve(bytes);
    }
    for (int64_t i = i0; i < i1; i++)
        if (!parse_bam_record(p, *out, base + offs[i], contig_ok)) return;
}

static int absorb_parts(cv_pileup *p, std::vector<sam_part> &parts, int64_t *kept);

extern "C" int cv_pileup_add_sam(cv_pileup *p, const char *text, int64_t nbytes, int final, int64_t *consumed,
                                 int64_t *kept)
{
    if (!p || (!text && nbytes > 0) || nbytes < 0) { cv_set_error("cv_pileup_add_sam: bad argument"); return 1; }
    // whole lines only (unless final)
    int64_t usable = nbytes;
    if (!final) {
        while (usable > 0 && text[usable - 1] != '\n') --usable;
    }
    if (consumed) *consumed = usable;
    if (kept) *kept = 0;
    if (usable == 0) return 0;
    // (1) parse, in line-aligned slices
    int T = p->threads > 1 && usable >= (1 << 20) ? p->threads : 1;
    std::vector<sam_part> parts((size_t)T);
    std::vector<const char *> cut((size_t)T + 1);
    cut[0] = text; cut[(size_t)T] = text + usable;
    for (int t = 1; t < T; ++t) {
        const char *c = text + usable * t / T;
        const char *nl = (const char *)memchr(c, '\n', (size_t)(text + usable - c));
        cut[(size_t)t] = nl ? nl + 1 : text + usable;
        if (cut[(size_t)t] < cut[(size_t)t - 1]) cut[(size_t)t] = cut[(size_t)t - 1];
    }
    if (T == 1) {
        parse_range(p, &parts[0], cut[0], cut[1]);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(parse_range, p, &parts[(size_t)t], cut[(size_t)t], cut[(size_t)t + 1]);
        for (auto &x : th) x.join();
    }
    return absorb_parts(p, parts, kept);
}

// (2) running state over the reads of the parser slices (in order), then append them to the queue
static int absorb_parts(cv_pileup *p, std::vector<sam_part> &parts, int64_t *kept)
{
    for (auto &part : parts)
        if (!part.err.empty()) { cv_set_error("%s", part.err.c_str()); return 1; }
    int64_t k = 0;
    for (auto &part : parts) {
        for (const read_rec &rr : part.reads) {
            int clear = 0;
            if (rr.ct) {                                              // CreateTensor.py:165-172
                if (p->prev_pos != rr.pos) { p->prev_pos = rr.pos; p->depth_cap = 0; }
                else if (++p->depth_cap >= p->dcov) clear |= F_CT;
            }
            bool late = false;
            if (rr.evc) {                                             // ExtractVariantCandidates.py:150,176
                late = p->evc_prev_pos == rr.pos;
                p->evc_prev_pos = rr.pos;
                p->evc_reads += 1;
            }
            if (rr.leading && !late) clear |= F_LATE;
            if (clear) {
                bool alive = false;
                for (uint32_t i = rr.seg0; i < rr.seg0 + rr.nseg; ++i) {
                    part.segs[i].info &= ~clear;
                    alive = alive || (part.segs[i].info & (F_CT | F_EVC));
                }
                if (alive) ++k;
            } else {
                ++k;
            }
        }
        if ((uint64_t)part.seq.size() >= 0xffffff00ull) {
            cv_set_error("cv_pileup_add_sam: more than 4 Gi query bases in one slice; feed smaller chunks");
            return 1;
        }
        p->pending_cols += part.cols;
        if (!part.segs.empty()) {
            part.reads.clear(); part.reads.shrink_to_fit();
            p->queue.push_back(std::move(part));
        }
    }
    if (kept) *kept = k;
    return 0;
}

// base/offs/n: a run of BAM records as cv_bam_view_records hands them out (record i at base + offs[i], refID
// first); contig_ok: the records' reference is the contig given to cv_pileup_set_contig (candidate pass).
extern "C" int cv_pileup_add_bam(cv_pileup *p, const uint8_t *base, const uint32_t *offs, int64_t n, int contig_ok,
                                 int64_t *kept)
{
    if (!p || n < 0 || (n > 0 && (!base || !offs))) { cv_set_error("cv_pileup_add_bam: bad argument"); return 1; }
    if (kept) *kept = 0;
    if (n == 0) return 0;
    int T = p->threads > 1 && n >= 4096 ? p->threads : 1;
    std::vector<sam_part> parts((size_t)T);
    if (T == 1) {
        parse_bam_range(p, &parts[0], base, offs, 0, n, contig_ok != 0);
    } else {
        // slices of about equal bytes (records differ in length): cut where the byte offset crosses t/T of the span
        std::vector<int64_t> cut((size_t)T + 1);
        cut[0] = 0; cut[(size_t)T] = n;
        const uint64_t span = (uint64_t)(offs[n - 1] - offs[0]) + 1;
        for (int t = 1; t < T; ++t) {
            const uint32_t want = offs[0] + (uint32_t)(span * (uint64_t)t / (uint64_t)T);
            cut[(size_t)t] = std::lower_bound(offs, offs + n, want) - offs;
        }
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back(parse_bam_range, p, &parts[(size_t)t], base, offs, cut[(size_t)t], cut[(size_t)t + 1], contig_ok != 0);
        for (auto &x : th) x.join();
    }
    return absorb_parts(p, parts, kept);
}

extern "C" int64_t cv_pileup_pending(const cv_pileup *p) { return p ? p->pending_cols : 0; }

static int launch_scatter(cv_pileup *p, const dev_batch &b, hipStream_t st)
{
    if (p->n <= 0 || b.nseg == 0) return 0;
    if (p->ev.size() >= 128 && drain(p->ev, p->ms_scatter)) return 1;
    {
        timed t(p->ev, st);
        pileup_scatter<<<(unsigned)((b.nseg + SC_SEGS - 1) / SC_SEGS), 1024, 0, st>>>(
            b.segs, (int64_t)b.nseg, b.seq, p->ref_dev, p->ref_first, p->ref_len, p->cand_dev, (int)p->n, p->bucket_dev,
            p->bucket_lo, p->nb, p->cnt_dev, p->touched_dev, p->left);
    }
    PL_HIP(hipGetLastError());
    p->launches += 1;
    return 0;
}

extern "C" int cv_pileup_flush(cv_pileup *p, void *stream)
{
    if (!p) { cv_set_error("cv_pileup_flush: null handle"); return 1; }
    if (p->queue.empty()) return 0;
    if (!p->ref_dev || (!p->cand_dev && !p->evc)) {
        cv_set_error("cv_pileup_flush: set the reference and the candidates first");
        return 1;
    }
    PL_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    // one device allocation for everything queued; every parser slice is uploaded to its own range (its q0 offsets
    // are relative to its own SEQ bytes: a small kernel rebases them in place), nothing is merged on the host
    size_t ns = 0, nq = 64;                 // a 64-byte tail keeps every lane's read in range
    for (const auto &part : p->queue) { ns += part.segs.size(); nq += (part.seq.size() + 15) / 16 * 16; }
    dev_batch fresh;
    dev_batch &b = p->retain ? fresh : p->work;
    if (ns > b.segs_cap) {
        PL_HIP(hipStreamSynchronize(st));
        hipFree(b.segs); b.segs = nullptr;
        b.segs_cap = p->retain ? ns : ns + ns / 4;
        PL_HIP(hipMalloc(&b.segs, b.segs_cap * sizeof(seg_t)));
    }
    if (nq > b.seq_cap) {
        PL_HIP(hipStreamSynchronize(st));
        hipFree(b.seq); b.seq = nullptr;
        b.seq_cap = p->retain ? nq : nq + nq / 4;
        PL_HIP(hipMalloc(&b.seq, b.seq_cap));
    }
    b.nseg = ns;
    if (p->evc && !p->pos_cnt) {
        const size_t bytes = (size_t)(p->ref_len > 0 ? p->ref_len : 1) * NPOS * sizeof(int32_t);
        PL_HIP(hipMalloc(&p->pos_cnt, bytes));
        PL_HIP(hipMemsetAsync(p->pos_cnt, 0, bytes, st));
    }
    if (nq >= 0xffffff00ull) { cv_set_error("cv_pileup_flush: more than 4 Gi query bases queued; flush more often"); return 1; }
    size_t so = 0, qo = 0;
    for (const auto &part : p->queue) {
        const size_t n1 = part.segs.size();
        PL_HIP(hipMemcpyAsync(b.segs + so, part.segs.data(), n1 * sizeof(seg_t), hipMemcpyHostToDevice, st));
        if (!part.seq.empty()) PL_HIP(hipMemcpyAsync(b.seq + qo, part.seq.data(), part.seq.size(), hipMemcpyHostToDevice, st));
        if (qo) rebase_q0<<<(unsigned)((n1 + 255) / 256), 256, 0, st>>>(b.segs + so, (int64_t)n1, (uint32_t)qo);
        so += n1; qo += (part.seq.size() + 15) / 16 * 16;
    }
    PL_HIP(hipGetLastError());
    if (p->evc) {
        if (p->eve.size() >= 128 && drain(p->eve, p->ms_evc)) return 1;
        {
            timed t(p->eve, st);
            evc_count<<<(unsigned)((ns + EVC_SEGS - 1) / EVC_SEGS), 1024, 0, st>>>(b.segs, (int64_t)ns, b.seq, p->ref_first,
                                                                                   p->ref_len, p->pos_cnt);
        }
        PL_HIP(hipGetLastError());
    }
    if (p->cand_dev && launch_scatter(p, b, st)) return 1;
    if (p->retain) p->kept.push_back(b);
    p->nsegs += (int64_t)ns;
    // pageable copies are staged before hipMemcpyAsync returns only for small sizes: wait for them
    PL_HIP(hipStreamSynchronize(st));
    p->cols += p->pending_cols;
    p->queue.clear(); p->pending_cols = 0;
    return 0;
}

extern "C" int cv_pileup_extract_candidates(cv_pileup *p, double threshold, double min_coverage, int has_region,
                                            int64_t ctg_start, int64_t ctg_end, const int64_t *bed_begin,
                                            const int64_t *bed_end, int64_t nbed, void *stream, int64_t *n_out)
{
    if (!p || !n_out) { cv_set_error("cv_pileup_extract_candidates: null argument"); return 1; }
    if (!p->evc) { cv_set_error("cv_pileup_extract_candidates: option 'evc' was not set"); return 1; }
    if (cv_pileup_flush(p, stream)) return 1;
    PL_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    p->sel_host.clear(); p->sel_counts.clear();
    *n_out = 0;
    if (!p->pos_cnt || p->ref_len <= 0) return 0;         // no read reached the candidate pass
    // BED intervals: sorted, merged -> disjoint (membership is all the reference asks of its interval tree)
    std::vector<std::pair<int64_t, int64_t>> iv;
    for (int64_t i = 0; i < nbed; ++i) if (bed_end[i] > bed_begin[i]) iv.emplace_back(bed_begin[i], bed_end[i]);
    std::sort(iv.begin(), iv.end());
    std::vector<int64_t> bb, be;
    for (auto &x : iv) {
        if (!bb.empty() && x.first <= be.back()) be.back() = std::max(be.back(), x.second);
        else { bb.push_back(x.first); be.push_back(x.second); }
    }
    int64_t *bb_dev = nullptr, *be_dev = nullptr;
    if (nbed >= 0) {
        const size_t nb1 = bb.size() ? bb.size() : 1;
        PL_HIP(hipMalloc(&bb_dev, nb1 * sizeof(int64_t)));
        PL_HIP(hipMalloc(&be_dev, nb1 * sizeof(int64_t)));
        if (!bb.empty()) {
            PL_HIP(hipMemcpy(bb_dev, bb.data(), bb.size() * sizeof(int64_t), hipMemcpyHostToDevice));
            PL_HIP(hipMemcpy(be_dev, be.data(), be.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        }
    }
    evc_pred pred{p->pos_cnt, p->ref_dev, p->ref_first, threshold, min_coverage, has_region, ctg_start, ctg_end,
                  bb_dev, be_dev, nbed >= 0 ? (int)bb.size() : -1};
    const int64_t items = p->ref_len * 2;
    int64_t *sel_dev = nullptr, *nsel_dev = nullptr;
    PL_HIP(hipMalloc(&sel_dev, (size_t)items * sizeof(int64_t)));
    PL_HIP(hipMalloc(&nsel_dev, sizeof(int64_t)));
    int64_t nsel = 0;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    // DeviceSelect takes an int item count: walk the positions in slices
    const int64_t SLICE = 1LL << 30;
    int64_t written = 0;
    for (int64_t off = 0; off < items; off += SLICE) {
        const int cnt = (int)std::min<int64_t>(SLICE, items - off);
        hipcub::CountingInputIterator<int64_t> it(off);
        size_t need = 0;
        PL_HIP(hipcub::DeviceSelect::If(nullptr, need, it, sel_dev + written, nsel_dev, cnt, pred, st));
        if (need > tmp_bytes) { hipFree(tmp); tmp = nullptr; PL_HIP(hipMalloc(&tmp, need)); tmp_bytes = need; }
        {
            timed t(p->eve, st);
            PL_HIP(hipcub::DeviceSelect::If(tmp, need, it, sel_dev + written, nsel_dev, cnt, pred, st));
        }
        PL_HIP(hipMemcpyAsync(&nsel, nsel_dev, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PL_HIP(hipStreamSynchronize(st));
        written += nsel;
    }
    p->sel_host.resize((size_t)written);
    p->sel_counts.resize((size_t)written * 7);
    if (written) {
        int32_t *c7 = nullptr;
        PL_HIP(hipMalloc(&c7, (size_t)written * 7 * sizeof(int32_t)));
        evc_gather<<<(unsigned)((written + 255) / 256), 256, 0, st>>>(sel_dev, written, p->pos_cnt, c7);
        PL_HIP(hipGetLastError());
        PL_HIP(hipMemcpyAsync(p->sel_host.data(), sel_dev, (size_t)written * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        PL_HIP(hipMemcpyAsync(p->sel_counts.data(), c7, (size_t)written * 7 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        PL_HIP(hipStreamSynchronize(st));
        hipFree(c7);
    }
    hipFree(tmp); hipFree(sel_dev); hipFree(nsel_dev); hipFree(bb_dev); hipFree(be_dev);
    *n_out = written;
    return 0;
}

extern "C" int cv_pileup_get_extracted(cv_pileup *p, int64_t *pos0, int32_t *late, int32_t *counts7,
                                       int64_t info[2])
{
    if (!p) { cv_set_error("cv_pileup_get_extracted: null handle"); return 1; }
    for (size_t i = 0; i < p->sel_host.size(); ++i) {
        if (pos0) pos0[i] = p->ref_first + (p->sel_host[i] >> 1);
        if (late) late[i] = (int32_t)(p->sel_host[i] & 1);
    }
    if (counts7 && !p->sel_counts.empty()) memcpy(counts7, p->sel_counts.data(), p->sel_counts.size() * sizeof(int32_t));
    if (info) { info[0] = p->evc_reads; info[1] = p->evc_prev_pos; }
    return 0;
}

extern "C" int cv_pileup_adopt_candidates(cv_pileup *p, int has_range, int64_t lo1, int64_t hi1, void *stream,
                                          int64_t *n_out)
{
    if (!p) { cv_set_error("cv_pileup_adopt_candidates: null handle"); return 1; }
    if (!p->retain) { cv_set_error("cv_pileup_adopt_candidates: option 'retain' was not set"); return 1; }
    if (cv_pileup_flush(p, stream)) return 1;
    PL_HIP(hipSetDevice(p->device));
    std::vector<int32_t> c32;
    for (size_t i = 0; i < p->sel_host.size(); ++i) {
        const int64_t c = p->ref_first + (p->sel_host[i] >> 1) + 1;          // the row prints pos+1 (:38)
        if (has_range && (c < lo1 || c > hi1)) continue;                      // CreateTensor.py:60-61
        if (!c32.empty() && c32.back() == (int32_t)c) continue;               // regular + late entry of one position
        c32.push_back((int32_t)c);
    }
    if (install_candidates(p, c32)) return 1;
    hipStream_t st = (hipStream_t)stream;
    for (const auto &b : p->kept) if (launch_scatter(p, b, st)) return 1;
    if (n_out) *n_out = (int64_t)c32.size();
    return 0;
}

extern "C" int cv_pileup_recount(cv_pileup *p, void *stream)
{
    if (!p) { cv_set_error("cv_pileup_recount: null handle"); return 1; }
    if (!p->evc || !p->retain) { cv_set_error("cv_pileup_recount: needs options 'evc' and 'retain'"); return 1; }
    if (cv_pileup_flush(p, stream)) return 1;
    if (!p->pos_cnt) return 0;
    PL_HIP(hipSetDevice(p->device));
    hipStream_t st = (hipStream_t)stream;
    if (p->eve.size() >= 128 && drain(p->eve, p->ms_evc)) return 1;
    PL_HIP(hipMemsetAsync(p->pos_cnt, 0, (size_t)p->ref_len * NPOS * sizeof(int32_t), st));
    for (const auto &b : p->kept) {
        timed t(p->eve, st);
        evc_count<<<(unsigned)((b.nseg + EVC_SEGS - 1) / EVC_SEGS), 1024, 0, st>>>(b.segs, (int64_t)b.nseg, b.seq,
                                                                                   p->ref_first, p->ref_len, p->pos_cnt);
    }
    PL_HIP(hipGetLastError());
    return 0;
}

extern "C" int cv_pileup_get_candidates(cv_pileup *p, int64_t *centers, int64_t cap, int64_t *n_out)
{
    if (!p) { cv_set_error("cv_pileup_get_candidates: null handle"); return 1; }
    if (n_out) *n_out = p->n;
    if (centers && p->n > 0) {
        if (cap < p->n) { cv_set_error("cv_pileup_get_candidates: buffer holds %lld, need %lld", (long long)cap, (long long)p->n); return 1; }
        std::vector<int32_t> c32((size_t)p->n);
        PL_HIP(hipSetDevice(p->device));
        PL_HIP(hipMemcpy(c32.data(), p->cand_dev, (size_t)p->n * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < p->n; ++i) centers[i] = c32[(size_t)i];
    }
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
        timed t(p->evf, st);
        const int64_t total = p->n * WIDTH;
        pileup_finalize<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(p->cnt_dev, p->cand_dev, p->n, p->ref_dev,
                                                                           p->ref_first, p->ref_len, tensors_dev,
                                                                           depth_dev, subtract);
    }
    PL_HIP(hipGetLastError());
    if (touched_dev)
        PL_HIP(hipMemcpyAsync(touched_dev, p->touched_dev, (size_t)p->n, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int cv_pileup_stats(cv_pileup *p, float ms[3], int64_t counts[3])
{
    if (!p) { cv_set_error("cv_pileup_stats: null handle"); return 1; }
    PL_HIP(hipSetDevice(p->device));
    if (drain(p->ev, p->ms_scatter) || drain(p->evf, p->ms_final) || drain(p->eve, p->ms_evc)) return 1;
    if (ms) { ms[0] = p->ms_scatter; ms[1] = p->ms_final; ms[2] = p->ms_evc; }
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
        if (w + 16 > cap) return -1;              // (only after values that print longer than the 16 bytes budgeted each)
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
            const int r = snprintf(dst + w, (size_t)(cap - w), "%0.1f", (double)v);
            if (r < 0 || r >= cap - w) return -1;  // "%0.1f" of 3e38 is 41 characters
            w += r;
        }
    }
    if (w >= cap) return -1;
    dst[w] = 0;
    return w;
}
