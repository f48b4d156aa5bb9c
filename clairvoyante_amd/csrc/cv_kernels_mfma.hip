// cv_kernels_mfma.hip -- gfx950 tile kernels of the forward path (impl 1).
//
// All contractions run on v_mfma_f32_16x16x4_f32 (exact fp32, bit-for-bit an
// ascending-k fmaf chain) in the TRANSPOSED form  D[feature][candidate] =
// W^T[feature][k] * act^T[k][candidate]:
//   * the B operand is a tile-major activation fragment (cv_internal.hpp): one
//     coalesced 16-byte load per lane = the operands of four MFMA steps;
//   * the A operand is a pre-packed weight fragment (same 1 KiB shape, rows
//     permuted by sigma) read from LDS with one conflict-free ds_read_b128;
//   * the D registers of a lane ARE the next layer's fragment, so bias + SELU +
//     max-pool run in registers and the result leaves with one coalesced
//     16-byte store per lane.  No transposes, no shuffles, no atomics.
// A wave owns 16 candidates and streams over the 33 pileup positions, keeping the
// kh-row window and the pooling window in registers (max-pool over positions is
// an element-wise max of successive accumulator tiles).
//
// Layers: /root/reference/clairvoyante/clairvoyante_v3.py:54-121 (and
// clairvoyante_v3_slim.py:53-101).  Padding: SAME, kw = 4 -> 1 left / 2 right,
// kh -> (kh-1)/2 on top; taps on padding are skipped (they add an exact zero).
#define CV_TILE_STAMPS          // the development wave stamps (cv_tile.hpp) are stored in this translation unit
#include "cv_tile.hpp"

namespace {

// ---------------------------------------------------------------------------
// weight packing (runs once per parameter change)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pack_conv1(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int cout)
{
    if (t >= 4 * 64) return;
    int lane = t & 63, kw = t >> 6;
    int i = lane & 15, kq = lane >> 4;
    int co = cv_sigma(i);
    wp[t] = co < cout ? w[((size_t)kw * 4 + kq) * cout + co] : 0.0f;
}

__device__ __forceinline__ void pack_conv(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int KH, int cin, int cout,
                          int CINB, int NT)
{
    int64_t total = (int64_t)NT * KH * 4 * CINB * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int frag = (int)(t >> 8);
    int cb = frag % CINB; frag /= CINB;
    int kw = frag % 4; frag /= 4;
    int kh = frag % KH;
    int nt = frag / KH;
    int i = lane & 15, kq = lane >> 4;
    int ci = 16 * cb + 4 * s + kq, co = 16 * nt + cv_sigma(i);
    wp[t] = (ci < cin && co < cout) ? w[(((size_t)kh * 4 + kw) * cin + ci) * cout + co] : 0.0f;
}

// dense [K][N] -> [kb][ob][lane][s];  input feature k = 16*kb + 4*s + kq.  NBP >= ceil(N/16)
// fragments per k step (pad fragments are zero: see dense_tm)
__device__ __forceinline__ void pack_dense(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int K, int N, int KB, int NBP)
{
    int64_t total = (int64_t)KB * NBP * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int ob = (int)(frag % NBP);
    int kb = (int)(frag / NBP);
    int i = lane & 15, kq = lane >> 4;
    int k = 16 * kb + 4 * s + kq, o = 16 * ob + cv_sigma(i);
    wp[t] = (k < K && o < N) ? w[(size_t)k * N + o] : 0.0f;
}

// forward weights of a small dense layer in PAIRS of k fragments: [kp][2 x NBH][lane][s] -- one 2*NBH-fragment stage of
// the LDS ring carries two k steps (dense_tm EPI 3 streams fc5 that way on fc4's ring: half the barriers)
__device__ __forceinline__ void pack_dense_kpairs(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int K, int N, int KB, int NBH,
                                                  int KS)
{
    const int KP = (KB + KS - 1) / KS;             // stages of KS k fragments x NBH output fragments
    int64_t total = (int64_t)KP * KS * NBH * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int f = (int)(frag % (KS * NBH));
    int kp = (int)(frag / (KS * NBH));
    int kb = KS * kp + f / NBH, ob = f % NBH;
    int i = lane & 15, kq = lane >> 4;
    int k = 16 * kb + 4 * s + kq, o = 16 * ob + cv_sigma(i);
    wp[t] = (kb < KB && k < K && o < N) ? w[(size_t)k * N + o] : 0.0f;
}

// forward weights of a dense layer in NSLAB slabs of NBS output fragments (padded to NBSP per k step):
// [slab][kb][NBSP][lane][s] -- small batches run one workgroup per (group block, slab), so that a layer
// with few groups still fills the chip
__device__ __forceinline__ void pack_dense_slabs(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int K, int N, int KB, int NBS,
                                 int NBSP, int NSLAB)
{
    int64_t total = (int64_t)NSLAB * KB * NBSP * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int obs = (int)(frag % NBSP); frag /= NBSP;
    int kb = (int)(frag % KB);
    int slab = (int)(frag / KB);
    int i = lane & 15, kq = lane >> 4;
    int k = 16 * kb + 4 * s + kq, o = 16 * (slab * NBS + obs) + cv_sigma(i);
    wp[t] = (obs < NBS && k < K && o < N) ? w[(size_t)k * N + o] : 0.0f;
}

// data-gradient weights of a dense layer: out feature = original input k, in feature =
// original output j; fragments [slab][jb][ob_in_slab][lane][s] (slabs of NBS output fragments, stored with a stride
// of NBSP >= NBS: dense_tm's count padded to its waves, the pad fragments zero)
__device__ __forceinline__ void pack_dense_dgrad(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int K, int N, int JB,
                                 int NBS, int NSLAB, int NBSP)
{
    int64_t total = (int64_t)NSLAB * JB * NBSP * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int obs = (int)(frag % NBSP); frag /= NBSP;
    int jb = (int)(frag % JB);
    int slab = (int)(frag / JB);
    int i = lane & 15, kq = lane >> 4;
    int j = 16 * jb + 4 * s + kq;                       // contraction index = original output unit
    int k = 16 * (slab * NBS + obs) + cv_sigma(i);      // result feature = original input unit
    wp[t] = (obs < NBS && j < N && k < K) ? w[(size_t)k * N + j] : 0.0f;
}

// data-gradient weights of fc4 for dense_dgrad_unpool: one column of the pooled conv3 map per workgroup, rows in
// sequence.  Result fragment f = r * NCOL + col (r = pooled row, col = base * tiles + tile) holds input units
// 16 f + sigma(i); fragments [col][r][jb (padded to JBP)][lane][s], contraction index j = 16 jb + 4 s + kq.
__device__ __forceinline__ void pack_dense_dgrad_rows(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int K, int N,
                                                      int JB, int JBP, int NCOL, int HO)
{
    int64_t total = (int64_t)NCOL * HO * JBP * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int jb = (int)(frag % JBP); frag /= JBP;
    int r = (int)(frag % HO);
    int col = (int)(frag / HO);
    int i = lane & 15, kq = lane >> 4;
    int j = 16 * jb + 4 * s + kq;
    int k = 16 * (r * NCOL + col) + cv_sigma(i);
    wp[t] = (jb < JB && j < N && k < K) ? w[(size_t)k * N + j] : 0.0f;
}

// data-gradient weights of a conv layer (see conv_tm MODE 2): flipped taps, channels swapped
//   Wd[nt'][kh'][kw'][cb'][lane][s] = W[KH-1-kh'][3-kw'][ci = 16 nt' + sigma(i)][co = 16 cb' + 4 s + kq]
__device__ __forceinline__ void pack_conv_dgrad(int64_t t, const float *__restrict__ w, float *__restrict__ wp, int KH, int cin, int cout,
                                int COB, int CIT)
{
    int64_t total = (int64_t)CIT * KH * 4 * COB * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int frag = (int)(t >> 8);
    int cb = frag % COB; frag /= COB;
    int kw = frag % 4; frag /= 4;
    int kh = frag % KH;
    int nt = frag / KH;
    int i = lane & 15, kq = lane >> 4;
    int co = 16 * cb + 4 * s + kq, ci = 16 * nt + cv_sigma(i);
    wp[t] = (ci < cin && co < cout) ? w[(((size_t)(KH - 1 - kh) * 4 + (3 - kw)) * cin + ci) * cout + co] : 0.0f;
}

// alpha-dropout on fc4 in TM layout
__global__ void dropout_tm(const float *__restrict__ h4, float *__restrict__ d4, float *__restrict__ amask,
                           int NB, int nunits, int64_t G, float rate, uint64_t seed, uint64_t step, int64_t cand0)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * NB * 256) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int ob = (int)(frag % NB);
    int64_t g = frag / NB;
    int c = lane & 15, kq = lane >> 4;
    int unit = 16 * ob + 4 * s + kq;
    float v = h4[t], mk;
    dropout_value(v, mk, unit, nunits, cand0 + g * 16 + c, rate, seed, step);
    d4[t] = v;
    amask[t] = mk;
}


// ---------------------------------------------------------------------------
// Training forward: which row of its window a pooled value came from.
// The backward pass routes the gradient of a pooled value to the FIRST maximum of its window (the canonical rule of this build, pool
// backward; the reference's tf.layers.max_pooling2d gradient).  Instead of keeping the pre-pool activations for that
// (0.68 MB per group of 16 candidates), the forward kernels record the window offset d of the first maximum: 4 bits per
// value, the 16 values a lane holds of a pooled row (4 bases x 4 registers) in one 64-bit word -- value (w, r) at bits
// 4 (4 w + r) .. +3 -- stored as [group][pooled row][tile][lane] (512 B per row and tile instead of 4 KiB).
// ---------------------------------------------------------------------------

// rows[0..P-2] = the P-1 older activated rows of the window (oldest first), v = the newest, o = their maximum
template <int P>
__device__ __forceinline__ unsigned pool_code4(const f4 (&older)[P > 1 ? P - 1 : 1], f4 v, f4 o)
{
    unsigned c = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int idx = P - 1;
#pragma unroll
        for (int d = P - 2; d >= 0; d--) idx = older[d][r] == o[r] ? d : idx;      // the lowest offset that holds the maximum
        (void)v;
        c |= (unsigned)idx << (4 * r);
    }
    return c;
}

// ---------------------------------------------------------------------------
// conv1 (k(1,4), cin 4) + SELU + max-pool(POOL,1): raw X [n,33,4,4] -> TM
// One wave per group of 16 candidates; per position 12 MFMA steps (K = 4 each).
// ---------------------------------------------------------------------------
template <int POOL, bool SAVE = false>
__global__ __launch_bounds__(256) void conv1_tm(const float *__restrict__ x, int64_t n,
                                                 const float *__restrict__ wp1,
                                                 const float *__restrict__ bias, int cout,
                                                 f4 *__restrict__ out_tm, int G, u32x2 *__restrict__ code_tm = nullptr)
{
    // SAVE (training forward): the window offset of every pooled value's first maximum goes to code_tm (pool_code4).
    // The layer has almost no arithmetic (12 MFMA steps per position) and a long dependent
    // chain per position (load -> MFMA -> SELU -> pool -> store), so it is latency-bound:
    // SPLIT waves share a group, each producing a contiguous range of pooled rows (and
    // recomputing the POOL-1 conv rows of overlap) -- 4x the waves in flight.
    constexpr int HIN = CV_INPUT_H, HOUT = HIN - POOL + 1, SPLIT = 4;
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = wv / SPLIT, part = wv % SPLIT;
    if (g >= G) return;
    const int r0 = (HOUT * part) / SPLIT, r1 = (HOUT * (part + 1)) / SPLIT;   // pooled rows [r0, r1)
    const int c = lane & 15, q = lane >> 4;
    int64_t cand = (int64_t)g * 16 + c;
    if (cand >= n) cand = n - 1;
    const float *xp = x + (size_t)cand * (HIN * 16) + q;   // B operand: lane (c, ci = q)
    float A[4];
#pragma unroll
    for (int kw = 0; kw < 4; kw++) A[kw] = wp1[kw * 64 + lane];
    const f4 b4 = load_bias4(bias, 0, q, cout);
    f4 pw[POOL > 1 ? POOL - 1 : 1][4];
#pragma unroll
    for (int j = 0; j < (POOL > 1 ? POOL - 1 : 1); j++)
#pragma unroll
        for (int w = 0; w < 4; w++) pw[j][w] = (f4){0.f, 0.f, 0.f, 0.f};
    float xw[4], xn[4];
#pragma unroll
    for (int w = 0; w < 4; w++) xw[w] = xp[r0 * 16 + w * 4];
    f4 *op = out_tm + (size_t)g * HOUT * 4 * 64 + lane;
    const int hend = r1 + POOL - 1;          // conv rows r0 .. r1+POOL-2
#pragma unroll 1
    for (int h = r0; h < hend; h++) {
        {
            const int hn = h + 1 < hend ? h + 1 : h;
#pragma unroll
            for (int w = 0; w < 4; w++) xn[w] = xp[hn * 16 + w * 4];
        }
        f4 acc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) acc[w] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kw = 0; kw < 4; kw++)
#pragma unroll
            for (int wo = 0; wo < 4; wo++) {
                const int wi = wo + kw - 1;
                if (wi < 0 || wi > 3) continue;
                acc[wo] = mfma4(A[kw], xw[wi], acc[wo]);
            }
        if constexpr (!SAVE && POOL > 1) {
            // inference: max-pool the PRE-activations (running maxima pw[j] = max of the last j+1 rows) and apply
            // SELU once per pooled row: SELU is monotone over all of fp32 (cv_selu_sweep), hence
            // max_j selu(a_j + b) == selu(max_j (a_j + b)) bit for bit
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f4 t = acc[w] + b4;      // (the sum, unlike a raw MFMA result, needs no canonicalising v_max x, x, x)
                const f4 o = max4(pw[POOL - 2][w], t);
#pragma unroll
                for (int j = POOL - 2; j > 0; j--) pw[j][w] = max4(pw[j - 1][w], t);
                pw[0][w] = t;
                if (h - r0 >= POOL - 1) op[(size_t)((h - (POOL - 1)) * 4 + w) * 64] = selu4(o);
            }
#pragma unroll
            for (int w = 0; w < 4; w++) xw[w] = xn[w];
            continue;
        }
        f4 v[4];
#pragma unroll
        for (int w = 0; w < 4; w++) v[w] = selu4(acc[w] + b4);
        if constexpr (POOL > 1) {
            f4 o[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                o[w] = v[w];
#pragma unroll
                for (int j = 0; j < POOL - 1; j++) o[w] = max4(o[w], pw[j][w]);
            }
            if constexpr (SAVE) {
                if (h - r0 >= POOL - 1) {
                    unsigned cw[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        f4 older[POOL - 1];
#pragma unroll
                        for (int j = 0; j < POOL - 1; j++) older[j] = pw[j][w];
                        cw[w] = pool_code4<POOL>(older, v[w], o[w]);
                    }
                    code_tm[((size_t)g * HOUT + (h - (POOL - 1))) * 64 + lane] = (u32x2){cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16)};
                }
            }
#pragma unroll
            for (int j = 0; j + 1 < POOL - 1; j++)
#pragma unroll
                for (int w = 0; w < 4; w++) pw[j][w] = pw[j + 1][w];
#pragma unroll
            for (int w = 0; w < 4; w++) pw[POOL - 2][w] = v[w];
            if (h - r0 >= POOL - 1) {
#pragma unroll
                for (int w = 0; w < 4; w++) op[(size_t)((h - (POOL - 1)) * 4 + w) * 64] = o[w];
            }
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) op[(size_t)(h * 4 + w) * 64] = v[w];
        }
#pragma unroll
        for (int w = 0; w < 4; w++) xw[w] = xn[w];
    }
}

// ---------------------------------------------------------------------------
// generic conv (k(KH,4), CINB*16 -> NT*16 channels) + SELU + max-pool(POOL,1) -> TM.
// One wave per (group, output tile nt); weights of the whole layer sit in LDS (loaded
// once per workgroup).  Per position and wave: KH*4*CINB ds_read_b128 feed
// KH*12*CINB*4 MFMA steps.
// Input rows come from a row source:
//   FRONT == 0: a TM buffer (one coalesced 16-byte load per fragment), or
//   FRONT  > 0: the raw pileup tensor X [n,33,4,4] pushed through conv1 k(1,4) + SELU +
//               max-pool(FRONT,1) on the fly (12 extra MFMA steps per position), so the
//               first layer never round-trips through HBM (CINB must be 1, HIN = 34-FRONT).
// ---------------------------------------------------------------------------
// selu4 of a fragment whose registers 2, 3 are channel padding (a layer of <= 8 channels in a 16-wide tile: weights
// and bias of those channels are zero, so the pre-activation is +0 and selu(+0) = +0): two activations, not four
__device__ __forceinline__ f4 selu4_low(f4 v)
{
    const cvm::f2v a = cvm::selu2((cvm::f2v){v[0], v[1]});
    return (f4){a[0], a[1], 0.0f, 0.0f};
}

// HALF: the first layer has <= 8 output channels (slim)
template <int FRONT, bool HALF = false>
struct front_source {
    static constexpr int NP = FRONT > 1 ? FRONT - 1 : 1;
    const float *xp;       // lane (c, ci = q): &X[cand][0][0][ci]
    float A[4];            // conv1 weight fragments, one MFMA step per kw
    f4 b4;
    f4 cw[NP][4];          // previous conv1 rows (after SELU) of the pooling window
    float xc[4], xn[4];    // current / prefetched input row
    int hx;                // next conv1 row

    __device__ __forceinline__ void load_x(float (&dst)[4], int h)
    {
        if (h < CV_INPUT_H) {
#pragma unroll
            for (int w = 0; w < 4; w++) dst[w] = xp[h * 16 + w * 4];
        }
    }
    __device__ __forceinline__ void conv1_row(f4 (&v)[4])
    {
        f4 acc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) acc[w] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kw = 0; kw < 4; kw++)
#pragma unroll
            for (int wo = 0; wo < 4; wo++) {
                const int wi = wo + kw - 1;
                if (wi < 0 || wi > 3) continue;
                acc[wo] = mfma4(A[kw], xc[wi], acc[wo]);
            }
        // pre-activations: SELU is applied to the pooled row (next()), see conv1_tm
#pragma unroll
        for (int w = 0; w < 4; w++) v[w] = acc[w] + b4;
#pragma unroll
        for (int w = 0; w < 4; w++) xc[w] = xn[w];
        hx++;
        load_x(xn, hx + 1);
    }
    // h0: the first pooled row next() will be asked for (a position part of the layer above starts there)
    __device__ __forceinline__ void init(const float *x, int64_t cand, int q, const float *wp1,
                                         const float *bias1, int cout1, int lane, int h0 = 0)
    {
        xp = x + (size_t)cand * (CV_INPUT_H * 16) + q;
#pragma unroll
        for (int kw = 0; kw < 4; kw++) A[kw] = wp1[kw * 64 + lane];
        b4 = load_bias4(bias1, 0, q, cout1);
        hx = h0;
        load_x(xc, h0);
        load_x(xn, h0 + 1);
        if constexpr (FRONT > 1) {
            // cw[j] = running maximum of the last j+1 pre-activation rows
#pragma unroll
            for (int r = 0; r < FRONT - 1; r++) {
                f4 v[4];
                conv1_row(v);
#pragma unroll
                for (int w = 0; w < 4; w++) {
#pragma unroll
                    for (int j = FRONT - 2; j > 0; j--) cw[j][w] = r == 0 ? v[w] : max4(cw[j - 1][w], v[w]);
                    cw[0][w] = v[w];
                }
            }
        }
    }
    // next pooled conv1 row (rows are requested in ascending order): max over the window of pre-activation rows,
    // then SELU once (monotone: same bits as pooling the activated rows)
    __device__ __forceinline__ void next(f4 (&row)[4][1])
    {
        f4 v[4];
        conv1_row(v);
        if constexpr (FRONT > 1) {
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f4 o = max4(cw[FRONT - 2][w], v[w]);
#pragma unroll
                for (int j = FRONT - 2; j > 0; j--) cw[j][w] = max4(cw[j - 1][w], v[w]);
                cw[0][w] = v[w];
                row[w][0] = HALF ? selu4_low(o) : selu4(o);
            }
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) row[w][0] = HALF ? selu4_low(v[w]) : selu4(v[w]);
        }
    }
};

// MODE 0: inference forward.  MODE 1: training forward -- additionally records, per pooled value, the window
// offset of its first maximum (act_tm viewed as [g][HOUT][NT][64] 64-bit code words, pool_code4) that the backward
// pass routes the pooling gradient with; nothing extra without pooling.  MODE 2: data-gradient pass -- the same kernel run as the transposed
// convolution  gIn[h][w][ci] = sum g[h-kh+pt][w-kw+1][co] W[kh][kw][ci][co]  on flipped,
// in/out-swapped packed weights (pack_conv_dgrad): padding 2 left / 1 right and
// KH-1-(KH-1)/2 on top, no bias, no activation, no pooling.
// HSPLIT > 1 (no fused first layer): HSPLIT waves share a (group, tile), each producing a contiguous range of
// (pooled) positions -- more waves in flight when a batch has few groups.
// KS4: MFMA steps per 16-channel input fragment (4 channels each); a layer whose last fragment holds fewer than 16
// real channels (slim conv2: 8) skips the steps that would multiply the zero padding -- they add an exact +0.
// (waves per group that do not make whole 4-wave workgroups, or more than one: see the XCD-aware numbering in the kernel)
#define CV_CONV_XCD_UNITS(W) ((W) != 1 && (W) != 2 && (W) != 4)
template <int KH, int CINB, int NT, int POOL, int HIN, int FRONT, int MODE, int HSPLIT = 1, int KS4 = 4>
__global__ __launch_bounds__(256, 2) void conv_tm(const f4 *__restrict__ in_tm, const float *__restrict__ x,
                                                   int64_t n, const float *__restrict__ wp1,
                                                   const float *__restrict__ bias1, int cout1,
                                                   const f4 *__restrict__ wp, const float *__restrict__ bias,
                                                   int cout, f4 *__restrict__ out_tm, f4 *__restrict__ act_tm, int G,
                                                   int rows_per = 0)
{
    static_assert(FRONT == 0 || CINB == 1, "the fused first layer feeds one 16-channel fragment");
    static_assert(MODE != 2 || (POOL == 1 && FRONT == 0), "the data-gradient pass has no pooling / first layer");
    static_assert(HSPLIT >= 1 || FRONT == 0, "flat ranges read their rows from a TM buffer (position parts may make them: front_source::init h0)");
    static_assert(HSPLIT != 0 || (FRONT == 0 && (MODE != 0 || POOL == 1)), "flat ranges: training kernels, and inference layers without pooling (slim small passes)");
    extern __shared__ __attribute__((aligned(16))) f4 ldsw[];
    constexpr int PADT = MODE == 2 ? KH - 1 - (KH - 1) / 2 : (KH - 1) / 2;
    constexpr int PADL = MODE == 2 ? 2 : 1;
    constexpr int HOUT = HIN - POOL + 1;
    constexpr int NFRAG = NT * KH * 4 * CINB;
    for (int i = threadIdx.x; i < NFRAG * 64; i += 256) ldsw[i] = wp[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if constexpr (HSPLIT >= 1 && CV_CONV_XCD_UNITS(NT * HSPLIT)) {
        // XCD-aware numbering (workgroup b runs on XCD b % 8, each XCD has its own L2): the NT x HSPLIT waves of a group read
        // the same input rows (tiles) or overlapping ones (parts); when they do not fill whole workgroups they sit in
        // consecutive workgroups = different XCDs, and the group's rows come in from HBM once per XCD (conv2's training
        // forward at 625 groups, 2 tiles x 3 parts: 318 MB per launch for a 74 MB input and 150 MB of output).  Here XCD x
        // owns the groups g = x (mod 8), as in conv3_rot; launch_conv pads the grid to whole XCD rows.  Speed only.
        if (gridDim.x >= 16) {
            const int x = blockIdx.x & 7;
            const int lw = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * 4 + (threadIdx.x >> 6));
            wv = ((lw / (NT * HSPLIT)) * 8 + x) * (NT * HSPLIT) + lw % (NT * HSPLIT);
        }
    }
    // HSPLIT >= 1: a wave owns part hs of the positions of ONE (group, tile).
    // HSPLIT == 0 (training, larger batches): a wave owns the output rows [r0, r1) of the flat (group, row) sequence
    // of its tile -- rows_per of them, whatever the group boundaries -- so that a launch is ONE round of equal waves
    // (parts of whole groups gave 3 750 waves for 2 048 slots at train.py's batch: a second round on a third of the
    // chip).  The rows of each group in the range are one segment of the loop below; same values row for row.
    constexpr int HSD = HSPLIT > 0 ? HSPLIT : 1;
    int nt, gF, gL, hs = 0, r0 = 0, r1 = 0;
    if constexpr (HSPLIT == 0) {
        nt = wv % NT;
        r0 = (wv / NT) * rows_per;
        r1 = r0 + rows_per < G * HOUT ? r0 + rows_per : G * HOUT;
        if (r0 >= r1) return;
        gF = r0 / HOUT; gL = (r1 - 1) / HOUT;
    } else {
        const int gt = wv / HSD;
        hs = wv % HSD; nt = gt % NT; gF = gL = gt / NT;
        if (gF >= G) return;
    }
    CV_STAMP_BEGIN
    const int q = lane >> 4;
    const f4 b4 = MODE == 2 ? (f4){0.f, 0.f, 0.f, 0.f} : load_bias4(bias, nt, q, cout);
    const f4 *wl = ldsw + (size_t)nt * (KH * 4 * CINB * 64) + lane;
#pragma unroll 1
    for (int g = gF; g <= gL; g++) {
    // positions [hbeg, hend): with pooling a part owns the pooled rows [HOUT*hs/HSPLIT, HOUT*(hs+1)/HSPLIT) and
    // computes the POOL-1 convolution rows behind them as well (recomputed by its neighbour: same values)
    int hbeg, hend;
    if constexpr (HSPLIT == 0) {
        const int oa = r0 - g * HOUT > 0 ? r0 - g * HOUT : 0, ob = r1 - g * HOUT < HOUT ? r1 - g * HOUT : HOUT;
        hbeg = oa; hend = POOL > 1 ? ob + POOL - 1 : ob;
    } else {
        hbeg = POOL > 1 ? HOUT * hs / HSD : HIN * hs / HSD;
        hend = POOL > 1 ? HOUT * (hs + 1) / HSD + POOL - 1 : HIN * (hs + 1) / HSD;
    }
    const f4 *inp = in_tm + (size_t)g * (HIN * 4 * CINB * 64) + lane;
    f4 *op = out_tm + (size_t)g * (HOUT * 4 * NT * 64) + (size_t)nt * 64 + lane;

    front_source<FRONT, (KS4 <= 2)> fs;          // KS4 <= 2: the fused first layer has <= 8 output channels
    if constexpr (FRONT > 0) {
        int64_t cand = (int64_t)g * 16 + (lane & 15);
        if (cand >= n) cand = n - 1;
        fs.init(x, cand, q, wp1, bias1, cout1, lane, hbeg - PADT > 0 ? hbeg - PADT : 0);
    }
    auto fetch_row = [&](int hr, f4 (&row)[4][CINB]) {     // rows are requested in ascending order
        if constexpr (FRONT > 0) {
            fs.next(row);
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int cb = 0; cb < CINB; cb++) row[w][cb] = inp[(size_t)((hr * 4 + w) * CINB + cb) * 64];
        }
    };
    // Without a fused first layer the row the NEXT position needs is loaded one fragment at a time between the MFMA
    // blocks of the first kernel row, from asm (scalar base + this lane's 16 bytes) so that the loads stay where they
    // are put: as a burst of 4 CINB loads at the top of the position the wave sits in the CU's vector-memory queue
    // behind the bursts of the other seven waves and multiplies nothing meanwhile (measured on wgrad_conv_cm).
    constexpr bool SPREAD = FRONT == 0;
    static_assert(CINB <= 3, "the counted wait below names at most 12 fragments");
    const f4 *const inp_s = in_tm + (size_t)__builtin_amdgcn_readfirstlane(g) * (HIN * 4 * CINB * 64);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto load_piece = [&](int hr, int w, int cb, f4 &dst) {
        const f4 *ps = inp_s + (size_t)((hr * 4 + w) * CINB + cb) * 64;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane16), "s"(ps));    // (no memory clobber: the
    };                                                       //  weight reads from LDS may move across it)

    f4 win[KH][4][CINB];   // win[kh] = input row h + kh - PADT
    f4 nxt[4][CINB];
    f4 pw[POOL > 1 ? POOL - 1 : 1][4];
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < (POOL > 1 ? POOL - 1 : 1); j++)
#pragma unroll
        for (int w = 0; w < 4; w++) pw[j][w] = zero;
    // prologue: rows -PADT .. KH-2-PADT -> win[0..KH-2]; row KH-1-PADT -> nxt
#pragma unroll
    for (int j = 0; j < KH; j++) {
        const int hr = hbeg + j - PADT;
        f4 tmp[4][CINB];
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) tmp[w][cb] = zero;
        if (hr >= 0 && hr < HIN) fetch_row(hr, tmp);
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) {
                if (j < KH - 1) win[j][w][cb] = tmp[w][cb]; else nxt[w][cb] = tmp[w][cb];
            }
    }
#pragma unroll 1
    for (int h = hbeg; h < hend; h++) {
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) win[KH - 1][w][cb] = nxt[w][cb];
        {   // fetch / produce the row the next position needs
            const int hr = h + 1 + (KH - 1) - PADT;
            if (!SPREAD && hr < HIN) fetch_row(hr, nxt);
        }
        // (past the last row the loads re-read it -- nobody uses the result: no branch around every load)
        const int hnext = h + 1 + (KH - 1) - PADT < HIN ? h + 1 + (KH - 1) - PADT : HIN - 1;
        f4 acc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) acc[w] = zero;
#pragma unroll
        for (int kh = 0; kh < KH; kh++) {
            const int hr = h + kh - PADT;
            const bool on = hr >= 0 && hr < HIN;     // wave-uniform; SAME padding rows are skipped
            if (on) {                                // (one branch per kernel row, not per block: the weight reads
#pragma unroll                                       //  run ahead of their MFMAs only inside a basic block)
                for (int kw = 0; kw < 4; kw++)
#pragma unroll
                    for (int cb = 0; cb < CINB; cb++) {
                        const f4 A = wl[(size_t)((kh * 4 + kw) * CINB + cb) * 64];
#pragma unroll
                        for (int s = 0; s < KS4; s++)
#pragma unroll
                            for (int wo = 0; wo < 4; wo++) {
                                const int wi = wo + kw - PADL;
                                if (wi < 0 || wi > 3) continue;
                                acc[wo] = mfma4(A[s], win[kh][wi][cb][s], acc[wo]);
                            }
                        if constexpr (SPREAD) {
                            if (kh == 0) load_piece(hnext, kw, cb, nxt[kw][cb]);
                        }
                    }
            } else if (SPREAD && kh == 0) {          // padding row on top: nothing to hide the loads under
#pragma unroll
                for (int kw = 0; kw < 4; kw++)
#pragma unroll
                    for (int cb = 0; cb < CINB; cb++) load_piece(hnext, kw, cb, nxt[kw][cb]);
            }
        }
        if constexpr (SPREAD) {            // the row has had the other kernel rows' MFMAs to land
            {
                if constexpr (CINB == 1)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt[0][0]), "+v"(nxt[1][0]), "+v"(nxt[2][0]), "+v"(nxt[3][0]) : : "memory");
                else if constexpr (CINB == 2)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt[0][0]), "+v"(nxt[1][0]), "+v"(nxt[2][0]), "+v"(nxt[3][0]),
                                 "+v"(nxt[0][1]), "+v"(nxt[1][1]), "+v"(nxt[2][1]), "+v"(nxt[3][1]) : : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt[0][0]), "+v"(nxt[1][0]), "+v"(nxt[2][0]), "+v"(nxt[3][0]),
                                 "+v"(nxt[0][1]), "+v"(nxt[1][1]), "+v"(nxt[2][1]), "+v"(nxt[3][1]),
                                 "+v"(nxt[0][CINB - 1]), "+v"(nxt[1][CINB - 1]), "+v"(nxt[2][CINB - 1]), "+v"(nxt[3][CINB - 1]) : : "memory");
            }
        }
        if constexpr (MODE == 0 && POOL > 1) {
            // inference: pool the PRE-activations (pw[j] = running maximum of the last j+1 rows), SELU once per
            // pooled row (monotone activation: bit-identical, see conv1_tm); the first POOL-1 positions of a
            // candidate produce no pooled row and skip the activation altogether
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f4 t = acc[w] + b4;
                const f4 o = max4(pw[POOL - 2][w], t);
#pragma unroll
                for (int j = POOL - 2; j > 0; j--) pw[j][w] = max4(pw[j - 1][w], t);
                pw[0][w] = t;
                if (h - hbeg >= POOL - 1) {
                    op[(size_t)((h - (POOL - 1)) * 4 + w) * (NT * 64)] = selu4(o);
                }
            }
#pragma unroll
            for (int j = 0; j + 1 < KH; j++)
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int cb = 0; cb < CINB; cb++) win[j][w][cb] = win[j + 1][w][cb];
            continue;
        }
        f4 v[4];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int w = 0; w < 4; w++) v[w] = acc[w];
            if (act_tm) {                    // the layer below has no pooling: its pre-activation gradient = this times selu'
                const f4 *yp = act_tm + (size_t)g * (HOUT * 4 * NT * 64) + (size_t)nt * 64 + lane;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const f4 y = yp[(size_t)(h * 4 + w) * (NT * 64)];
#pragma unroll
                    for (int k = 0; k < 4; k++) v[w][k] *= cv_selu_grad_from_out(y[k]);
                }
            }
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) v[w] = selu4(acc[w] + b4);
        }
        if constexpr (POOL > 1) {
            f4 o[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                o[w] = v[w];
#pragma unroll
                for (int j = 0; j < POOL - 1; j++) o[w] = max4(o[w], pw[j][w]);
            }
            if constexpr (MODE == 1) {
                if (h - hbeg >= POOL - 1) {
                    unsigned cw[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        f4 older[POOL - 1];
#pragma unroll
                        for (int j = 0; j < POOL - 1; j++) older[j] = pw[j][w];
                        cw[w] = pool_code4<POOL>(older, v[w], o[w]);
                    }
                    u32x2 *cp = reinterpret_cast<u32x2 *>(act_tm);
                    cp[(((size_t)g * HOUT + (h - (POOL - 1))) * NT + nt) * 64 + lane] = (u32x2){cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16)};
                }
            }
#pragma unroll
            for (int j = 0; j + 1 < POOL - 1; j++)
#pragma unroll
                for (int w = 0; w < 4; w++) pw[j][w] = pw[j + 1][w];
#pragma unroll
            for (int w = 0; w < 4; w++) pw[POOL - 2][w] = v[w];
            if (h - hbeg >= POOL - 1) {
#pragma unroll
                for (int w = 0; w < 4; w++) op[(size_t)((h - (POOL - 1)) * 4 + w) * (NT * 64)] = o[w];
            }
        } else {
#pragma unroll
            for (int w = 0; w < 4; w++) op[(size_t)(h * 4 + w) * (NT * 64)] = v[w];
        }
#pragma unroll
        for (int j = 0; j + 1 < KH; j++)
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int cb = 0; cb < CINB; cb++) win[j][w][cb] = win[j + 1][w][cb];
    }
    }                                      // segments (groups) of this wave
    CV_STAMP_END(MODE == 2 && KH == 3 && CINB == 3, 2);
    CV_STAMP_END(MODE == 1 && KH == 2 && CINB == 1 && FRONT == 0, 6);
}

// ---------------------------------------------------------------------------
// conv1 k(1,4) + pool(5) + conv2 k(2,4) + pool(4) of the full topology with the FIRST LAYER SHARED between the two
// waves of a group (variant bit 6).  conv_tm<2,1,2,4,29,5> gives each of the two output tiles of conv2 its own wave,
// and both waves push the whole first layer through their registers: its 33 MFMA rows, pooling windows and -- the
// expensive part -- 29 rows of SELU are computed twice.  Here a workgroup (4 waves = 2 groups x 2 tiles) walks the 29
// pooled first-layer rows in chunks of CH: in phase A the two waves of a group each produce HALF of the chunk's rows
// (pre-activations of CH/2 + 4 input rows, running maxima, one SELU per pooled row) into an LDS row buffer laid out
// as B fragments; after a barrier both waves run conv2 over the chunk from LDS (phase B: 96 MFMA steps per
// position, kh-row and pooling state kept in registers across chunks), second barrier, next chunk.  Per position a
// wave now evaluates 8 + 16 SELU'd values x lanes instead of 16 + 16.  Arithmetic and order per output value are
// those of conv_tm: bit-identical.  LDS: 16 KB conv2 weights + 2 groups x CH x 4 KB rows (CH = 6: 64 KB, two
// workgroups per CU).
// ---------------------------------------------------------------------------
// FLAT (round 6): a workgroup is ONE pair of waves and owns the pooled conv2 rows [r0, r1) of the flat (group, row)
// sequence -- rows_per of them, whatever the group boundaries -- so that a launch whose groups do not fill the chip's
// workgroup slots evenly still gives every SIMD the same work.  The rows of each group in the range are one segment:
// pooled rows [oa, ob) need the conv2 rows [oa, ob + 3) and those the first-layer rows [oa, ob + 3] (a whole group: 26,
// 29 and 29 rows -- a segment pays 3 conv2 rows and 4 first-layer rows for its first window).  Same values row for row.
template <int CH, bool FLAT = false>
__global__ __launch_bounds__(FLAT ? 128 : 256, 2) void front2_tm(const float *__restrict__ x, int64_t n,
                                                     const float *__restrict__ wp1, const float *__restrict__ bias1,
                                                     int cout1, const f4 *__restrict__ wp,
                                                     const float *__restrict__ bias, int cout,
                                                     f4 *__restrict__ out_tm, int G, int rows_per = 0)
{
    constexpr int P1 = 5, H1 = CV_INPUT_H - P1 + 1;      // 29 pooled first-layer rows
    constexpr int NT = 2, P2 = 4, H2 = H1 - P2 + 1;      // conv2: 29 rows -> 26 pooled rows
    constexpr int NW = NT * 2 * 4 * 64;                  // f4 of packed conv2 weights [nt][kh][kw][64]
    extern __shared__ __attribute__((aligned(16))) f4 lds[];
    f4 *rows = lds + NW;                                 // [group in workgroup][CH][w][64]
    for (int i = threadIdx.x; i < NW; i += (FLAT ? 128 : 256)) lds[i] = wp[i];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int gl = FLAT ? 0 : wid >> 1, nt = wid & 1;
    int gF, gL, r0 = 0, r1 = 0;
    bool live = true;
    if constexpr (FLAT) {
        r0 = (int)blockIdx.x * rows_per;
        r1 = r0 + rows_per < G * H2 ? r0 + rows_per : G * H2;
        if (r0 >= r1) return;                            // (the whole workgroup: both waves own the same range)
        gF = r0 / H2; gL = (r1 - 1) / H2;
    } else {
        const int gq = blockIdx.x * 2 + gl;
        live = gq < G;                                   // a spare half workgroup still takes part in the barriers
        gF = gL = live ? gq : G - 1;
    }
    const int q = lane >> 4;
    float A1[4];
#pragma unroll
    for (int kw = 0; kw < 4; kw++) A1[kw] = wp1[kw * 64 + lane];
    const f4 b1 = load_bias4(bias1, 0, q, cout1);
    const f4 b2 = load_bias4(bias, nt, q, cout);
    const f4 *wl = lds + (size_t)nt * (2 * 4 * 64) + lane;
    f4 *myrows = rows + (size_t)gl * (CH * 4 * 64) + lane;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 prev[4], m1[4], m2[4], m3[4];
#pragma unroll
    for (int w = 0; w < 4; w++) { prev[w] = zero; m1[w] = zero; m2[w] = zero; m3[w] = zero; }
    __syncthreads();                                     // conv2 weights are in LDS
    CV_PHASE_BEGIN
#pragma unroll 1
    for (int g = gF; g <= gL; g++) {
    // this segment: pooled conv2 rows [oa, ob) of group g (a whole group: 0, H2)
    int oa = 0, ob = H2;
    if constexpr (FLAT) {
        oa = r0 - g * H2 > 0 ? r0 - g * H2 : 0;
        ob = r1 - g * H2 < H2 ? r1 - g * H2 : H2;
    }
    const int pend = ob + P2 < H1 ? ob + P2 : H1;        // first-layer rows [oa, pend)
    int64_t cand = (int64_t)g * 16 + (lane & 15);
    if (cand >= n) cand = n - 1;
    const float *xp = x + (size_t)cand * (CV_INPUT_H * 16) + q;      // lane (c, ci = q)
    f4 *op = out_tm + (size_t)g * (H2 * 4 * NT * 64) + (size_t)nt * 64 + lane;

    // conv2 output row h from input rows h (prev) and h + 1 (cur; absent below the last row), pooled over 4 rows
    // (the running maxima of a segment's first three rows hold values of the segment before: they are never stored)
    auto out_row = [&](int h, const f4 (&cur)[4], bool has_cur) {
        f4 acc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) acc[w] = zero;
#pragma unroll
        for (int kw = 0; kw < 4; kw++) {
            const f4 A = wl[(size_t)(0 * 4 + kw) * 64];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                for (int wo = 0; wo < 4; wo++) {
                    const int wi = wo + kw - 1;
                    if (wi < 0 || wi > 3) continue;
                    acc[wo] = mfma4(A[s4], prev[wi][s4], acc[wo]);
                }
        }
        if (has_cur) {
#pragma unroll
            for (int kw = 0; kw < 4; kw++) {
                const f4 A = wl[(size_t)(1 * 4 + kw) * 64];
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                    for (int wo = 0; wo < 4; wo++) {
                        const int wi = wo + kw - 1;
                        if (wi < 0 || wi > 3) continue;
                        acc[wo] = mfma4(A[s4], cur[wi][s4], acc[wo]);
                    }
            }
        }
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const f4 t = acc[w] + b2;
            const f4 o = max4(m3[w], t);
            m3[w] = max4(m2[w], t);
            m2[w] = max4(m1[w], t);
            m1[w] = t;
            if (h - oa >= P2 - 1 && live) op[(size_t)((h - (P2 - 1)) * 4 + w) * (NT * 64)] = selu4(o);
        }
    };

#pragma unroll 1
    for (int c0 = oa; c0 < pend; c0 += CH) {
        const int cn = pend - c0 < CH ? pend - c0 : CH;
        // ---- phase A: this wave's half of the chunk's first-layer rows
        const int half = (cn + 1) >> 1;
        const int a0 = c0 + nt * half;
        const int a1 = a0 + half < c0 + cn ? a0 + half : c0 + cn;
        if (a0 < a1) {
            // HALF pooled rows need HALF + 4 pre-activation rows; all of them are kept in registers so that the 5-row
            // windows share their middle: c = max3(t2,t3,t4), rows = max3(t0,t1,c), max3(t1,c,t5), max3(c,t5,t6) --
            // four v_max3 per value for three rows, where a running-maximum walk takes four v_max per value and ROW
            static_assert(CH == 6, "the block pooling below is written for three rows per wave");
            constexpr int HALF = CH / 2, NRAW = HALF + P1 - 1;
            float xr[NRAW][4];
#pragma unroll
            for (int r = 0; r < NRAW; r++) {
                const int rr = a0 + r < CV_INPUT_H ? a0 + r : CV_INPUT_H - 1;   // rows past the input feed unused outputs
#pragma unroll
                for (int w = 0; w < 4; w++) xr[r][w] = xp[rr * 16 + w * 4];
            }
            CV_PHASE(0);
            CV_PHASE_DRAIN();                            // (development probe: the raw rows' round trip on its own)
            CV_PHASE(3);
            f4 t[NRAW][4];
#pragma unroll
            for (int r = 0; r < NRAW; r++) {
                f4 acc[4];
#pragma unroll
                for (int w = 0; w < 4; w++) acc[w] = zero;
#pragma unroll
                for (int kw = 0; kw < 4; kw++)
#pragma unroll
                    for (int wo = 0; wo < 4; wo++) {
                        const int wi = wo + kw - 1;
                        if (wi < 0 || wi > 3) continue;
                        acc[wo] = mfma4(A1[kw], xr[r][wi], acc[wo]);
                    }
#pragma unroll
                for (int w = 0; w < 4; w++) t[r][w] = acc[w] + b1;
            }
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const f4 c = max3_4(t[2][w], t[3][w], t[4][w]);
                const f4 o0 = max3_4(t[0][w], t[1][w], c);
                const f4 o1 = max3_4(t[1][w], c, t[5][w]);
                const f4 o2 = max3_4(c, t[5][w], t[6][w]);
                myrows[(size_t)((a0 - c0 + 0) * 4 + w) * 64] = selu4(o0);
                if (a0 + 1 < a1) myrows[(size_t)((a0 - c0 + 1) * 4 + w) * 64] = selu4(o1);
                if (a0 + 2 < a1) myrows[(size_t)((a0 - c0 + 2) * 4 + w) * 64] = selu4(o2);
            }
        }
        CV_PHASE(0);
        __syncthreads();
        CV_PHASE(2);
        // ---- phase B: conv2 over the chunk's rows
#pragma unroll 1
        for (int p = c0; p < c0 + cn; p++) {
            f4 cur[4];
#pragma unroll
            for (int w = 0; w < 4; w++) cur[w] = myrows[(size_t)((p - c0) * 4 + w) * 64];
            if (p > oa) out_row(p - 1, cur, true);
#pragma unroll
            for (int w = 0; w < 4; w++) prev[w] = cur[w];
        }
        CV_PHASE(1);
        __syncthreads();
        CV_PHASE(2);
    }
    if (ob == H2) out_row(H1 - 1, prev, false);          // the row below the last one is SAME padding
    }                                                    // segments (groups) of this pair of waves
    CV_PHASE(1);
    CV_PHASE_END(true, wid);
}

// ---------------------------------------------------------------------------
// conv3-class layer, register-lean form (variant bit 3): the KH = 3 row window lives in THREE
// rotating register slots (the position loop is unrolled by 3 so every slot index is a
// compile-time constant: no shift copies), the row a position needs next is loaded straight
// into the slot that just retired and is consumed by the LAST kh of that position (its load
// overlaps the first two thirds of the MFMA block), and the pooling window is two running
// maxima.  184 VGPRs (conv_tm: 252).  Three waves per SIMD would need <= 168: hipcc then spills
// 17 dwords per lane into the loop and the kernel is 27 % slower, so it runs at two (measured
// -2.4 % on conv3 against conv_tm).  Same arithmetic in the same order: bit-identical results.
// ---------------------------------------------------------------------------
// SAVE (training forward): every row is activated as it is produced (the backward pass routes the pooling gradient
// by the ACTIVATED values), the window's maximum is taken over the three activated rows in the rotating slots and
// the window offset of its first occurrence goes to code_tm (pool_code4) -- conv_tm MODE 1 with the rotating window.
template <int CINB, int NT, int HIN, int WAVES, int MINW, bool SAVE = false>
__global__ __launch_bounds__(WAVES * 64, MINW) void conv3_rot(const f4 *__restrict__ in_tm, const f4 *__restrict__ wp,
                                                            const float *__restrict__ bias, int cout,
                                                            f4 *__restrict__ out_tm, int G, u32x2 *__restrict__ code_tm = nullptr,
                                                            int rows_per = 0)
{
    constexpr int KH = 3, PADT = 1, POOL = 3, HOUT = HIN - POOL + 1;
    extern __shared__ __attribute__((aligned(16))) f4 ldsw[];
    constexpr int NFRAG = NT * KH * 4 * CINB;
    for (int i = threadIdx.x; i < NFRAG * 64; i += WAVES * 64) ldsw[i] = wp[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + (threadIdx.x >> 6));
    // rows_per == 0: one wave per (group, tile), all HIN positions.  rows_per > 0 (round 6: also the inference pass, when a
    // whole-group launch would leave part of the chip idle or need a round more -- launch_conv3_rot): the wave owns the
    // POOLED rows [r0, r1) of the flat (group, row) sequence of its tile, so that a launch is one round of equal waves
    // (conv_tm HSPLIT == 0); the rows of each group in the range are one segment of the loop below -- positions
    // [hbeg, hend) = its pooled rows and the POOL - 1 behind them; same values row for row.
    const bool flat = rows_per > 0;
    if (gridDim.x >= 16) {
        // XCD-aware mapping (workgroup b runs on XCD b % 8, each XCD has its own L2): the NT waves of a group (of a range)
        // read the same input rows, and with WAVES = 4, NT = 3 every other one has its waves in two consecutive workgroups
        // = two XCDs, which then both fetch the map from HBM (measured: 1.44 x the input per launch).  Here XCD x owns the
        // units u = x (mod 8): the waves of the workgroups b = x, x + 8, x + 16, ... are numbered in that order, so a
        // unit's waves sit in workgroups of ONE XCD.  A speed-only assumption: the values do not depend on it.
        const int x = blockIdx.x & 7;
        const int lw = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * WAVES + (threadIdx.x >> 6));
        wv = ((lw / NT) * 8 + x) * NT + lw % NT;
    }
    const int nt = wv % NT;
    int gF = wv / NT, gL = gF, r0 = 0, r1 = 0;
    if (flat) {
        r0 = (wv / NT) * rows_per;
        r1 = r0 + rows_per < G * HOUT ? r0 + rows_per : G * HOUT;
        if (r0 >= r1) return;
        gF = r0 / HOUT; gL = (r1 - 1) / HOUT;
    } else if (gF >= G) return;
    CV_STAMP_BEGIN
    const int q = lane >> 4;
    const f4 b4 = load_bias4(bias, nt, q, cout);
    const f4 *wl = ldsw + (size_t)nt * (KH * 4 * CINB * 64) + lane;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    const unsigned lane16 = (unsigned)lane * 16u;
#pragma unroll 1
    for (int g = gF; g <= gL; g++) {
    int hbeg = 0, hend = HIN;
    if (flat) {
        hbeg = r0 - g * HOUT > 0 ? r0 - g * HOUT : 0;
        hend = (r1 - g * HOUT < HOUT ? r1 - g * HOUT : HOUT) + POOL - 1;
    }
    const f4 *inp = in_tm + (size_t)g * (HIN * 4 * CINB * 64) + lane;
    f4 *op = out_tm + (size_t)g * (HOUT * 4 * NT * 64) + (size_t)nt * 64 + lane;
    // slots are relative to the segment: position hbeg + j runs as R = j % 3 and input row hbeg + j sits in slot (j + 1) % 3
    f4 win[3][4][CINB];
    f4 tp[3][4];                 // rows h-2, h-1, h of the pooling window (slot R of their position)
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int w = 0; w < 4; w++) tp[j][w] = zero;
    auto load_row = [&](int hr, f4 (&row)[4][CINB]) {
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) row[w][cb] = inp[(size_t)((hr * 4 + w) * CINB + cb) * 64];
    };
    load_row(hbeg, win[1]);      // first row -> slot 1 ; the row above it -> slot 0 (padding, never read, when hbeg == 0)
    if (hbeg > 0) load_row(hbeg - 1, win[0]);
    // The row the last kernel row needs is loaded one fragment at a time between the MFMA blocks of the FIRST kernel
    // row, from asm (scalar base + this lane's 16 bytes) so that the loads stay where they are put: as a burst of
    // 4 CINB loads the wave queues on the CU's vector-memory port behind the other waves' bursts (see conv_tm).
    static_assert(CINB == 2, "the counted wait below names 8 fragments");
    const f4 *const inp_s = in_tm + (size_t)g * (HIN * 4 * CINB * 64);
    auto load_piece = [&](int hr, int w, int cb, f4 &dst) {
        const f4 *ps = inp_s + (size_t)((hr * 4 + w) * CINB + cb) * 64;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane16), "s"(ps));    // (no memory clobber: the
    };                                                       //  weight reads from LDS may move across it)
    // one position; R = (h - hbeg) % 3 is a compile-time constant so that slot indices are static
    auto step = [&](auto Rc, int h) {
        constexpr int R = decltype(Rc)::value;
        // rows h-1, h, h+1 are in slots R, (R+1)%3, (R+2)%3; row h+1 is fetched under the first kernel row and used last
        // (past the last row the loads re-read it -- nobody uses the result: no branch around every load)
        const int hnext = h + 1 < HIN ? h + 1 : HIN - 1;
        f4 acc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) acc[w] = zero;
#pragma unroll
        for (int kh = 0; kh < KH; kh++) {
            const int hr = h + kh - PADT;
            const bool on = hr >= 0 && hr < HIN;
            if (kh == KH - 1) {
                f4 (&nw)[4][CINB] = win[(R + 2) % 3];
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nw[0][0]), "+v"(nw[1][0]), "+v"(nw[2][0]), "+v"(nw[3][0]),
                             "+v"(nw[0][1]), "+v"(nw[1][1]), "+v"(nw[2][1]), "+v"(nw[3][1]) : : "memory");
            }
            if (on) {                                // (one branch per kernel row, not per block: the weight reads
#pragma unroll                                       //  run ahead of their MFMAs only inside a basic block)
                for (int kw = 0; kw < 4; kw++)
#pragma unroll
                    for (int cb = 0; cb < CINB; cb++) {
                        const f4 A = wl[(size_t)((kh * 4 + kw) * CINB + cb) * 64];
#pragma unroll
                        for (int s = 0; s < 4; s++)
#pragma unroll
                            for (int wo = 0; wo < 4; wo++) {
                                const int wi = wo + kw - 1;
                                if (wi < 0 || wi > 3) continue;
                                acc[wo] = mfma4(A[s], win[(R + kh) % 3][wi][cb][s], acc[wo]);
                            }
                        if (kh == 0) load_piece(hnext, kw, cb, win[(R + 2) % 3][kw][cb]);
                    }
            } else if (kh == 0) {                    // padding row on top (h == 0): nothing to hide the loads under
#pragma unroll
                for (int kw = 0; kw < 4; kw++)
#pragma unroll
                    for (int cb = 0; cb < CINB; cb++) load_piece(hnext, kw, cb, win[(R + 2) % 3][kw][cb]);
            }
        }
        // max-pool on the PRE-activations, SELU once per pooled row (SELU is monotone over all of fp32 --
        // cv_selu_sweep -- so max_j selu(a_j + b) == selu(max_j (a_j + b)) bit for bit): 24 activated rows per
        // candidate instead of 26
#pragma unroll
        // the three pre-activation rows of the window sit in rotating slots (like the input window): one v_max3 per
        // value and row
        for (int w = 0; w < 4; w++) {
            if constexpr (SAVE) tp[R][w] = selu4(acc[w] + b4);
            else tp[R][w] = acc[w] + b4;             // (a sum needs no canonicalising v_max x, x, x; a raw MFMA result would)
            if (h - hbeg >= POOL - 1) {
                if constexpr (SAVE) op[(size_t)((h - (POOL - 1)) * 4 + w) * (NT * 64)] = max3_4(tp[0][w], tp[1][w], tp[2][w]);
                else op[(size_t)((h - (POOL - 1)) * 4 + w) * (NT * 64)] = selu4(max3_4(tp[0][w], tp[1][w], tp[2][w]));
            }
        }
        if constexpr (SAVE) {
            if (h - hbeg >= POOL - 1) {              // rows h-2, h-1, h sit in slots (R+1)%3, (R+2)%3, R
                unsigned cw[4];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const f4 older[2] = {tp[(R + 1) % 3][w], tp[(R + 2) % 3][w]};
                    cw[w] = pool_code4<3>(older, tp[R][w], max3_4(tp[0][w], tp[1][w], tp[2][w]));
                }
                code_tm[(((size_t)g * HOUT + (h - (POOL - 1))) * NT + nt) * 64 + lane] = (u32x2){cw[0] | (cw[1] << 16), cw[2] | (cw[3] << 16)};
            }
        }
    };
#pragma unroll 1
    for (int h0 = hbeg; h0 < hend; h0 += 3) {
        step(std::integral_constant<int, 0>{}, h0);
        __builtin_amdgcn_sched_barrier(0);     // keep the three positions apart: register budget
        if (h0 + 1 < hend) step(std::integral_constant<int, 1>{}, h0 + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (h0 + 2 < hend) step(std::integral_constant<int, 2>{}, h0 + 2);
        __builtin_amdgcn_sched_barrier(0);
    }
    }                                          // segments (groups) of this wave
    CV_STAMP_END(SAVE, 1);
}

// ---------------------------------------------------------------------------
// heads (v3.py:124-138): one wave per group of 16 candidates.
//   tile 0 (input fc4 side, K = NB4*16): rows 0..3  = base logits -> sigmoid
//   tile 1 (input fc5,      K = NB5*16): rows 0..1  = zygosity, rows 4..7 = variant type,
//                                        rows 8..13 = indel length -> softmax(selu(.)+1e-10)
// Rows are NOT sigma-permuted (identity), so lane (c, q) holds rows 4q..4q+3 of candidate c:
// q=0: base[0..3] and zyg[0..1];  q=1: type[0..3];  q=2: len[0..3];  q=3: len[4..5].
// The 6-way softmax spans lanes c+32 / c+48; its sum is formed in index order
// ((((e0+e1)+e2)+e3)+e4)+e5 by passing the partial sum across.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pack_heads(int64_t t, const float *__restrict__ wb, const float *__restrict__ wz,
                           const float *__restrict__ wt, const float *__restrict__ wl, int K4, int K5,
                           int NB4, int NB5, float *__restrict__ wp0, float *__restrict__ wp1, float *__restrict__ w12)
{
    int tot0 = NB4 * 256, tot1 = NB5 * 256;
    if (t >= tot0 + tot1) {          // the fc5-side head weights of a unit side by side [k][zygosity 2 | type 4 | length 6]: what the
        const int u = (int)t - tot0 - tot1;      // training heads stage in LDS for their data gradient (one coalesced copy)
        if (u >= NB5 * 16 * 12 || !w12) return;
        const int k = u / 12, jj = u % 12;
        float v = 0.0f;
        if (k < K5) v = jj < 2 ? wz[(size_t)k * 2 + jj] : (jj < 6 ? wt[(size_t)k * 4 + (jj - 2)] : wl[(size_t)k * 6 + (jj - 6)]);
        w12[u] = v;
        return;
    }
    if (t < tot0) {
        int s = t & 3, lane = (t >> 2) & 63, kb = t >> 8;
        int i = lane & 15, kq = lane >> 4, k = 16 * kb + 4 * s + kq;
        wp0[t] = (i < 4 && k < K4) ? wb[(size_t)k * 4 + i] : 0.0f;
    } else if (t < tot0 + tot1) {
        int u = t - tot0;
        int s = u & 3, lane = (u >> 2) & 63, kb = u >> 8;
        int i = lane & 15, kq = lane >> 4, k = 16 * kb + 4 * s + kq;
        float v = 0.0f;
        if (k < K5) {
            if (i < 2) v = wz[(size_t)k * 2 + i];
            else if (i >= 4 && i < 8) v = wt[(size_t)k * 4 + (i - 4)];
            else if (i >= 8 && i < 14) v = wl[(size_t)k * 6 + (i - 8)];
        }
        wp1[u] = v;
    }
}

// Epilogue of the heads: a0 = base-head tile (rows 0..3 on q = 0), a1 = zygosity / type / length tile; sigmoid,
// softmax(selu(.) + 1e-10) per head in registers (the 6-way softmax spans lanes c+32 / c+48), 16 outputs per candidate.
__device__ __forceinline__ void heads_finish(f4 a0, f4 a1, const float *__restrict__ bb, const float *__restrict__ bz,
                                             const float *__restrict__ bt, const float *__restrict__ bl, int64_t n,
                                             float *__restrict__ out16, int g, int lane)
{
    const int c = lane & 15, q = lane >> 4;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    // biases of this lane's rows
    f4 bias1 = zero;
    if (q == 0) { bias1[0] = bz[0]; bias1[1] = bz[1]; }
    else if (q == 1) { bias1[0] = bt[0]; bias1[1] = bt[1]; bias1[2] = bt[2]; bias1[3] = bt[3]; }
    else if (q == 2) { bias1[0] = bl[0]; bias1[1] = bl[1]; bias1[2] = bl[2]; bias1[3] = bl[3]; }
    else { bias1[0] = bl[4]; bias1[1] = bl[5]; }
    f4 lg;
#pragma unroll
    for (int r = 0; r < 4; r++) lg[r] = cvm::selu(a1[r] + bias1[r]) + 1e-10f;
    const int64_t cand = (int64_t)g * 16 + c;
    float *o = out16 + (size_t)cand * 16;
    // 6-way softmax across lanes q=2 (len0..3) and q=3 (len4..5)
    float m_loc = q == 3 ? fmaxf(lg[0], lg[1]) : fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
    float m_oth = __shfl_xor(m_loc, 16);
    const float m6 = fmaxf(m_loc, m_oth);
    if (q == 0) {
        if (cand < n) {
            float4 b;
            b.x = cvm::sigmoid(a0[0] + bb[0]); b.y = cvm::sigmoid(a0[1] + bb[1]);
            b.z = cvm::sigmoid(a0[2] + bb[2]); b.w = cvm::sigmoid(a0[3] + bb[3]);
            *reinterpret_cast<float4 *>(o) = b;
            float l2[2] = {lg[0], lg[1]}, p2[2];
            cvm::softmax<2>(l2, p2);
            o[4] = p2[0]; o[5] = p2[1];
        }
    } else if (q == 1) {
        if (cand < n) {
            float l4[4] = {lg[0], lg[1], lg[2], lg[3]}, p4v[4];
            cvm::softmax<4>(l4, p4v);
            o[6] = p4v[0]; o[7] = p4v[1]; o[8] = p4v[2]; o[9] = p4v[3];
        }
    }
    // all lanes take part in the exchange below (shuffles need the full wave)
    float e[4];
#pragma unroll
    for (int r = 0; r < 4; r++) e[r] = cvm::expf_fixed(lg[r] - m6);
    float s03 = ((e[0] + e[1]) + e[2]) + e[3];              // meaningful on q == 2
    float s03_from2 = __shfl_xor(s03, 16);                  // q == 3 receives q == 2's partial sum
    float tot = (s03_from2 + e[0]) + e[1];                  // meaningful on q == 3
    float tot_from3 = __shfl_xor(tot, 16);                  // q == 2 receives the total
    if (cand < n) {
        if (q == 2) {
            o[10] = e[0] / tot_from3; o[11] = e[1] / tot_from3; o[12] = e[2] / tot_from3; o[13] = e[3] / tot_from3;
        } else if (q == 3) {
            o[14] = e[0] / tot; o[15] = e[1] / tot;
        }
    }}

__global__ __launch_bounds__(256) void heads_tm(const f4 *__restrict__ h4, const f4 *__restrict__ h5, int NB4,
                                                 int NB5, const f4 *__restrict__ wp0, const f4 *__restrict__ wp1,
                                                 const float *__restrict__ bb, const float *__restrict__ bz,
                                                 const float *__restrict__ bt, const float *__restrict__ bl,
                                                 int64_t n, float *__restrict__ out16, int G)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 a0 = zero, a1 = zero;
    const f4 *p4 = h4 + (size_t)g * NB4 * 64 + lane;
    const f4 *p5 = h5 + (size_t)g * NB5 * 64 + lane;
#pragma unroll 7                           // (two loads per k fragment, 14 in flight: the 21-step chain is latency, not work)
    for (int kb = 0; kb < NB4; kb++) {
        const f4 B = p4[(size_t)kb * 64];
        const f4 A = wp0[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) a0 = mfma4(A[s], B[s], a0);
    }
#pragma unroll 3
    for (int kb = 0; kb < NB5; kb++) {
        const f4 B = p5[(size_t)kb * 64];
        const f4 A = wp1[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) a1 = mfma4(A[s], B[s], a1);
    }
    heads_finish(a0, a1, bb, bz, bt, bl, n, out16, g, lane);
}

// operands of the heads when they ride on the fc5 kernel (dense_tm EPI 2)
struct heads_args {
    const f4 *wp0, *wp1;                 // packed head weights (pack_heads): base head over fc4, the others over fc5
    const float *bb, *bz, *bt, *bl;      // biases
    int64_t n;
    float *out16;
    const f4 *dact = nullptr;            // EPI 1 only: the output is multiplied by selu'-from-output of this map (same layout)
    // EPI 1, fc5's data gradient of a training pass (round 5): the base head's contribution, the dropout factor and
    // selu'(fc4 output) follow on the store -- b_head_dgrad_tm's mode 1 arithmetic, one launch and one round trip of the
    // map less.  hg_g16 == NULL: none.  g16 [n][16] head pre-activation gradients, wb [K][4] base-head weights,
    // mask / act: tile-major maps in the output's layout.
    const float *hg_g16 = nullptr, *hg_wb = nullptr; const f4 *hg_mask = nullptr, *hg_act = nullptr; int64_t hg_n = 0; int hg_K = 0;
    // EPI 3 only (fc4 with fc5 and the heads on its tail): fc5's weights in k PAIRS [kp][24][64] (pack_dense_kpairs),
    // its bias / width, and where its output goes (kept for cv_get_activation and the parity tests)
    const f4 *wp5p = nullptr; const float *bias5 = nullptr; int nout5 = 0;
    int keep = 0;                        // option keep_activations: also store the maps only cv_get_activation reads
    f4 *h5_out = nullptr;
    // EPI 3: the kernel reads all of the above from this DEVICE copy on its tail, so that the two dozen scalars stay out
    // of the main loop's register budget (by value they are loaded at kernel entry and live across the whole kernel)
    const heads_args *tail = nullptr;
    // EPI 0, three-slab form (fc4 of a training pass): the alpha-dropout of the value follows in the same thread
    // (dropout_tm's arithmetic, one launch and one round trip of the map less -- the step time does not move, 2.107 against
    // 2.108 ms at 10 000: the 12 us pass ran next to the weight packing on the side stream); d4 == NULL: none
    cv_dropout_args drop = cv_dropout_args();
};

// ---------------------------------------------------------------------------
// Slim topology: conv3 k(5,4) 16 -> 32 (no pooling) FUSED with fc4 (4 224 -> 36), variant bit 8.
// Separate kernels write the 16.9 KB conv3 map of every candidate to HBM and read it back for a 36-wide
// contraction: slim fc4 sits on the HBM roof (1.1 GB in 0.26 ms), not on the matrix cores.  Here ONE wave owns a
// group and computes BOTH output tiles of conv3, so after bias + SELU its registers hold, position by position,
// exactly the fragments fc4 contracts over, in fc4's own order: kb = (h*4 + w)*2 + nt ascending = flatten order
// (v3_slim.py:84-87).  They feed the three fc4 accumulator tiles straight from registers -- the conv3 map never
// exists in memory.  fc4's weights (24 KB per position) are DMA'd global -> LDS two positions ahead into a 3-slot
// ring shared by the 8 waves of the workgroup (each wave moves the 3 fragments of one kb), one barrier per
// position (~600 MFMAs apart).  Same ascending-k chain per output value as conv_tm + dense_tm: bit-identical.
// LDS: 40 KB conv3 weights + 3 x 24 KB.  All VMEM from inline asm with one counted wait per position (dense_tm).
// ---------------------------------------------------------------------------
// WAVES (round 6) = groups per workgroup, 8 or 4: the 112 KB of LDS allow one workgroup per CU whatever its size, so a pass
// of up to 2 048 groups took the time of 2 048 (8 groups on each of G / 8 CUs, the other CUs idle); with fewer waves per
// workgroup the same groups spread over more CUs (each wave then stages 8 / WAVES of a position's k fragments).
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void conv3fc4_slim(const f4 *__restrict__ in_tm, const f4 *__restrict__ wp3,
                                                        const float *__restrict__ bias3, int cout3,
                                                        const f4 *__restrict__ wp4, const float *__restrict__ bias4,
                                                        int nout4, f4 *__restrict__ out_h4, int G,
                                                        const heads_args *__restrict__ tail = nullptr, int64_t n_cand = 0,
                                                        float *__restrict__ out16 = nullptr)
{
    // tail != nullptr (variant bit 10): fc5 (36 -> 18: 3 k fragments x 2 tiles) and the four heads follow on the same
    // wave from the fc4 fragments in its registers -- 44 MFMAs instead of two more launches; weights straight from L2
    constexpr int KH = 5, PADT = 2, HIN = CV_INPUT_H, NT = 2, NB4 = 3, NBP4 = 4;
    static_assert(WAVES == 8 || WAVES == 4, "a wave stages 8 / WAVES k fragments of a position");
    constexpr int KPW = 8 / WAVES;                        // k fragments of a position's fc4 slab each wave stages
    constexpr int NW3 = NT * KH * 4 * 64;                 // f4 of packed conv3 weights [nt][kh][kw][64]
    constexpr int SLOT = 8 * NB4 * 64;                    // f4 per ring slot: 8 k fragments x 3 output fragments
    extern __shared__ __attribute__((aligned(16))) f4 lds[];
    f4 *ring = lds + NW3;
    for (int i = threadIdx.x; i < NW3; i += WAVES * 64) lds[i] = wp3[i];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gq = blockIdx.x * WAVES + wid;
    const bool live = gq < G;
    const int g = live ? gq : G - 1;
    const int q = lane >> 4;
    const f4 b3[NT] = {load_bias4(bias3, 0, q, cout3), load_bias4(bias3, 1, q, cout3)};
    const f4 *inp = in_tm + (size_t)g * (HIN * 4 * 64) + lane;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) f4 *)ring;
    // this wave's share of the fc4 slab of position h: k fragments (h*8 + wid*KPW ..), their 3 real output fragments
    auto stage_async = [&](int h, int slot) {
        const int hc = h < HIN ? h : HIN - 1;              // surplus stages of the last positions re-read valid data
#pragma unroll
        for (int kf = 0; kf < KPW; kf++)
#pragma unroll
        for (int ob = 0; ob < NB4; ob++) {
            const f4 *gp = wp4 + ((size_t)(hc * 8 + wid * KPW + kf) * NBP4 + ob) * 64 + lane;
            const unsigned ldst = ring_base + (unsigned)(((slot * 8 + wid * KPW + kf) * NB4 + ob) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
        }
    };
    auto load_frag = [&](const f4 *ptr) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    f4 win[KH][4];            // win[kh] = input row h + kh - PADT
    f4 nxt[4];
    f4 acc4[NB4];
#pragma unroll
    for (int ob = 0; ob < NB4; ob++) acc4[ob] = zero;
    stage_async(0, 0);
    stage_async(1, 1);
    // prologue: rows -2..1 -> win[0..3], row 2 -> nxt
#pragma unroll
    for (int j = 0; j < KH; j++) {
        const int hr = j - PADT;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const f4 v = hr >= 0 ? load_frag(inp + (size_t)(hr * 4 + w) * 64) : zero;
            if (j < KH - 1) win[j][w] = v; else nxt[w] = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < KH - 1; j++)
#pragma unroll
        for (int w = 0; w < 4; w++) asm volatile("" : "+v"(win[j][w]));
#pragma unroll
    for (int w = 0; w < 4; w++) asm volatile("" : "+v"(nxt[w]));
    __syncthreads();          // conv3 weights and ring slots 0, 1 are in LDS
    int slot = 0;
#pragma unroll 1
    for (int h = 0; h < HIN; h++) {
#pragma unroll
        for (int w = 0; w < 4; w++) win[KH - 1][w] = nxt[w];
        {   // row h + 3 for the next position, fc4 slab of position h + 2
            const int hr = h + 1 + (KH - 1) - PADT;
            const int hc = hr < HIN ? hr : HIN - 1;
#pragma unroll
            for (int w = 0; w < 4; w++) nxt[w] = load_frag(inp + (size_t)(hc * 4 + w) * 64);
            int wslot = slot + 2; if (wslot >= 3) wslot -= 3;
            stage_async(h + 2, wslot);
        }
        f4 v[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f4 acc[4];
#pragma unroll
            for (int w = 0; w < 4; w++) acc[w] = zero;
            const f4 *wl = lds + (size_t)nt * (KH * 4 * 64) + lane;
#pragma unroll
            for (int kh = 0; kh < KH; kh++) {
                const int hr = h + kh - PADT;
                if (hr >= 0 && hr < HIN) {             // wave-uniform; SAME padding rows are skipped
#pragma unroll
                    for (int kw = 0; kw < 4; kw++) {
                        const f4 A = wl[(size_t)(kh * 4 + kw) * 64];
#pragma unroll
                        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                            for (int wo = 0; wo < 4; wo++) {
                                const int wi = wo + kw - 1;
                                if (wi < 0 || wi > 3) continue;
                                acc[wo] = mfma4(A[s4], win[kh][wi][s4], acc[wo]);
                            }
                    }
                }
            }
#pragma unroll
            for (int w = 0; w < 4; w++) v[nt][w] = selu4(acc[w] + b3[nt]);
        }
        // fc4: k fragments of this position in flatten order (w, nt), weights from the ring slot
        const f4 *rl = ring + (size_t)slot * SLOT + lane;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                f4 A[NB4];
#pragma unroll
                for (int ob = 0; ob < NB4; ob++) A[ob] = rl[(size_t)((w * NT + nt) * NB4 + ob) * 64];
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                    for (int ob = 0; ob < NB4; ob++) acc4[ob] = mfma4(A[ob][s4], v[nt][w][s4], acc4[ob]);
            }
        __builtin_amdgcn_sched_barrier(0);
        // counted wait: this position's 3 DMA pieces (slab h + 2) stay in flight; the row loads issued before them
        // and the pieces of slab h + 1 (issued one position ago) have landed; the barrier publishes slab h + 1
        if constexpr (KPW == 1) asm volatile("s_waitcnt vmcnt(3)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(6)" : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]) : : "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j + 1 < KH; j++)
#pragma unroll
            for (int w = 0; w < 4; w++) win[j][w] = win[j + 1][w];
        slot = slot + 1 == 3 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the surplus stages before the wave retires
    if (!live) return;
    f4 *op = out_h4 + (size_t)g * NB4 * 64 + lane;
    f4 h4[NB4];
    const bool store_maps = !tail || tail->keep;           // with the tail below the maps are for cv_get_activation only
#pragma unroll
    for (int ob = 0; ob < NB4; ob++) { h4[ob] = selu4(acc4[ob] + load_bias4(bias4, ob, q, nout4)); if (store_maps) op[ob * 64] = h4[ob]; }
    if (!tail) return;
    const heads_args hd = *tail;
    constexpr int NB5 = 2, NBP5 = 4;                       // fc5's packed weights: [kb][4][64] (two real tiles)
    f4 a0 = zero, h5[NB5];
#pragma unroll
    for (int ob = 0; ob < NB5; ob++) {
        f4 a = zero;
#pragma unroll
        for (int kb = 0; kb < NB4; kb++) {
            const f4 A = hd.wp5p[((size_t)kb * NBP5 + ob) * 64 + lane];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) a = mfma4(A[s4], h4[kb][s4], a);
        }
        h5[ob] = selu4(a + load_bias4(hd.bias5, ob, q, hd.nout5));
        if (hd.keep) hd.h5_out[((size_t)g * NB5 + ob) * 64 + lane] = h5[ob];
    }
#pragma unroll
    for (int kb = 0; kb < NB4; kb++) {                     // base head over the fc4 output
        const f4 A = hd.wp0[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) a0 = mfma4(A[s4], h4[kb][s4], a0);
    }
    f4 a1 = zero;
#pragma unroll
    for (int ob = 0; ob < NB5; ob++) {
        const f4 A = hd.wp1[(size_t)ob * 64 + lane];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) a1 = mfma4(A[s4], h5[ob][s4], a1);
    }
    heads_finish(a0, a1, hd.bb, hd.bz, hd.bt, hd.bl, n_cand, out16, g, lane);
}


// ---------------------------------------------------------------------------
// dense (KB*16 -> NB*16) + bias + SELU, TM -> TM.  One wave per group of 16
// candidates holds all NB accumulator tiles; the workgroup streams the packed
// weight matrix through a 3-stage LDS ring (one barrier per 16-deep k step),
// each wave streams its own activation fragments straight from HBM/L2.
// ---------------------------------------------------------------------------
// EPI 0: + bias, SELU (forward layer).  EPI 1: raw accumulators (data-gradient pass: the same
// kernel on transposed packed weights; blockIdx.y selects a slab of NB output fragments of a
// wider result with NBT fragments per group).
// EPI 2 (fc5 of an inference pass): EPI 0 plus the four heads (v3.py:124-138) on the same wave -- the layer's input
// fragments (the fc4 output, which the base head contracts over) stream through the wave anyway and its output
// tiles are, after SELU, the fragments the other three heads contract over: two more accumulator tiles, 4 MFMAs
// per k step + 4 per output tile, then heads_finish.  No separate heads launch, the fc5 output is not re-read.
// GR = groups per wave (1 or 2): with 2 a wave keeps two sets of accumulator tiles and every weight fragment read
// from LDS feeds both -- twice the MFMA work per barrier and per LDS read, at 2 waves per SIMD.
// (measured, round 4: a FOUR-slot ring filled three k steps ahead, so that a wave reads the first two weight fragments of
// step k + 1 while it still multiplies step k, carries them across the barrier in registers and issues this step's loads /
// DMA pieces behind its first MFMA block -- no LDS round trip between a barrier and the first MFMA.  Bit-identical; the
// training step did not move: 2.1163 against 2.1167 ms at 10 000, three alternating runs each on one box,
// profiles/r04/train_ab_fc4_forward_ring.txt.  The step behind a barrier is not what the kernel waits for; removed.)
template <int NB, int WAVES, int EPI = 0, int GR = 1>
__global__ __launch_bounds__(WAVES * 64, (GR == 2 ? 2 : WAVES / 2)) void dense_tm(const f4 *__restrict__ in_tm, int KB,
                                                        const f4 *__restrict__ wp_all,
                                                        const float *__restrict__ bias, int nout,
                                                        f4 *__restrict__ out_tm, int G, int NBT = NB,
                                                        heads_args hd = heads_args())
{
    static_assert(EPI != 2 || GR == 1, "the fused heads keep one group per wave");
    static_assert(EPI != 3 || (GR == 2 && NB == 21 && WAVES == 8), "the fc5 + heads tail is written for the full topology's fc4");
    // The packed weight matrix holds NBP = roundup(NB, WAVES) fragments per k step (the pad
    // fragments are zero and never multiplied), so every thread stages exactly PER 16-byte
    // pieces per step: no conditional loads in the loop, which lets the waits sit at the
    // LDS writes (after the MFMAs) instead of right behind the load issue.
    extern __shared__ __attribute__((aligned(16))) f4 ring[];
    constexpr int NBP = (NB + WAVES - 1) / WAVES * WAVES;
    constexpr int STAGE = NBP * 64;              // f4 per stage
    constexpr int PER = NBP / WAVES;             // fragments each wave stages per k step
    // gridDim.z > 1 (training forward of tiny batches): the contraction is split into gridDim.z ranges of k
    // fragments, each workgroup leaves the raw partial accumulators of its range in out_tm ([z][g][NBT] fragments)
    // and dense_ksum adds the ranges in ascending order, + bias, SELU.  A fixed order (reproducible), but not the
    // single ascending-k chain of the inference path -- used where the step is latency-bound (288 dependent k
    // steps for fc4) and parity is a tolerance, never for cv_forward.
    const int KS = gridDim.z, kz = blockIdx.z;
    const int kb0 = KS > 1 ? KB * kz / KS : 0, KBA = KB;
    if (KS > 1) KB = KBA * (kz + 1) / KS - kb0;
    const f4 *wp = wp_all + ((size_t)blockIdx.y * KBA + kb0) * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = (blockIdx.x * WAVES + wid) * GR;
    CV_STAMP_BEGIN
    // this wave's activation fragments: byte offsets from in_tm (a scalar base + a 32-bit vector offset per group
    // instead of a 64-bit pointer: 2 VGPRs less per group, which the fc5 + heads tail of EPI 3 needs; a pass is at most
    // 4 096 groups x 288 fragments = 1.2 GB)
    unsigned bo[GR];
#pragma unroll
    for (int r = 0; r < GR; r++) {
        const int gl = g + r < G ? g + r : G - 1;
        bo[r] = (unsigned)((((size_t)gl * KBA + kb0) * 64 + lane) * sizeof(f4));
    }
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[GR][NB];
#pragma unroll
    for (int r = 0; r < GR; r++)
#pragma unroll
        for (int ob = 0; ob < NB; ob++) acc[r][ob] = zero;
    // global -> LDS DMA (global_load_lds_dwordx4): a wave moves one 1 KiB fragment per
    // instruction, destination = wave-uniform LDS base (M0) + lane*16 = the fragment layout
    // itself.  Issued from inline asm so that hipcc does not fence every following ds_read
    // behind it (it cannot tell the ring slots apart); completion is waited for explicitly
    // (vmcnt) before the barrier that publishes the slot.
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) f4 *)ring;
    auto stage_async = [&](int kb, int slot) {
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const f4 *gp = wp + ((size_t)kb * NBP + wid * PER + p) * 64 + lane;
            const unsigned ldst = ring_base + (unsigned)((slot * NBP + wid * PER + p) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
        }
    };
    // The activation fragments are loaded from asm as well: with no compiler-visible VMEM in
    // the loop hipcc emits no vmcnt waits of its own (its counted waits would also drain the
    // DMA pieces queued behind them); every VMEM completion is the explicit wait below.
    auto load_frag = [&](const f4 *ptr) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    auto load_frag_off = [&](unsigned byte_off) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(in_tm) : "memory");
        return v;
    };
    stage_async(0, 0);
    stage_async(KB > 1 ? 1 : 0, 1);
    f4 hacc0 = zero, hA = zero;                 // EPI 2: base-head tile and its weight fragment of the current k step
    if constexpr (EPI == 2) hA = load_frag(hd.wp0 + lane);
    f4 B[GR];
#pragma unroll
    for (int r = 0; r < GR; r++) B[r] = load_frag_off(bo[r]);
#pragma unroll
    for (int r = 0; r < GR; r++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(B[r]) : : "memory");
    if constexpr (EPI == 2) asm volatile("" : "+v"(hA) : : "memory");
    __syncthreads();
    int slot = 0;
#pragma unroll 1
    for (int kb = 0; kb < KB; kb++) {
        // stage kb+2 and activation fragment kb+1 (indices clamped: the surplus loads of the
        // last two steps re-read valid data and land in ring slots nobody reads again)
        const int ks = kb + 2 < KB ? kb + 2 : KB - 1;
        const int kn = kb + 1 < KB ? kb + 1 : KB - 1;
        int wslot = slot + 2; if (wslot >= 3) wslot -= 3;
        f4 Bn[GR];
#pragma unroll
        for (int r = 0; r < GR; r++) Bn[r] = load_frag_off(bo[r] + (unsigned)kn * 1024u);
        f4 hAn = zero;
        if constexpr (EPI == 2) hAn = load_frag(hd.wp0 + (size_t)kn * 64 + lane);      // issued before this step's DMA pieces
        stage_async(ks, wslot);          // slot (kb+2)%3 was last read in step kb-1 (barrier passed)
        // (measured, round 3: these loads and pieces issued one per MFMA block instead of here -- what helped the
        // convolution kernels -- makes this ring slower: training step 2.15 -> 2.21 ms, inference 18.41 -> 18.28 M/s on
        // one box; a wave issues at most 5 of them per step, and the barrier needs them early)
        const f4 *wl = ring + slot * STAGE + lane;
        constexpr int AB = EPI == 3 ? 2 : 3;      // weight fragments read ahead of their MFMAs (EPI 3 is short of 4 VGPRs)
#pragma unroll
        for (int ob = 0; ob < NB; ob += AB) {
            f4 A[AB];
#pragma unroll
            for (int j = 0; j < AB; j++)
                if (ob + j < NB) A[j] = wl[(ob + j) * 64];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int j = 0; j < AB; j++)
#pragma unroll
                    for (int r = 0; r < GR; r++)
                        if (ob + j < NB) acc[r][ob + j] = mfma4(A[j][s], B[r][s], acc[r][ob + j]);
        }
        if constexpr (EPI == 2) {
#pragma unroll
            for (int s = 0; s < 4; s++) hacc0 = mfma4(hA[s], B[0][s], hacc0);
        }
        __builtin_amdgcn_sched_barrier(0);                  // keep the MFMAs above the wait
        // Counted wait: leave THIS step's PER DMA pieces (stage kb+2, first read two steps from
        // now) in flight; everything older -- Bn and the pieces of stage kb+1 issued one step
        // ago -- has landed.  The barrier then publishes stage kb+1 to the whole workgroup.
        if constexpr (PER == 6) asm volatile("s_waitcnt vmcnt(6)" : "+v"(Bn[0]) : : "memory");
        else if constexpr (PER == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(Bn[0]) : : "memory");
        else if constexpr (PER == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(Bn[0]) : : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(Bn[0]) : : "memory");
        if constexpr (GR == 2) asm volatile("" : "+v"(Bn[1]) : : "memory");      // the same wait covers the second fragment
        if constexpr (EPI == 2) { asm volatile("" : "+v"(hAn) : : "memory"); hA = hAn; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GR; r++) B[r] = Bn[r];
        slot = slot + 1 == 3 ? 0 : slot + 1;
    }
    CV_STAMP_END(NB == 7 && EPI == 0, 3);
    const int q = lane >> 4;
    if constexpr (EPI == 3) {
        // ---- fc5 and the four heads on the tail of fc4 (inference, variant bit 10).  After bias + SELU the wave's
        // accumulators ARE fc5's k fragments (and the base head's): they never leave the registers.  fc5's weights come
        // through the same ring, two k fragments per stage (24 fragments, the same three DMA pieces per wave and the same
        // counted wait as the main loop), one group of the wave at a time (11 accumulator tiles next to the 42 fragments
        // of fc4 output that stay live).  Per output value the chain is dense_tm<11,..>'s and heads_tm's: same bits.
        constexpr int NB5 = 11, NBH = 12, KS5 = 4, KP = (NB + KS5 - 1) / KS5, ST5 = KS5 * NBH * 64, PER5 = KS5 * NBH / WAVES;
        const heads_args *tp = hd.tail;
        const int64_t n_cand = hd.n;                 // by value: they change from call to call
        float *const out16 = hd.out16;
        asm volatile("" : "+s"(tp));                 // the loads below stay below
        const heads_args hd = *tp;                   // (shadows the by-value argument from here on)
        // lane id recomputed here (v_mbcnt) instead of carried through the main loop in a register: the loop is at its
        // register limit, and a value that is live across it would be spilled and reloaded on every one of its 288 steps
        int lane_t;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
        const int q_t = lane_t >> 4;
#pragma unroll
        for (int r = 0; r < GR; r++)
#pragma unroll
            for (int ob = 0; ob < NB; ob++) {
                // (nout == 16 NB here -- checked by the launcher --, so no bounds test per lane: 168 predicates less)
                const float *bq = bias + 16 * ob + q_t;
                acc[r][ob] = selu4(acc[r][ob] + (f4){bq[0], bq[4], bq[8], bq[12]});
                // the second group's fragments are parked in the fc4 map while the first group's tail runs (re-read
                // below); the first group's are stored for cv_get_activation only (option keep_activations)
                if (g + r < G && (r > 0 || hd.keep)) out_tm[((size_t)(g + r) * NBT + ob) * 64 + lane_t] = acc[r][ob];
            }
        f4 hacc0[GR], hacc1[GR];
#pragma unroll
        for (int r = 0; r < GR; r++) { hacc0[r] = zero; hacc1[r] = zero; }
#pragma unroll
        for (int kb = 0; kb < NB; kb++) {                    // base head over the fc4 output (v3.py:124-126)
            const f4 A = hd.wp0[(size_t)kb * 64 + lane_t];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int r = 0; r < GR; r++) hacc0[r] = mfma4(A[s], acc[r][kb][s], hacc0[r]);
        }
        auto stage5 = [&](int kp, int sl) {
#pragma unroll
            for (int p = 0; p < PER5; p++) {
                const f4 *gp = hd.wp5p + ((size_t)kp * (KS5 * NBH) + wid * PER5 + p) * 64 + lane_t;
                const unsigned ldst = ring_base + (unsigned)((sl * (KS5 * NBH) + wid * PER5 + p) * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
            }
        };
#pragma unroll
        for (int r = 0; r < GR; r++) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of the previous pipeline is in flight,
            __syncthreads();                                       // nobody still reads the ring
            if (r > 0) {
                // The fc4 fragments of this group come back from the map they were stored to above (this wave's own
                // stores, complete after the wait): while the previous group's tail ran they did not occupy 84 registers,
                // which lets the compiler read weight fragments ahead of their MFMAs there.  Rows of a group past the
                // batch are read from the last real group (their results are never stored).
                const int gr = g + r < G ? g + r : G - 1;
#pragma unroll
                for (int kb = 0; kb < NB; kb++) acc[r][kb] = out_tm[((size_t)gr * NBT + kb) * 64 + lane_t];
            }
            stage5(0, 0);
            stage5(1, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            f4 acc5[NB5];
#pragma unroll
            for (int ob = 0; ob < NB5; ob++) acc5[ob] = zero;
            // The stage loop stays ROLLED (unrolled, the compiler hoists the weight reads of all stages and spills 165
            // registers): the four k fragments of a step are always acc[r][0..3]; the fragments are rotated down by four
            // at the end of a step.
            int sl = 0;
#pragma unroll 1
            for (int kp = 0; kp < KP; kp++) {
                int wsl = sl + 2; if (wsl >= 3) wsl -= 3;
                stage5(kp + 2 < KP ? kp + 2 : KP - 1, wsl);
                const f4 *wl5 = ring + sl * ST5 + lane_t;
#pragma unroll
                for (int half = 0; half < KS5; half++) {
                    if (KS5 * kp + half < NB) {                    // wave-uniform: the last stage holds one k fragment
#pragma unroll
                        for (int ob = 0; ob < NB5; ob += 3) {
                            f4 A[3];
#pragma unroll
                            for (int j = 0; j < 3; j++)
                                if (ob + j < NB5) A[j] = wl5[(half * NBH + ob + j) * 64];
#pragma unroll
                            for (int s = 0; s < 4; s++)
#pragma unroll
                                for (int j = 0; j < 3; j++)
                                    if (ob + j < NB5) acc5[ob + j] = mfma4(A[j][s], acc[r][half][s], acc5[ob + j]);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i + KS5 < NB; i++) acc[r][i] = acc[r][i + KS5];
                __builtin_amdgcn_sched_barrier(0);
                static_assert(PER5 == 6, "counted wait below");
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // stage kp + 2 (6 pieces per wave) stays in flight, kp + 1 has landed
                __syncthreads();
                sl = sl + 1 == 3 ? 0 : sl + 1;
            }
            // fc5 output of this group: bias + SELU, kept for cv_get_activation, and straight into the three fc5-side heads
#pragma unroll
            for (int ob = 0; ob < NB5; ob++) {
                const f4 h = selu4(acc5[ob] + load_bias4(hd.bias5, ob, q_t, hd.nout5));
                if (hd.keep && g + r < G) hd.h5_out[((size_t)(g + r) * NB5 + ob) * 64 + lane_t] = h;
                const f4 W = hd.wp1[(size_t)ob * 64 + lane_t];
#pragma unroll
                for (int s = 0; s < 4; s++) hacc1[r] = mfma4(W[s], h[s], hacc1[r]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // surplus DMA pieces of the last stages
#pragma unroll
        for (int r = 0; r < GR; r++)
            if (g + r < G) heads_finish(hacc0[r], hacc1[r], hd.bb, hd.bz, hd.bt, hd.bl, n_cand, out16, g + r, lane_t);
        return;
    }
#pragma unroll
    for (int r = 0; r < GR; r++) {
        if (g + r >= G) break;
        f4 *op = out_tm + (((size_t)kz * G + (size_t)(g + r)) * NBT + (size_t)blockIdx.y * NB) * 64 + lane;
        if (KS > 1) {
#pragma unroll
            for (int ob = 0; ob < NB; ob++) op[ob * 64] = acc[r][ob];
            continue;
        }
        if constexpr (EPI == 2) {
            // the other three heads: contraction over this layer's output tiles, straight from the registers
            f4 hW[NB];
#pragma unroll
            for (int ob = 0; ob < NB; ob++) hW[ob] = load_frag(hd.wp1 + (size_t)ob * 64 + lane);
            f4 hacc1 = zero;
#pragma unroll
            for (int ob = 0; ob < NB; ob++) {
                const f4 b4 = load_bias4(bias, ob, q, nout);
                const f4 h = selu4(acc[r][ob] + b4);
                op[ob * 64] = h;                             // kept for cv_get_activation
                if (ob == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hW[0]) : : "memory");
                asm volatile("" : "+v"(hW[ob]) : : "memory");
#pragma unroll
                for (int s = 0; s < 4; s++) hacc1 = mfma4(hW[ob][s], h[s], hacc1);
            }
            heads_finish(hacc0, hacc1, hd.bb, hd.bz, hd.bt, hd.bl, hd.n, hd.out16, g + r, lane);
            continue;
        }
#pragma unroll
        for (int ob = 0; ob < NB; ob++) {
            if constexpr (EPI == 0) {
                const f4 b4 = load_bias4(bias, (int)blockIdx.y * NB + ob, q, nout);
                const f4 h = selu4(acc[r][ob] + b4);
                op[ob * 64] = h;
                if constexpr (NB == 7 && GR == 1) {
                    if (hd.drop.d4) {
                        f4 d, mk;
#pragma unroll
                        for (int s = 0; s < 4; s++) {
                            float x = h[s], k;
                            dropout_value(x, k, 16 * ((int)blockIdx.y * NB + ob) + 4 * s + q, hd.drop.nunits,
                                          hd.drop.cand0 + (int64_t)(g + r) * 16 + (lane & 15), hd.drop.rate, hd.drop.seed, hd.drop.step);
                            d[s] = x; mk[s] = k;
                        }
                        const size_t t = (size_t)(op - out_tm) + ob * 64;
                        reinterpret_cast<f4 *>(hd.drop.d4)[t] = d;
                        reinterpret_cast<f4 *>(hd.drop.amask)[t] = mk;
                    }
                }
            } else {
                f4 v = acc[r][ob];
                if (hd.dact) {               // data gradient times selu' of the layer below (a layer without pooling)
                    const f4 y = hd.dact[(op - out_tm) + ob * 64];
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] *= cv_selu_grad_from_out(y[k]);
                }
                if (hd.hg_g16) {             // + base head, * dropout factor, * selu'(fc4 output): b_head_dgrad_tm mode 1
                    const size_t t = (size_t)(op - out_tm) + ob * 64;
                    const f4 mk = hd.hg_mask[t], y = hd.hg_act[t];
                    const int64_t cand = (int64_t)(g + r) * 16 + (lane & 15);
                    const float *gi = hd.hg_g16 + (size_t)(cand < hd.hg_n ? cand : 0) * 16;
#pragma unroll
                    for (int s = 0; s < 4; s++) {
                        const int k = 16 * ((int)blockIdx.y * NB + ob) + 4 * s + q;
                        float a = 0.0f;
                        if (cand < hd.hg_n && k < hd.hg_K) {
#pragma unroll
                            for (int j = 0; j < 4; j++) a = __builtin_fmaf(gi[j], hd.hg_wb[(size_t)k * 4 + j], a);
                        }
                        float gact = v[s] + a;
                        gact *= mk[s];
                        v[s] = gact * cv_selu_grad_from_out(y[s]);
                    }
                }
                op[ob * 64] = v;
            }
        }
    }
}



// ---------------------------------------------------------------------------
// dense layer in output slabs with RAGGED waves (round 6): time proportional to the work at every batch size.
//
// dense_tm<7, 8> in three slabs hands every wave ONE group x the 7 output tiles of its slab, 8 waves to a workgroup: a
// workgroup is 14 tile-units of matrix work per SIMD (one tile-unit = one 16 x 16 output tile over all k = 288 x 4 MFMAs,
// 18 us of a SIMD) whatever the batch, so a launch costs ceil(workgroups / CUs) x 254 us: 768 groups (288 workgroups on
// 256 CUs) take the time of 1 365.  Here the (group, tile) pairs of a slab form ONE flat sequence u = group * NBS + tile,
// cut into equal pieces: the first four waves of a workgroup take `ca` consecutive pairs each, the last four `cb`
// (ca - cb <= 1; waves w and w + 4 share a SIMD, so every SIMD of the workgroup gets s = ca + cb tile-units, any s from 2 to
// 2 NBS).  The launcher picks s so that ceil(workgroups / CUs) x s is as close to 21 G / 1024 as it gets
// (dense_rag_shape).  A piece of c <= NBS pairs touches at most two groups: its first n0 tiles are tiles t0 .. of group
// g0, the rest tiles 0 .. of g0 + 1.  The accumulators are indexed by the POSITION in the piece (static registers), the
// LDS offset of a position's weight fragment is a wave-uniform scalar, and which group's activation fragment a position
// multiplies is decided at COMPILE time: the k loop exists once per (c, n0) -- a wave jumps to its copy before the loop
// (branches around single MFMA blocks cost the compiler's accumulator copies and 40 % of the kernel: first version of this
// kernel, profiles/r06/dense_rag_first_version.txt).  Weights through the same 3-slot LDS-DMA ring as dense_tm (one
// fragment per wave and k step), one barrier per k step.  Per output value the chain is dense_tm's: ascending k, + bias,
// SELU -- the same bits whatever the shape.  drop.d4 != NULL: the alpha-dropout of the value follows on the store (fc4 of
// a training pass, as dense_tm<7, 8>'s three-slab form does).
// ---------------------------------------------------------------------------
template <int NBS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 2) void dense_rag(const f4 *__restrict__ in_tm, int KB,
                                                                   const f4 *__restrict__ wp_all,
                                                                   const float *__restrict__ bias, int nout,
                                                                   f4 *__restrict__ out_tm, int G, int NBT, int ca, int cb,
                                                                   int wgs, int nslab, cv_dropout_args drop)
{
    static_assert(WAVES == 8 && NBS == 7, "one padded stage of WAVES fragments per k step, one DMA piece per wave; the dispatch below names 7 tiles");
    extern __shared__ __attribute__((aligned(16))) f4 ring[];
    constexpr int STAGE = WAVES * 64;            // f4 per stage (NBS real fragments + zero pad)
    constexpr int HW = WAVES / 2;
    // XCD-aware numbering of a one-dimensional grid (workgroup b runs on XCD b % 8, each XCD has its own L2): the nslab
    // workgroups that multiply the SAME activation fragments -- one per output slab -- are b = 8 (nslab t + y) + x % 8, i.e.
    // on one XCD and dispatched together, so that two of three reads of the layer's input hit that L2 (as a (pieces, slabs)
    // grid they ran whole launches apart: the 184 MB pool3 map of a 10 000-candidate pass came in from outside the L2 three
    // times).  A speed-only assumption: the values do not depend on it.
    const int xcd = blockIdx.x & 7, tq = blockIdx.x >> 3;
    const int slab = tq % nslab, bx = (tq / nslab) * 8 + xcd;
    if (bx >= wgs) return;                       // (padding of the grid to whole XCD rows: the whole workgroup leaves)
    const f4 *wp = wp_all + (size_t)slab * KB * STAGE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // this wave's piece of the slab's (group, tile) sequence
    const int u0 = bx * HW * (ca + cb) + (wid < HW ? wid * ca : HW * ca + (wid - HW) * cb);
    const int g0 = u0 / NBS, t0 = u0 - g0 * NBS;
    int c = wid < HW ? ca : cb;
    int n0 = c < NBS - t0 ? c : NBS - t0;
    if (g0 + 1 >= G) c = n0;                     // the piece ends with the batch
    if (g0 >= G) { c = 0; n0 = 0; }              // a spare wave: it still stages its fragment and takes part in the barriers
    c = __builtin_amdgcn_readfirstlane(c); n0 = __builtin_amdgcn_readfirstlane(n0);
    int tj[NBS];                                 // tile of position i (wave-uniform)
#pragma unroll
    for (int i = 0; i < NBS; i++) tj[i] = __builtin_amdgcn_readfirstlane(i < n0 ? t0 + i : (i < c ? i - n0 : 0));
    const int ga = g0 < G ? g0 : G - 1, gb = g0 + 1 < G ? g0 + 1 : G - 1;
    const unsigned bo0 = (unsigned)((((size_t)ga * KB) * 64 + lane) * sizeof(f4));
    const unsigned bo1 = (unsigned)((((size_t)gb * KB) * 64 + lane) * sizeof(f4));
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[NBS];
#pragma unroll
    for (int j = 0; j < NBS; j++) acc[j] = zero;
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) f4 *)ring;
    auto stage_async = [&](int kb, int slot) {       // one 1 KiB fragment per wave (see dense_tm)
        const f4 *gp = wp + ((size_t)kb * WAVES + wid) * 64 + lane;
        const unsigned ldst = ring_base + (unsigned)((slot * WAVES + wid) * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
    };
    auto load_frag_off = [&](unsigned byte_off) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(in_tm) : "memory");
        return v;
    };
    // the k loop for a piece of CC positions of which the first N0 belong to the first group (both compile-time).
    // Unrolled by three so that the ring slot and the activation registers of a step are static: the activation fragments
    // are fetched TWO steps ahead into three rotating register sets (a step of a short piece is ~0.25 us of MFMAs, less
    // than the L2 round trip of a fragment requested at its start), no copies, no address arithmetic in the loop.
    auto run = [&](auto CCc, auto N0c) {
        constexpr int CC = decltype(CCc)::value, N0 = decltype(N0c)::value;
        constexpr bool TWO = N0 < CC;
        const int K1 = KB - 1;
        stage_async(0, 0);
        stage_async(K1 < 1 ? K1 : 1, 1);
        f4 B0[3], B1[3];
#pragma unroll
        for (int r = 0; r < 3; r++) { B0[r] = zero; B1[r] = zero; }
        B0[0] = load_frag_off(bo0);
        if constexpr (TWO) B1[0] = load_frag_off(bo1);
        B0[1] = load_frag_off(bo0 + (unsigned)(K1 < 1 ? K1 : 1) * 1024u);
        if constexpr (TWO) B1[1] = load_frag_off(bo1 + (unsigned)(K1 < 1 ? K1 : 1) * 1024u);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(B0[0]), "+v"(B1[0]), "+v"(B0[1]), "+v"(B1[1]) : : "memory");
        __syncthreads();
        const f4 *wl = ring + lane;
        auto step = [&](auto Rc, int kb) {
            constexpr int R = decltype(Rc)::value, RN = (R + 2) % 3;
            const int k2 = kb + 2 < KB ? kb + 2 : K1;           // (clamped: the surplus loads of the last steps re-read valid data)
            B0[RN] = load_frag_off(bo0 + (unsigned)k2 * 1024u);
            if constexpr (TWO) B1[RN] = load_frag_off(bo1 + (unsigned)k2 * 1024u);
            stage_async(k2, RN);                 // slot (kb + 2) % 3 was last read in step kb - 1 (barrier passed)
            constexpr int AB = 3;                // weight fragments read ahead of their MFMAs
#pragma unroll
            for (int i = 0; i < CC; i += AB) {
                f4 A[AB];
#pragma unroll
                for (int j = 0; j < AB; j++)
                    if (i + j < CC) A[j] = wl[R * STAGE + tj[i + j] * 64];
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                    for (int j = 0; j < AB; j++)
                        if (i + j < CC) acc[i + j] = mfma4(A[j][s4], (i + j < N0 ? B0[R] : B1[R])[s4], acc[i + j]);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs above the wait
            // counted wait: this step's loads (1 or 2 fragments + the DMA piece: stage and fragments kb + 2) stay in flight;
            // what the previous step issued -- stage kb + 1, which the barrier publishes, and fragments kb + 1 -- has landed
            constexpr int R1 = (R + 1) % 3;
            if constexpr (TWO) asm volatile("s_waitcnt vmcnt(3)" : "+v"(B0[R1]), "+v"(B1[R1]) : : "memory");
            else asm volatile("s_waitcnt vmcnt(2)" : "+v"(B0[R1]) : : "memory");
            __syncthreads();      // (without it -- wrong results, a timing ceiling -- the kernel is 2-3 % shorter: the step is MFMA-bound)
        };
#pragma unroll 1
        for (int kb = 0; kb < KB; kb += 3) {
            step(std::integral_constant<int, 0>{}, kb);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < KB) step(std::integral_constant<int, 1>{}, kb + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 2 < KB) step(std::integral_constant<int, 2>{}, kb + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the surplus loads of the last steps
    };
#define CV_RAG_N0(CCV, N0V) case N0V: if constexpr (N0V <= CCV) run(std::integral_constant<int, CCV>{}, std::integral_constant<int, N0V>{}); break;
#define CV_RAG_C(CCV) case CCV: switch (n0) { CV_RAG_N0(CCV, 1) CV_RAG_N0(CCV, 2) CV_RAG_N0(CCV, 3) CV_RAG_N0(CCV, 4) CV_RAG_N0(CCV, 5) CV_RAG_N0(CCV, 6) CV_RAG_N0(CCV, 7) default: break; } break;
    switch (c) {
    CV_RAG_C(1) CV_RAG_C(2) CV_RAG_C(3) CV_RAG_C(4) CV_RAG_C(5) CV_RAG_C(6) CV_RAG_C(7)
    default: run(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); break;
    }
#undef CV_RAG_C
#undef CV_RAG_N0
    const int q = lane >> 4;
#pragma unroll
    for (int i = 0; i < NBS; i++) {
        if (i >= c) break;
        const int g = i < n0 ? g0 : g0 + 1;
        const int ob = slab * NBS + tj[i];
        const size_t t = ((size_t)g * NBT + ob) * 64 + lane;
        const f4 h = selu4(acc[i] + load_bias4(bias, ob, q, nout));
        out_tm[t] = h;
        if (drop.d4) {
            f4 d, mk;
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                float x = h[s4], k;
                dropout_value(x, k, 16 * ob + 4 * s4 + q, drop.nunits, drop.cand0 + (int64_t)g * 16 + (lane & 15), drop.rate,
                              drop.seed, drop.step);
                d[s4] = x; mk[s4] = k;
            }
            reinterpret_cast<f4 *>(drop.d4)[t] = d;
            reinterpret_cast<f4 *>(drop.amask)[t] = mk;
        }
    }
}

// ---------------------------------------------------------------------------
// fc4 data gradient FUSED with the max-pool backward + SELU' of conv3 (training step, full topology).
//   gF[k] = sum_j g4pre[j] W4[k][j]      (the gradient of the pooled conv3 map, k = flatten index (h, w, c))
//   gpre3 = unpool(gF) * selu'           (cv_unpool.hpp)
// As separate kernels (dense_tm EPI 1 + the element-wise pass) the 18.4 KB-per-candidate gradient map is written,
// read back together with the pre-pool activations, and written again: 0.9 MB of HBM traffic per group for zero
// FLOPs.  Pooling runs along positions only, so the work is cut by COLUMN (base w, tile nt) of the map instead of by
// slabs of output features: a workgroup owns WAVES * GR groups and one column; each wave keeps the NB fragments of
// its groups' g4pre in registers for the whole kernel (they are the B operands of every row) and walks the HO pooled
// rows in order -- per row NB x 4 MFMA steps per group on the row's weight fragments, streamed through the same 3-slot
// LDS-DMA ring as dense_tm (one barrier per row), then the row's gradient enters the P-row unpool window and one
// finished pre-activation gradient row leaves.  Per output value the contraction is the single ascending-j chain of
// dense_tm EPI 1.  Loads and DMA from inline asm; per iteration: GR stores (row r-1), 2 GR loads (pooled output and codes
// of row r), PER DMA pieces (weights of row r+2), one counted wait that leaves only the DMA pieces in flight.
// ---------------------------------------------------------------------------
#ifndef CV_DGRAD_KLATE
#define CV_DGRAD_KLATE 9            // head length of the second wave of a SIMD (measured: 9 against 15, step 2.109 against 2.115 ms)
#endif
template <int NB, int P, int WAVES, int GR>
__global__ __launch_bounds__(WAVES * 64, 2) void dense_dgrad_unpool(const f4 *__restrict__ g_tm, const f4 *__restrict__ wpr,
                                                                  const f4 *__restrict__ pooled, const u32x2 *__restrict__ codes,
                                                                  f4 *__restrict__ gpre, int G, int HO, int NT)
{
    extern __shared__ __attribute__((aligned(16))) f4 ring[];
    constexpr int NBP = (NB + WAVES - 1) / WAVES * WAVES;
    constexpr int STAGE = NBP * 64;
    constexpr int PER = NBP / WAVES;
    const int NCOL = 4 * NT;
    const int col = blockIdx.y, w = col / NT, nt = col % NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = (blockIdx.x * WAVES + wid) * GR;
    CV_STAMP_BEGIN
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) f4 *)ring;
    const f4 *wcol = wpr + (size_t)col * HO * STAGE;
    auto stage_async = [&](int r, int slot) {
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const f4 *gp = wcol + ((size_t)r * NBP + wid * PER + p) * 64 + lane;
            const unsigned ldst = ring_base + (unsigned)((slot * NBP + wid * PER + p) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
        }
    };
    auto load_f4 = [&](const f4 *ptr) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    // per-row traffic of the unpool window with SCALAR base addresses (group, column and row are wave-uniform) + one lane
    // offset for everything -- no 64-bit vector pointers live across the loop (used with one group per wave, see below).
    // (The bases are recomputed by scalar instructions right in front of these statements, and the compiler cannot see
    // that the asm is a memory instruction: the wait states it would insert are written out -- 5 between a scalar write
    // of an SGPR and a vector-memory instruction that uses it as address, 2 (gfx940 and later; 1 before) behind a store of
    // more than 8 bytes before its data registers may be overwritten.  With one wait state behind the store the very next
    // instruction -- the unpool arithmetic of the wave's second group -- rewrote half of the stored fragment: the step was
    // wrong, differently from run to run, and only tests/test_gpu_train_parity.py at 10 000+ candidates said so.)
    const unsigned lane16 = (unsigned)lane * 16u, lane8 = (unsigned)lane * 8u;
    auto load_f4_s = [&](const f4 *sbase) {
        f4 v;
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lane16), "s"(sbase) : "memory");
        return v;
    };
    auto load_u1_s = [&](const unsigned *sbase) {          // one dword of every lane's 8-byte code word
        unsigned v;
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(v) : "v"(lane8), "s"(sbase) : "memory");
        return v;
    };
    auto store_f4_s = [&](f4 *sbase, f4 v) {
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(lane16), "v"(v), "s"(sbase) : "memory");
    };
    auto load_u2 = [&](const u32x2 *ptr) {
        u32x2 v;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
        return v;
    };
    int gl[GR]; bool live[GR];
#pragma unroll
    for (int r = 0; r < GR; r++) { live[r] = g0 + r < G; gl[r] = live[r] ? g0 + r : G - 1; }
    // gridDim.z row parts (tiny batches: a shorter chain per workgroup): part z owns the OUTPUT rows [pa, pb) of the
    // HO + P - 1 and walks the windows [lo, hi] -- the P - 1 windows in front of its rows are recomputed (same values)
    const int HP = HO + P - 1;
    const int pa = HP * (int)blockIdx.z / (int)gridDim.z, pb = HP * ((int)blockIdx.z + 1) / (int)gridDim.z;
    const int lo = pa - (P - 1) > 0 ? pa - (P - 1) : 0;
    const int hi = pb - 1 < HO - 1 ? pb - 1 : HO - 1;
    stage_async(lo, 0);
    stage_async(lo + 1 <= hi ? lo + 1 : hi, 1);
    f4 B[GR][NB];
#pragma unroll
    for (int r = 0; r < GR; r++)
#pragma unroll
        for (int kb = 0; kb < NB; kb++) B[r][kb] = load_f4(g_tm + ((size_t)gl[r] * NB + kb) * 64 + lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < GR; r++)
#pragma unroll
        for (int kb = 0; kb < NB; kb++) asm volatile("" : "+v"(B[r][kb]));
    __syncthreads();
    unpool_col<P> U[GR];
#pragma unroll
    for (int r = 0; r < GR; r++) U[r].init();
    // Two groups per wave keep the round-3 form of this traffic: 64-bit vector pointers, the whole code word, a
    // compiler-visible store.  With the scalar bases below the kernel needs 232 registers instead of 244 and is 9 us faster
    // on its own (296 -> 287 us), but the STEP is 45 us slower (2.16 against 2.11 ms, same box, profiles/r04/
    // train_10000_timeline_{r03_head,scalar_addressing}.txt): at 232 registers the small kernels at the head of the side
    // stream fit next to this kernel's workgroups on a CU instead of queueing behind it, the side stream runs ahead, fc4's
    // weight gradient arrives before conv3's data gradient and takes the CUs from the data-gradient chain.  (Side-stream
    // priorities and three other enqueue orders did not restore the old schedule.)  One group per wave (small batches,
    // nothing to compete with): the scalar form, 0.606 -> 0.592 ms per step at 1 250.
    constexpr bool SCALAR_ADDR = GR == 1;
    const f4 *pp[GR]; const u32x2 *cp[GR]; f4 *op[GR];
#pragma unroll
    for (int r = 0; r < GR; r++) {
        pp[r] = pooled + ((size_t)gl[r] * HO * NCOL + col) * 64 + lane;
        cp[r] = codes + ((size_t)gl[r] * HO * NT + nt) * 64 + lane;
        op[r] = gpre + ((size_t)gl[r] * (HO + P - 1) * NCOL + col) * 64 + lane;
    }
    // lane 0's element of (group r, this column, row): scalar pointers
    auto pooled_at = [&](int r, int row) { return pooled + ((size_t)gl[r] * HO * NCOL + col + (size_t)row * NCOL) * 64; };
    auto gpre_at = [&](int r, int row) { return gpre + ((size_t)gl[r] * (HO + P - 1) * NCOL + col + (size_t)row * NCOL) * 64; };
    auto code_at = [&](int r, int row) {                   // the dword that holds base w's 16 code bits (cv_code16)
        return reinterpret_cast<const unsigned *>(codes + ((size_t)gl[r] * HO * NT + nt + (size_t)row * NT) * 64) + (w >> 1);
    };
    f4 acc[GR], yv[GR]; u32x2 cv[GR]; unsigned cs[GR];      // code words: whole (vector form) / the dword of base w (scalar form)
#pragma unroll
    for (int r = 0; r < GR; r++) { acc[r] = zero; yv[r] = zero; cv[r] = (u32x2){0u, 0u}; cs[r] = 0u; }
    // row `row` leaves the accumulators (a copy: the next row is already being multiplied): into the unpool window, one
    // finished row out
    auto finish_row = [&](int row, const f4 (&done)[GR]) {
#pragma unroll
        for (int r = 0; r < GR; r++) {
            const unsigned cw = SCALAR_ADDR ? cs[r] : (w < 2 ? cv[r][0] : cv[r][1]);      // (read here, behind the counted wait)
            U[r].push(done[r], yv[r], (cw >> (16 * (w & 1))) & 0xFFFFu);
            const f4 o = U[r].emit();
            if (live[r] && row >= pa) {                  // (older than the DMA pieces the counted wait leaves in flight)
                if constexpr (SCALAR_ADDR) store_f4_s(gpre_at(r, row), o);
                else op[r][(size_t)row * NCOL * 64] = o;
            }
        }
    };
    // the MFMAs of the k fragments [k0, k1) of the current row
    // the MFMAs of the k fragments [k0, k1) of the current row.  (Measured, round 4: the reads of fragments kb + 2, kb + 3
    // issued from inline asm BEFORE the MFMAs of kb, kb + 1 with counted lgkmcnt waits -- hipcc sinks every ds_read to
    // just in front of its MFMAs, "2 reads, wait, 8 MFMAs, wait, 8 MFMAs" -- changed nothing: 295.1 -> 294.1 us, the
    // partner wave covers the LDS round trips; profiles/r04/lib_ab_dgrad_lds_read_ahead.txt.  Nor do the register hops
    // hipcc makes the two accumulators take (destination quad != addend quad, padded with s_nop 4..7 in front of the
    // next LDS read) cost anything measurable: with the MFMAs issued from inline asm on tied registers the stream is
    // clean and the kernel no faster, 292 against 287 us -- and wrong, the compiler no longer pads the hazards around
    // instructions it cannot see.)
    auto multiply = [&](const f4 *wl, int k0, int k1) {
#pragma unroll
        for (int kb = k0; kb < k1; kb += 3) {
            f4 A[3];
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (kb + j < k1) A[j] = wl[(kb + j) * 64];
#pragma unroll
            for (int j = 0; j < 3; j++)
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                    for (int r = 0; r < GR; r++)
                        if (kb + j < k1) acc[r] = mfma4(A[j][s4], B[r][kb + j][s4], acc[r]);
        }
    };
    // A row opens with MFMAs, not with the bookkeeping of the row before: behind a barrier both waves of a SIMD are at
    // the same place, and ~150 vector / scalar / memory instructions each (unpool window, store, this row's loads, the
    // DMA pieces) in front of the first MFMA left the matrix pipe idle for about a tenth of a row.  Now a wave multiplies
    // a HEAD of the row's fragments first; the finished row of the previous iteration (its accumulators live on in
    // `done`), this row's loads and the DMA pieces follow in that order -- the order the counted wait below relies on --,
    // then the rest of the fragments.  The two waves that share a SIMD (waves w and w + WAVES/2 of a workgroup) take
    // heads of different length, 3 and 9 of the 21 fragments, so that one of them always has MFMAs for the pipe while the
    // other does its bookkeeping.  Same chain per value.
    constexpr int KEARLY = NB >= 6 ? 3 : NB, KLATE = NB >= 18 ? CV_DGRAD_KLATE : KEARLY;
    const bool late = wid >= WAVES / 2;
    int slot = 0;
    CV_PHASE_BEGIN
#pragma unroll 1
    for (int row = lo; row <= hi; row++) {
        f4 done[GR];
#pragma unroll
        for (int r = 0; r < GR; r++) { done[r] = acc[r]; acc[r] = zero; }
        const f4 *wl = ring + slot * STAGE + lane;
        const int rs = row + 2 <= hi ? row + 2 : hi;
        int wslot = slot + 2; if (wslot >= 3) wslot -= 3;
        auto bookkeeping = [&]() {
            if (row > lo) finish_row(row - 1, done);
#pragma unroll
            for (int r = 0; r < GR; r++) {
                if constexpr (SCALAR_ADDR) {
                    yv[r] = load_f4_s(pooled_at(r, row));
                    cs[r] = load_u1_s(code_at(r, row));
                } else {
                    yv[r] = load_f4(pp[r] + (size_t)row * NCOL * 64);
                    cv[r] = load_u2(cp[r] + (size_t)row * NT * 64);
                }
            }
            stage_async(rs, wslot);
        };
        multiply(wl, 0, KEARLY);
        if (!late) bookkeeping();
        if constexpr (KLATE > KEARLY) multiply(wl, KEARLY, KLATE);
        if (late) bookkeeping();
        multiply(wl, KLATE, NB);
        __builtin_amdgcn_sched_barrier(0);
        CV_PHASE(0);                                        // (development probe: cycles up to here = issue of the row's work)
        // counted wait: only this iteration's PER DMA pieces (weights of row + 2) stay in flight -- VMEM operations
        // complete in order, and the pieces are the newest ones; the loads of this row and the store of the previous
        // one are done.  The barrier then publishes the weights of row + 1.
        if constexpr (PER == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if constexpr (PER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int r = 0; r < GR; r++) { asm volatile("" : "+v"(yv[r])); asm volatile("" : "+v"(cv[r])); asm volatile("" : "+v"(cs[r])); }
        CV_PHASE(1);                                        // ... waiting for this wave's loads / the DMA pieces of the next row
        __syncthreads();
        CV_PHASE(2);                                        // ... waiting for the other waves at the barrier
        slot = slot + 1 == 3 ? 0 : slot + 1;
    }
    CV_PHASE_END(GR == 2, wid);
    finish_row(hi, acc);
    if (hi == HO - 1) {                                    // the last P - 1 output rows start no window
        for (int row = HO; row < pb; row++) {
#pragma unroll
            for (int r = 0; r < GR; r++) {
                U[r].push_none();
                const f4 o = U[r].emit();
                if (live[r] && row >= pa) {
                    if constexpr (SCALAR_ADDR) store_f4_s(gpre_at(r, row), o);
                    else op[r][(size_t)row * NCOL * 64] = o;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // surplus DMA pieces of the last rows, the stores
    CV_STAMP_END(true, 4);
}

// ---------------------------------------------------------------------------
// dense layer for FEW groups (a predict() call of the reference's batch of 1 000 is 63 groups): dense_tm streams the
// weight matrix through an LDS ring with one workgroup barrier per k fragment -- ~0.9 us per step whatever the
// batch, 254 us for fc4's 288 dependent steps, half of a small call.  Here nothing is shared: ONE WAVE owns a
// (group, slab of NBW output fragments) pair and reads its operands straight from L2 -- the activation fragment
// and NBW weight fragments per step, D steps ahead through a register ring (the loop is unrolled by D, so slot
// indices are constants and the compiler's counted vmcnt waits leave the younger loads in flight).  A step is
// NBW x 4 MFMAs with no barrier and no LDS round trip.  The contraction is the same single ascending-k chain per
// output value: bit-identical to dense_tm.  Weights: [slab][kb][NBW][64] fragments (pack_dense_slabs).
// ---------------------------------------------------------------------------
template <int NBW, int D, int EPI = 0>
__global__ __launch_bounds__(256) void dense_small(const f4 *__restrict__ in_tm, int KB, const f4 *__restrict__ wp_all,
                                                    const float *__restrict__ bias, int nout, f4 *__restrict__ out_tm,
                                                    int G, int NSLAB, int NBT)
{
    // NBT = output fragments per group (<= NBW * NSLAB: the last slab may be padded with zero fragments)
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= G * NSLAB) return;
    const int g = unit / NSLAB, slab = unit % NSLAB;
    const f4 *bp = in_tm + (size_t)g * KB * 64 + lane;
    const f4 *wp = wp_all + (size_t)slab * KB * (NBW * 64) + lane;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[NBW];
#pragma unroll
    for (int j = 0; j < NBW; j++) acc[j] = zero;
    f4 A[D][NBW], B[D];
    // KB is a multiple of D (launcher): the loop body never needs a bounds test, and the operand pointers just advance
    auto fetch = [&](const f4 *pb, const f4 *pw, int d) {
        B[d] = pb[(size_t)d * 64];
#pragma unroll
        for (int j = 0; j < NBW; j++) A[d][j] = pw[((size_t)d * NBW + j) * 64];
    };
    auto step = [&](int d) {
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int j = 0; j < NBW; j++) acc[j] = mfma4(A[d][j][s4], B[d][s4], acc[j]);
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(bp, wp, d);
#pragma unroll 1
    for (int kb0 = D; kb0 < KB; kb0 += D) {
        bp += (size_t)D * 64; wp += (size_t)D * NBW * 64;
#pragma unroll
        for (int d = 0; d < D; d++) {
            step(d);
            fetch(bp, wp, d);
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) step(d);
    const int q = lane >> 4;
    f4 *op = out_tm + ((size_t)g * NBT + (size_t)slab * NBW) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NBW; j++) {
        if (slab * NBW + j >= NBT) break;
        if constexpr (EPI == 0) op[j * 64] = selu4(acc[j] + load_bias4(bias, slab * NBW + j, q, nout));
        else op[j * 64] = acc[j];
    }
}

// second pass of a k-split dense layer: out = selu(sum_z part[z] + bias), ranges added in ascending z; with dr.d4 set
// (fc4 of a training pass) the alpha-dropout of the value follows in the same thread -- dropout_tm's arithmetic, one launch less
__global__ void dense_ksum(const f4 *__restrict__ part, int KS, int G, int NBT, const float *__restrict__ bias, int nout,
                           f4 *__restrict__ out_tm, cv_dropout_args dr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)G * NBT * 64;
    if (t >= per) return;
    f4 v = part[t];
    for (int z = 1; z < KS; z++) v += part[(size_t)z * per + t];
    const int lane = (int)(t & 63), ob = (int)((t >> 6) % NBT);
    const f4 h = selu4(v + load_bias4(bias, ob, lane >> 4, nout));
    out_tm[t] = h;
    if (dr.d4) {
        const int64_t g = t / ((int64_t)64 * NBT);
        f4 d, mk;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            float x = h[s], k;
            dropout_value(x, k, 16 * ob + 4 * s + (lane >> 4), dr.nunits, dr.cand0 + g * 16 + (lane & 15), dr.rate, dr.seed, dr.step);
            d[s] = x; mk[s] = k;
        }
        reinterpret_cast<f4 *>(dr.d4)[t] = d;
        reinterpret_cast<f4 *>(dr.amask)[t] = mk;
    }
}

// training: the same two tile products, stored as pre-activations (+ bias) in candidate-major [n][16] order
// (base 0..3 | zygosity 4..5 | type 6..9 | length 10..15); loss and head gradients follow in t_heads_loss
__global__ __launch_bounds__(256) void heads_pre_tm(const f4 *__restrict__ h4, const f4 *__restrict__ h5, int NB4,
                                                     int NB5, const f4 *__restrict__ wp0, const f4 *__restrict__ wp1,
                                                     const float *__restrict__ bb, const float *__restrict__ bz,
                                                     const float *__restrict__ bt, const float *__restrict__ bl,
                                                     int64_t n, float *__restrict__ pre16, int G)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int c = lane & 15, q = lane >> 4;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 a0 = zero, a1 = zero;
    const f4 *p4 = h4 + (size_t)g * NB4 * 64 + lane;
    const f4 *p5 = h5 + (size_t)g * NB5 * 64 + lane;
#pragma unroll 3
    for (int kb = 0; kb < NB4; kb++) {
        const f4 B = p4[(size_t)kb * 64];
        const f4 A = wp0[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) a0 = mfma4(A[s], B[s], a0);
    }
#pragma unroll 3
    for (int kb = 0; kb < NB5; kb++) {
        const f4 B = p5[(size_t)kb * 64];
        const f4 A = wp1[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) a1 = mfma4(A[s], B[s], a1);
    }
    const int64_t cand = (int64_t)g * 16 + c;
    if (cand >= n) return;
    float *o = pre16 + (size_t)cand * 16;
    // rows of the second tile: q 0 = zygosity (2), q 1 = type (4), q 2 = length 0..3, q 3 = length 4..5
    if (q == 0) {
        *reinterpret_cast<float4 *>(o) = make_float4(a0[0] + bb[0], a0[1] + bb[1], a0[2] + bb[2], a0[3] + bb[3]);
        o[4] = a1[0] + bz[0]; o[5] = a1[1] + bz[1];
    } else if (q == 1) {
        o[6] = a1[0] + bt[0]; o[7] = a1[1] + bt[1]; o[8] = a1[2] + bt[2]; o[9] = a1[3] + bt[3];
    } else if (q == 2) {
        o[10] = a1[0] + bl[0]; o[11] = a1[1] + bl[1]; o[12] = a1[2] + bl[2]; o[13] = a1[3] + bl[3];
    } else {
        o[14] = a1[0] + bl[4]; o[15] = a1[1] + bl[5];
    }
}


// ---------------------------------------------------------------------------
// Heads of the TRAINING pass in one kernel (was: heads_pre_tm, then the loss kernel, then the head data-gradient pass):
//   1. the two tile products (base head over the dropped-out fc4 output, zygosity / type / length heads over fc5) on the
//      matrix cores, one wave per group, as heads_pre_tm;
//   2. the 16 pre-activations of the group's 16 candidates through LDS to a (candidate, head) lane mapping: losses
//      (v3.py:140-149: squared error of the sigmoid head, cross-entropy of softmax(selu(.) + 1e-10) for the others) and
//      the gradients w.r.t. the pre-activations, written to g16 [n][16] (the heads' weight gradients and the base
//      head's data gradient read them later) and kept in LDS;
//   3. the data gradient of the three fc5-side heads, times selu'(fc5 output) -- the fc5 fragments are still in the
//      registers they were loaded into for step 1 -- straight into the tile-major pre-activation gradient of fc5.
// The arithmetic per value is that of the three kernels it replaces (same order): same bits.
// ---------------------------------------------------------------------------
template <int NB5>
__global__ __launch_bounds__(256) void heads_train_tm(const f4 *__restrict__ d4, const f4 *__restrict__ h5, int NB4,
                                                       const f4 *__restrict__ wp0, const f4 *__restrict__ wp1,
                                                       const float *__restrict__ bb, const float *__restrict__ bz,
                                                       const float *__restrict__ bt, const float *__restrict__ bl,
                                                       const float *__restrict__ wz, const float *__restrict__ wt,
                                                       const float *__restrict__ wl, int K5, const float *__restrict__ y,
                                                       int64_t n, int want_grad, float *__restrict__ g16,
                                                       f4 *__restrict__ g5pre_tm, double *__restrict__ loss_rows, int G,
                                                       const float *__restrict__ w12)
{
    __shared__ float sh[4][16][17];
    __shared__ double part[4][4];          // [wave][head]: the four loss sums of a wave's 16 candidates
    __shared__ __attribute__((aligned(16))) float shw[NB5 * 16][12];   // fc5-side head weights of a unit side by side: zygosity 2 | type 4 | length 6
    if (g5pre_tm && want_grad) {
        // (w12: the same [k][12] array packed once per weight change -- one coalesced copy instead of nine dependent
        // strided loads per thread with a division each, which were ~10 us of this kernel at any batch)
        for (int i = threadIdx.x; i < NB5 * 16 * 3; i += 256)
            reinterpret_cast<f4 *>(&shw[0][0])[i] = reinterpret_cast<const f4 *>(w12)[i];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x * 4 + wave;
    const bool live = g < G;
    const int gc = live ? g : G - 1;
    const int c = lane & 15, q = lane >> 4;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 a0 = zero, a1 = zero;
    const f4 *p4 = d4 + (size_t)gc * NB4 * 64 + lane;
    const f4 *p5 = h5 + (size_t)gc * NB5 * 64 + lane;
#pragma unroll 3
    for (int kb = 0; kb < NB4; kb++) {
        const f4 B = p4[(size_t)kb * 64];
        const f4 A = wp0[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) a0 = mfma4(A[s], B[s], a0);
    }
    f4 H5[NB5];
#pragma unroll
    for (int kb = 0; kb < NB5; kb++) {
        H5[kb] = p5[(size_t)kb * 64];
        const f4 A = wp1[(size_t)kb * 64 + lane];
#pragma unroll
        for (int s = 0; s < 4; s++) a1 = mfma4(A[s], H5[kb][s], a1);
    }
    float (*S)[17] = sh[wave];
    // rows of the second tile: q 0 = zygosity (2), q 1 = type (4), q 2 = length 0..3, q 3 = length 4..5
    if (q == 0) {
        S[c][0] = a0[0] + bb[0]; S[c][1] = a0[1] + bb[1]; S[c][2] = a0[2] + bb[2]; S[c][3] = a0[3] + bb[3];
        S[c][4] = a1[0] + bz[0]; S[c][5] = a1[1] + bz[1];
    } else if (q == 1) {
        S[c][6] = a1[0] + bt[0]; S[c][7] = a1[1] + bt[1]; S[c][8] = a1[2] + bt[2]; S[c][9] = a1[3] + bt[3];
    } else if (q == 2) {
        S[c][10] = a1[0] + bl[0]; S[c][11] = a1[1] + bl[1]; S[c][12] = a1[2] + bl[2]; S[c][13] = a1[3] + bl[3];
    } else {
        S[c][14] = a1[0] + bl[4]; S[c][15] = a1[1] + bl[5];
    }
    __syncthreads();
    {   // losses and gradients: lane -> (candidate lane >> 2 of the group, head lane & 3)
        const int cc = lane >> 2, j = lane & 3;
        const int64_t cand = (int64_t)g * 16 + cc;
        double l = 0.0;
        if (live && cand < n) {
            const float *yi = y + (size_t)cand * 16;
            float *gl = S[cc];
            float *go = g16 + (size_t)cand * 16;
            if (j == 0) {
                float v[4];
                for (int k = 0; k < 4; k++) v[k] = gl[k];
                for (int k = 0; k < 4; k++) {
                    float sg = cvm::sigmoid(v[k]);
                    float d = sg - yi[k];
                    l += (double)d * d;
                    if (want_grad) { const float gr = 2.0f * d * sg * (1.0f - sg); gl[k] = gr; go[k] = gr; }
                }
            } else {
                const int off = j == 1 ? 4 : (j == 2 ? 6 : 10);
                const int cnt = j == 1 ? 2 : (j == 2 ? 4 : 6);
                float v[6], lg[6], p[6];
                float mx = -__builtin_inff();
                for (int k = 0; k < cnt; k++) { v[k] = gl[off + k]; lg[k] = cvm::selu(v[k]) + 1e-10f; mx = fmaxf(mx, lg[k]); }
                float se = 0.0f, ysum = 0.0f;
                for (int k = 0; k < cnt; k++) { p[k] = cvm::expf_fixed(lg[k] - mx); se += p[k]; ysum += yi[off + k]; }
                float lse = mx + logf(se);
                for (int k = 0; k < cnt; k++) {
                    l += -(double)yi[off + k] * (double)(lg[k] - lse);
                    if (want_grad) { const float gr = (p[k] / se * ysum - yi[off + k]) * cvm::selu_grad(v[k]); gl[off + k] = gr; go[off + k] = gr; }
                }
            }
        }
        // No atomics: the 16 candidates of a wave are added in a fixed tree (lanes with the same head: xor 4, 8, 16, 32),
        // the four waves of the block in order, and the block's four sums go to ITS row of loss_rows -- t_loss_finish adds
        // the rows in a fixed order.  The loss sums of a step are the same bits from run to run, like its gradients.
#pragma unroll
        for (int d = 4; d < 64; d <<= 1) l += __shfl_xor(l, d);
        if (lane < 4) part[wave][lane] = l;
    }
    __syncthreads();
    if (threadIdx.x < 4)
        loss_rows[(size_t)blockIdx.x * 4 + threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    if (!g5pre_tm || !want_grad || !live) return;
    // fc5-side head data gradients (zygosity, type, length; k = fc5 unit), times selu'(fc5 output)
    const bool cand_ok = (int64_t)g * 16 + c < n;
    const float *gi = S[c];
#pragma unroll
    for (int kb = 0; kb < NB5; kb++) {
        f4 o;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = 16 * kb + 4 * s + q;
            float acc = 0.0f;
            if (cand_ok && k < K5) {                 // (weights from LDS: staged at the top, published by the barriers above)
                const f4 *wk4 = reinterpret_cast<const f4 *>(shw[k]);      // three 16-byte reads instead of twelve words
                const f4 w0 = wk4[0], w1 = wk4[1], w2 = wk4[2];
                const float wk[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
                for (int jj = 0; jj < 12; jj++) acc = __builtin_fmaf(gi[4 + jj], wk[jj], acc);
            }
            o[s] = acc * cv_selu_grad_from_out(H5[kb][s]);
        }
        g5pre_tm[((size_t)g * NB5 + kb) * 64 + lane] = o;
    }
}

// ---------------------------------------------------------------------------
// Tail of the TRAINING forward pass at tiny batches in one kernel (full topology; round 5): second pass of the k-split
// fc4 (dense_ksum: partial sums added in order, bias, SELU, alpha-dropout), fc5 (dense_small<4, 7>), and the heads
// of the training pass (heads_train_tm: products, losses, head gradients, fc5-side data gradient times selu').  As
// three launches they were 8 + 12 + 28 us of a 14-kernel chain at 79 groups (a rank's share of train.py's batch on
// 8 GPUs), each a latency chain of global loads on a fraction of the chip; here a workgroup of eight waves owns one
// group of 16 candidates and the operands of every step come from LDS or registers:
//   1. wave w sums fragments w, w + 8, w + 16 of the eight k ranges, + bias, SELU -> fc4 output (stored: the backward
//      pass takes selu' from it), dropout -> d4 / mask (stored) and d4 into LDS;
//   2. waves 0..2: one slab of 4 fc5 tiles each over the 21 d4 fragments (weights from L2 through a register ring,
//      dense_small's loop), bias + SELU -> fc5 output (stored, LDS, and kept in registers); wave 3: the base head's
//      product over the same fragments;
//   3. wave 0: the other heads' product from LDS, then -- lane = (candidate, head) -- losses and head gradients;
//   4. waves 0..2: the fc5-side data gradient of their own tiles times selu'(fc5 output) from the registers of step 2.
// Arithmetic and order per value are those of the three kernels (same bits); the loss sums leave as ONE ROW PER GROUP
// (heads_train_tm: one per four groups), which t_loss_header adds in its fixed order.
// ---------------------------------------------------------------------------
// NWV waves per workgroup; PART: the fc4 output arrives as k-range partial sums (tiny batches) -- else (larger batches,
// train_sched bit 10) fc4's own kernel has stored the dropped-out output (dr.d4) and step 1 only brings it into LDS.
// (four-wave form: three workgroups per CU -- 168 registers, 40 dwords of them spilled -- so that the 625 workgroups of
// train.py's batch are ONE round on 256 CUs instead of two: 52.7 -> 40.6 us, the step 2.060 -> 2.054 ms, 12 288: 2.574 ->
// 2.559; profiles/r06/train_tail_occupancy_ab.txt)
template <int NB4, int NB5, int NWV, bool PART>
__global__ __launch_bounds__(NWV * 64, (PART ? 1 : 3)) void train_tail_tm(const f4 *__restrict__ part, int KS, int G, const float *__restrict__ bias4,
                                                      int nout4, f4 *__restrict__ h4_out, cv_dropout_args dr,
                                                      const f4 *__restrict__ w5s, const float *__restrict__ bias5, int nout5,
                                                      f4 *__restrict__ h5_out, const f4 *__restrict__ wp0,
                                                      const f4 *__restrict__ wp1, const float *__restrict__ bb,
                                                      const float *__restrict__ bz, const float *__restrict__ bt,
                                                      const float *__restrict__ bl, const float *__restrict__ wz,
                                                      const float *__restrict__ wt, const float *__restrict__ wl,
                                                      const float *__restrict__ y, int64_t n, int want_grad,
                                                      float *__restrict__ g16, f4 *__restrict__ g5pre_tm,
                                                      double *__restrict__ loss_rows, const float *__restrict__ w12)
{
    constexpr int NBW = 4, D = 7;                 // fc5 slab width and operand ring depth of dense_small<4, 7>
    static_assert(NB4 % D == 0 && NB5 <= 3 * NBW, "three slabs of four fc5 tiles, 21 k fragments in rings of 7");
    __shared__ __attribute__((aligned(16))) f4 sd4[NB4][64];
    __shared__ __attribute__((aligned(16))) f4 sh5[NB5][64];
    __shared__ __attribute__((aligned(16))) f4 sa0[64];
    __shared__ float S[16][17];
    __shared__ __attribute__((aligned(16))) float shw[NB5 * 16][12];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x;
    const int c = lane & 15, q = lane >> 4;
    const int K5 = nout5;
    const bool grads = g5pre_tm && want_grad;
    __shared__ float sy[16][16];                  // the group's label rows (one coalesced load)
    if (threadIdx.x < 256) {
        const int64_t yc = (int64_t)g * 16 + (threadIdx.x >> 4);
        sy[threadIdx.x >> 4][threadIdx.x & 15] = yc < n ? y[(size_t)yc * 16 + (threadIdx.x & 15)] : 0.0f;
    }
    (void)wz; (void)wt; (void)wl;
    // ---- 1. fc4: k ranges added in ascending order, + bias, SELU, alpha-dropout (dense_ksum).  Eight waves: at most
    // three fragments each, the partial sums of all of them in flight at once (the step is a latency chain: 79 workgroups
    // on 256 CUs)
    constexpr int NW = NWV, MAXF = (NB4 + NW - 1) / NW;
    const int64_t per = (int64_t)G * NB4 * 64;
    // (sixteen ranges instead of eight were measured: the k-range kernel in front takes the same 40 us -- it is not bound by
    // its number of barrier steps -- and the step does not move: profiles/r05/step_ab_session7_join_latefc4_kranges.txt)
    constexpr int KSF = PART ? CV_DENSE_KSPLIT : 1;
    f4 pz[MAXF][KSF];
    const bool fast = !PART || KS == KSF;
    if constexpr (!PART) {
#pragma unroll
        for (int i = 0; i < MAXF; i++) {
            const int ob = wave + NW * i;
            if (ob < NB4) pz[i][0] = reinterpret_cast<const f4 *>(dr.d4)[((int64_t)g * NB4 + ob) * 64 + lane];
        }
    } else if (fast) {
#pragma unroll
        for (int i = 0; i < MAXF; i++) {
            const int ob = wave + NW * i;
            if (ob < NB4) {
                const int64_t t = ((int64_t)g * NB4 + ob) * 64 + lane;
#pragma unroll
                for (int z = 0; z < KSF; z++) pz[i][z] = part[(size_t)z * per + t];
            }
        }
    }
    // Operands of step 2 that depend on nothing computed here are requested NOW, behind the partial sums: the first ring of
    // fc5 weight fragments (waves 0..2), all of the base head's (wave 3), the other heads' (wave 0) -- they land while
    // step 1 computes, instead of opening step 2 with a round trip to L2 each
    // (ONE register array for both roles -- wave 3's 21 base-head fragments live where waves 0..2 keep their ring of
    // 7 x 4: as two arrays the kernel spilled)
    static_assert(NB4 <= D * NBW, "the base head's fragments fit the ring's registers");
    const f4 *wp5 = w5s + (size_t)(wave < 3 ? wave : 0) * NB4 * (NBW * 64) + lane;
    f4 A[D][NBW];
    {
        const f4 *src = wave == 3 ? wp0 + lane : wp5;
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int j = 0; j < NBW; j++) {
                const int f = d * NBW + j;
                if (wave <= 3 && (wave < 3 || f < NB4)) A[d][j] = src[(size_t)f * 64];      // (wave-uniform)
            }
    }
    if (grads) {
        for (int i = threadIdx.x; i < NB5 * 16 * 3; i += NW * 64)
            reinterpret_cast<f4 *>(&shw[0][0])[i] = reinterpret_cast<const f4 *>(w12)[i];
    }
#pragma unroll
    for (int i = 0; i < MAXF; i++) {
        const int ob = wave + NW * i;
        if (ob >= NB4) break;
        const int64_t t = ((int64_t)g * NB4 + ob) * 64 + lane;
        if constexpr (!PART) { sd4[ob][lane] = pz[i][0]; continue; }
        f4 v;
        if (fast) {
            v = pz[i][0];
#pragma unroll
            for (int z = 1; z < KSF; z++) v += pz[i][z];
        } else {
            v = part[t];
            for (int z = 1; z < KS; z++) v += part[(size_t)z * per + t];
        }
        const f4 h = selu4(v + load_bias4(bias4, ob, q, nout4));
        h4_out[t] = h;
        f4 d, mk;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            float x = h[s4], k;
            dropout_value(x, k, 16 * ob + 4 * s4 + q, dr.nunits, dr.cand0 + (int64_t)g * 16 + c, dr.rate, dr.seed, dr.step);
            d[s4] = x; mk[s4] = k;
        }
        reinterpret_cast<f4 *>(dr.d4)[t] = d;
        reinterpret_cast<f4 *>(dr.amask)[t] = mk;
        sd4[ob][lane] = d;
    }
    __syncthreads();
    // ---- 2. fc5 slabs (waves 0..2) and the base head (wave 3)
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 H5[NBW];                                   // this wave's fc5 output tiles (waves 0..2)
#pragma unroll
    for (int j = 0; j < NBW; j++) H5[j] = zero;
    f4 W1[NB5];                                   // wave 0: the fc5-side heads' fragments, landing under its fc5 slab
    if (wave == 0) {
#pragma unroll
        for (int kb = 0; kb < NB5; kb++) W1[kb] = wp1[(size_t)kb * 64 + lane];
    }
    if (wave < 3) {
        const int slab = wave;
        const f4 *wp = wp5;
        f4 acc[NBW];
#pragma unroll
        for (int j = 0; j < NBW; j++) acc[j] = zero;
        auto fetch = [&](const f4 *pw, int d) {
#pragma unroll
            for (int j = 0; j < NBW; j++) A[d][j] = pw[((size_t)d * NBW + j) * 64];
        };
        auto step = [&](int kb, int d) {
            const f4 B = sd4[kb][lane];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                for (int j = 0; j < NBW; j++) acc[j] = mfma4(A[d][j][s4], B[s4], acc[j]);
        };
#pragma unroll 1
        for (int kb0 = D; kb0 < NB4; kb0 += D) {          // (the first ring was requested at the top of the kernel)
            wp += (size_t)D * NBW * 64;
#pragma unroll
            for (int d = 0; d < D; d++) {
                step(kb0 - D + d, d);
                fetch(wp, d);
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) step(NB4 - D + d, d);
#pragma unroll
        for (int j = 0; j < NBW; j++) {
            const int ob = slab * NBW + j;
            if (ob >= NB5) break;
            H5[j] = selu4(acc[j] + load_bias4(bias5, ob, q, nout5));
            h5_out[((size_t)g * NB5 + ob) * 64 + lane] = H5[j];
            sh5[ob][lane] = H5[j];
        }
    } else if (wave == 3) {
        f4 a0 = zero;
#pragma unroll
        for (int kb = 0; kb < NB4; kb++) {
            const f4 B = sd4[kb][lane];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) a0 = mfma4(A[kb / NBW][kb % NBW][s4], B[s4], a0);
        }
        sa0[lane] = a0;
    }
    __syncthreads();
    // ---- 3. the fc5-side heads' product, losses and head gradients (wave 0; heads_train_tm)
    if (wave == 0) {
        f4 a1 = zero;
#pragma unroll
        for (int kb = 0; kb < NB5; kb++) {
            const f4 B = sh5[kb][lane];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) a1 = mfma4(W1[kb][s4], B[s4], a1);
        }
        const f4 a0 = sa0[lane];
        // rows of the second tile: q 0 = zygosity (2), q 1 = type (4), q 2 = length 0..3, q 3 = length 4..5
        if (q == 0) {
            S[c][0] = a0[0] + bb[0]; S[c][1] = a0[1] + bb[1]; S[c][2] = a0[2] + bb[2]; S[c][3] = a0[3] + bb[3];
            S[c][4] = a1[0] + bz[0]; S[c][5] = a1[1] + bz[1];
        } else if (q == 1) {
            S[c][6] = a1[0] + bt[0]; S[c][7] = a1[1] + bt[1]; S[c][8] = a1[2] + bt[2]; S[c][9] = a1[3] + bt[3];
        } else if (q == 2) {
            S[c][10] = a1[0] + bl[0]; S[c][11] = a1[1] + bl[1]; S[c][12] = a1[2] + bl[2]; S[c][13] = a1[3] + bl[3];
        } else {
            S[c][14] = a1[0] + bl[4]; S[c][15] = a1[1] + bl[5];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // losses and gradients: lane -> (candidate lane >> 2 of the group, head lane & 3)
        const int cc = lane >> 2, j = lane & 3;
        const int64_t cand = (int64_t)g * 16 + cc;
        double l = 0.0;
        if (cand < n) {
            const float *yi = sy[cc];
            float *gl = S[cc];
            float *go = g16 + (size_t)cand * 16;
            if (j == 0) {
                float v[4];
                for (int k = 0; k < 4; k++) v[k] = gl[k];
                for (int k = 0; k < 4; k++) {
                    float sg = cvm::sigmoid(v[k]);
                    float dd = sg - yi[k];
                    l += (double)dd * dd;
                    if (want_grad) { const float gr = 2.0f * dd * sg * (1.0f - sg); gl[k] = gr; go[k] = gr; }
                }
            } else {
                const int off = j == 1 ? 4 : (j == 2 ? 6 : 10);
                const int cnt = j == 1 ? 2 : (j == 2 ? 4 : 6);
                float v[6], lg[6], p[6];
                float mx = -__builtin_inff();
                for (int k = 0; k < cnt; k++) { v[k] = gl[off + k]; lg[k] = cvm::selu(v[k]) + 1e-10f; mx = fmaxf(mx, lg[k]); }
                float se = 0.0f, ysum = 0.0f;
                for (int k = 0; k < cnt; k++) { p[k] = cvm::expf_fixed(lg[k] - mx); se += p[k]; ysum += yi[off + k]; }
                float lse = mx + logf(se);
                for (int k = 0; k < cnt; k++) {
                    l += -(double)yi[off + k] * (double)(lg[k] - lse);
                    if (want_grad) { const float gr = (p[k] / se * ysum - yi[off + k]) * cvm::selu_grad(v[k]); gl[off + k] = gr; go[off + k] = gr; }
                }
            }
        }
        // the 16 candidates of the group in heads_train_tm's fixed tree (lanes with the same head: xor 4, 8, 16, 32)
#pragma unroll
        for (int d = 4; d < 64; d <<= 1) l += __shfl_xor(l, d);
        if (lane < 4) loss_rows[(size_t)g * 4 + lane] = l;
    }
    __syncthreads();
    if (!grads || wave >= 3) return;
    // ---- 4. fc5-side head data gradients (zygosity, type, length; k = fc5 unit), times selu'(fc5 output)
    const bool cand_ok = (int64_t)g * 16 + c < n;
    const float *gi = S[c];
#pragma unroll
    for (int j = 0; j < NBW; j++) {
        const int kb = wave * NBW + j;
        if (kb >= NB5) break;
        f4 o;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const int k = 16 * kb + 4 * s4 + q;
            float acc = 0.0f;
            if (cand_ok && k < K5) {
                const f4 *wk4 = reinterpret_cast<const f4 *>(shw[k]);
                const f4 w0 = wk4[0], w1 = wk4[1], w2 = wk4[2];
                const float wk[12] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3], w2[0], w2[1], w2[2], w2[3]};
#pragma unroll
                for (int jj = 0; jj < 12; jj++) acc = __builtin_fmaf(gi[4 + jj], wk[jj], acc);
            }
            o[s4] = acc * cv_selu_grad_from_out(H5[j][s4]);
        }
        g5pre_tm[((size_t)g * NB5 + kb) * 64 + lane] = o;
    }
}

// ---------------------------------------------------------------------------
// Small inference passes of the full topology (round 6): fc5 and the four heads in ONE launch, a workgroup of four
// waves per group -- train_tail_tm's steps 2 and 3 without the losses.  As two launches (dense_small<4, 7> + heads_tm)
// they were 12.6 + 17.6 us of a six-kernel chain at 63 groups, each a latency chain of its own: heads_tm re-reads from
// L2 what dense_small has just stored.  Here the group's 21 fc4 fragments go to LDS once; waves 0..2 run one slab of 4
// fc5 tiles each (dense_small's loop: weights from L2 through a register ring), wave 3 the base head's product over
// the same fragments; the fc5 outputs meet in LDS and wave 0 finishes the heads (heads_finish).  Per value the chains
// are dense_small's and heads_tm's: the same bits.
// ---------------------------------------------------------------------------
template <int NB4, int NB5>
__global__ __launch_bounds__(256) void infer_tail_tm(const f4 *__restrict__ h4, const f4 *__restrict__ w5s,
                                                    const float *__restrict__ bias5, int nout5, f4 *__restrict__ h5_out,
                                                    const f4 *__restrict__ wp0, const f4 *__restrict__ wp1,
                                                    const float *__restrict__ bb, const float *__restrict__ bz,
                                                    const float *__restrict__ bt, const float *__restrict__ bl, int64_t n,
                                                    float *__restrict__ out16)
{
    constexpr int NBW = 4, D = 7;
    static_assert(NB4 % D == 0 && NB5 <= 3 * NBW && NB4 <= D * NBW, "three slabs of four fc5 tiles, 21 k fragments in rings of 7");
    __shared__ __attribute__((aligned(16))) f4 sd4[NB4][64];
    __shared__ __attribute__((aligned(16))) f4 sh5[NB5][64];
    __shared__ __attribute__((aligned(16))) f4 sa0[64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x;
    const int q = lane >> 4;
    // the group's fc4 fragments: wave w brings w, w + 4, ...; the operands that depend on nothing computed here are
    // requested right behind them (the first ring of fc5 weights / all of the base head's, ONE register array for both roles)
    constexpr int MAXF = (NB4 + 3) / 4;
    f4 pz[MAXF];
#pragma unroll
    for (int i = 0; i < MAXF; i++) {
        const int ob = wave + 4 * i;
        if (ob < NB4) pz[i] = h4[((size_t)g * NB4 + ob) * 64 + lane];
    }
    const f4 *wp5 = w5s + (size_t)(wave < 3 ? wave : 0) * NB4 * (NBW * 64) + lane;
    f4 A[D][NBW];
    {
        const f4 *src = wave == 3 ? wp0 + lane : wp5;
#pragma unroll
        for (int d = 0; d < D; d++)
#pragma unroll
            for (int j = 0; j < NBW; j++) {
                const int f = d * NBW + j;
                if (wave < 3 || f < NB4) A[d][j] = src[(size_t)f * 64];      // (wave-uniform)
            }
    }
    f4 W1[NB5];
    if (wave == 0) {
#pragma unroll
        for (int kb = 0; kb < NB5; kb++) W1[kb] = wp1[(size_t)kb * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < MAXF; i++) {
        const int ob = wave + 4 * i;
        if (ob < NB4) sd4[ob][lane] = pz[i];
    }
    __syncthreads();
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    if (wave < 3) {
        const int slab = wave;
        const f4 *wp = wp5;
        f4 acc[NBW];
#pragma unroll
        for (int j = 0; j < NBW; j++) acc[j] = zero;
        auto fetch = [&](const f4 *pw, int d) {
#pragma unroll
            for (int j = 0; j < NBW; j++) A[d][j] = pw[((size_t)d * NBW + j) * 64];
        };
        auto step = [&](int kb, int d) {
            const f4 B = sd4[kb][lane];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                for (int j = 0; j < NBW; j++) acc[j] = mfma4(A[d][j][s4], B[s4], acc[j]);
        };
#pragma unroll 1
        for (int kb0 = D; kb0 < NB4; kb0 += D) {
            wp += (size_t)D * NBW * 64;
#pragma unroll
            for (int d = 0; d < D; d++) {
                step(kb0 - D + d, d);
                fetch(wp, d);
            }
        }
#pragma unroll
        for (int d = 0; d < D; d++) step(NB4 - D + d, d);
#pragma unroll
        for (int j = 0; j < NBW; j++) {
            const int ob = slab * NBW + j;
            if (ob >= NB5) break;
            const f4 h = selu4(acc[j] + load_bias4(bias5, ob, q, nout5));
            h5_out[((size_t)g * NB5 + ob) * 64 + lane] = h;      // (kept: cv_get_activation layer 5)
            sh5[ob][lane] = h;
        }
    } else {
        f4 a0 = zero;
#pragma unroll
        for (int kb = 0; kb < NB4; kb++) {
            const f4 B = sd4[kb][lane];
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) a0 = mfma4(A[kb / NBW][kb % NBW][s4], B[s4], a0);
        }
        sa0[lane] = a0;
    }
    __syncthreads();
    if (wave != 0) return;
    f4 a1 = zero;
#pragma unroll
    for (int kb = 0; kb < NB5; kb++) {
        const f4 B = sh5[kb][lane];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) a1 = mfma4(W1[kb][s4], B[s4], a1);
    }
    heads_finish(sa0[lane], a1, bb, bz, bt, bl, n, out16, g, lane);
}

// up to this many groups (16 candidates each) fc4 runs as 3 output slabs per group block: 8-wave workgroups
// x 3 slabs fill the 256 CUs from ~700 groups on; above the threshold one workgroup keeps all 21 tiles
constexpr int CV_FC4_SLAB_MAX_G = 2048;
// "tiny" batches of the training step (cv_model::tiny_g, option "train_tiny_groups", default 400 groups = 6 400
// candidates): the step is a chain of latency-bound kernels on a fraction of the chip; the layers then split their
// serial loops over more waves.  (Rounds 2-4 drew the line at 160 groups, tuned at 79; a sweep over eight batch sizes
// with the line lifted, profiles/r05/step_ab_session14_small_batch_regime.txt: at 161 groups 0.951 -> 0.735 ms, at 313
// groups 1.260 -> 1.215, break-even near 400, +7 % at 625.)

template <int KH, int CINB, int NT, int POOL, int HIN, int FRONT, int MODE = 0, int HSPLIT = 1, int KS4 = 4>
int launch_conv(const float *in, const float *x, int64_t n, const float *wp1, const float *bias1, int cout1,
                const float *wp, const float *bias, int cout, float *out, int G, hipStream_t st,
                float *act = nullptr)
{
    auto k = conv_tm<KH, CINB, NT, POOL, HIN, FRONT, MODE, HSPLIT, KS4>;
    size_t lds = (size_t)NT * KH * 4 * CINB * 1024;
    if (set_lds(k, lds)) return 1;
    unsigned grid = nblk((int64_t)G * NT * HSPLIT, 4);
    if constexpr (HSPLIT >= 1 && CV_CONV_XCD_UNITS(NT * HSPLIT))
        if (grid >= 16) grid = 8 * nblk((int64_t)((G + 7) / 8) * NT * HSPLIT, 4);       // per-XCD group numbering (see the kernel)
    int rows_per = 0;
    if constexpr (HSPLIT == 0) {
        // flat ranges: one round of equal waves.  slots = resident waves of this kernel (4-wave workgroups per CU by its
        // registers and LDS, asked once); at least 4 output rows per wave (a pooled layer recomputes POOL - 1 per segment).
        // (cached per DEVICE of this template instance, written once with an atomic store: the range boundaries -- and with
        // them the summation order of the weight gradients -- depend on this value, so it must be the calling device's own)
        static std::atomic<int> slots_by_dev[64];
        int dev = 0;
        CV_HIP(hipGetDevice(&dev));
        int slots = slots_by_dev[dev & 63].load(std::memory_order_relaxed);
        if (slots == 0) {
            int nb = 0, cus = 0;
            CV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), 256, lds));
            CV_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            slots = (nb > 0 ? nb : 2) * 4 * (cus > 0 ? cus : 256);
            slots_by_dev[dev & 63].store(slots, std::memory_order_relaxed);
        }
        const int64_t rows = (int64_t)G * (HIN - POOL + 1);
        const int per_tile = slots / NT;
        rows_per = (int)((rows + per_tile - 1) / per_tile);
        if (rows_per < 4) rows_per = 4;
        grid = nblk((rows + rows_per - 1) / rows_per * NT, 4);
    }
    k<<<grid, 256, lds, st>>>((const f4 *)in, x, n, wp1, bias1, cout1, (const f4 *)wp, bias, cout, (f4 *)out,
                              (f4 *)act, G, rows_per);
    CV_HIP(hipGetLastError());
    return 0;
}

// How many waves should share the positions of a (group, tile)?  The chip runs 2 048 conv waves at a time (256 CUs x
// 4 SIMDs x 2); a layer takes rounds x (rows per wave), where a pooled layer's parts recompute `overlap` rows.
// train.py's batch of 10 000 is 625 groups: conv3's data gradient with 2 tiles x 2 parts = 2 500 waves is two rounds
// of 13 rows (the second a fifth full), with 3 parts 1.8 rounds of 9.  Ties go to fewer parts (less redundancy).
static int pick_hsplit(int G, int NT, int rows, int overlap, int max_parts)
{
    static const int cand[6] = {1, 2, 3, 4, 6, 8};
    int best = 1;
    long best_cost = -1;
    for (int i = 0; i < 6; i++) {
        const int hs = cand[i];
        if (hs > max_parts || hs > rows) break;
        const long waves = (long)G * NT * hs;
        const long rounds = (waves + 2047) / 2048;
        const long cost = rounds * ((rows + hs - 1) / hs + overlap);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = hs; }
    }
    return best;
}

// Training layers: position parts of whole groups for small batches (pick_hsplit), flat ranges (0) beyond
// train_tiny_groups -- there a launch would take more than one round of waves.  dbg 9 at the call site's switch keeps
// the parts (development A/B).
static int conv_parts(const cv_model *m, int dbg, int G, int NT, int rows, int overlap, int max_parts)
{
    if (m->tiny_g <= 0) return 1;
    // (the position parts stop at CV_TINY_PARTS_MAX_G groups whatever the option says: they win from 60 to 79 groups
    // -- 23 us of the step at 79, a rank's share on 8 GPUs -- and lose 25 us from 88 groups on, slim 29 us at 157;
    // the rest of the small-batch kernel set pays up to 400 groups)
    const int parts_g = m->tiny_g < CV_TINY_PARTS_MAX_G ? m->tiny_g : CV_TINY_PARTS_MAX_G;
    if ((G > parts_g && dbg != 9) || dbg == 7) return 0;      // (dbg 7: flat ranges for a small batch too)
    return pick_hsplit(G, NT, rows, overlap, max_parts);
}

// launch_conv with the number of position parts chosen at run time (1, 2, 3, 4, 6 or 8; 0 = flat ranges)
template <int KH, int CINB, int NT, int POOL, int HIN, int MODE, int KS4 = 4>
int launch_conv_parts(int hs, const float *in, const float *x, int64_t n, const float *wp, const float *bias, int cout,
                      float *out, int G, hipStream_t st, float *act = nullptr)
{
#define CV_PARTS(H) return launch_conv<KH, CINB, NT, POOL, HIN, 0, MODE, H, KS4>(in, x, n, nullptr, nullptr, 0, wp, bias, cout, out, G, st, act)
    switch (hs) {
    case 0: CV_PARTS(0);          // flat ranges of the (group, row) sequence
    case 2: CV_PARTS(2);
    case 3: CV_PARTS(3);
    case 4: CV_PARTS(4);
    case 6: CV_PARTS(6);
    case 8: CV_PARTS(8);
    default: CV_PARTS(1);
    }
#undef CV_PARTS
}

// inference pass of a POOLED layer over 2, 4 or 8 position parts (full topology, small passes; parts recompute the window overlap)
// (FRONT > 0: the first layer fused in -- every part makes the pooled first-layer rows it needs from the raw X: x, wp1, bias1)
template <int KH, int CINB, int NT, int POOL, int HIN, int FRONT = 0>
int launch_conv_parts_pooled(int hs, const float *in, int64_t n, const float *wp, const float *bias, int cout, float *out, int G,
                             hipStream_t st, const float *x = nullptr, const float *wp1 = nullptr, const float *bias1 = nullptr,
                             int cout1 = 0)
{
#define CV_PARTS(H) return launch_conv<KH, CINB, NT, POOL, HIN, FRONT, 0, H>(in, x, n, wp1, bias1, cout1, wp, bias, cout, out, G, st)
    switch (hs) {
    case 2: CV_PARTS(2);
    case 8: CV_PARTS(8);
    default: CV_PARTS(4);
    }
#undef CV_PARTS
}
static int pooled_parts(int G, int NT, int rows, int overlap)
{
    int best = 4;
    long best_cost = -1;
    for (int hs = 2; hs <= 8; hs *= 2) {
        const long waves = (long)G * NT * hs;
        // (per SIMD, not per wave slot: measured at 63 groups, conv3 in 8 parts = 1 512 waves 57.8 us against 50.5 in 4 = 756 waves;
        // conv2 in 8 parts = 1 008 waves 21.7 us against 27.7 -- two of these waves on a SIMD take twice the time of one)
        const long cost = ((waves + 1023) / 1024) * ((rows + hs - 1) / hs + overlap + 1);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = hs; }
    }
    return best;
}

// inference pass of a layer WITHOUT pooling over position parts (slim, small passes): 1, 2, 4 or 8 waves per (group, tile)
template <int KH, int CINB, int NT, int HIN, int KS4 = 4>
int launch_conv_parts_infer(int hs, const float *in, int64_t n, const float *wp, const float *bias, int cout, float *out, int G,
                            hipStream_t st)
{
#define CV_PARTS(H) return launch_conv<KH, CINB, NT, 1, HIN, 0, 0, H, KS4>(in, nullptr, n, nullptr, nullptr, 0, wp, bias, cout, out, G, st)
    switch (hs) {
    case 0: CV_PARTS(0);          // equal ranges of the flat (group, row) sequence
    case 2: CV_PARTS(2);
    case 4: CV_PARTS(4);
    case 8: CV_PARTS(8);
    default: CV_PARTS(1);
    }
#undef CV_PARTS
}
// ... and how many: a launch costs (waves per slot, rounded up) x (positions per wave + what a wave pays before its first
// position: the workgroup's weight fragments into LDS, its first window).  slots = the chip's SIMDs for a layer whose wave
// keeps the matrix pipe of its SIMD busy on its own (slim conv3: 240 MFMAs per position, two waves on a SIMD take twice
// the time of one), twice that for a layer whose positions are mostly activation arithmetic (slim conv2: 72 MFMAs).
static int infer_parts(int G, int NT, int rows, int slots)
{
    // enough rows for equal ranges of at least 4 (launch_conv's flat form, two waves per SIMD): time proportional to the
    // pass whatever the number of groups -- whole parts step where G NT hs passes a multiple of the chip (257 groups:
    // 219 us against 143 at 256)
    if ((long)G * rows * NT >= 4L * 2048) return 0;
    int best = 1;
    double best_cost = -1.0;
    for (int hs = 1; hs <= 8; hs *= 2) {
        const long waves = (long)G * NT * hs;
        const double cost = (double)((waves + slots - 1) / slots) * ((rows + hs - 1) / hs + 3.0);
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = hs; }
    }
    return best;
}

// flat_slots > 0 (training forward): flat ranges sized for that many resident waves.  flat_slots < 0 (inference, round 6):
// whole groups or flat ranges, whichever the model says is shorter.  One wave alone on a SIMD already takes 93 % of its
// matrix pipe (26 positions: 131 us alone, 245 us for two waves side by side), so what a launch costs is the number of
// wave-positions its busiest SIMD has to run:
//   whole groups: ceil(G NT / SIMDs) waves of HIN positions (768 groups: 2 304 waves = 3 on some SIMDs, 2 on others);
//   flat ranges:  two equal waves per SIMD, each rows_per = ceil(G HOUT / (slots / NT)) pooled rows + the POOL - 1 rows
//                 every one of its ~ rows_per / HOUT + 1 segments computes for its first window
// (+ 0.7 of a position per segment / group for the prologue of its window; a single wave per SIMD pays 7 % for the idle
// issue slots).  Measured against this model at 13 sizes: profiles/r06/conv3_flat_ab.txt.
// *chose_flat (optional): which one ran (development probes).
template <int CINB, int NT, int HIN, int WAVES, int MINW, bool SAVE = false>
int launch_conv3_rot(const float *in, const float *wp, const float *bias, int cout, float *out, int G, hipStream_t st,
                     float *codes = nullptr, int flat_slots = 0, bool *chose_flat = nullptr)
{
    auto k = conv3_rot<CINB, NT, HIN, WAVES, MINW, SAVE>;
    size_t lds = (size_t)NT * 3 * 4 * CINB * 1024;
    if (set_lds(k, lds)) return 1;
    constexpr int HOUT = HIN - 2;
    unsigned grid = nblk((int64_t)G * NT, WAVES);
    if (grid >= 16) grid = 8 * nblk((int64_t)((G + 7) / 8) * NT, WAVES);      // per-XCD wave numbering (see the kernel)
    int rows_per = 0;
    if (chose_flat) *chose_flat = false;
    if (SAVE && flat_slots > 0) {            // training forward: one round of equal waves (see launch_conv)
        const int64_t rows = (int64_t)G * HOUT;
        const int per_tile = flat_slots / NT / 8 * 8;      // (a multiple of 8: the per-XCD numbering pads the units to one)
        rows_per = (int)((rows + per_tile - 1) / per_tile);
        if (rows_per < 4) rows_per = 4;
        const int64_t units = (rows + rows_per - 1) / rows_per;
        grid = nblk(units * NT, WAVES);
        if (grid >= 16) grid = 8 * nblk((units + 7) / 8 * NT, WAVES);
    } else if (!SAVE && flat_slots < 0) {
        static std::atomic<int> slots_by_dev[64];
        int dev = 0;
        CV_HIP(hipGetDevice(&dev));
        int slots = slots_by_dev[dev & 63].load(std::memory_order_relaxed);
        if (slots == 0) {
            int nb = 0, cus = 0;
            CV_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), WAVES * 64, lds));
            CV_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
            slots = (nb > 0 ? nb : 2) * WAVES * (cus > 0 ? cus : 256);
            slots_by_dev[dev & 63].store(slots, std::memory_order_relaxed);
        }
        const int64_t rows = (int64_t)G * HOUT;
        const int per_tile = slots / NT / 8 * 8;       // (a multiple of 8: the per-XCD numbering pads the units to one)
        int rp = (int)((rows + per_tile - 1) / per_tile);
        if (rp < 4) rp = 4;
        const double seg = 0.7 + 2.0;
        const int simds = slots / 2;                    // (the kernel runs two waves per SIMD)
        const int64_t per_simd = ((int64_t)G * NT + simds - 1) / simds;
        const double cost_flat = 2.0 * (rp + seg * ((double)rp / HOUT + 1.0));
        const double cost_whole = (double)per_simd * (HIN + 0.7) * (per_simd == 1 ? 1.07 : 1.0);
        if (flat_slots == -2 || (flat_slots == -1 && cost_flat < 0.98 * cost_whole)) {
            rows_per = rp;
            const int64_t units = (rows + rp - 1) / rp;
            grid = nblk(units * NT, WAVES);
            if (grid >= 16) grid = 8 * nblk((units + 7) / 8 * NT, WAVES);
            if (chose_flat) *chose_flat = true;
        }
    }
    k<<<grid, WAVES * 64, lds, st>>>((const f4 *)in, (const f4 *)wp, bias, cout, (f4 *)out, G, (u32x2 *)codes, rows_per);
    CV_HIP(hipGetLastError());
    return 0;
}

template <int NB, int WAVES, int EPI = 0, int GR = 1>
int launch_dense(const float *in, int KB, const float *wp, const float *bias, int nout, float *out, int G,
                 hipStream_t st, int slabs = 1, int ksplit = 1, float *part = nullptr, heads_args hd = heads_args(),
                 cv_dropout_args dr = cv_dropout_args())
{
    auto k = dense_tm<NB, WAVES, EPI, GR>;
    size_t lds = (size_t)3 * ((NB + WAVES - 1) / WAVES * WAVES) * 1024;
    if (EPI == 3) lds = (size_t)3 * 48 * 1024;          // the tail streams fc5 in stages of 4 k fragments x 12 output fragments
    if (set_lds(k, lds)) return 1;
    if (ksplit > 1) {       // partial sums per k range, then dense_ksum (EPI 0 layers only)
        static_assert(EPI == 0 || true, "");
        k<<<dim3(nblk(G, WAVES * GR), slabs, ksplit), WAVES * 64, lds, st>>>((const f4 *)in, KB, (const f4 *)wp, bias, nout,
                                                                        (f4 *)part, G, NB * slabs, heads_args());
        dense_ksum<<<nblk((int64_t)G * NB * slabs * 64, 256), 256, 0, st>>>((const f4 *)part, ksplit, G, NB * slabs, bias, nout,
                                                                          (f4 *)out, dr);
        CV_HIP(hipGetLastError());
        return 0;
    }
    k<<<dim3(nblk(G, WAVES * GR), slabs), WAVES * 64, lds, st>>>((const f4 *)in, KB, (const f4 *)wp, bias, nout,
                                                            (f4 *)out, G, NB * slabs, hd);
    CV_HIP(hipGetLastError());
    return 0;
}

template <int NBW, int D, int EPI = 0>
int launch_dense_small(const float *in, int KB, const float *wp, const float *bias, int nout, float *out, int G, int nslab,
                       hipStream_t st, int nbt = 0)
{
    if (nbt == 0) nbt = NBW * nslab;
    if (KB % D != 0 || KB < D) { cv_set_error("dense_small: %d k fragments are not a multiple of %d", KB, D); return 1; }
    dense_small<NBW, D, EPI><<<nblk((int64_t)G * nslab, 4), 256, 0, st>>>((const f4 *)in, KB, (const f4 *)wp, bias, nout,
                                                                          (f4 *)out, G, nslab, nbt);
    CV_HIP(hipGetLastError());
    return 0;
}

// Shape of a dense_rag launch over G groups: s = tile-units per SIMD and workgroup (ca + cb), from the model
//   time ~ ceil(workgroups per XCD / CUs per XCD) x (s + OV + ODD [s odd]) x UNIT    workgroups per XCD = nslab x ceil(ceil(NBS G / (4 s)) / 8)
// UNIT = 15.5 us: a tile-unit (288 x 4 MFMAs) on a SIMD at the clock the chip holds under this load; OV = 2: what a
// round costs besides its MFMAs; ODD = 0.5: odd shapes (ca = cb + 1) run a little over the trend.  Calibrated on the
// (G, s) table of tools/gpu_dense_rag_probe.py (profiles/r06/dense_rag_calibration.txt): the shape the model picks is
// within 2.2 % of the best measured one at all 17 sizes from 289 to 4 096 groups.  Ties go to the larger s (fewer
// workgroups stream the weight slab from L2).  force > 0: that s (development A/B).
struct rag_shape { int s, ca, cb, wgs; double us; };
constexpr double CV_RAG_UNIT_US = 15.5;
static rag_shape dense_rag_shape(int G, int NBS, int nslab, int cus, int force)
{
    const double OV = 2.0, ODD = 0.5;
    int best = 2 * NBS;
    double best_cost = -1.0;
    for (int s = 2 * NBS; s >= 4; s--) {
        if (force >= 4 && force <= 2 * NBS && s != force) continue;
        // (workgroups per XCD: the kernel numbers its grid so that the nslab workgroups of a piece share an XCD -- 8 pieces to
        // a row of the grid -- and an XCD runs cus / 8 workgroups of a round)
        const long wg = (long)nslab * ((((long)NBS * G + 4 * s - 1) / (4 * s) + 7) / 8);
        const long per_xcd = cus / 8 > 0 ? cus / 8 : 1;
        const double cost = (double)((wg + per_xcd - 1) / per_xcd) * (s + OV + ((s & 1) ? ODD : 0.0));
        if (best_cost < 0 || cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    rag_shape r;
    r.s = best; r.ca = (best + 1) / 2; r.cb = best / 2;
    r.wgs = (int)(((long)NBS * G + 4 * best - 1) / (4 * best));
    r.us = best_cost * CV_RAG_UNIT_US;
    return r;
}

// fc4 of the full topology (3 slabs of 7 output tiles, cv_model::wps_fc4) on ragged waves
static int launch_dense_rag(const float *in, int KB, const float *wps, const float *bias, int nout, float *out, int G, int nbt,
                            int force_s, hipStream_t st, cv_dropout_args dr = cv_dropout_args())
{
    auto k = dense_rag<7, 8>;
    const size_t lds = (size_t)3 * 8 * 1024;
    if (set_lds(k, lds)) return 1;
    int cus = 256;
    if (device_cus(&cus)) return 1;
    const rag_shape sh = dense_rag_shape(G, 7, 3, cus, force_s);
    k<<<8 * 3 * ((sh.wgs + 7) / 8), 512, lds, st>>>((const f4 *)in, KB, (const f4 *)wps, bias, nout, (f4 *)out, G, nbt, sh.ca, sh.cb, sh.wgs, 3, dr);
    CV_HIP(hipGetLastError());
    return 0;
}

// (the size options of an inference pass are cv_model::inf_small_g / inf_fc4_small_g / inf_slab_g: cv_api.hip, cv_mfma_forward)

}  // namespace

// ---- all weight packing of a parameter change in ONE launch ------------------------------------------------
// A training step re-packs every layer after its Adam update (forward fragments, slabs, transposed data-gradient
// fragments: 14 small jobs).  As separate launches they are 14 x ~5 us of dependent-launch latency at the head of
// the step -- 6 % of the step at config 4's per-rank batch; here the jobs share one grid (a block finds its job
// from the table) and cost one launch.

struct pack_job {
    int kind;                      // 0 conv1, 1 conv, 2 dense, 3 dense slabs, 4 dense dgrad, 5 conv dgrad, 6 heads, 7 dense dgrad by rows
    const float *src[4];
    float *dst[3];
    int i[8];
    unsigned first;                // first block of the job
};
struct pack_tab { pack_job j[16]; int n; };

__global__ __launch_bounds__(256) void pack_all(pack_tab tab)
{
    const unsigned b = blockIdx.x;
    int k = 0;
    while (k + 1 < tab.n && b >= tab.j[k + 1].first) k++;
    const pack_job &J = tab.j[k];
    const int64_t t = (int64_t)(b - J.first) * 256 + threadIdx.x;
    switch (J.kind) {
    case 0: pack_conv1(t, J.src[0], J.dst[0], J.i[0]); break;
    case 1: pack_conv(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3], J.i[4]); break;
    case 2: pack_dense(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3]); break;
    case 3: pack_dense_slabs(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3], J.i[4], J.i[5]); break;
    case 4: pack_dense_dgrad(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3], J.i[4], J.i[5]); break;
    case 5: pack_conv_dgrad(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3], J.i[4]); break;
    case 7: pack_dense_dgrad_rows(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3], J.i[4], J.i[5]); break;
    case 8: pack_dense_kpairs(t, J.src[0], J.dst[0], J.i[0], J.i[1], J.i[2], J.i[3], J.i[4]); break;
    default: pack_heads(t, J.src[0], J.src[1], J.src[2], J.src[3], J.i[0], J.i[1], J.i[2], J.i[3], J.dst[0], J.dst[1], J.dst[2]); break;
    }
}

struct pack_builder {
    pack_tab tab; unsigned blocks;
    pack_builder() : blocks(0) { tab.n = 0; }
    pack_job &add(int kind, int64_t threads)
    {
        pack_job &J = tab.j[tab.n++];
        J = pack_job();
        J.kind = kind; J.first = blocks;
        blocks += (unsigned)((threads + 255) / 256);
        return J;
    }
};

// Packs the layouts of `mask` (CVL_* bits; a layout the topology does not have is skipped) in ONE launch on `st` and
// marks them current.
static int pack_launch(cv_model *m, hipStream_t st, unsigned mask)
{
    const float *P = m->params;
    const int64_t *o = m->poff;
    const cv_shapes &s = m->sh;
    const cv_arch &a = m->arch;
    pack_builder pb;
    if (mask & CVL_CONV) {
        { pack_job &J = pb.add(0, 256); J.src[0] = P + o[0]; J.dst[0] = m->wp_conv1; J.i[0] = a.cout[0]; }
        for (int l = 1; l < 3; l++) {
            pack_job &J = pb.add(1, (int64_t)s.ntile[l] * a.kh[l] * 4 * s.cinb[l] * 256);
            J.src[0] = P + o[2 * l]; J.dst[0] = m->wp_conv[l];
            J.i[0] = a.kh[l]; J.i[1] = s.cin[l]; J.i[2] = a.cout[l]; J.i[3] = s.cinb[l]; J.i[4] = s.ntile[l];
        }
    }
    const int nbp4 = (s.nb4 + 3) / 4 * 4, nbp5 = (s.nb5 + 3) / 4 * 4;   // launch_dense: WAVES = 4
    if (mask & CVL_FC4) {
        pack_job &J = pb.add(2, (int64_t)s.kb4 * nbp4 * 256); J.src[0] = P + o[6]; J.dst[0] = m->wp_fc4;
        J.i[0] = s.flat; J.i[1] = a.fc4; J.i[2] = s.kb4; J.i[3] = nbp4;
    }
    if (mask & CVL_FC5) {
        pack_job &J = pb.add(2, (int64_t)s.nb4 * nbp5 * 256); J.src[0] = P + o[8]; J.dst[0] = m->wp_fc5;
        J.i[0] = a.fc4; J.i[1] = a.fc5; J.i[2] = s.nb4; J.i[3] = nbp5;
    }
    if ((mask & CVL_FC5P) && m->wp5p_fc5) {      // full topology: fc5 in k pairs for the tail of the large-pass fc4 kernel (dense_tm EPI 3)
        pack_job &J = pb.add(8, (int64_t)((s.nb4 + 3) / 4) * 48 * 256); J.src[0] = P + o[8]; J.dst[0] = m->wp5p_fc5;
        J.i[0] = a.fc4; J.i[1] = a.fc5; J.i[2] = s.nb4; J.i[3] = 12; J.i[4] = 4;
    }
    if ((mask & CVL_FC4S3) && m->wps_fc4) {      // full topology: fc4 in 3 slabs of 7 fragments for small batches
        pack_job &J = pb.add(3, (int64_t)3 * s.kb4 * 8 * 256); J.src[0] = P + o[6]; J.dst[0] = m->wps_fc4;
        J.i[0] = s.flat; J.i[1] = a.fc4; J.i[2] = s.kb4; J.i[3] = 7; J.i[4] = 8; J.i[5] = 3;
    }
    if ((mask & CVL_FC5S3) && m->wps3_fc5) {     // fc5 in 3 slabs of 4 fragments (11 -> 12, the last one zero) for dense_small
        pack_job &J = pb.add(3, (int64_t)3 * s.nb4 * 4 * 256); J.src[0] = P + o[8]; J.dst[0] = m->wps3_fc5;
        J.i[0] = a.fc4; J.i[1] = a.fc5; J.i[2] = s.nb4; J.i[3] = 4; J.i[4] = 4; J.i[5] = 3;
    }
    if ((mask & CVL_FC4S7) && m->wps7_fc4) {     // ... and in 7 slabs of 3 fragments for the one-wave-per-slab kernel of very small batches
        // (slim, 3 output fragments: 3 slabs of one)
        const int per = s.nb4 == 21 ? 3 : 1, nsl = s.nb4 == 21 ? 7 : s.nb4;
        pack_job &J = pb.add(3, (int64_t)nsl * s.kb4 * per * 256); J.src[0] = P + o[6]; J.dst[0] = m->wps7_fc4;
        J.i[0] = s.flat; J.i[1] = a.fc4; J.i[2] = s.kb4; J.i[3] = per; J.i[4] = per; J.i[5] = nsl;
    }
    if ((mask & CVL_FC4S21) && m->wps21_fc4) {   // ... and in 21 slabs of one (inference passes of up to ~100 groups)
        pack_job &J = pb.add(3, (int64_t)21 * s.kb4 * 256); J.src[0] = P + o[6]; J.dst[0] = m->wps21_fc4;
        J.i[0] = s.flat; J.i[1] = a.fc4; J.i[2] = s.kb4; J.i[3] = 1; J.i[4] = 1; J.i[5] = 21;
    }
    if (mask & CVL_HEADS) {
        pack_job &J = pb.add(6, (int64_t)(s.nb4 + s.nb5) * 256 + (int64_t)s.nb5 * 16 * 12);
        J.src[0] = P + o[10]; J.src[1] = P + o[12]; J.src[2] = P + o[14]; J.src[3] = P + o[16];
        J.i[0] = a.fc4; J.i[1] = a.fc5; J.i[2] = s.nb4; J.i[3] = s.nb5; J.dst[0] = m->wp_heads0; J.dst[1] = m->wp_heads1;
        J.dst[2] = m->wp_heads12;
    }
    if (mask & CVL_DCONV) {
        for (int l = 1; l < 3; l++) {
            pack_job &J = pb.add(5, (int64_t)s.cinb[l] * a.kh[l] * 4 * s.ntile[l] * 256);
            J.src[0] = P + o[2 * l]; J.dst[0] = m->wpd_conv[l];
            J.i[0] = a.kh[l]; J.i[1] = s.cin[l]; J.i[2] = a.cout[l]; J.i[3] = s.ntile[l]; J.i[4] = s.cinb[l];
        }
    }
    if (mask & CVL_DFC4) {
        if (m->wpr_fc4 && m->dbg[3] != 1) {     // full: by column and pooled row, for dense_dgrad_unpool
            const int ncol = 4 * s.ntile[2];
            pack_job &J = pb.add(7, (int64_t)ncol * s.hp[2] * 24 * 256); J.src[0] = P + o[6]; J.dst[0] = m->wpr_fc4;
            J.i[0] = s.flat; J.i[1] = a.fc4; J.i[2] = s.nb4; J.i[3] = 24; J.i[4] = ncol; J.i[5] = s.hp[2];
        } else {
            pack_job &J = pb.add(4, (int64_t)(s.kb4 / 24) * s.nb4 * 24 * 256); J.src[0] = P + o[6]; J.dst[0] = m->wpd_fc4;
            J.i[0] = s.flat; J.i[1] = a.fc4; J.i[2] = s.nb4; J.i[3] = 24; J.i[4] = s.kb4 / 24; J.i[5] = 24;
        }
    }
    if (mask & CVL_DFC5) {   // fc5: full = 3 slabs of 7 fragments (stride 8 = dense_tm<7, 8>'s padded count), slim = one slab of 3 (stride 4)
        const int nbs = is_full(a) ? 7 : s.nb4, nbsp = is_full(a) ? 8 : 4, nslab = is_full(a) ? 3 : 1;
        pack_job &J = pb.add(4, (int64_t)nslab * s.nb5 * nbsp * 256); J.src[0] = P + o[8]; J.dst[0] = m->wpd_fc5;
        J.i[0] = a.fc4; J.i[1] = a.fc5; J.i[2] = s.nb5; J.i[3] = nbs; J.i[4] = nslab; J.i[5] = nbsp;
    }
    m->packed_valid |= mask;
    if (pb.blocks == 0) return 0;
    pack_all<<<pb.blocks, 256, 0, st>>>(pb.tab);
    CV_HIP(hipGetLastError());
    return 0;
}

int cv_layout_current(const cv_model *m, unsigned layout, const char *who)
{
    if ((m->packed_valid & layout) == layout) return 0;
    cv_set_error("%s: packed weight layout 0x%x is stale (0x%x current) -- internal: the pass did not pack what it reads", who, layout, m->packed_valid);
    return 1;
}

// inference: every forward layout that is stale
int cv_pack_weights(cv_model *m, hipStream_t st)
{
    return pack_launch(m, st, CVL_FORWARD & ~m->packed_valid);
}

// Which packed layout the fc4 / fc5 of a TRAINING pass over G groups read: cv_tile_dense_fwd takes its kernel from
// these, and cv_pack_for_training packs by them.
static unsigned fc4_train_layout(const cv_model *m, int G)
{
    if (!is_full(m->arch)) return CVL_FC4;
    if (m->train_ksplit && G <= m->tiny_g) return CVL_FC4S3;          // eight k ranges of the 3-slab form
    if (G <= m->tiny_g && (m->variant & 128)) return CVL_FC4S7;       // one wave per (group, slab of 3)
    if (G <= CV_FC4_SLAB_MAX_G) return CVL_FC4S3;
    return CVL_FC4;
}
static unsigned fc5_train_layout(const cv_model *m, int G)
{
    if (!is_full(m->arch)) return CVL_FC5;
    return (G <= CV_FC4_SLAB_MAX_G && (m->variant & 128)) ? CVL_FC5S3 : CVL_FC5;
}

// Training pass over G groups: the layouts ITS kernels read and that are stale -- two of the four fc4 layouts, one of
// the three fc5 layouts (round 5: all of them were re-packed every step, 30 MB; a step now packs ~14 MB and an
// inference pass that follows packs what it reads).  One launch on `st`; or, with a side stream, the convolution
// fragments on `st` (the first kernels need them) and the dense / data-gradient fragments on `sw` next to the
// convolution forward pass -- *wait_before_dense is then the event `st` has to wait for before the first dense layer.
// phase 0: all of it; 1: only the convolution fragments on st; 2: only the rest on sw (the caller has ordered sw)
int cv_pack_for_training(cv_model *m, hipStream_t st, bool backward, int G, hipStream_t sw, hipEvent_t fork, hipEvent_t done,
                         bool *wait_before_dense, bool sw_ordered, int phase)
{
    if (wait_before_dense) *wait_before_dense = false;
    unsigned need = CVL_CONV | CVL_HEADS | fc4_train_layout(m, G) | fc5_train_layout(m, G);
    if (!(m->sched & 16)) need = CVL_FORWARD;          // every forward layout, as before round 5
    if (backward) need |= CVL_BACKWARD;
    const unsigned todo = need & ~m->packed_valid;
    if (!todo) return 0;
    if (sw == st || !sw || !wait_before_dense) return pack_launch(m, st, todo);
    if (!sw_ordered && phase != 1) {
        CV_HIP(hipEventRecord(fork, st));              // behind the optimizer update of the previous step
        CV_HIP(hipStreamWaitEvent(sw, fork, 0));
    }
    if (phase != 2 && pack_launch(m, st, todo & CVL_CONV)) return 1;
    if (phase == 1) return 0;
    if (todo & ~CVL_CONV) {
        if (pack_launch(m, sw, todo & ~CVL_CONV)) return 1;
        CV_HIP(hipEventRecord(done, sw));
        *wait_before_dense = true;
    }
    return 0;
}

static int mfma_alloc(cv_model *m, int64_t cap)
{
    cap = (cap + 15) / 16 * 16;
    if (m->ws_cap >= cap) return 0;
    float **bufs[5] = {&m->tm_p1, &m->tm_p2, &m->tm_p3, &m->tm_h4, &m->tm_h5};
    for (auto b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    m->ws_cap = 0;
    const cv_shapes &s = m->sh;
    size_t per[5] = {(size_t)s.hp[0] * 4 * s.ntile[0] * 16, (size_t)s.hp[1] * 4 * s.ntile[1] * 16,
                     (size_t)s.hp[2] * 4 * s.ntile[2] * 16, (size_t)s.nb4 * 16, (size_t)s.nb5 * 16};
    for (int i = 0; i < 5; i++) CV_HIP(hipMalloc(bufs[i], sizeof(float) * per[i] * cap));
    m->ws_cap = cap;
    return 0;
}


// Device copy of the per-model (stable) tail arguments of the fused kernels: refreshed synchronously, and only when a
// pointer changed (first pass, a workspace that grew).  What changes from call to call -- the number of candidates and
// the output pointer -- travels by value, so that no launch ever depends on host memory a later call may overwrite.
static int tail_args_refresh(cv_model *m, heads_args h, hipStream_t st, const heads_args **dev)
{
    static_assert(sizeof(heads_args) <= sizeof(m->tail_host), "cv_model::tail_host holds a heads_args");
    unsigned char clean[sizeof(heads_args)];            // field by field into zeroed bytes: padding must not make two equal sets differ
    memset(clean, 0, sizeof(clean));
    heads_args *c = reinterpret_cast<heads_args *>(clean);
    c->wp0 = h.wp0; c->wp1 = h.wp1; c->bb = h.bb; c->bz = h.bz; c->bt = h.bt; c->bl = h.bl;
    c->wp5p = h.wp5p; c->bias5 = h.bias5; c->nout5 = h.nout5; c->h5_out = h.h5_out; c->keep = h.keep;
    if (!m->tail_dev) CV_HIP(hipMalloc(&m->tail_dev, sizeof(heads_args)));
    if (memcmp(m->tail_host, clean, sizeof(heads_args)) != 0) {
        CV_HIP(hipStreamSynchronize(st));               // no kernel in flight reads the old copy
        memcpy(m->tail_host, clean, sizeof(heads_args));
        CV_HIP(hipMemcpy(m->tail_dev, m->tail_host, sizeof(heads_args), hipMemcpyHostToDevice));
    }
    *dev = (const heads_args *)m->tail_dev;
    return 0;
}

// one chunk (n <= chunk) through the tile kernels
// slim topology: does a pass over G groups run the small-pass kernel set?  (option "slim_small_groups")
// Default (-1): whichever an estimate says is shorter.  The unfused set runs on equal flat ranges, its time is linear in the
// pass, 0.46 us per group + 25; the fused pair is a staircase -- the front kernel 65 us per started 1 024 groups + 20, conv3 +
// fc4 one 112 KB workgroup per CU, 337 us per round of 4-group workgroups or 614 us per round of 8-group ones.  The fused
// pair wins where a round is nearly full (1 024, 2 048, 3 072, 4 096 groups and the few hundred below each), the unfused
// set everywhere else up to ~3 100 groups (profiles/r06/slim_small_pass.txt).
static bool slim_small_pass(const cv_model *m, int G)
{
    if (!is_slim(m->arch) || !(m->variant & 128) || m->wps7_fc4 == nullptr) return false;
    if (m->inf_slim_small_g >= 0) return G <= m->inf_slim_small_g;
    const double unfused = 0.46 * G + 25.0;
    const long r8 = ((G + 7) / 8 + 255) / 256, r4 = ((G + 3) / 4 + 255) / 256;
    const double k8 = 614.0 * r8, k4 = 337.0 * r4;
    const double fused = 65.0 * ((G + 1023) / 1024) + 20.0 + (k8 < k4 ? k8 : k4);
    return unfused < 0.97 * fused;
}

int cv_mfma_forward(cv_model *m, const float *x, int64_t n, float *out16, hipStream_t st)
{
    if (n <= 0) return 0;
    const cv_arch &a = m->arch;
    const bool full = arch_is(a, 1, 2, 3, 16, 32, 48, 5, 4, 3, 336, 168);
    const bool slim = arch_is(a, 1, 3, 5, 8, 16, 32, 1, 1, 1, 36, 18);
    if (!full && !slim) {
        cv_set_error("tile kernels cover the v3 full and v3 slim topologies only; set option impl=0");
        return 1;
    }
    if (mfma_alloc(m, n)) return 1;
    for (int i = 0; i < CV_NUM_STAGES; i++) m->stage_kernel[i] = nullptr;
    m->last_maps = 1;
    if (cv_pack_weights(m, st)) return 1;
    const float *P = m->params;
    const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    const cv_shapes &s = m->sh;
    int rc = 0;
    const bool fuse_front = (m->variant & 1) != 0;
    const float *W1 = m->wp_conv1, *B1 = P + o[1];
    bool heads_done = false;             // the heads rode on the fc5 kernel (variant bit 9)
    bool tail_done = false;              // fc5 and the heads rode on the fc4 kernel (variant bit 10)
    heads_args hd;
    hd.wp0 = (const f4 *)m->wp_heads0; hd.wp1 = (const f4 *)m->wp_heads1;
    hd.bb = P + o[11]; hd.bz = P + o[13]; hd.bt = P + o[15]; hd.bl = P + o[17];
    hd.n = n; hd.out16 = out16;
    if (full) {
        // very small passes (a predict() call of the reference's batch of 1 000 is 63 groups) are latency-bound by the
        // serial position loop of one (group, tile): the layers are launched unfused with their positions split over
        // four waves (pooled layers recompute the window overlap) -- the same values row for row
        const bool small_pass = (m->variant & 128) && G <= m->inf_small_g;
        if (small_pass && fuse_front && m->dbg[0] != 5 && (G <= 96 || m->dbg[0] == 6)) {
            // conv1 + pool1 inside conv2's position parts (round 6): each part makes the pooled first-layer rows it needs from
            // the raw X in registers (front_source from its first row: 4 + 1 more conv1 rows than conv2 input rows) -- a launch
            // and the 7.4 KB-per-candidate pool1 map less: 63 groups 30.6 us against 16.0 + 21.4, a pass of 1 000 candidates
            // 148 -> 144 us; from 160 groups on the recomputed rows cost more than the launch (59.4 against 16.4 + 40.6):
            // up to 96 groups.  dbg0 = 5: as two kernels, 6: fused at every small-pass size
            cv_prof_begin(m, 1, st);
            const int hs2 = pooled_parts(G, 2, 26, 3 + 2);
            m->stage_kernel[1] = hs2 == 8 ? "conv_tm<2, 1, 2, 4, 29, 5, 0, 8>" : hs2 == 2 ? "conv_tm<2, 1, 2, 4, 29, 5, 0, 2>" : "conv_tm<2, 1, 2, 4, 29, 5, 0, 4>";
            rc |= launch_conv_parts_pooled<2, 1, 2, 4, 29, 5>(hs2, nullptr, n, m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st, x, W1, B1, a.cout[0]);
            cv_prof_end(m, 1, st);
        } else if (small_pass) {
            cv_prof_begin(m, 0, st);
            m->stage_kernel[0] = "conv1_tm<5, false>";
            conv1_tm<5><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, W1, B1, a.cout[0], (f4 *)m->tm_p1, G);
            cv_prof_end(m, 0, st);
            cv_prof_begin(m, 1, st);
            // (round 6: 2, 4 or 8 parts by the number of groups -- 63 groups: 8 parts of 4 + 3 rows instead of 4 of 7 + 3)
            const int hs2 = m->dbg[0] == 4 ? 4 : pooled_parts(G, 2, 26, 3);
            m->stage_kernel[1] = hs2 == 8 ? "conv_tm<2, 1, 2, 4, 29, 0, 0, 8>" : hs2 == 2 ? "conv_tm<2, 1, 2, 4, 29, 0, 0, 2>" : "conv_tm<2, 1, 2, 4, 29, 0, 0, 4>";
            rc |= launch_conv_parts_pooled<2, 1, 2, 4, 29>(hs2, m->tm_p1, n, m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
            cv_prof_end(m, 1, st);
        } else if (fuse_front && (m->variant & 64)) {
            cv_prof_begin(m, 1, st);
            // whole groups (a workgroup = 2 groups x 2 tiles, two workgroups per CU: ceil(G / 512) waves per SIMD of 29 + 29
            // rows; 91 us alone, 156 us for two side by side) or flat ranges (FLAT: one pair of waves per workgroup, 1 024
            // resident, a segment pays ~4 rows for its first windows) -- whichever the model says is shorter
            int cusf = 256;
            if (device_cus(&cusf)) return 1;
            const int per_simd = (G + 2 * cusf - 1) / (2 * cusf);
            const int rpf = (int)(((int64_t)G * 26 + 4 * cusf - 1) / (4 * cusf));
            const double cost_whole = per_simd * 30.0 * (per_simd == 1 ? 1.17 : 1.0);
            const double cost_flat = 2.12 * (rpf + 4.0 * (rpf / 26.0 + 1.0));      // (2.12: the two-wave workgroups run 6 % under the four-wave ones on a full chip)
            if (m->inf_flat == 2 || (m->inf_flat == 1 && rpf >= 6 && cost_flat < 0.98 * cost_whole)) {
                m->stage_kernel[1] = "front2_tm<6, true>";
                auto k = front2_tm<6, true>;
                const size_t lds = (size_t)(16 + 6 * 4) * 1024;
                if (set_lds(k, lds)) return 1;
                const int rp = rpf < 6 ? 6 : rpf;
                k<<<nblk((int64_t)G * 26, rp), 128, lds, st>>>(x, n, W1, B1, a.cout[0], (const f4 *)m->wp_conv[1], P + o[3], a.cout[1],
                                                               (f4 *)m->tm_p2, G, rp);
            } else {
                m->stage_kernel[1] = "front2_tm<6, false>";
                auto k = front2_tm<6>;
                const size_t lds = (size_t)(16 + 2 * 6 * 4) * 1024;
                if (set_lds(k, lds)) return 1;
                k<<<nblk(G, 2), 256, lds, st>>>(x, n, W1, B1, a.cout[0], (const f4 *)m->wp_conv[1], P + o[3], a.cout[1],
                                                (f4 *)m->tm_p2, G, 0);
            }
            cv_prof_end(m, 1, st);
        } else if (fuse_front) {
            cv_prof_begin(m, 1, st);
            m->stage_kernel[1] = "conv_tm<2, 1, 2, 4, 29, 5, 0, 1>";
            rc |= launch_conv<2, 1, 2, 4, 29, 5>(nullptr, x, n, W1, B1, a.cout[0], m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
            cv_prof_end(m, 1, st);
        } else {
            cv_prof_begin(m, 0, st);
            m->stage_kernel[0] = "conv1_tm<5, false>";
            conv1_tm<5><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, W1, B1, a.cout[0], (f4 *)m->tm_p1, G);
            cv_prof_end(m, 0, st);
            cv_prof_begin(m, 1, st);
            m->stage_kernel[1] = "conv_tm<2, 1, 2, 4, 29, 0, 0, 1>";
            rc |= launch_conv<2, 1, 2, 4, 29, 0>(m->tm_p1, x, n, W1, B1, a.cout[0], m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
            cv_prof_end(m, 1, st);
        }
        cv_prof_begin(m, 2, st);
        if (small_pass) {
            const int hs3 = m->dbg[0] == 4 ? 4 : pooled_parts(G, 3, 24, 2);
            m->stage_kernel[2] = hs3 == 8 ? "conv_tm<3, 2, 3, 3, 26, 0, 0, 8>" : hs3 == 2 ? "conv_tm<3, 2, 3, 3, 26, 0, 0, 2>" : "conv_tm<3, 2, 3, 3, 26, 0, 0, 4>";
            rc |= launch_conv_parts_pooled<3, 2, 3, 3, 26>(hs3, m->tm_p2, n, m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st);
        }
        else if (m->variant & 8) { m->stage_kernel[2] = "conv3_rot<2, 3, 26, 4, 2, false>"; rc |= launch_conv3_rot<2, 3, 26, 4, 2>(m->tm_p2, m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st, nullptr, m->inf_flat == 0 ? 0 : (m->inf_flat == 2 ? -2 : -1)); }
        else { m->stage_kernel[2] = "conv_tm<3, 2, 3, 3, 26, 0, 0, 1>"; rc |= launch_conv<3, 2, 3, 3, 26, 0>(m->tm_p2, x, n, W1, B1, a.cout[0], m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st); }
        cv_prof_end(m, 2, st);
        cv_prof_begin(m, 3, st);
        // fc4: which kernel form runs is an ESTIMATE of each form's time at this number of groups (round 6; rounds 1-5 drew
        // fixed lines at 288 / 3 400 groups) -- the same bits whichever runs:
        //   dense_small (one wave per group and slab of 3 tiles, weights from L2): 83 us per round of 1 024 waves;
        //   dense_rag (three slabs on ragged waves): its shape model, + fc5 and the heads as kernels of their own;
        //   dense_tm<21, 8, 3, 2> (fc4 + fc5 + heads, 16 groups per workgroup, one workgroup per CU): 1 525 us per round.
        // Options: infer_fc4_small_groups = the size up to which dense_small is considered at all (288: beyond it the weight
        // matrix is read from L2 7 x G times), infer_slab_groups >= 0 = a fixed line between dense_rag and the fused kernel
        // instead of the estimate (A/B, tests), dense_rag -1 = the round-5 three-slab kernel in dense_rag's place.
        int cus = 256;
        if (device_cus(&cus)) return 1;
        const bool can_small = G <= m->inf_fc4_small_g && (m->variant & 128) && m->wps7_fc4;
        const bool can_fused = (m->variant & 1024) && (m->variant & 32) && m->wp5p_fc5;
        const bool can_rag = m->wps_fc4 != nullptr;
        const rag_shape rsh = dense_rag_shape(G, 7, 3, cus, m->inf_rag_s);
        const double us_rag = rsh.us + 22.0 + 0.019 * G;                          // + fc5 with the heads on its tail (dense_tm<11, 4, 2>)
        const double us_small = 83.0 * (double)(((long)7 * G + 4 * cus - 1) / (4 * cus)) + 12.0;
        const double us_wide = 1525.0 * (double)(((G + 15) / 16 + cus - 1) / cus) + (can_fused ? -17.0 : 85.0);   // (the launches the fused tail saves the pass / fc5 + heads on their own)
        int form;                                                                  // 0 small, 1 three slabs, 2 all 21 tiles per wave (fused with fc5 + heads by variant bit 10)
        if (can_small && (!can_rag || us_small <= us_rag + 12.0)) form = 0;       // (fc5 follows on dense_small too: 12 us less)
        else if (!can_rag) form = 2;
        else if (m->inf_slab_g >= 0) form = G <= m->inf_slab_g ? 1 : 2;
        else form = us_wide < us_rag ? 2 : 1;
        if (form == 0 && m->wps21_fc4 && G <= m->inf_fc4_one_g) {
            // the smallest passes: one wave per (group, output fragment) -- a wave's chain is 288 x 4 MFMAs instead of 288 x 12
            // (16 groups 31.7 us against 79.5, 63 groups 63.6 against 82.1, 100 groups 97.7 against 83.5: option
            // infer_fc4_one_groups, 80)
            // operand ring depth by the number of groups (same-box ladder, 16 / 40 / 63 / 80 groups, us: depth 4: 41 / 39 / 55 / 56,
            // 8: 32 / 31 / 63 / 64, 12: 27 / 28 / 71 / 71, 16: 29 / 30 / 68 / 69): up to 48 groups (1 008 waves: one per SIMD) a wave
            // is alone with its load latency and a deeper ring hides more of it; beyond, two waves share a SIMD and a CU's
            // 64 B per clock of vector loads, and the shallow ring's smaller register set lets them overlap
            if (G <= 48) { m->stage_kernel[3] = "dense_small<1, 12, 0>"; rc |= launch_dense_small<1, 12>(m->tm_p3, s.kb4, m->wps21_fc4, P + o[7], a.fc4, m->tm_h4, G, 21, st); }
            else { m->stage_kernel[3] = "dense_small<1, 4, 0>"; rc |= launch_dense_small<1, 4>(m->tm_p3, s.kb4, m->wps21_fc4, P + o[7], a.fc4, m->tm_h4, G, 21, st); }
        }
        else if (form == 0) { m->stage_kernel[3] = "dense_small<3, 8, 0>"; rc |= launch_dense_small<3, 8>(m->tm_p3, s.kb4, m->wps7_fc4, P + o[7], a.fc4, m->tm_h4, G, 7, st); }
        else if (form == 1 && m->inf_rag_s >= 0) { m->stage_kernel[3] = "dense_rag<7, 8>"; rc |= launch_dense_rag(m->tm_p3, s.kb4, m->wps_fc4, P + o[7], a.fc4, m->tm_h4, G, s.nb4, m->inf_rag_s, st); }
        else if (form == 1) { m->stage_kernel[3] = "dense_tm<7, 8, 0, 1>"; rc |= launch_dense<7, 8>(m->tm_p3, s.kb4, m->wps_fc4, P + o[7], a.fc4, m->tm_h4, G, st, 3); }
        else if (can_fused) {      // fc4 + fc5 + heads as one kernel
            m->stage_kernel[3] = "dense_tm<21, 8, 3, 2>";
            heads_args h3 = hd;
            h3.wp5p = (const f4 *)m->wp5p_fc5; h3.bias5 = P + o[9]; h3.nout5 = a.fc5; h3.h5_out = (f4 *)m->tm_h5;
            h3.keep = m->keep_act;
            if (a.fc4 != 16 * s.nb4) { cv_set_error("fused fc4 tail: fc4 width must be a whole number of tiles"); return 1; }
            heads_args hk;                           // by value: the pointer to the device copy + what changes per call
            if (tail_args_refresh(m, h3, st, &hk.tail)) return 1;
            hk.n = n; hk.out16 = out16;
            rc |= launch_dense<21, 8, 3, 2>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st, 1, 1, nullptr, hk);
            tail_done = true;
        }
        else if (m->variant & 32) { m->stage_kernel[3] = "dense_tm<21, 8, 0, 2>"; rc |= launch_dense<21, 8, 0, 2>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st); }
        else if (m->variant & 4) { m->stage_kernel[3] = "dense_tm<21, 8, 0, 1>"; rc |= launch_dense<21, 8>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st); }
        else { m->stage_kernel[3] = "dense_tm<21, 4, 0, 1>"; rc |= launch_dense<21, 4>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st); }
        cv_prof_end(m, 3, st);
        if (tail_done) {
            if (rc) return 1;
            CV_HIP(hipGetLastError());
            m->last_n = n; m->last_impl = 1; m->last_variant = m->variant; m->last_maps = m->keep_act;
            return 0;
        }
        cv_prof_begin(m, 4, st);
        if (G <= (m->inf_fc4_small_g > 560 ? m->inf_fc4_small_g : 560) && (m->variant & 128) && (m->variant & 512) && (m->variant & 2) && m->wps3_fc5 && s.nb4 == 21 && s.nb5 == 11) {
            // fc5 + the heads as one launch of four-wave workgroups, one per group (was dense_small<4, 7> + heads_tm in small
            // passes: 30 -> 18 us).  Up to 560 groups it also beats the ring kernel with the heads on its tail (dense_tm<11, 4, 2>:
            // 257 / 320 / 512 groups 23 / 24 / 25 us against 31 / 31 / 33; from 625 on 34 against 32 and growing with the pass)
            m->stage_kernel[4] = "infer_tail_tm<21, 11>";
            infer_tail_tm<21, 11><<<G, 256, 0, st>>>((const f4 *)m->tm_h4, (const f4 *)m->wps3_fc5, P + o[9], a.fc5, (f4 *)m->tm_h5,
                                                     (const f4 *)m->wp_heads0, (const f4 *)m->wp_heads1, P + o[11], P + o[13],
                                                     P + o[15], P + o[17], n, out16);
            heads_done = true;
        } else if (G <= m->inf_fc4_small_g && (m->variant & 128)) {
            m->stage_kernel[4] = "dense_small<4, 7, 0>";
            rc |= launch_dense_small<4, 7>(m->tm_h4, s.nb4, m->wps3_fc5, P + o[9], a.fc5, m->tm_h5, G, 3, st, s.nb5);
        } else if (m->variant & 512) {      // fc5 + heads as one kernel
            m->stage_kernel[4] = "dense_tm<11, 4, 2, 1>";
            rc |= launch_dense<11, 4, 2>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st, 1, 1, nullptr, hd);
            heads_done = true;
        } else {                            // (two groups per wave, as for fc4, measured: 79.5 -> 75.3 us; not worth a variant)
            m->stage_kernel[4] = "dense_tm<11, 4, 0, 1>";
            rc |= launch_dense<11, 4>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st);
        }
        cv_prof_end(m, 4, st);
    } else if (slim_small_pass(m, G)) {
        // Small passes of the slim topology (round 6).  The fused kernels walk the 33 positions of a group in ONE wave: the
        // first two layers take 82 us and conv3 + fc4 330 us whether a pass has 63 groups or 1 024 (one 112 KB workgroup per
        // CU) -- a predict() call of the reference's batch of 1 000 took 400 us, twice the full topology's.  Here, as in the
        // full topology's small pass, the layers run unfused with their positions split over up to 8 waves (no pooling in
        // this topology: nothing is recomputed), fc4 as one wave per (group, output fragment) straight from L2; fc5 and the
        // heads as in larger passes.  Same values row for row.
        int cusm = 256;
        if (device_cus(&cusm)) return 1;
        cv_prof_begin(m, 0, st);
        m->stage_kernel[0] = "conv1_tm<1, false>";
        conv1_tm<1><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, W1, B1, a.cout[0], (f4 *)m->tm_p1, G);
        cv_prof_end(m, 0, st);
        cv_prof_begin(m, 1, st);
        static const char *const n2[5] = {"conv_tm<3, 1, 1, 1, 33, 0, 0, 0, 2>", "conv_tm<3, 1, 1, 1, 33, 0, 0, 1, 2>", "conv_tm<3, 1, 1, 1, 33, 0, 0, 2, 2>", "conv_tm<3, 1, 1, 1, 33, 0, 0, 4, 2>", "conv_tm<3, 1, 1, 1, 33, 0, 0, 8, 2>"};
        static const char *const n3[5] = {"conv_tm<5, 1, 2, 1, 33, 0, 0, 0, 4>", "conv_tm<5, 1, 2, 1, 33, 0, 0, 1, 4>", "conv_tm<5, 1, 2, 1, 33, 0, 0, 2, 4>", "conv_tm<5, 1, 2, 1, 33, 0, 0, 4, 4>", "conv_tm<5, 1, 2, 1, 33, 0, 0, 8, 4>"};
        auto slot_of = [](int hs) { return hs == 8 ? 4 : hs == 4 ? 3 : hs == 2 ? 2 : hs == 1 ? 1 : 0; };
        const int hs2 = infer_parts(G, 1, 33, 8 * cusm), hs3 = infer_parts(G, 2, 33, 4 * cusm);
        m->stage_kernel[1] = n2[slot_of(hs2)];
        rc |= launch_conv_parts_infer<3, 1, 1, 33, 2>(hs2, m->tm_p1, n, m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
        cv_prof_end(m, 1, st);
        cv_prof_begin(m, 2, st);
        m->stage_kernel[2] = n3[slot_of(hs3)];
        rc |= launch_conv_parts_infer<5, 1, 2, 33>(hs3, m->tm_p2, n, m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st);
        cv_prof_end(m, 2, st);
        cv_prof_begin(m, 3, st);
        // (operand ring depth as for the full topology's smallest passes: 12 while every wave has a SIMD to itself, 4 beyond)
        if ((long)G * s.nb4 <= 4L * cusm) { m->stage_kernel[3] = "dense_small<1, 12, 0>"; rc |= launch_dense_small<1, 12>(m->tm_p3, s.kb4, m->wps7_fc4, P + o[7], a.fc4, m->tm_h4, G, s.nb4, st); }
        else { m->stage_kernel[3] = "dense_small<1, 4, 0>"; rc |= launch_dense_small<1, 4>(m->tm_p3, s.kb4, m->wps7_fc4, P + o[7], a.fc4, m->tm_h4, G, s.nb4, st); }
        cv_prof_end(m, 3, st);
        cv_prof_begin(m, 4, st);
        if (m->variant & 512) {
            m->stage_kernel[4] = "dense_tm<2, 4, 2, 1>";
            rc |= launch_dense<2, 4, 2>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st, 1, 1, nullptr, hd);
            heads_done = true;
        } else {
            m->stage_kernel[4] = "dense_tm<2, 4, 0, 1>";
            rc |= launch_dense<2, 4>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st);
        }
        cv_prof_end(m, 4, st);
    } else {
        if (fuse_front) {
            cv_prof_begin(m, 1, st);
            m->stage_kernel[1] = "conv_tm<3, 1, 1, 1, 33, 1, 0, 1, 2>";
            rc |= launch_conv<3, 1, 1, 1, 33, 1, 0, 1, 2>(nullptr, x, n, W1, B1, a.cout[0], m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
            cv_prof_end(m, 1, st);
        } else {
            cv_prof_begin(m, 0, st);
            m->stage_kernel[0] = "conv1_tm<1, false>";
            conv1_tm<1><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, W1, B1, a.cout[0], (f4 *)m->tm_p1, G);
            cv_prof_end(m, 0, st);
            cv_prof_begin(m, 1, st);
            m->stage_kernel[1] = "conv_tm<3, 1, 1, 1, 33, 0, 0, 1, 2>";
            rc |= launch_conv<3, 1, 1, 1, 33, 0, 0, 1, 2>(m->tm_p1, x, n, W1, B1, a.cout[0], m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
            cv_prof_end(m, 1, st);
        }
        if (m->variant & 256) {
            cv_prof_begin(m, 2, st);
            {
                const size_t lds = (size_t)(40 + 3 * 24) * 1024;
                // groups per workgroup by the estimate rounds of CUs x time per position of a workgroup alone on its CU: 18.4 us
                // with two waves per SIMD (8 groups), 10.2 us with one (4 groups; 2 groups: the same 10.2 -- not built).
                // Measured at 14 sizes: profiles/r06/slim_waves_ab.txt (8 192 candidates: 0.69 -> 0.42 ms per pass)
                int cuss = 256;
                if (device_cus(&cuss)) return 1;
                int wv = 8;
                if (m->inf_slim_waves == 4 || m->inf_slim_waves == 8) wv = m->inf_slim_waves;
                else {
                    const double t8 = (double)(((G + 7) / 8 + cuss - 1) / cuss) * 18.4, t4 = (double)(((G + 3) / 4 + cuss - 1) / cuss) * 10.2;
                    wv = t4 < 0.97 * t8 ? 4 : 8;
                }
                const heads_args *tail = nullptr;
                if (m->variant & 1024) {            // fc5 + heads on the kernel's tail: arguments through a device copy
                    heads_args h3 = hd;
                    h3.wp5p = (const f4 *)m->wp_fc5; h3.bias5 = P + o[9]; h3.nout5 = a.fc5; h3.h5_out = (f4 *)m->tm_h5;
                    h3.keep = m->keep_act;
                    if (tail_args_refresh(m, h3, st, &tail)) return 1;
                    tail_done = true;
                }
#define CV_SLIM_LAUNCH(W) do { if (set_lds(conv3fc4_slim<W>, lds)) return 1; \
                    conv3fc4_slim<W><<<nblk(G, W), W * 64, lds, st>>>((const f4 *)m->tm_p2, (const f4 *)m->wp_conv[2], P + o[5], a.cout[2], \
                                                          (const f4 *)m->wp_fc4, P + o[7], a.fc4, (f4 *)m->tm_h4, G, tail, n, out16); } while (0)
                if (wv == 8) { m->stage_kernel[2] = "conv3fc4_slim<8>"; CV_SLIM_LAUNCH(8); }
                else { m->stage_kernel[2] = "conv3fc4_slim<4>"; CV_SLIM_LAUNCH(4); }
#undef CV_SLIM_LAUNCH
            }
            cv_prof_end(m, 2, st);
            if (tail_done) {
                if (rc) return 1;
                CV_HIP(hipGetLastError());
                m->last_n = n; m->last_impl = 1; m->last_variant = m->variant; m->last_maps = m->keep_act;
                return 0;
            }
        } else {
        cv_prof_begin(m, 2, st);
        m->stage_kernel[2] = "conv_tm<5, 1, 2, 1, 33, 0, 0, 1>";
        rc |= launch_conv<5, 1, 2, 1, 33, 0>(m->tm_p2, x, n, W1, B1, a.cout[0], m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st);
        cv_prof_end(m, 2, st);
        cv_prof_begin(m, 3, st);
        m->stage_kernel[3] = "dense_tm<3, 4, 0, 1>";
        rc |= launch_dense<3, 4>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st);
        cv_prof_end(m, 3, st);
        }
        cv_prof_begin(m, 4, st);
        if (m->variant & 512) {
            m->stage_kernel[4] = "dense_tm<2, 4, 2, 1>";
            rc |= launch_dense<2, 4, 2>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st, 1, 1, nullptr, hd);
            heads_done = true;
        } else {
            m->stage_kernel[4] = "dense_tm<2, 4, 0, 1>";
            rc |= launch_dense<2, 4>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st);
        }
        cv_prof_end(m, 4, st);
    }
    if (rc) return 1;
    CV_HIP(hipGetLastError());
    m->last_n = n;
    m->last_impl = 1;
    m->last_variant = m->variant;
    if (heads_done) return 0;
    cv_prof_begin(m, 5, st);
    if (m->variant & 2) {
        m->stage_kernel[5] = "heads_tm";
        heads_tm<<<nblk(G, 4), 256, 0, st>>>((const f4 *)m->tm_h4, (const f4 *)m->tm_h5, s.nb4, s.nb5,
                                            (const f4 *)m->wp_heads0, (const f4 *)m->wp_heads1, P + o[11], P + o[13],
                                            P + o[15], P + o[17], n, out16, G);
        rc = 0;
    } else {
        m->stage_kernel[5] = "heads_kernel";
        rc = cv_launch_heads(m, m->tm_h4, m->tm_h5, 1, n, out16, st);
    }
    cv_prof_end(m, 5, st);
    CV_HIP(hipGetLastError());
    return rc;
}

// ---------------------------------------------------------------------------
// tile-kernel entry points of the training step (cv_train.hip)
// ---------------------------------------------------------------------------

bool cv_tile_supported(const cv_model *m) { return is_full(m->arch) || is_slim(m->arch); }

int cv_pack_train_weights(cv_model *m, hipStream_t st)
{
    return pack_launch(m, st, CVL_BACKWARD & ~m->packed_valid);
}

// conv1..conv3 (+pools) with the pre-pool activations kept; buffers are TM
int cv_tile_train_convs(cv_model *m, const float *x, int64_t n, float *p1, float *a1, float *p2, float *a2,
                        float *p3, float *a3, hipStream_t st, const std::function<int()> *after_conv1)
{
    const cv_arch &a = m->arch;
    const float *P = m->params;
    const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    int rc = 0;
    // small batches: the positions of a (group, tile) are split over as many waves as fills the chip best (pick_hsplit;
    // config 4's per-rank batch of 1 250 is 79 groups -> 4 parts); beyond train_tiny_groups: equal ranges of the flat
    // (group, row) sequence (conv_parts); same values row for row.  Option train_tiny_groups = 0 keeps one wave per
    // (group, tile).
    if (is_full(a) && m->dbg[1] > 0 && m->dbg[1] < 7) {          // development: forced number of position parts
        conv1_tm<5, true><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, m->wp_conv1, P + o[1], a.cout[0], (f4 *)p1, G, (u32x2 *)a1);
        if (after_conv1 && (*after_conv1)()) return 1;       // (side-stream work that should not share the chip with conv1)
        rc |= launch_conv_parts<2, 1, 2, 4, 29, 1>(m->dbg[1], p1, x, n, m->wp_conv[1], P + o[3], a.cout[1], p2, G, st, a2);
        rc |= launch_conv_parts<3, 2, 3, 3, 26, 1>(m->dbg[1], p2, x, n, m->wp_conv[2], P + o[5], a.cout[2], p3, G, st, a3);
        CV_HIP(hipGetLastError());
        return rc;
    }
    if (is_full(a)) {
        conv1_tm<5, true><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, m->wp_conv1, P + o[1], a.cout[0], (f4 *)p1, G, (u32x2 *)a1);
        if (after_conv1 && (*after_conv1)()) return 1;       // (side-stream work that should not share the chip with conv1)
        // conv2 stays on position parts (measured at train.py's batch on one box: flat ranges -- dbg1 = 8 -- make the
        // kernel 4 us shorter on its own and the step 40 us longer: its 3 waves per SIMD then hold every slot to the
        // end and the weight packing on the side stream waits)
        rc |= launch_conv_parts<2, 1, 2, 4, 29, 1>(conv_parts(m, m->dbg[1] == 8 ? 0 : 9, G, 2, 26, 3, 4), p1, x, n, m->wp_conv[1], P + o[3], a.cout[1], p2, G, st, a2);
        const int hs3 = conv_parts(m, m->dbg[1], G, 3, 24, 2, 4);
        // one wave per (group, tile) or flat ranges: the rotating-window kernel (dbg7 = 1: conv_tm; dbg7 = 2 / 3: flat
        // ranges sized for 2 / 3 waves per SIMD)
        if ((hs3 == 1 || hs3 == 0) && m->dbg[7] != 1)
            rc |= launch_conv3_rot<2, 3, 26, 4, 2, true>(p2, m->wp_conv[2], P + o[5], a.cout[2], p3, G, st, a3,
                                                         hs3 == 0 ? (m->dbg[7] == 3 ? 3072 : 2048) : 0);
        else
            rc |= launch_conv_parts<3, 2, 3, 3, 26, 1>(hs3, p2, x, n, m->wp_conv[2], P + o[5], a.cout[2], p3, G, st, a3);
    } else {
        conv1_tm<1, true><<<nblk((int64_t)G * 4, 4), 256, 0, st>>>(x, n, m->wp_conv1, P + o[1], a.cout[0], (f4 *)p1, G, (u32x2 *)a1);
        if (after_conv1 && (*after_conv1)()) return 1;       // (side-stream work that should not share the chip with conv1)
        rc |= launch_conv_parts<3, 1, 1, 1, 33, 1, 2>(conv_parts(m, m->dbg[1], G, 1, 33, 0, 4), p1, x, n, m->wp_conv[1], P + o[3], a.cout[1], p2, G, st, a2);
        rc |= launch_conv_parts<5, 1, 2, 1, 33, 1>(conv_parts(m, m->dbg[1], G, 2, 33, 0, 4), p2, x, n, m->wp_conv[2], P + o[5], a.cout[2], p3, G, st, a3);
    }
    CV_HIP(hipGetLastError());
    return rc;
}

int cv_tile_dense_fwd(cv_model *m, int layer, const float *in_tm, float *out_tm, int64_t n, hipStream_t st, float *part,
                      const cv_train_dropout *drop, bool *drop_done)
{
    if (drop_done) *drop_done = false;
    const cv_arch &a = m->arch;
    const float *P = m->params;
    const int64_t *o = m->poff;
    const cv_shapes &s = m->sh;
    const int G = (int)((n + 15) / 16);
    // (the layout each branch reads is the one cv_pack_for_training packed for this G: fc4_train_layout / fc5_train_layout)
    if (is_full(a)) {
        if (layer == 4) {
            const unsigned lay = fc4_train_layout(m, G);
            if (cv_layout_current(m, lay, "fc4 forward (training pass)")) return 1;
            // tiny batches: 288 dependent k steps at ~0.9 us each are the longest kernel of the step; eight k ranges
            // (CV_DENSE_KSPLIT) run side by side instead and a second pass adds them up in order
            // (option train_ksplit 0 with variant bit 7 cleared also reads the 3-slab layout at these sizes: that is the
            // single-chain three-slab kernel below, not this branch)
            if (lay == CVL_FC4S3 && G <= m->tiny_g && m->train_ksplit) {
                if (!part) { cv_set_error("k-split fc4 forward without its scratch (internal)"); return 1; }
                cv_dropout_args dr = cv_dropout_args();
                if (drop && drop_done) {            // the alpha-dropout of fc4 rides on the second pass of the k-split
                    dr.d4 = drop->d4; dr.amask = drop->amask; dr.nunits = a.fc4; dr.rate = drop->rate; dr.seed = drop->seed;
                    dr.step = drop->step; dr.cand0 = drop->cand0;
                    *drop_done = true;
                }
                return launch_dense<7, 8>(in_tm, s.kb4, m->wps_fc4, P + o[7], a.fc4, out_tm, G, st, 3, CV_DENSE_KSPLIT, part,
                                          heads_args(), dr);
            }
            // (dense_small beyond the tiny range loses: at 625 groups its 4 375 waves re-read the weight matrix from L2
            // 7 x 625 times -- 2.63 against 2.41 ms per step)
            if (lay == CVL_FC4S7)
                return launch_dense_small<3, 8>(in_tm, s.kb4, m->wps7_fc4, P + o[7], a.fc4, out_tm, G, 7, st);
            // (measured at train.py's batch of 10 000, no gain: two k ranges of the 3-slab form; 3 / 4 / 6 / 8 k ranges of the
            // two-groups-per-wave, all-21-tiles form -- 2.47 / 2.36 / 2.25 / 2.39 ms per step against 2.25)
            if (lay == CVL_FC4S3) {
                heads_args hd = heads_args();
                if (drop && drop_done && m->dbg[2] != 3) {      // the alpha-dropout rides on the kernel's store (dbg2 = 3: dropout_tm)
                    hd.drop.d4 = drop->d4; hd.drop.amask = drop->amask; hd.drop.nunits = a.fc4; hd.drop.rate = drop->rate;
                    hd.drop.seed = drop->seed; hd.drop.step = drop->step; hd.drop.cand0 = drop->cand0;
                    *drop_done = true;
                }
                if (m->inf_rag_s >= 0) return launch_dense_rag(in_tm, s.kb4, m->wps_fc4, P + o[7], a.fc4, out_tm, G, s.nb4, m->inf_rag_s, st, hd.drop);
                return launch_dense<7, 8>(in_tm, s.kb4, m->wps_fc4, P + o[7], a.fc4, out_tm, G, st, 3, 1, nullptr, hd);
            }
            return launch_dense<21, 8, 0, 2>(in_tm, s.kb4, m->wp_fc4, P + o[7], a.fc4, out_tm, G, st);      // slices of more than 2 048 groups: the inference kernel
        }
        // fc5 (21 k fragments): one wave per (group, slab of 4 output fragments), no barriers, weights straight from L2 --
        // up to a slice of 2 048 groups (dense_tm<11, 4> with its 4-group workgroups took 25 us at 625 groups)
        const unsigned lay5 = fc5_train_layout(m, G);
        if (cv_layout_current(m, lay5, "fc5 forward (training pass)")) return 1;
        if (lay5 == CVL_FC5S3)
            return launch_dense_small<4, 7>(in_tm, s.nb4, m->wps3_fc5, P + o[9], a.fc5, out_tm, G, 3, st, s.nb5);
        return launch_dense<11, 4>(in_tm, s.nb4, m->wp_fc5, P + o[9], a.fc5, out_tm, G, st);
    }
    if (cv_layout_current(m, layer == 4 ? CVL_FC4 : CVL_FC5, "dense forward (training pass)")) return 1;
    if (layer == 4) {
        // slim fc4 is 396 dependent k steps of 12 MFMAs on 4-wave workgroups, one barrier each: 133 us at 625 groups for
        // 40 us of matrix work, the longest kernel of the slim step.  With the k-split scratch (training passes, option
        // train_ksplit) eight k ranges run side by side at ANY batch and dense_ksum adds them in order (+ bias, SELU and the
        // alpha-dropout): a fixed order, within the gradient tolerance of the single chain, never used by cv_forward.
        if (part) {
            cv_dropout_args dr = cv_dropout_args();
            if (drop && drop_done) {
                dr.d4 = drop->d4; dr.amask = drop->amask; dr.nunits = a.fc4; dr.rate = drop->rate; dr.seed = drop->seed;
                dr.step = drop->step; dr.cand0 = drop->cand0;
                *drop_done = true;
            }
            return launch_dense<3, 4>(in_tm, s.kb4, m->wp_fc4, P + o[7], a.fc4, out_tm, G, st, 1, CV_DENSE_KSPLIT, part,
                                      heads_args(), dr);
        }
        return launch_dense<3, 4>(in_tm, s.kb4, m->wp_fc4, P + o[7], a.fc4, out_tm, G, st);
    }
    return launch_dense<2, 4>(in_tm, s.nb4, m->wp_fc5, P + o[9], a.fc5, out_tm, G, st);
}

// Tiny batches of the full topology with the k-split fc4 forward: fc4's eight k ranges, then ONE kernel for their sum,
// bias, SELU, alpha-dropout, fc5, the heads, the losses, the head gradients and the fc5-side data gradient
// (train_tail_tm).  *done = false: not this regime (or dbg2 = 5) -- the caller runs the layers one by one.
int cv_tile_train_tail(cv_model *m, const float *p3_tm, float *h4_tm, float *h5_tm, const float *y, int64_t n, int want_grad,
                       float *g16, float *g5pre_tm, float *part, const cv_train_dropout *drop, hipStream_t st, bool *done)
{
    *done = false;
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const float *P = m->params; const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    if (G <= 0 || !is_full(a) || !part || !drop || G > m->tiny_g || m->dbg[2] == 5) return 0;
    if (fc4_train_layout(m, G) != CVL_FC4S3 || fc5_train_layout(m, G) != CVL_FC5S3 || s.nb4 != 21 || s.nb5 != 11) return 0;
    if (cv_layout_current(m, CVL_FC4S3 | CVL_FC5S3 | CVL_HEADS, "training forward tail")) return 1;
    if (m->loss_rows_used + G > m->loss_rows_cap) { cv_set_error("train_tail_tm: loss row buffer too small (internal)"); return 1; }
    double *rows = m->loss_rows + (size_t)m->loss_rows_used * 4;
    m->loss_rows_used += G;                       // one row per group (cv_train.hip t_loss_header adds the rows in order)
    const int KR = CV_DENSE_KSPLIT;
    {
        auto k = dense_tm<7, 8, 0, 1>;
        const size_t lds = (size_t)3 * 8 * 1024;
        if (set_lds(k, lds)) return 1;
        k<<<dim3(nblk(G, 8), 3, KR), 512, lds, st>>>((const f4 *)p3_tm, s.kb4, (const f4 *)m->wps_fc4, P + o[7], a.fc4,
                                                      (f4 *)part, G, 21, heads_args());
    }
    cv_dropout_args dr = cv_dropout_args();
    dr.d4 = drop->d4; dr.amask = drop->amask; dr.nunits = a.fc4; dr.rate = drop->rate; dr.seed = drop->seed;
    dr.step = drop->step; dr.cand0 = drop->cand0;
    train_tail_tm<21, 11, 8, true><<<G, 512, 0, st>>>((const f4 *)part, KR, G, P + o[7], a.fc4, (f4 *)h4_tm, dr, (const f4 *)m->wps3_fc5,
                                             P + o[9], a.fc5, (f4 *)h5_tm, (const f4 *)m->wp_heads0, (const f4 *)m->wp_heads1, P + o[11],
                                             P + o[13], P + o[15], P + o[17], P + o[12], P + o[14], P + o[16], y, n, want_grad, g16,
                                             (f4 *)g5pre_tm, rows, m->wp_heads12);
    CV_HIP(hipGetLastError());
    *done = true;
    return 0;
}

// Larger batches of the full topology (up to 2 048 groups; train_sched bit 10): fc5, the heads, losses, head gradients and
// the fc5-side data gradient in one kernel behind fc4's own (which has stored the dropped-out output d4_tm) -- the same
// kernel as the tiny-batch tail without its first step, four waves per group.  *done = false: not this regime.
int cv_tile_train_fc5_heads(cv_model *m, float *d4_tm, float *h5_tm, const float *y, int64_t n, int want_grad, float *g16,
                            float *g5pre_tm, hipStream_t st, bool *done)
{
    *done = false;
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const float *P = m->params; const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    if (G <= 0 || !is_full(a) || !(m->sched & 1024) || fc5_train_layout(m, G) != CVL_FC5S3 || s.nb4 != 21 || s.nb5 != 11) return 0;
    if (cv_layout_current(m, CVL_FC5S3 | CVL_HEADS, "training forward fc5 + heads")) return 1;
    if (m->loss_rows_used + G > m->loss_rows_cap) { cv_set_error("train_tail_tm: loss row buffer too small (internal)"); return 1; }
    double *rows = m->loss_rows + (size_t)m->loss_rows_used * 4;
    m->loss_rows_used += G;
    cv_dropout_args dr = cv_dropout_args();
    dr.d4 = d4_tm;
    train_tail_tm<21, 11, 4, false><<<G, 256, 0, st>>>(nullptr, 1, G, P + o[7], a.fc4, nullptr, dr, (const f4 *)m->wps3_fc5,
                                                       P + o[9], a.fc5, (f4 *)h5_tm, (const f4 *)m->wp_heads0, (const f4 *)m->wp_heads1,
                                                       P + o[11], P + o[13], P + o[15], P + o[17], P + o[12], P + o[14], P + o[16], y, n,
                                                       want_grad, g16, (f4 *)g5pre_tm, rows, m->wp_heads12);
    CV_HIP(hipGetLastError());
    *done = true;
    return 0;
}

// full topology: fc4 data gradient + max-pool backward + SELU' of conv3 in one kernel (dense_dgrad_unpool):
// g_tm = fc4 pre-activation gradient, pooled / codes = conv3's pooled output and window-offset codes, gpre = conv3's
// pre-activation gradient (hc[2] rows).  Few groups: one group per wave, more workgroups.
int cv_tile_fc4_dgrad_unpool(cv_model *m, const float *g_tm, const float *pooled, const float *codes, float *gpre, int64_t n,
                             hipStream_t st)
{
    const cv_shapes &s = m->sh;
    const int G = (int)((n + 15) / 16);
    if (!is_full(m->arch) || !m->wpr_fc4) { cv_set_error("cv_tile_fc4_dgrad_unpool: full topology only"); return 1; }
    if (cv_layout_current(m, CVL_DFC4, "fc4 data gradient")) return 1;
    const size_t lds = (size_t)3 * 24 * 1024;
    const int HO = s.hp[2], NT = s.ntile[2];
    // (workgroup shape measured at 625 groups: 8 waves x 2 groups 298 us, 4 waves x 2 groups 298 us, 8 waves x 1 group 292 us;
    // round 6: the twelve column workgroups of a block of groups numbered onto one XCD so that they share the block's gradient
    // fragments in its L2 -- what halved fc4's forward traffic -- RAISES this kernel's traffic, 520 -> 541 MB per step, and the
    // step by 40 us: twelve columns then stream twelve different 576 KB weight slabs through one 4 MB L2 at a time, where the
    // (blocks, columns) grid runs one column's workgroups together.  Not kept.)
    if (G > 512) {
        auto k = dense_dgrad_unpool<21, 3, 8, 2>;
        if (set_lds(k, lds)) return 1;
        k<<<dim3(nblk(G, 16), 4 * NT), 512, lds, st>>>((const f4 *)g_tm, (const f4 *)m->wpr_fc4, (const f4 *)pooled,
                                                       (const u32x2 *)codes, (f4 *)gpre, G, HO, NT);
    } else {
        // few groups: row parts (gridDim.z; each recomputes the two windows in front of its rows) shorten the 24-row walk of a
        // workgroup, 4-wave workgroups put one wave on each SIMD of twice as many CUs.  Same values row for row.  Same-box
        // steps (profiles/r06/dgrad_unpool_parts_ab.txt; 8 waves x 1 part / the form kept):  20 groups 0.424 / 0.379 ms
        // (4 waves x 4 parts), 40 groups 0.449 / 0.403 (4 waves x 2 parts), 60 groups 0.465 / 0.451 and 79 groups 0.498 /
        // 0.485 (8 waves x 2 parts); from 100 groups one part is the fastest again (0.587 / 0.597): the weight gradients on the
        // side streams use the other CUs meanwhile.  dbg6 = parts (+ 100: 4-wave workgroups) for A/B.
        int parts = G <= 24 ? 4 : G <= 80 ? 2 : 1;
        bool four = G <= 48;
        if (m->dbg[6] > 0) { parts = m->dbg[6] % 100; four = m->dbg[6] >= 100; }
        if (parts > 6) parts = 6;
        if (parts < 1) parts = 1;
        if (four) {
            auto k4 = dense_dgrad_unpool<21, 3, 4, 1>;
            if (set_lds(k4, lds)) return 1;
            k4<<<dim3(nblk(G, 4), 4 * NT, parts), 256, lds, st>>>((const f4 *)g_tm, (const f4 *)m->wpr_fc4, (const f4 *)pooled,
                                                                  (const u32x2 *)codes, (f4 *)gpre, G, HO, NT);
        } else {
            auto k = dense_dgrad_unpool<21, 3, 8, 1>;
            if (set_lds(k, lds)) return 1;
            k<<<dim3(nblk(G, 8), 4 * NT, parts), 512, lds, st>>>((const f4 *)g_tm, (const f4 *)m->wpr_fc4, (const f4 *)pooled,
                                                                 (const u32x2 *)codes, (f4 *)gpre, G, HO, NT);
        }
    }
    CV_HIP(hipGetLastError());
    return 0;
}

// gF[k] = sum_j g4pre[j] W4[k][j]  (input TM with nb4 fragments, output TM with kb4 fragments)
int cv_tile_fc4_dgrad(cv_model *m, const float *g_tm, float *gin_tm, int64_t n, hipStream_t st, const float *act_below)
{
    const cv_shapes &s = m->sh;
    const int G = (int)((n + 15) / 16);
    heads_args hd;
    hd.dact = (const f4 *)act_below;        // not null: the result is already the pre-activation gradient of conv3 (no pooling)
    if (cv_layout_current(m, CVL_DFC4, "fc4 data gradient")) return 1;
    return launch_dense<24, 8, 1>(g_tm, s.nb4, m->wpd_fc4, nullptr, 0, gin_tm, G, st, s.kb4 / 24, 1, nullptr, hd);
}

// g(d4)[k] = sum_j g5pre[j] W5[k][j]  (input TM with nb5 fragments, output TM with nb4 fragments)
// g16 != NULL: the result is already fc4's PRE-ACTIVATION gradient -- the base head's contribution (g16, the base head's
// weights), the dropout factor (mask_tm) and selu'(fc4 output act_tm) ride on the store
int cv_tile_fc5_dgrad(cv_model *m, const float *g_tm, float *gin_tm, int64_t n, hipStream_t st, const float *g16,
                      const float *mask_tm, const float *act_tm)
{
    const cv_shapes &s = m->sh;
    const int G = (int)((n + 15) / 16);
    if (cv_layout_current(m, CVL_DFC5, "fc5 data gradient")) return 1;
    heads_args hd;
    if (g16) {
        hd.hg_g16 = g16; hd.hg_wb = m->params + m->poff[10]; hd.hg_mask = (const f4 *)mask_tm; hd.hg_act = (const f4 *)act_tm;
        hd.hg_n = n; hd.hg_K = m->arch.fc4;
    }
    // full: three slabs of 7 output fragments -- as one workgroup per 8 groups with all 21 the kernel took 32 us at ANY
    // batch (2 waves x 11 k steps x 84 MFMAs per SIMD on 10 .. 79 CUs); the values do not depend on the slab width
    if (is_full(m->arch)) return launch_dense<7, 8, 1>(g_tm, s.nb5, m->wpd_fc5, nullptr, 0, gin_tm, G, st, 3, 1, nullptr, hd);
    return launch_dense<3, 4, 1>(g_tm, s.nb5, m->wpd_fc5, nullptr, 0, gin_tm, G, st, 1, 1, nullptr, hd);
}

// layer 1 = conv2, 2 = conv3: gradient w.r.t. the layer input from the pre-activation gradient
int cv_tile_conv_dgrad(cv_model *m, int layer, const float *g_tm, float *gin_tm, int64_t n, hipStream_t st, const float *act_below)
{
    float *act = const_cast<float *>(act_below);      // conv_tm MODE 2 reads it (selu' factor of a layer without pooling)
    const cv_arch &a = m->arch;
    const int G = (int)((n + 15) / 16);
    const float *W = m->wpd_conv[layer];
    if (cv_layout_current(m, CVL_DCONV, "convolution data gradient")) return 1;
    if (is_full(a) && m->dbg[0] > 0 && m->dbg[0] < 7) {          // development: forced number of position parts
        if (layer == 2) return launch_conv_parts<3, 3, 2, 1, 26, 2>(m->dbg[0], g_tm, nullptr, n, W, nullptr, 0, gin_tm, G, st);
        return launch_conv_parts<2, 2, 1, 1, 29, 2>(m->dbg[0], g_tm, nullptr, n, W, nullptr, 0, gin_tm, G, st);
    }
    if (is_full(a)) {
        if (layer == 2)
            return launch_conv_parts<3, 3, 2, 1, 26, 2>(conv_parts(m, m->dbg[0], G, 2, 26, 0, 8), g_tm, nullptr, n, W, nullptr, 0, gin_tm, G, st);
        return launch_conv_parts<2, 2, 1, 1, 29, 2>(conv_parts(m, m->dbg[0], G, 1, 29, 0, 8), g_tm, nullptr, n, W, nullptr, 0, gin_tm, G, st);
    }
    if (layer == 2)
        return launch_conv_parts<5, 2, 1, 1, 33, 2>(conv_parts(m, m->dbg[0], G, 1, 33, 0, 8), g_tm, nullptr, n, W, nullptr, 0, gin_tm, G, st, act);
    return launch_conv_parts<3, 1, 1, 1, 33, 2>(conv_parts(m, m->dbg[0], G, 1, 33, 0, 8), g_tm, nullptr, n, W, nullptr, 0, gin_tm, G, st, act);
}

// heads of the training pass in one launch: products, losses (added to loss[0..3]), gradients w.r.t. the 16
// pre-activations (g16 [n][16], when want_grad) and the fc5-side data gradient times selu'(fc5) (g5pre_tm, when not null)
int cv_tile_heads_train(cv_model *m, const float *d4_tm, const float *h5_tm, const float *y, int64_t n, int want_grad,
                        float *g16, float *g5pre_tm, hipStream_t st)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const float *P = m->params; const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    if (G <= 0) return 0;
#define CV_HT(NB5) heads_train_tm<NB5><<<nblk(G, 4), 256, 0, st>>>((const f4 *)d4_tm, (const f4 *)h5_tm, s.nb4, (const f4 *)m->wp_heads0, \
        (const f4 *)m->wp_heads1, P + o[11], P + o[13], P + o[15], P + o[17], P + o[12], P + o[14], P + o[16], a.fc5, y, n, want_grad, \
        g16, (f4 *)g5pre_tm, rows, G, m->wp_heads12)
    // the block sums of this slice: rows [loss_rows_used, + blocks) of the step's row buffer (cv_train.hip t_loss_finish)
    const int64_t blocks = nblk(G, 4);
    if (m->loss_rows_used + blocks > m->loss_rows_cap) { cv_set_error("heads_train_tm: loss row buffer too small (internal)"); return 1; }
    double *rows = m->loss_rows + (size_t)m->loss_rows_used * 4;
    m->loss_rows_used += blocks;
    if (s.nb5 == 11) CV_HT(11);
    else if (s.nb5 == 2) CV_HT(2);
    else { cv_set_error("heads_train_tm: %d fc5 fragments not instantiated", s.nb5); return 1; }
#undef CV_HT
    CV_HIP(hipGetLastError());
    return 0;
}

// heads of the training pass: pre-activations of the 16 outputs from the dropped-out fc4 output and fc5 (tile-major)
int cv_tile_heads_pre(cv_model *m, const float *d4_tm, const float *h5_tm, int64_t n, float *pre16, hipStream_t st)
{
    const cv_shapes &s = m->sh;
    const float *P = m->params; const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    if (G <= 0) return 0;
    heads_pre_tm<<<nblk(G, 4), 256, 0, st>>>((const f4 *)d4_tm, (const f4 *)h5_tm, s.nb4, s.nb5, (const f4 *)m->wp_heads0,
                                            (const f4 *)m->wp_heads1, P + o[11], P + o[13], P + o[15], P + o[17], n, pre16, G);
    CV_HIP(hipGetLastError());
    return 0;
}

int cv_dropout_tm(cv_model *m, const float *h4, float *d4, float *amask, int64_t n, float rate, uint64_t seed,
                  uint64_t step, int64_t cand0, hipStream_t st)
{
    int64_t G = (n + 15) / 16;
    dropout_tm<<<nblk(G * m->sh.nb4 * 256, 256), 256, 0, st>>>(h4, d4, amask, m->sh.nb4, m->arch.fc4, G, rate, seed,
                                                              step, cand0);
    CV_HIP(hipGetLastError());
    return 0;
}

// layer 4 = fc4 (x = pool3 TM, g = fc4 pre-activation gradient TM), 5 = fc5.  The candidate range is split over