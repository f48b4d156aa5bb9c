// cv_train.hip -- loss, backward and dropout of the training step.
//
// Replaces the session.run((loss, training_op, ...)) of train / trainNoRT and
// session.run(loss) of getLoss / getLossNoRT
// (/root/reference/clairvoyante/clairvoyante_v3.py:183-227):
//   loss (v3.py:140-151)  = sum (sigmoid - y)^2 + sum -y*log_softmax(logits) x3
//                           + lambda * sum_{kernels} sum(w^2)/2      -- SUMS over the batch
//   alpha-dropout on fc4 (selu.py:34-69): keep mask from a counter-based hash of
//   (seed, step, candidate, unit) -- the reference's stream is unseeded TF state, so
//   only the distribution can match.
// Forward pass, data gradients and weight gradients all run on the MFMA tile kernels
// (cv_kernels_mfma.hip: conv_tm MODE 1/2, dense_tm EPI 1, wgrad_*_cm); this file holds the
// element-wise steps on tile-major buffers (SELU', max-pool routing, dropout, loss, head data
// gradients) and the sequence of the step.  Gradients accumulate into the flat buffer
// (cv_grad_buffer) slice by slice in stream order; every weight gradient is reduced in a
// fixed order (no float atomics), so a step is reproducible bit for bit.  Architectures the
// tile kernels do not cover fall back to the all-plain path (train_slice_plain, atomics).
#include "cv_internal.hpp"
#include "cv_math.hpp"
#include "cv_unpool.hpp"

namespace {

inline unsigned nblk(int64_t total, int bs) { return (unsigned)((total + bs - 1) / bs); }

__device__ __forceinline__ uint32_t hash_u32(uint64_t a)
{
    // splitmix64 finaliser
    a += 0x9E3779B97F4A7C15ull;
    a = (a ^ (a >> 30)) * 0xBF58476D1CE4E5B9ull;
    a = (a ^ (a >> 27)) * 0x94D2049BB133111Bull;
    a = a ^ (a >> 31);
    return (uint32_t)(a >> 32);
}

// ---- forward pieces ------------------------------------------------------------
__global__ void t_conv_pre(const float *__restrict__ in, const float *__restrict__ w,
                           const float *__restrict__ bias, float *__restrict__ pre, int64_t n, int H,
                           int cin, int kh, int cout)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * H * 4 * cout) return;
    int co = (int)(t % cout);
    int64_t r = t / cout;
    int wo = (int)(r % 4); r /= 4;
    int h = (int)(r % H);
    int64_t i = r / H;
    const int padt = (kh - 1) / 2;
    const float *xi = in + (size_t)i * H * 4 * cin;
    float acc = 0.0f;
    for (int a = 0; a < kh; a++) {
        int hi = h + a - padt;
        if (hi < 0 || hi >= H) continue;
        for (int b = 0; b < 4; b++) {
            int wi = wo + b - 1;
            if (wi < 0 || wi >= 4) continue;
            const float *xr = xi + ((size_t)hi * 4 + wi) * cin;
            const float *wr = w + ((size_t)(a * 4 + b) * cin) * cout + co;
            for (int ci = 0; ci < cin; ci++) acc = __builtin_fmaf(xr[ci], wr[(size_t)ci * cout], acc);
        }
    }
    pre[t] = acc + bias[co];
}

// pooled = max over the window of selu(pre)
__global__ void t_pool_selu(const float *__restrict__ pre, float *__restrict__ out, int64_t n, int H, int c,
                            int p)
{
    int Ho = H - p + 1, row = 4 * c;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Ho * row) return;
    int e = (int)(t % row);
    int64_t r = t / row;
    int h = (int)(r % Ho);
    int64_t i = r / Ho;
    const float *b = pre + ((size_t)i * H + h) * row + e;
    float m = b[0];
    for (int d = 1; d < p; d++) m = fmaxf(m, b[(size_t)d * row]);
    out[t] = cvm::selu(m);     // selu is monotone: max(selu(x)) == selu(max(x))
}

__global__ void t_dense_pre(const float *__restrict__ x, const float *__restrict__ w,
                            const float *__restrict__ bias, float *__restrict__ y, int64_t n, int K, int N)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N) return;
    int j = (int)(t % N);
    int64_t i = t / N;
    const float *xi = x + (size_t)i * K;
    float acc = 0.0f;
    for (int k = 0; k < K; k++) acc = __builtin_fmaf(xi[k], w[(size_t)k * N + j], acc);
    y[t] = acc + bias[j];
}

// d4 = alpha-dropout(selu(fc4pre)); mask stored as a*keep (0 when dropped)
__global__ void t_fc4_act(const float *__restrict__ pre, float *__restrict__ d4, float *__restrict__ amask,
                          int64_t n, int N, float rate, uint64_t seed, uint64_t step, int64_t cand0)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N) return;
    float v = cvm::selu(pre[t]);
    if (rate > 0.0f) {
        const float ap = -1.7580993408473766f;
        float q = 1.0f - rate;
        float a = sqrtf(1.0f / (q * ((1.0f - q) * (ap * ap) + 1.0f)));
        float b = 0.0f - a * ((1.0f - q) * ap);
        uint64_t ctr = (seed * 0x9E3779B97F4A7C15ull) ^ (step << 40) ^ (uint64_t)(cand0 * N + t);
        float u = (float)(hash_u32(ctr) >> 8) * (1.0f / 16777216.0f);   // [0,1)
        float keep = floorf(q + u);                                     // selu.py:53-56
        v = a * (v * keep + ap * (1.0f - keep)) + b;
        amask[t] = a * keep;
    } else {
        amask[t] = 1.0f;
    }
    d4[t] = v;
}

__global__ void t_selu_act(const float *__restrict__ pre, float *__restrict__ act, int64_t total)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) act[t] = cvm::selu(pre[t]);
}

// heads: pre-activations, outputs, losses and d loss / d head-pre-activation.
// 16 candidates x 16 outputs per block.
__global__ __launch_bounds__(256) void t_heads(
    const float *__restrict__ d4, const float *__restrict__ h5, int K4, int K5,
    const float *__restrict__ wb, const float *__restrict__ bb, const float *__restrict__ wz,
    const float *__restrict__ bz, const float *__restrict__ wt, const float *__restrict__ bt,
    const float *__restrict__ wl, const float *__restrict__ bl, const float *__restrict__ y, int64_t n,
    float *__restrict__ ghpre, double *__restrict__ loss)
{
    __shared__ float pre[16][17];
    __shared__ double part[4];
    int c = threadIdx.x >> 4, j = threadIdx.x & 15;
    if (threadIdx.x < 4) part[threadIdx.x] = 0.0;
    int64_t cand = (int64_t)blockIdx.x * 16 + c;
    int64_t cl = cand < n ? cand : n - 1;
    const float *w; const float *b; int idx, nh, K; const float *src;
    if (j < 4)       { w = wb; b = bb; idx = j;      nh = 4; K = K4; src = d4; }
    else if (j < 6)  { w = wz; b = bz; idx = j - 4;  nh = 2; K = K5; src = h5; }
    else if (j < 10) { w = wt; b = bt; idx = j - 6;  nh = 4; K = K5; src = h5; }
    else             { w = wl; b = bl; idx = j - 10; nh = 6; K = K5; src = h5; }
    float acc = 0.0f;
    const float *xi = src + (size_t)cl * K;
    for (int k = 0; k < K; k++) acc = __builtin_fmaf(xi[k], w[(size_t)k * nh + idx], acc);
    pre[c][j] = acc + b[idx];
    __syncthreads();
    if (cand < n && j < 4) {
        const float *yi = y + (size_t)cand * 16;
        float *g = ghpre ? ghpre + (size_t)cand * 16 : nullptr;
        double l = 0.0;
        if (j == 0) {
            for (int k = 0; k < 4; k++) {
                float s = cvm::sigmoid(pre[c][k]);
                float d = s - yi[k];
                l += (double)d * d;
                if (g) g[k] = 2.0f * d * s * (1.0f - s);
            }
        } else {
            const int off = j == 1 ? 4 : (j == 2 ? 6 : 10);
            const int cnt = j == 1 ? 2 : (j == 2 ? 4 : 6);
            float lg[6], p[6];
            float mx = -__builtin_inff();
            for (int k = 0; k < cnt; k++) { lg[k] = cvm::selu(pre[c][off + k]) + 1e-10f; mx = fmaxf(mx, lg[k]); }
            float se = 0.0f, ysum = 0.0f;
            for (int k = 0; k < cnt; k++) { p[k] = cvm::expf_fixed(lg[k] - mx); se += p[k]; ysum += yi[off + k]; }
            float lse = mx + logf(se);
            for (int k = 0; k < cnt; k++) {
                l += -(double)yi[off + k] * (double)(lg[k] - lse);
                if (g) g[off + k] = (p[k] / se * ysum - yi[off + k]) * cvm::selu_grad(pre[c][off + k]);
            }
        }
        atomicAdd(&part[j], l);
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&loss[threadIdx.x], part[threadIdx.x]);
}

struct l2_args { const float *w[CV_NUM_PARAMS / 2]; int64_t count[CV_NUM_PARAMS / 2]; };

// sum of squares / 2 of every kernel (blockIdx.y), one launch
__global__ void t_l2(l2_args a, double *__restrict__ out, double *__restrict__ rows)
{
    __shared__ double sh[256];
    const float *__restrict__ w = a.w[blockIdx.y];
    const int64_t count = a.count[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < count; i += 4 * stride) {          // four loads in flight
        const float a = w[i], b = w[i + stride], c = w[i + 2 * stride], d = w[i + 3 * stride];
        s0 += (double)a * (double)a; s1 += (double)b * (double)b; s2 += (double)c * (double)c; s3 += (double)d * (double)d;
    }
    for (; i < count; i += stride) s0 += (double)w[i] * (double)w[i];
    sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    // rows != NULL (tile path): this block's sum to its own slot [kernel][block] -- t_loss_finish adds the slots in a
    // fixed order; else one atomic per block (the all-plain fallback path)
    if (threadIdx.x == 0) {
        if (rows) rows[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sh[0] * 0.5;
        else if (sh[0] != 0.0) atomicAdd(out, sh[0] * 0.5);
    }
}

// ---- backward pieces -------------------------------------------------------------
// dW[k][j] += sum_n x[n][k] * g[n][j] ; db[j] += sum_n g[n][j] (row k == K)
// grid: (ceil((K+1)*N / 256), nslices); each slice covers a range of n.
__global__ void b_dense_wgrad(const float *__restrict__ x, int ldx, const float *__restrict__ g, int ldg,
                              int64_t n, int K, int N, float *__restrict__ dw, float *__restrict__ db)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)(K + 1) * N) return;
    int j = (int)(t % N);
    int k = (int)(t / N);
    int64_t per = (n + gridDim.y - 1) / gridDim.y;
    int64_t n0 = per * blockIdx.y, n1 = n0 + per < n ? n0 + per : n;
    float acc = 0.0f;
    if (k < K) {
        for (int64_t i = n0; i < n1; i++) acc = __builtin_fmaf(x[(size_t)i * ldx + k], g[(size_t)i * ldg + j], acc);
        atomicAdd(&dw[(size_t)k * N + j], acc);
    } else {
        for (int64_t i = n0; i < n1; i++) acc += g[(size_t)i * ldg + j];
        atomicAdd(&db[j], acc);
    }
}

// gx[n][k] (+)= sum_j g[n][j] * w[k][j]
__global__ void b_dense_dgrad(const float *__restrict__ g, int ldg, const float *__restrict__ w, int64_t n,
                              int K, int N, float *__restrict__ gx, int accumulate)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * K) return;
    int k = (int)(t % K);
    int64_t i = t / K;
    const float *gi = g + (size_t)i * ldg;
    const float *wr = w + (size_t)k * N;
    float acc = 0.0f;
    for (int j = 0; j < N; j++) acc = __builtin_fmaf(gi[j], wr[j], acc);
    gx[t] = accumulate ? gx[t] + acc : acc;
}

__device__ __forceinline__ float selu_grad_from_out(float y) { return cv_selu_grad_from_out(y); }

// g_pre = g_act * selu'(pre) (optionally * amask for the dropout layer)
__global__ void b_selu(const float *__restrict__ gact, const float *__restrict__ pre,
                       const float *__restrict__ amask, float *__restrict__ gpre, int64_t total)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    float g = gact[t];
    if (amask) g *= amask[t];
    gpre[t] = g * cvm::selu_grad(pre[t]);
}

// max-pool backward (route to the first maximum of each window) fused with selu':
// gpre[n][h][e] = selu'(pre) * sum_{ho: argmax(window ho) == h} gpool[n][ho][e]
__global__ void b_pool_selu(const float *__restrict__ gpool, const float *__restrict__ pre,
                            float *__restrict__ gpre, int64_t n, int H, int c, int p)
{
    int Ho = H - p + 1, row = 4 * c;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * H * row) return;
    int e = (int)(t % row);
    int64_t r = t / row;
    int h = (int)(r % H);
    int64_t i = r / H;
    const float *b = pre + (size_t)i * H * row + e;
    float me = b[(size_t)h * row];
    // the pooling compares ACTIVATIONS (v3.py:59-67: selu, then max_pooling2d): two pre-activations an ulp apart can share
    // one activation (selu' < 1 below -0.56), and the window's gradient then goes to the first of them
    const float me_act = cvm::selu(me);
    float acc = 0.0f;
    for (int ho = h - p + 1; ho <= h; ho++) {
        if (ho < 0 || ho >= Ho) continue;
        bool win = true;
        for (int d = 0; d < p; d++) {
            float v = cvm::selu(b[(size_t)(ho + d) * row]);
            int hh = ho + d;
            if (hh < h ? v >= me_act : v > me_act) { win = false; break; }   // first maximum wins
        }
        if (win) acc += gpool[((size_t)i * Ho + ho) * row + e];
    }
    gpre[t] = acc * cvm::selu_grad(me);
}

// conv weight / bias gradient: one thread per (kh,kw,ci,co) [+ cout bias threads],
// blockIdx.y slices the batch.
__global__ void b_conv_wgrad(const float *__restrict__ in, const float *__restrict__ gpre, int64_t n, int H,
                             int cin, int kh, int cout, float *__restrict__ dw, float *__restrict__ db)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t nw = (int64_t)kh * 4 * cin * cout;
    if (t >= nw + cout) return;
    int64_t per = (n + gridDim.y - 1) / gridDim.y;
    int64_t n0 = per * blockIdx.y, n1 = n0 + per < n ? n0 + per : n;
    const int padt = (kh - 1) / 2;
    float acc = 0.0f;
    if (t < nw) {
        int co = (int)(t % cout);
        int64_t r = t / cout;
        int ci = (int)(r % cin); r /= cin;
        int b = (int)(r % 4);
        int a = (int)(r / 4);
        for (int64_t i = n0; i < n1; i++) {
            const float *xi = in + (size_t)i * H * 4 * cin;
            const float *gi = gpre + (size_t)i * H * 4 * cout;
            for (int h = 0; h < H; h++) {
                int hi = h + a - padt;
                if (hi < 0 || hi >= H) continue;
                for (int wo = 0; wo < 4; wo++) {
                    int wi = wo + b - 1;
                    if (wi < 0 || wi >= 4) continue;
                    acc = __builtin_fmaf(xi[((size_t)hi * 4 + wi) * cin + ci], gi[((size_t)h * 4 + wo) * cout + co], acc);
                }
            }
        }
        atomicAdd(&dw[t], acc);
    } else {
        int co = (int)(t - nw);
        for (int64_t i = n0; i < n1; i++) {
            const float *gi = gpre + (size_t)i * H * 4 * cout;
            for (int e = 0; e < H * 4; e++) acc += gi[(size_t)e * cout + co];
        }
        atomicAdd(&db[co], acc);
    }
}

// conv input gradient: gin[n][hi][wi][ci] = sum_{kh,kw,co} gpre[n][hi-kh+pad][wi-kw+1][co] * w[kh][kw][ci][co]
__global__ void b_conv_dgrad(const float *__restrict__ gpre, const float *__restrict__ w, int64_t n, int H,
                             int cin, int kh, int cout, float *__restrict__ gin)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * H * 4 * cin) return;
    int ci = (int)(t % cin);
    int64_t r = t / cin;
    int wi = (int)(r % 4); r /= 4;
    int hi = (int)(r % H);
    int64_t i = r / H;
    const int padt = (kh - 1) / 2;
    const float *gi = gpre + (size_t)i * H * 4 * cout;
    float acc = 0.0f;
    for (int a = 0; a < kh; a++) {
        int h = hi - a + padt;
        if (h < 0 || h >= H) continue;
        for (int b = 0; b < 4; b++) {
            int wo = wi - b + 1;
            if (wo < 0 || wo >= 4) continue;
            const float *gr = gi + ((size_t)h * 4 + wo) * cout;
            const float *wr = w + (((size_t)a * 4 + b) * cin + ci) * cout;
            for (int co = 0; co < cout; co++) acc = __builtin_fmaf(gr[co], wr[co], acc);
        }
    }
    gin[t] = acc;
}

// ---- element-wise backward steps directly on tile-major buffers ----------------------------
typedef float tf4 __attribute__((ext_vector_type(4)));

// max-pool backward + SELU' on TM maps from the forward pass's window-offset codes (cv_unpool.hpp): one thread per
// (group, base, tile, lane) streams over the pooled rows -- per row one gradient fragment, one pooled-output fragment
// and 8 code bytes are read, one finished row is written; the pre-pool activations are never needed.
// gpool / pooled: HO rows of 4*NT fragments; codes: [g][HO][NT][64] 64-bit words; gpre: HO + P - 1 rows.
template <int P>
__global__ void b_unpool_tm(const tf4 *__restrict__ gpool, const tf4 *__restrict__ pooled, const uint2 *__restrict__ codes,
                            tf4 *__restrict__ gpre, int64_t G, int HO, int NT)
{
    const int cols = 4 * NT * 64;                     // f4 columns per group row
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * cols) return;
    const int col = (int)(t % cols);
    const int64_t g = t / cols;
    const int lane = col & 63, nt = (col >> 6) % NT, w = (col >> 6) / NT;
    const int H = HO + P - 1;
    const tf4 *gp = gpool + (size_t)g * HO * cols + col;
    const tf4 *pp = pooled + (size_t)g * HO * cols + col;
    const uint2 *cp = codes + ((size_t)g * HO * NT + nt) * 64 + lane;
    tf4 *o = gpre + (size_t)g * H * cols + col;
    unpool_col<P> U;
    U.init();
    for (int ho = 0; ho < HO; ho++) {
        const uint2 c = cp[(size_t)ho * NT * 64];
        U.push(gp[(size_t)ho * cols], pp[(size_t)ho * cols], cv_code16(c.x, c.y, w));
        o[(size_t)ho * cols] = U.emit();
    }
#pragma unroll
    for (int d = 0; d + 1 < P; d++) {                 // the last P-1 rows
        U.push_none();
        o[(size_t)(HO + d) * cols] = U.emit();
    }
}

// The same with one thread per (group, column, ROW): row h collects the <= P windows that hold it -- P small reads per
// thread instead of a serial walk over the rows, for batches of few groups.  Same terms in the same order: same bits.
template <int P>
__global__ void b_unpool_rows(const tf4 *__restrict__ gpool, const tf4 *__restrict__ pooled, const uint2 *__restrict__ codes,
                              tf4 *__restrict__ gpre, int64_t G, int HO, int NT)
{
    const int cols = 4 * NT * 64;
    const int H = HO + P - 1;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * H * cols) return;
    const int col = (int)(t % cols);
    const int h = (int)((t / cols) % H);
    const int64_t g = t / ((int64_t)cols * H);
    const int lane = col & 63, nt = (col >> 6) % NT, w = (col >> 6) / NT;
    const tf4 *gp = gpool + (size_t)g * HO * cols + col;
    const tf4 *pp = pooled + (size_t)g * HO * cols + col;
    const uint2 *cp = codes + ((size_t)g * HO * NT + nt) * 64 + lane;
    tf4 acc = (tf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int d = P - 1; d >= 0; d--) {                // windows ho = h - d in ascending order
        const int ho = h - d;
        if (ho < 0 || ho >= HO) continue;
        const uint2 c = cp[(size_t)ho * NT * 64];
        const unsigned c16 = cv_code16(c.x, c.y, w);
        const tf4 gv = gp[(size_t)ho * cols], y = pp[(size_t)ho * cols];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float gs = gv[k] * cv_selu_grad_from_out(y[k]);
            acc[k] += ((c16 >> (4 * k)) & 15u) == (unsigned)d ? gs : 0.0f;
        }
    }
    gpre[(size_t)(g * H + h) * cols + col] = acc;
}

// Between the two for batches of few groups: one thread per (group, column, SEGMENT of rows).  A segment walks its rows
// with b_unpool_tm's register window after P - 1 warm-up windows that are pushed but emit nothing -- every value is read
// (1 + (P - 1) NSEG / HO) times instead of P times and a thread has H / NSEG dependent steps instead of H.  Row for row
// the terms and their order are b_unpool_tm's: same bits.  (79 groups, conv2's map: 61 us thread-per-row -> see
// profiles/r05/train_1250_timeline_*.txt)
template <int P, int NSEG>
__global__ void b_unpool_seg(const tf4 *__restrict__ gpool, const tf4 *__restrict__ pooled, const uint2 *__restrict__ codes,
                             tf4 *__restrict__ gpre, int64_t G, int HO, int NT)
{
    const int cols = 4 * NT * 64;
    const int H = HO + P - 1;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * NSEG * cols) return;
    const int col = (int)(t % cols);
    const int seg = (int)((t / cols) % NSEG);
    const int64_t g = t / ((int64_t)cols * NSEG);
    const int lane = col & 63, nt = (col >> 6) % NT, w = (col >> 6) / NT;
    const int h0 = H * seg / NSEG, h1 = H * (seg + 1) / NSEG;
    const tf4 *gp = gpool + (size_t)g * HO * cols + col;
    const tf4 *pp = pooled + (size_t)g * HO * cols + col;
    const uint2 *cp = codes + ((size_t)g * HO * NT + nt) * 64 + lane;
    tf4 *o = gpre + (size_t)g * H * cols + col;
    unpool_col<P> U;
    U.init();
    for (int ho = h0 - (P - 1); ho < h1; ho++) {
        if (ho >= 0 && ho < HO) {
            const uint2 c = cp[(size_t)ho * NT * 64];
            U.push(gp[(size_t)ho * cols], pp[(size_t)ho * cols], cv_code16(c.x, c.y, w));
        } else {
            U.push_none();
        }
        if (ho >= h0) o[(size_t)ho * cols] = U.emit();
    }
}

// gpre = gact * selu'(act) for a layer without pooling (slim): act is the layer output
__global__ void b_selu_out_tm(const tf4 *__restrict__ gact, const tf4 *__restrict__ act, tf4 *__restrict__ gpre, int64_t nf4)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nf4) return;
    const tf4 g = gact[t], a = act[t];
    tf4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = g[k] * cv_selu_grad_from_out(a[k]);
    gpre[t] = r;
}

// gpool / pooled: HO = H - p + 1 rows; gpre: H rows
static int launch_unpool(const float *gpool, const float *pooled, const float *codes, float *gpre, int64_t G, int H, int NT,
                         int p, hipStream_t st, int tiny_g, bool rows_form)
{
    const tf4 *gi = (const tf4 *)gpool, *pi = (const tf4 *)pooled;
    const uint2 *ci = (const uint2 *)codes;
    tf4 *go = (tf4 *)gpre;
    if (p == 1) {
        const int64_t nf4 = G * H * 4 * NT * 64;
        b_selu_out_tm<<<nblk(nf4, 256), 256, 0, st>>>(gi, pi, go, nf4);
        return 0;
    }
    const int HO = H - p + 1;
    if (G <= tiny_g && !rows_form) {       // tiny batches (cv_model::tiny_g): four row segments per column
        const unsigned grid = (unsigned)((G * 4 * 4 * NT * 64 + 255) / 256);
        switch (p) {
        case 2: b_unpool_seg<2, 4><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        case 3: b_unpool_seg<3, 4><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        case 4: b_unpool_seg<4, 4><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        case 5: b_unpool_seg<5, 4><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        default: break;
        }
    }
    if (G <= tiny_g) {       // (dbg2 = 1 / 4: one thread per row)
        const unsigned grid = (unsigned)((G * H * 4 * NT * 64 + 255) / 256);
        switch (p) {
        case 2: b_unpool_rows<2><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        case 3: b_unpool_rows<3><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        case 4: b_unpool_rows<4><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        case 5: b_unpool_rows<5><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); return 0;
        default: break;
        }
    }
    const unsigned grid = (unsigned)((G * 4 * NT * 64 + 255) / 256);
    switch (p) {
    case 2: b_unpool_tm<2><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); break;
    case 3: b_unpool_tm<3><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); break;
    case 4: b_unpool_tm<4><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); break;
    case 5: b_unpool_tm<5><<<grid, 256, 0, st>>>(gi, pi, ci, go, G, HO, NT); break;
    default: cv_set_error("pooling window %d is not supported by the tile backward pass", p); return 1;
    }
    return 0;
}

// heads: data gradients in TM layout, fused with the SELU' (and dropout) factor of the layer they flow into.
//   mode 0: g5pre = (sum over the three fc5-side heads) * selu'(h5)                      -- all entries written, padding = 0
//   mode 1: g4pre = (gd4 + base-head contribution) * amask * selu'(h4)                   -- gd4 = fc5's data gradient
// (one pass instead of a head-gradient pass plus an element-wise pass per layer)
__global__ void b_head_dgrad_tm(const float *__restrict__ ghpre, const float *__restrict__ wb,
                                const float *__restrict__ wz, const float *__restrict__ wt,
                                const float *__restrict__ wl, int K, int KB, int64_t n, int64_t G, int mode,
                                const float *__restrict__ gin_tm, const float *__restrict__ act_tm,
                                const float *__restrict__ mask_tm, float *__restrict__ out_tm)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * KB * 256) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int kb = (int)(frag % KB);
    int64_t g = frag / KB;
    int c = lane & 15, kq = lane >> 4;
    int k = 16 * kb + 4 * s + kq;
    int64_t cand = g * 16 + c;
    float acc = 0.0f;
    if (cand < n && k < K) {
        const float *gi = ghpre + (size_t)cand * 16;
        if (mode == 0) {
            for (int j = 0; j < 2; j++) acc = __builtin_fmaf(gi[4 + j], wz[(size_t)k * 2 + j], acc);
            for (int j = 0; j < 4; j++) acc = __builtin_fmaf(gi[6 + j], wt[(size_t)k * 4 + j], acc);
            for (int j = 0; j < 6; j++) acc = __builtin_fmaf(gi[10 + j], wl[(size_t)k * 6 + j], acc);
        } else {
            for (int j = 0; j < 4; j++) acc = __builtin_fmaf(gi[j], wb[(size_t)k * 4 + j], acc);
        }
    }
    float gact = mode == 0 ? acc : gin_tm[t] + acc;
    if (mask_tm) gact *= mask_tm[t];
    out_tm[t] = gact * selu_grad_from_out(act_tm[t]);
}

struct slab {
    float *base; size_t used, cap;
    float *take(size_t nfloat) { float *p = base + used; used += (nfloat + 63) / 64 * 64; return used <= cap ? p : nullptr; }
};

}  // namespace

static size_t train_floats_per_cand(const cv_model *m)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    size_t f = 0;
    for (int l = 0; l < 3; l++) {
        f += 2 * (size_t)s.hc[l] * 4 * a.cout[l];   // pre, gpre
        f += 2 * (size_t)s.hp[l] * 4 * a.cout[l];   // pooled, gpooled
    }
    f += 5 * (size_t)a.fc4 + 4 * (size_t)a.fc5 + 32;
    // tile path: TM copies of activations, pre-pool activations and three gradient maps,
    // padded channel counts, plus the dense TM buffers
    for (int l = 0; l < 3; l++) f += (size_t)(2 * s.hc[l] + 2 * s.hp[l]) * 4 * s.ntile[l] * 16;
    f += (6 + CV_DENSE_KSPLIT) * (size_t)s.nb4 * 16 + 2 * (size_t)s.nb5 * 16;
    return f + 64 * 80;
}

// forward (+ optional backward) of one slice of the batch: all-plain path
static int train_slice_plain(cv_model *m, const float *x, const float *y, int64_t n, int64_t cand0, bool backward,
                       float drop4, uint64_t seed, uint64_t step, hipStream_t st)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const float *P = m->params; float *G = m->grads; const int64_t *o = m->poff;
    slab sb{m->t_buf, 0, m->t_bytes / sizeof(float)};
    float *pre[3], *pool[3], *gpre[3], *gpool[3];
    for (int l = 0; l < 3; l++) {
        pre[l] = sb.take((size_t)n * s.hc[l] * 4 * a.cout[l]);
        pool[l] = sb.take((size_t)n * s.hp[l] * 4 * a.cout[l]);
        gpre[l] = sb.take((size_t)n * s.hc[l] * 4 * a.cout[l]);
        gpool[l] = sb.take((size_t)n * s.hp[l] * 4 * a.cout[l]);
    }
    float *fc4pre = sb.take((size_t)n * a.fc4), *d4 = sb.take((size_t)n * a.fc4), *amask = sb.take((size_t)n * a.fc4);
    float *gd4 = sb.take((size_t)n * a.fc4), *gfc4pre = sb.take((size_t)n * a.fc4);
    float *fc5pre = sb.take((size_t)n * a.fc5), *h5 = sb.take((size_t)n * a.fc5);
    float *gh5 = sb.take((size_t)n * a.fc5), *gfc5pre = sb.take((size_t)n * a.fc5);
    float *ghpre = sb.take((size_t)n * 16);
    if (!ghpre) { cv_set_error("training workspace too small"); return 1; }
    // ---- forward
    const float *in = x;
    for (int l = 0; l < 3; l++) {
        int H = s.hc[l], C = a.cout[l];
        t_conv_pre<<<nblk(n * H * 4 * C, 256), 256, 0, st>>>(in, P + o[2 * l], P + o[2 * l + 1], pre[l], n, H,
                                                            s.cin[l], a.kh[l], C);
        t_pool_selu<<<nblk(n * s.hp[l] * 4 * C, 256), 256, 0, st>>>(pre[l], pool[l], n, H, C, a.pool[l]);
        in = pool[l];
    }
    t_dense_pre<<<nblk(n * a.fc4, 256), 256, 0, st>>>(pool[2], P + o[6], P + o[7], fc4pre, n, s.flat, a.fc4);
    t_fc4_act<<<nblk(n * a.fc4, 256), 256, 0, st>>>(fc4pre, d4, amask, n, a.fc4, backward ? drop4 : 0.0f, seed,
                                                    step, cand0);
    m->last_tr_d4 = d4; m->last_tr_mask = amask; m->last_tr_n = n; m->last_tr_tile = 0;
    for (int l = 0; l < 3; l++) { m->last_tr_pool[l] = pool[l]; m->last_tr_gpre[l] = backward ? gpre[l] : nullptr; }
    t_dense_pre<<<nblk(n * a.fc5, 256), 256, 0, st>>>(d4, P + o[8], P + o[9], fc5pre, n, a.fc4, a.fc5);
    t_selu_act<<<nblk(n * a.fc5, 256), 256, 0, st>>>(fc5pre, h5, n * a.fc5);
    t_heads<<<nblk(n, 16), 256, 0, st>>>(d4, h5, a.fc4, a.fc5, P + o[10], P + o[11], P + o[12], P + o[13],
                                         P + o[14], P + o[15], P + o[16], P + o[17], y, n,
                                         backward ? ghpre : nullptr, m->loss_dev);
    CV_HIP(hipGetLastError());
    if (!backward) return 0;
    // ---- backward
    const int NS = 128;     // candidate-range slices of the weight-gradient reductions (short serial loops, few atomics)
    // heads: columns of ghpre: 0..3 base (input d4), 4..5 / 6..9 / 10..15 (input h5)
    b_dense_wgrad<<<dim3(nblk((a.fc4 + 1) * 4, 256), NS), 256, 0, st>>>(d4, a.fc4, ghpre + 0, 16, n, a.fc4, 4, G + o[10], G + o[11]);
    b_dense_wgrad<<<dim3(nblk((a.fc5 + 1) * 2, 256), NS), 256, 0, st>>>(h5, a.fc5, ghpre + 4, 16, n, a.fc5, 2, G + o[12], G + o[13]);
    b_dense_wgrad<<<dim3(nblk((a.fc5 + 1) * 4, 256), NS), 256, 0, st>>>(h5, a.fc5, ghpre + 6, 16, n, a.fc5, 4, G + o[14], G + o[15]);
    b_dense_wgrad<<<dim3(nblk((a.fc5 + 1) * 6, 256), NS), 256, 0, st>>>(h5, a.fc5, ghpre + 10, 16, n, a.fc5, 6, G + o[16], G + o[17]);
    b_dense_dgrad<<<nblk(n * a.fc4, 256), 256, 0, st>>>(ghpre + 0, 16, P + o[10], n, a.fc4, 4, gd4, 0);
    b_dense_dgrad<<<nblk(n * a.fc5, 256), 256, 0, st>>>(ghpre + 4, 16, P + o[12], n, a.fc5, 2, gh5, 0);
    b_dense_dgrad<<<nblk(n * a.fc5, 256), 256, 0, st>>>(ghpre + 6, 16, P + o[14], n, a.fc5, 4, gh5, 1);
    b_dense_dgrad<<<nblk(n * a.fc5, 256), 256, 0, st>>>(ghpre + 10, 16, P + o[16], n, a.fc5, 6, gh5, 1);
    // fc5
    b_selu<<<nblk(n * a.fc5, 256), 256, 0, st>>>(gh5, fc5pre, nullptr, gfc5pre, n * a.fc5);
    b_dense_wgrad<<<dim3(nblk((int64_t)(a.fc4 + 1) * a.fc5, 256), NS), 256, 0, st>>>(d4, a.fc4, gfc5pre, a.fc5, n, a.fc4, a.fc5, G + o[8], G + o[9]);
    b_dense_dgrad<<<nblk(n * a.fc4, 256), 256, 0, st>>>(gfc5pre, a.fc5, P + o[8], n, a.fc4, a.fc5, gd4, 1);
    // dropout4 + selu'
    b_selu<<<nblk(n * a.fc4, 256), 256, 0, st>>>(gd4, fc4pre, amask, gfc4pre, n * a.fc4);
    // fc4
    b_dense_wgrad<<<dim3(nblk((int64_t)(s.flat + 1) * a.fc4, 256), 8), 256, 0, st>>>(pool[2], s.flat, gfc4pre, a.fc4, n, s.flat, a.fc4, G + o[6], G + o[7]);
    b_dense_dgrad<<<nblk(n * s.flat, 256), 256, 0, st>>>(gfc4pre, a.fc4, P + o[6], n, s.flat, a.fc4, gpool[2], 0);
    // conv stack
    for (int l = 2; l >= 0; l--) {
        int H = s.hc[l], C = a.cout[l];
        b_pool_selu<<<nblk(n * H * 4 * C, 256), 256, 0, st>>>(gpool[l], pre[l], gpre[l], n, H, C, a.pool[l]);
        const float *lin = l == 0 ? x : pool[l - 1];
        int64_t nw = (int64_t)a.kh[l] * 4 * s.cin[l] * C + C;
        b_conv_wgrad<<<dim3(nblk(nw, 64), 64), 64, 0, st>>>(lin, gpre[l], n, H, s.cin[l], a.kh[l], C, G + o[2 * l], G + o[2 * l + 1]);
        if (l > 0)
            b_conv_dgrad<<<nblk(n * H * 4 * s.cin[l], 256), 256, 0, st>>>(gpre[l], P + o[2 * l], n, H, s.cin[l], a.kh[l], C, gpool[l - 1]);
    }
    CV_HIP(hipGetLastError());
    return 0;
}


// forward (+ optional backward) of one slice of the batch on the tile kernels; every
// intermediate stays tile-major, the only natural-layout tensors are X, Y and the 16 head
// gradients per candidate
// Weight gradients leave the critical path: the data-gradient chain (heads -> fc5 -> fc4 -> conv3 -> conv2 -> conv1)
// runs on `st`, every weight-gradient kernel pair on the model's side stream `sw` behind an event recorded after
// the kernel that produces its gradient operand.  All weight gradients share one scratch buffer and one stream, so
// they stay in a fixed order (bit-reproducible); at train.py's batch and below the grids do not fill the chip and
// the two chains overlap.  sw == st: everything in stream order (option "train_overlap" = 0).
// Up to CV_TR_SIDES side streams: the weight gradients of different layers touch different gradient tensors and
// different scratch regions, so they are independent of each other as well; at small batches (a kernel covers a
// fraction of the chip) one side stream made them a second critical path as long as the data-gradient chain.
struct tr_fork {
    cv_model *m; hipStream_t st; hipStream_t side[CV_TR_SIDES]; int nside; int k; bool used[CV_TR_SIDES];
    bool tail_only;        // large batches: the second side stream takes the launch sites from tail_first on (see train_slice_tile)
    int tail_first;
    hipEvent_t mark;       // the newest marker recorded on st; mark_fresh: nothing was enqueued on st since
    bool mark_fresh;
    bool share;            // launch sites at the same point of st share a marker (train_sched bit 3; see cv_internal.hpp)
    hipEvent_t next_event() { return m->tr_ev[k++ % (CV_TR_EVENTS - 1)]; }
    int side_of(int site) const { return tail_only ? (site >= tail_first && nside > 1 ? 1 : 0) : site % nside; }
    // side stream of launch site `site` (0 heads, 1 fc5, 2 fc4, 3 conv3, 4 conv2, 5 conv1), made to wait for
    // everything enqueued on st so far; st itself when the step runs in stream order.  A marker costs the main stream
    // ~6 us (a barrier packet between two kernels: profiles/r05/train_1250_timeline_before.txt), so two launch sites with
    // no kernel of st between them share one (`same_point`).
    int to_side(int site, hipStream_t *out, bool same_point = false)
    {
        if (nside == 0) { *out = st; return 0; }
        const int i = side_of(site);
        if (!(same_point && mark_fresh && share)) {
            mark = next_event();
            CV_HIP(hipEventRecord(mark, st));
            mark_fresh = true;
        }
        CV_HIP(hipStreamWaitEvent(side[i], mark, 0));
        used[i] = true;
        *out = side[i];
        return 0;
    }
    void st_moved() { mark_fresh = false; }      // the caller enqueued something on st
    // stream `to` continues behind everything enqueued so far on the side streams other than `to`
    int gather(hipStream_t to)
    {
        for (int i = 0; i < nside; i++) {
            if (!used[i] || side[i] == to) continue;
            hipEvent_t e = next_event();
            CV_HIP(hipEventRecord(e, side[i]));
            CV_HIP(hipStreamWaitEvent(to, e, 0));
        }
        return 0;
    }
    // st continues behind all side streams: one wait per side stream ON st.  (Round 5 measured the alternative -- the side
    // streams chained among themselves, one wait on st: when the last side stream finishes just ahead of st, as at
    // train.py's batch, the chain puts two dependent hand-overs in series, 19 us against 11 at the tail of the step;
    // a wait for an event that has already fired costs st next to nothing.)
    int join() { return nside == 0 ? 0 : gather(st); }
    // Tiny batches, where the last kernels of the backward pass run on st itself: the side streams finished long before st
    // gets here, so they are chained among themselves (off the critical path) and st takes ONE wait, for an event that has
    // fired by then
    int join_chained()
    {
        if (nside == 0) return 0;
        int last = -1;
        for (int i = 0; i < nside; i++) if (used[i]) last = i;
        if (last < 0) return 0;
        if (gather(side[last])) return 1;
        hipEvent_t e = next_event();
        CV_HIP(hipEventRecord(e, side[last]));
        CV_HIP(hipStreamWaitEvent(st, e, 0));
        return 0;
    }
};

// the fixed-order loss sums + the bucket's loss header (t_loss_header, below) launched from inside a slice
struct tr_header { double lambda; bool l2; };
// train_sched bit 11: the side stream's work ahead of the backward pass (L2 term, the dense / data-gradient weight packing)
// is forked behind conv1's forward kernel (first slice of a step) instead of at the head of the step
struct tr_defer { bool on; float lambda; bool tile_path; };
static int launch_l2(cv_model *m, hipStream_t sw, bool tile_path);

// forward (+ optional backward) of one slice of the batch on the tile kernels; every
// intermediate stays tile-major, the only natural-layout tensors are X, Y and the 16 head
// gradients per candidate.  dense_ready (last slice of a backward pass): recorded once the
// gradients of fc4, fc5 and the heads -- the contiguous tail of the flat gradient buffer, 95 %
// of its bytes -- are final, so that the caller's exchange of that part runs under the conv
// backward pass.
__global__ __launch_bounds__(256) void t_loss_header(double *__restrict__ loss, double lambda, float *__restrict__ hdr,
                                                     const double *__restrict__ rows, int64_t nrows,
                                                     const double *__restrict__ l2_rows, int l2_kernels, int overwrite);

// hdr_now (single-slice step with side streams): the loss header is launched on the side stream of launch site 0 right
// behind the heads kernel -- it needs the heads' block sums and the L2 sums (that stream carries t_l2), nothing of the
// backward pass -- instead of at the tail of the step on st (8 us + a launch off the critical path).
static int train_slice_tile(cv_model *m, const float *x, const float *y, int64_t n, int64_t cand0, bool backward,
                            float drop4, uint64_t seed, uint64_t step, hipStream_t st, hipStream_t sw,
                            hipEvent_t dense_ready, bool sw_ordered, const tr_header *hdr_now, const tr_defer *defer,
                            bool *hdr_launched)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const float *P = m->params; const int64_t *o = m->poff;
    const int64_t np = (n + 15) / 16 * 16;             // TM buffers hold whole groups
    const int64_t Gn = np / 16;
    slab sb{m->t_buf, 0, m->t_bytes / sizeof(float)};
    size_t fp[3], fa[3];                               // floats per candidate: pooled / pre-pool maps
    for (int l = 0; l < 3; l++) { fp[l] = (size_t)s.hp[l] * 4 * s.ntile[l] * 16; fa[l] = (size_t)s.hc[l] * 4 * s.ntile[l] * 16; }
    const size_t f4u = (size_t)s.nb4 * 16, f5u = (size_t)s.nb5 * 16;
    float *tp[3], *ta[3];          // pooled maps; window-offset codes of the pooled values (pooled layers only)
    for (int l = 0; l < 3; l++) {
        tp[l] = sb.take(np * fp[l]);
        ta[l] = a.pool[l] > 1 ? sb.take((size_t)Gn * s.hp[l] * s.ntile[l] * 128) : nullptr;
    }
    float *th4 = sb.take(np * f4u), *td4 = sb.take(np * f4u), *tmask = sb.take(np * f4u), *th5 = sb.take(np * f5u);
    float *ghpre = sb.take((size_t)n * 16);
    float *kpart = sb.take((size_t)CV_DENSE_KSPLIT * np * f4u);      // partial sums of the k-split fc4 forward
    if (!ghpre || !kpart) { cv_set_error("training workspace too small"); return 1; }
    // ---- forward
    bool pack_wait = false;       // dbg5 = 1: all packing in one launch on st, as before
    if (defer && defer->on && sw != st && m->dbg[5] != 1) {
        // convolution fragments now; the marker, the L2 term and the rest of the packing behind conv1's kernel
        if (cv_pack_for_training(m, st, backward, (int)Gn, sw, m->tr_pack_fork, m->tr_pack_done, &pack_wait, false, 1)) return 1;
        const std::function<int()> hook = [&]() -> int {
            CV_HIP(hipEventRecord(m->tr_ev[CV_TR_EVENTS - 1], st));
            CV_HIP(hipStreamWaitEvent(sw, m->tr_ev[CV_TR_EVENTS - 1], 0));
            if (defer->lambda != 0.0f && launch_l2(m, sw, defer->tile_path)) return 1;
            return cv_pack_for_training(m, st, backward, (int)Gn, sw, m->tr_pack_fork, m->tr_pack_done, &pack_wait, true, 2);
        };
        if (cv_tile_train_convs(m, x, n, tp[0], ta[0], tp[1], ta[1], tp[2], ta[2], st, &hook)) return 1;
    } else {
        if (cv_pack_for_training(m, st, backward, (int)Gn, m->dbg[5] == 1 ? st : sw, m->tr_pack_fork, m->tr_pack_done, &pack_wait, sw_ordered)) return 1;
        if (cv_tile_train_convs(m, x, n, tp[0], ta[0], tp[1], ta[1], tp[2], ta[2], st)) return 1;
    }
    if (pack_wait) CV_HIP(hipStreamWaitEvent(st, m->tr_pack_done, 0));
    const cv_train_dropout drop{td4, tmask, backward ? drop4 : 0.0f, seed, step, cand0};
    float *tg5pre = backward ? sb.take(np * f5u) : nullptr;
    if (backward && !tg5pre) { cv_set_error("training workspace too small"); return 1; }
    // tiny batches (full topology, k-split fc4): everything behind fc4's k ranges -- their sum, dropout, fc5, the heads,
    // losses, head gradients, fc5-side data gradient -- is one kernel
    bool tail_done = false;
    if (cv_tile_train_tail(m, tp[2], th4, th5, y, n, backward ? 1 : 0, ghpre, tg5pre, m->train_ksplit ? kpart : nullptr, &drop, st, &tail_done)) return 1;
    if (!tail_done) {
        bool drop_done = false;
        if (cv_tile_dense_fwd(m, 4, tp[2], th4, n, st, m->train_ksplit ? kpart : nullptr, &drop, &drop_done)) return 1;
        if (!drop_done && cv_dropout_tm(m, th4, td4, tmask, n, drop.rate, seed, step, cand0, st)) return 1;
        bool fused_fc5 = false;
        if (Gn > m->tiny_g && cv_tile_train_fc5_heads(m, td4, th5, y, n, backward ? 1 : 0, ghpre, tg5pre, st, &fused_fc5)) return 1;
        if (!fused_fc5) {
            if (cv_tile_dense_fwd(m, 5, td4, th5, n, st)) return 1;
            // heads: products, losses, head gradients and the fc5-side data gradient (times selu'(h5)) in one launch
            if (cv_tile_heads_train(m, td4, th5, y, n, backward ? 1 : 0, ghpre, tg5pre, st)) return 1;
        }
    }
    m->last_tr_d4 = td4; m->last_tr_mask = tmask; m->last_tr_n = n; m->last_tr_tile = 1;
    for (int l = 0; l < 3; l++) { m->last_tr_pool[l] = tp[l]; m->last_tr_gpre[l] = nullptr; }
    CV_HIP(hipGetLastError());
    if (!backward) return 0;
    // ---- backward buffers (TM gradients; the weight-gradient kernels transpose their operands on the way in)
    float *tgd4 = sb.take(np * f4u), *tg4pre = sb.take(np * f4u);
    float *tgpre[3], *tgin[3];
    for (int l = 0; l < 3; l++) { tgpre[l] = sb.take(np * fa[l]); tgin[l] = sb.take(np * fp[l]); }
    if (!tgin[2]) { cv_set_error("training workspace too small"); return 1; }
    for (int l = 0; l < 3; l++) m->last_tr_gpre[l] = tgpre[l];
    tr_fork f;
    f.m = m; f.st = st; f.k = 0; f.nside = 0; f.tail_only = Gn > m->tiny_g; f.tail_first = m->wpr_fc4 ? 4 : 3;
    f.mark = nullptr; f.mark_fresh = false;
    // (measured, one box, alternating: at 625 groups of the full topology a shared marker lets the main stream run 6 us
    // ahead and the step comes out 39 us LONGER -- the weight gradients then meet the fc4 / conv3 data gradients on the
    // CUs at another moment; 2 500 and 5 000 candidates and the slim topology gain with it)
    f.share = (m->sched & 8) && (Gn <= 512 || !m->wpr_fc4);
    for (int i = 0; i < CV_TR_SIDES; i++) { f.side[i] = nullptr; f.used[i] = false; }
    if (sw != st) {
        f.side[f.nside++] = sw;
        // Three side streams, launch sites round robin, only for tiny batches: at train.py's 625 groups every kernel fills
        // the chip and three concurrent weight-gradient kernels just take CUs from the data-gradient chain (2.25 -> 2.34 ms,
        // profiles/r03).  A large batch gets ONE more stream for the last layers: behind conv3's weight gradient on the
        // single side stream conv2 and conv1 started 280 us after their inputs were ready and ran on past the end of the
        // data-gradient chain (the last 100 us of the step had two small kernels on the chip).  Full topology: conv2 and
        // conv1 (with conv3 as well 2.13 -> 2.17 ms: its kernel fills the chip next to fc4's); slim: conv3, conv2, conv1
        // (1.233 / 1.176 / 1.167 ms with one stream / two layers / three layers on the second; profiles/r03).
        const int want = Gn <= m->tiny_g ? m->train_sides : (m->train_sides >= 2 ? 2 : 1);
        for (int i = 0; i < 2 && f.nside < want; i++) f.side[f.nside++] = m->tr_side_more[i];
        f.used[0] = true;                    // sw already carries the L2 term / the weight packing of this step
    }
    hipStream_t sx = st;
    // heads: weight gradients on the matrix cores (inputs tile-major, the 16 gradients as they lie), data
    // gradients written to TM, times selu'(h5)
    if (f.to_side(0, &sx)) return 1;
    if (hdr_now && f.nside > 0) {
        // tiny batches: on sw, behind the heads' marker (sw carries the L2 kernel of this step: stream order covers it).
        // Larger batches: sw's weight-gradient chain is as long as the main one (a header at its head cost the step 33 us),
        // but the SECOND side stream idles until conv2's weight gradient: the header goes there, behind the same marker
        // and behind the event recorded after the L2 kernel
        hipStream_t hs = nullptr;
        if (Gn <= m->tiny_g && sx == sw) hs = sx;
        else if (Gn > m->tiny_g && f.nside > 1 && f.mark) {
            hs = f.side[1];
            CV_HIP(hipStreamWaitEvent(hs, f.mark, 0));
            CV_HIP(hipStreamWaitEvent(hs, m->tr_l2_done, 0));
            f.used[1] = true;
        }
        if (hs) {
            t_loss_header<<<1, 256, 0, hs>>>(m->loss_dev, hdr_now->lambda, m->grads - CV_GRAD_HEADER, m->loss_rows, m->loss_rows_used,
                                             hdr_now->l2 ? m->l2_rows : nullptr, CV_NUM_PARAMS / 2, 1);
            *hdr_launched = true;
        }
    }
    if (cv_tile_heads_wgrad(m, td4, th5, ghpre, n, sx)) return 1;
    // fc5 (its pre-activation gradient came out of the heads kernel: the same point of st as the heads' launch site)
    if (f.to_side(1, &sx, true)) return 1;
    if (cv_tile_dense_wgrad(m, 5, td4, tg5pre, n, sx)) return 1;
    f.st_moved();
    // fc5's data gradient + the base head's contribution, then dropout4 + selu' (h4 is the SELU output before dropout): on
    // the kernel's store, or (train_sched bit 5 off) as an element-wise pass behind it -- the same operations per value
    if (m->sched & 32) {
        if (cv_tile_fc5_dgrad(m, tg5pre, tg4pre, n, st, ghpre, tmask, th4)) return 1;
    } else {
        if (cv_tile_fc5_dgrad(m, tg5pre, tgd4, n, st)) return 1;
        b_head_dgrad_tm<<<nblk(Gn * s.nb4 * 256, 256), 256, 0, st>>>(ghpre, P + o[10], P + o[12], P + o[14], P + o[16], a.fc4,
                                                                   s.nb4, n, Gn, 1, tgd4, th4, tmask, tg4pre);
    }
    f.st_moved();
    // fc4's data gradient; full topology: fused with conv3's max-pool backward + SELU' (dbg3 = 1: as two kernels)
    const bool fused3 = m->wpr_fc4 != nullptr && m->dbg[3] != 1;
    // fc4's weight gradient.  (Launched one kernel later at tiny batches, at the marker of conv3's weight gradient -- a marker
    // less on the main stream -- it was 14 us slower at 79 groups and 14 us faster at 157: profiles/r05/
    // step_ab_session6_sched_bit7.txt; the schedule bit was removed in round 6.)
    auto fc4_wgrad = [&](bool same_point) -> int {
        if (f.to_side(2, &sx, same_point)) return 1;
        if (cv_tile_dense_wgrad(m, 4, tp[2], tg4pre, n, sx)) return 1;
        if (dense_ready) {                   // heads, fc5 and fc4 gradients final: behind all three launch sites
            if (f.nside > 0 && f.gather(sx)) return 1;
            CV_HIP(hipEventRecord(dense_ready, sx));
        }
        return 0;
    };
    if (fc4_wgrad(false)) return 1;
    // layers without pooling (slim): the selu' factor of the layer below rides on the data-gradient kernel's store
    // (dbg4 = 3: as a separate element-wise pass)
    const bool nopool_fused = a.pool[0] == 1 && a.pool[1] == 1 && a.pool[2] == 1 && m->dbg[4] != 3;
    f.st_moved();
    if (fused3) { if (cv_tile_fc4_dgrad_unpool(m, tg4pre, tp[2], ta[2], tgpre[2], n, st)) return 1; }
    else if (nopool_fused) { if (cv_tile_fc4_dgrad(m, tg4pre, tgpre[2], n, st, tp[2])) return 1; }
    else if (cv_tile_fc4_dgrad(m, tg4pre, tgin[2], n, st)) return 1;
    // conv stack.  (A convolution data gradient fused with the unpool below it -- the pre-activation gradient of layer l - 1
    // written directly -- was built in round 3 and lost at every batch: 385 us against 254 + 79 for conv3 at 625 groups,
    // 0.698 against 0.672 ms per step at 79; removed in round 6, profiles/HISTORY.md.)
    for (int l = 2; l >= 0; l--) {
        const int H = s.hc[l], NT = s.ntile[l];
        const bool have_gpre = (l == 2 && fused3) || nopool_fused;
        // the first layer's unpool rides inside its weight-gradient kernel (its gradient map has no other reader)
        const bool conv1_fused = l == 0 && !have_gpre && a.pool[0] == 5 && s.ntile[0] == 1 && m->dbg[4] != 4;
        if (!have_gpre && !conv1_fused && launch_unpool(tgin[l], tp[l], ta[l], tgpre[l], Gn, H, NT, a.pool[l], st, (m->dbg[2] == 1 || m->dbg[2] == 6) ? (1 << 30) : (m->dbg[2] == 2 ? 0 : m->tiny_g), m->dbg[2] == 1 || m->dbg[2] == 4)) return 1;
        f.st_moved();
        // The first layer's weight gradient is the LAST work of the backward pass: nothing of st is left to run beside it.
        // At tiny batches it stays on st (a marker, the hand-over to the side stream and the wait for it back cost ~25 us
        // of an otherwise idle chip for a 17 us kernel); at large ones the side stream keeps it off the chain's tail.
        if (l == 0 && ((Gn <= m->tiny_g && (m->sched & 2)) || (m->sched & 512))) sx = st;
        else {
            if (f.to_side(5 - l, &sx)) return 1;
        }
        if (l == 0) {        // first layer: X viewed as [33][16] fragments, read in place
            bool done1 = false;
            if (conv1_fused && cv_tile_conv1_wgrad_unpool(m, x, tgin[0], tp[0], ta[0], n, sx, &done1)) return 1;
            if (!done1 && cv_tile_conv1_wgrad(m, x, tgpre[0], n, sx)) return 1;
        } else {
            if (cv_tile_conv_wgrad(m, l, tp[l - 1], tgpre[l], n, sx)) return 1;
            if (nopool_fused) { if (cv_tile_conv_dgrad(m, l, tgpre[l], tgpre[l - 1], n, st, tp[l - 1])) return 1; }
            else if (cv_tile_conv_dgrad(m, l, tgpre[l], tgin[l - 1], n, st)) return 1;
        }
    }
    if ((Gn <= m->tiny_g && Gn <= CV_TINY_PARTS_MAX_G && (m->sched & 2) && (m->sched & 128)) ? f.join_chained() : f.join()) return 1;
    CV_HIP(hipGetLastError());
    return 0;
}

static int train_slice(cv_model *m, const float *x, const float *y, int64_t n, int64_t cand0, bool backward,
                       float drop4, uint64_t seed, uint64_t step, hipStream_t st, hipStream_t sw, hipEvent_t dense_ready,
                       bool sw_ordered, const tr_header *hdr_now, const tr_defer *defer, bool *hdr_launched)
{
    if (m->impl == 1 && cv_tile_supported(m))
        return train_slice_tile(m, x, y, n, cand0, backward, drop4, seed, step, st, sw, dense_ready, sw_ordered, hdr_now, defer, hdr_launched);
    if (train_slice_plain(m, x, y, n, cand0, backward, drop4, seed, step, st)) return 1;
    if (dense_ready) CV_HIP(hipEventRecord(dense_ready, st));
    return 0;
}

// Loss header of the gradient bucket (cv_grad_bucket): 16 floats in front of the flat gradient.  Slots 2k, 2k+1 =
// loss k (k = 0..3: base, zygosity, type, length) of this pass as a (hi, lo) pair of floats whose sum is the
// double the kernels accumulated; slots 8, 9 = lambda * sum(w^2)/2 likewise; slot 10 = 1 (counts the ranks when
// the bucket is all-reduced); the rest 0.  A SUM all-reduce of the header therefore keeps ~48 bits of every loss.
//
// Before that the kernel (one block of 256 threads) finishes the loss sums of the tile path in a FIXED order, so that a
// step's losses are the same bits from run to run: thread t adds the heads kernel's block rows t, t + 256, ... (4 doubles
// each, `nrows` rows, slice after slice) and, when l2_rows is given, the L2 kernel's block sums [kernel][t] over the
// kernels; a fixed binary tree over the threads follows; the results are ADDED to loss[0..3] / loss[4] (which hold what
// the all-plain fallback path accumulated, normally 0).  hdr == NULL (cv_loss): the sums only.
// overwrite (tile path: nothing else accumulates into loss[]): the sums REPLACE loss[0..4] -- no memset of them needed.
__global__ __launch_bounds__(256) void t_loss_header(double *__restrict__ loss, double lambda, float *__restrict__ hdr,
                                                     const double *__restrict__ rows, int64_t nrows,
                                                     const double *__restrict__ l2_rows, int l2_kernels, int overwrite)
{
    __shared__ double sh[5][256];
    const int t = threadIdx.x;
    double a[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t r = t; r < nrows; r += 256)
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] += rows[r * 4 + j];
    if (l2_rows)
        for (int p = 0; p < l2_kernels; p++) a[4] += l2_rows[(size_t)p * 256 + t];
#pragma unroll
    for (int j = 0; j < 5; j++) sh[j][t] = a[j];
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (t < k)
#pragma unroll
            for (int j = 0; j < 5; j++) sh[j][t] += sh[j][t + k];
        __syncthreads();
    }
    if (t < 5) loss[t] = overwrite ? sh[t][0] : loss[t] + sh[t][0];
    __syncthreads();
    if (t >= 16 || !hdr) return;
    __threadfence_block();
    float v = 0.0f;
    if (t < 10) {
        const int k = t >> 1;
        const double d = k < 4 ? loss[k] : loss[4] * lambda;        // (written above by threads of this block, behind the barrier)
        const float hi = (float)d;
        v = (t & 1) ? (float)(d - (double)hi) : hi;
    } else if (t == 10) v = 1.0f;
    hdr[t] = v;
}

// acc[0..3] += data losses of the header, acc[4] += L2 term (divided by the rank count: it is identical on all
// ranks), acc[6] += 1 (steps accumulated)
__global__ void t_loss_accumulate(const float *__restrict__ hdr, double *__restrict__ acc)
{
    const int t = threadIdx.x;
    if (t < 4) acc[t] += (double)hdr[2 * t] + (double)hdr[2 * t + 1];
    else if (t == 4) acc[4] += ((double)hdr[8] + (double)hdr[9]) / (double)(hdr[10] > 0.5f ? hdr[10] : 1.0f);
    else if (t == 6) acc[6] += 1.0;
}

static int launch_l2(cv_model *m, hipStream_t sw, bool tile_path)
{
    l2_args la;
    for (int p = 0; p < CV_NUM_PARAMS; p += 2) { la.w[p / 2] = m->params + m->poff[p]; la.count[p / 2] = m->psize[p]; }
    t_l2<<<dim3(256, CV_NUM_PARAMS / 2), 256, 0, sw>>>(la, m->loss_dev + 4, tile_path ? m->l2_rows : nullptr);
    CV_HIP(hipGetLastError());
    CV_HIP(hipEventRecord(m->tr_l2_done, sw));       // (a loss header on another side stream waits for it)
    return 0;
}

static int train_workspace(cv_model *m, int64_t n, int64_t *slice_out)
{
    // one pass for train.py's batch of 10 000; larger batches go in equal slices of at most 65 536 candidates
    // (the kernels are at their best on thousands of groups, and HBM has room: ~0.4 MB of workspace per candidate)
    const int64_t max_slice = 65536;
    const int64_t nslice = n > 0 ? (n + max_slice - 1) / max_slice : 1;
    const int64_t slice = n > 0 ? ((n + nslice - 1) / nslice + 15) / 16 * 16 : 16;
    const size_t need = train_floats_per_cand(m) * (size_t)(slice + 16) * sizeof(float);
    if (m->t_bytes < need) {
        CV_HIP(hipDeviceSynchronize());
        if (m->t_buf) CV_HIP(hipFree(m->t_buf));
        m->t_buf = nullptr; m->t_bytes = 0; m->last_tr_n = 0;
        CV_HIP(hipMalloc(&m->t_buf, need));
        m->t_bytes = need;
    }
    // block rows of the heads kernel: 4 groups per block, slice after slice
    // (heads_train_tm: one row per four groups; train_tail_tm: one per group)
    const int64_t rows_need = nslice * (slice / 16 + 4);
    if (m->loss_rows_cap < rows_need) {
        CV_HIP(hipDeviceSynchronize());
        if (m->loss_rows) CV_HIP(hipFree(m->loss_rows));
        m->loss_rows = nullptr; m->loss_rows_cap = 0;
        CV_HIP(hipMalloc(&m->loss_rows, sizeof(double) * 4 * (size_t)rows_need));
        m->loss_rows_cap = rows_need;
    }
    if (!m->l2_rows) CV_HIP(hipMalloc(&m->l2_rows, sizeof(double) * 256 * (CV_NUM_PARAMS / 2)));
    m->loss_rows_used = 0;
    if (!m->tr_side) {
        CV_HIP(hipStreamCreateWithFlags(&m->tr_side, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) CV_HIP(hipStreamCreateWithFlags(&m->tr_side_more[i], hipStreamNonBlocking));
        // (side streams at the lowest priority, so that the data-gradient chain is dispatched first: measured 2.265 against
        // 2.249 ms at 10 000, no difference at 1 250 -- every large kernel fills the chip either way)
        for (int i = 0; i < CV_TR_EVENTS; i++) CV_HIP(hipEventCreateWithFlags(&m->tr_ev[i], hipEventDisableTiming));
        CV_HIP(hipEventCreateWithFlags(&m->tr_dense_ready, hipEventDisableTiming));
        CV_HIP(hipEventCreateWithFlags(&m->tr_l2_done, hipEventDisableTiming));
        CV_HIP(hipEventRecord(m->tr_l2_done, m->tr_side));       // (recorded once, so that a wait before any L2 kernel passes)
        CV_HIP(hipEventCreateWithFlags(&m->tr_pack_fork, hipEventDisableTiming));
        CV_HIP(hipEventCreateWithFlags(&m->tr_pack_done, hipEventDisableTiming));
    }
    *slice_out = slice;
    return 0;
}

// enqueue forward (+ backward) of the whole batch; no host synchronisation
static int train_enqueue(cv_model *m, const float *x, const float *y, int64_t n, bool backward, float drop4,
                         float lambda, uint64_t seed, uint64_t step, hipStream_t st, hipStream_t comm)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (n < 0) { cv_set_error("negative batch"); return 1; }
    if (n > 0 && (!x || !y)) { cv_set_error("null buffer"); return 1; }
    if (drop4 < 0.0f || drop4 >= 1.0f) { cv_set_error("dropout rate must be in [0,1)"); return 1; }
    CV_HIP(hipSetDevice(m->device));
    int64_t slice = 16;
    if (train_workspace(m, n, &slice)) return 1;
    if (backward && cv_wgrad_scratch_reserve(m)) return 1;
    // the side stream exists for the tile path only (its slices join it back into st before they return); the
    // all-plain path runs everything, the L2 term included, in stream order
    const bool tile_path = m->impl == 1 && cv_tile_supported(m);
    hipStream_t sw = (backward && m->train_overlap && tile_path) ? m->tr_side : st;
    // Tile path: every gradient element is written by the second pass of its layer's weight gradient, which STORES for
    // the first slice of a step and adds for the later ones (cv_model::tr_accumulate), and t_loss_header replaces the
    // loss sums -- nothing needs zeroing (the 6.5 MB memset was the first 5-7 us of every step).  An empty batch (a rank
    // without candidates) runs no kernel: its gradient is zeroed here.  All-plain path: atomics into zeroed buffers.
    const bool no_memset = tile_path && backward && n > 0 && (m->sched & 64);
    m->tr_accumulate = no_memset ? 0 : 1;
    if (no_memset) {
        /* nothing */
    } else if (backward && m->grads == m->grads_own + CV_GRAD_HEADER) {
        // gradients and the loss sums behind them (cv_create) in one memset
        const size_t bytes = (size_t)(reinterpret_cast<char *>(m->loss_dev + 8) - reinterpret_cast<char *>(m->grads));
        CV_HIP(hipMemsetAsync(m->grads, 0, bytes, st));
    } else {
        // (tile path: t_loss_header REPLACES the loss sums, nothing accumulates into them -- no memset of their own)
        if (!tile_path) CV_HIP(hipMemsetAsync(m->loss_dev, 0, sizeof(double) * 8, st));
        if (backward) CV_HIP(hipMemsetAsync(m->grads, 0, sizeof(float) * m->poff[CV_NUM_PARAMS], st));      // (a caller's bucket)
    }
    // ONE marker for everything the side stream does ahead of the backward pass -- the L2 term and the weight packing
    // both depend on the weights alone, i.e. on the optimizer update of the previous step (cv_pack_for_training is
    // told that sw is ordered already)
    const bool one_marker = (m->sched & 4) != 0;
    // (full topology up to 512 groups: -20 us at 79 groups, -5 at 313, +8 at 625 -- profiles/r05/step_ab_session11_side_work_behind_conv1.txt;
    // slim, whose first layer is a quarter of the work and whose fc4 needs its weights sooner: +25 us at 79 groups, so not there)
    const tr_defer defer{(m->sched & 2048) != 0 && one_marker && tile_path && sw != st && n > 0 && m->dbg[5] != 1 &&
                         m->wpr_fc4 != nullptr && (n < slice ? n : slice) <= 512 * 16, lambda, tile_path};
    const bool sw_ordered = sw != st && n > 0 && (one_marker || lambda != 0.0f);
    if (sw_ordered && !defer.on) {
        CV_HIP(hipEventRecord(m->tr_ev[CV_TR_EVENTS - 1], st));
        CV_HIP(hipStreamWaitEvent(sw, m->tr_ev[CV_TR_EVENTS - 1], 0));
    }
    // lambda * sum(w^2)/2: with a side stream it runs there, next to the forward pass (the slices join the side stream
    // before they return), instead of at the tail of the step
    bool l2_done = false;
    if (lambda != 0.0f && sw_ordered) {
        if (!defer.on && launch_l2(m, sw, tile_path)) return 1;       // (deferred: the first slice launches it behind conv1)
        l2_done = true;
    }
    // single-slice step on the tile path with side streams: the loss header rides behind the heads kernel on the side stream
    const tr_header hdr_early{(double)lambda, lambda != 0.0f};
    // (tiny batches: behind the heads kernel on sw; larger ones: on the second side stream -- at train.py's batch sw is as
    // long as the main chain, and 14 us of header at its head made the step 33 us longer, profiles/r05/step_ab_session3.txt)
    const bool early = backward && tile_path && sw_ordered && n <= slice && (l2_done || lambda == 0.0f) && (m->sched & 1);
    bool hdr_launched = false;
    // option keep_activations and several slices: the dropout maps of every slice are kept (cv_get_activation 6 / 7 then
    // covers the whole batch, and the oracle tests can feed a multi-slice step's own keep mask back); one slice: in place
    const size_t keep_per = tile_path ? (size_t)m->sh.nb4 * 16 : (size_t)m->arch.fc4;       // floats per candidate of a map
    const bool keep_all = m->keep_act && n > slice;
    const size_t keep_half = keep_all ? (size_t)(n + 16) * keep_per : 0;
    if (keep_all && m->tr_keep_floats < 2 * keep_half) {
        CV_HIP(hipDeviceSynchronize());
        if (m->tr_keep) CV_HIP(hipFree(m->tr_keep));
        m->tr_keep = nullptr; m->tr_keep_floats = 0;
        CV_HIP(hipMalloc(&m->tr_keep, sizeof(float) * 2 * keep_half));
        m->tr_keep_floats = 2 * keep_half;
    }
    bool recorded = false;
    for (int64_t off = 0; off < n; off += slice) {
        int64_t cn = n - off < slice ? n - off : slice;
        const bool last = off + slice >= n;
        // (no communication stream -- one rank, or the whole bucket exchanged behind the step: nobody waits for "dense
        // gradients final", so the side streams are not gathered for it)
        hipEvent_t ev = (backward && last && comm) ? m->tr_dense_ready : nullptr;
        if (train_slice(m, x + (size_t)off * (CV_INPUT_H * 16), y + (size_t)off * 16, cn, off, backward, drop4,
                        seed, step, st, sw, ev, sw_ordered && one_marker, early ? &hdr_early : nullptr, off == 0 ? &defer : nullptr,
                        &hdr_launched))
            return 1;
        recorded = recorded || ev != nullptr;
        m->tr_accumulate = 1;                     // the slices behind the first one add
        if (keep_all && m->last_tr_d4) {          // (slices are multiples of 16 candidates: the tile-major maps concatenate)
            const size_t cnt = (size_t)(tile_path ? (cn + 15) / 16 * 16 : cn) * keep_per;
            CV_HIP(hipMemcpyAsync(m->tr_keep + (size_t)off * keep_per, m->last_tr_mask, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
            CV_HIP(hipMemcpyAsync(m->tr_keep + keep_half + (size_t)off * keep_per, m->last_tr_d4, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
        }
    }
    if (keep_all && m->last_tr_d4) { m->last_tr_mask = m->tr_keep; m->last_tr_d4 = m->tr_keep + keep_half; m->last_tr_n = n; }
    if (backward && comm) {
        if (!recorded) CV_HIP(hipEventRecord(m->tr_dense_ready, st));      // empty batch
        CV_HIP(hipStreamWaitEvent(comm, m->tr_dense_ready, 0));
    }
    if (lambda != 0.0f && !l2_done) {
        l2_args la;
        for (int p = 0; p < CV_NUM_PARAMS; p += 2) { la.w[p / 2] = m->params + m->poff[p]; la.count[p / 2] = m->psize[p]; }
        t_l2<<<dim3(256, CV_NUM_PARAMS / 2), 256, 0, st>>>(la, m->loss_dev + 4, tile_path ? m->l2_rows : nullptr);
    }
    // the fixed-order loss sums (and, for a training step, the header of the gradient bucket) -- unless the slice launched them
    if (!hdr_launched)
        t_loss_header<<<1, 256, 0, st>>>(m->loss_dev, (double)lambda, backward ? m->grads - CV_GRAD_HEADER : nullptr, m->loss_rows,
                                         m->loss_rows_used, (lambda != 0.0f && tile_path) ? m->l2_rows : nullptr, CV_NUM_PARAMS / 2,
                                         tile_path ? 1 : 0);
    CV_HIP(hipGetLastError());
    return 0;
}

static int train_pass(cv_model *m, const float *x, const float *y, int64_t n, bool backward, float drop4,
                      float lambda, uint64_t seed, uint64_t step, double *losses_host, hipStream_t st)
{
    if (train_enqueue(m, x, y, n, backward, drop4, lambda, seed, step, st, nullptr)) return 1;
    double h[8];
    CV_HIP(hipMemcpyAsync(h, m->loss_dev, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
    CV_HIP(hipStreamSynchronize(st));
    if (losses_host) {
        for (int k = 0; k < 4; k++) losses_host[k] = h[k];
        losses_host[4] = h[4] * (double)lambda;
        losses_host[5] = h[0] + h[1] + h[2] + h[3] + losses_host[4];
    }
    return 0;
}

extern "C" int cv_loss(cv_model *m, const float *x_dev, const float *y_dev, int64_t n, double *losses_host,
                       void *stream)
{
    return train_pass(m, x_dev, y_dev, n, false, 0.0f, 0.0f, 0, 0, losses_host, (hipStream_t)stream);
}

extern "C" int cv_grad(cv_model *m, const float *x_dev, const float *y_dev, int64_t n, float drop4,
                       float lambda, uint64_t seed, uint64_t step, double *losses_host, void *stream)
{
    return train_pass(m, x_dev, y_dev, n, true, drop4, lambda, seed, step, losses_host, (hipStream_t)stream);
}

extern "C" int cv_grad_async(cv_model *m, const float *x_dev, const float *y_dev, int64_t n, float drop4,
                             float lambda, uint64_t seed, uint64_t step, void *stream, void *comm_stream)
{
    return train_enqueue(m, x_dev, y_dev, n, true, drop4, lambda, seed, step, (hipStream_t)stream,
                         (hipStream_t)comm_stream);
}

extern "C" int cv_loss_accumulate(cv_model *m, void *stream)
{
    if (!m) { cv_set_error("null model"); return 1; }
    CV_HIP(hipSetDevice(m->device));
    t_loss_accumulate<<<1, 64, 0, (hipStream_t)stream>>>(m->grads - CV_GRAD_HEADER, m->loss_acc);
    CV_HIP(hipGetLastError());
    return 0;
}

extern "C" int cv_loss_read(cv_model *m, double losses_host[6], int64_t *steps, int reset, void *stream)
{
    if (!m || !losses_host) { cv_set_error("cv_loss_read: null argument"); return 1; }
    CV_HIP(hipSetDevice(m->device));
    hipStream_t st = (hipStream_t)stream;
    double h[8];
    CV_HIP(hipMemcpyAsync(h, m->loss_acc, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
    if (reset) CV_HIP(hipMemsetAsync(m->loss_acc, 0, sizeof(double) * 8, st));
    CV_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 5; k++) losses_host[k] = h[k];
    losses_host[5] = h[0] + h[1] + h[2] + h[3] + h[4];
    if (steps) *steps = (int64_t)(h[6] + 0.5);
    return 0;
}
