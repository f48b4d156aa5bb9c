// cv_kernels_ref.hip -- plain one-thread-per-output kernels (impl 0).
//
// The simplest possible statement of the canonical arithmetic on the GPU: every
// output element is one fmaf chain in ascending (kh, kw, ci) / k order.  Used as
// the on-device cross-check of the MFMA tile kernels (they must agree bit for
// bit), as the VALU arm of the VALU-vs-MFMA comparison in DESIGN.md, and for
// architectures the tile kernels do not cover.  Natural (reference) layouts:
// activations [n,h,4,c] NHWC, dense [n,units].
//
// Graph followed: /root/reference/clairvoyante/clairvoyante_v3.py:54-138.
#include "cv_internal.hpp"
#include "cv_math.hpp"

namespace {

// conv2d SAME (kw = 4: pad 1 left / 2 right; kh: before = (kh-1)/2) + bias + selu
__global__ void ref_conv_selu(const float *__restrict__ in, const float *__restrict__ w,
                              const float *__restrict__ bias, float *__restrict__ out, int64_t n,
                              int H, int cin, int kh, int cout)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * H * 4 * cout;
    if (t >= total) return;
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
    out[t] = cvm::selu(acc + bias[co]);
}

// max_pooling2d (p,1), stride 1, VALID
__global__ void ref_pool(const float *__restrict__ in, float *__restrict__ out, int64_t n, int H,
                         int c, int p)
{
    int Ho = H - p + 1, row = 4 * c;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * Ho * row;
    if (t >= total) return;
    int e = (int)(t % row);
    int64_t r = t / row;
    int h = (int)(r % Ho);
    int64_t i = r / Ho;
    const float *b = in + ((size_t)i * H + h) * row + e;
    float m = b[0];
    for (int d = 1; d < p; d++) m = fmaxf(m, b[(size_t)d * row]);
    out[t] = m;
}

// dense + bias + selu
__global__ void ref_dense_selu(const float *__restrict__ x, const float *__restrict__ w,
                               const float *__restrict__ bias, float *__restrict__ y, int64_t n, int K,
                               int N)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N) return;
    int j = (int)(t % N);
    int64_t i = t / N;
    const float *xi = x + (size_t)i * K;
    float acc = 0.0f;
    for (int k = 0; k < K; k++) acc = __builtin_fmaf(xi[k], w[(size_t)k * N + j], acc);
    y[t] = cvm::selu(acc + bias[j]);
}

// The four heads (v3.py:124-138): 16 candidates x 16 outputs per block.
//   base     = sigmoid(d4 . Wb + bb)            (reads the fc4 side, v3.py:125)
//   zyg/type/len = softmax(selu(fc5 . W + b) + 1e-10)
template <bool TM>
__global__ __launch_bounds__(256) void heads_kernel(
    const float *__restrict__ h4, const float *__restrict__ h5, int K4, int K5, int KB4, int KB5,
    const float *__restrict__ wb, const float *__restrict__ bb, const float *__restrict__ wz,
    const float *__restrict__ bz, const float *__restrict__ wt, const float *__restrict__ bt,
    const float *__restrict__ wl, const float *__restrict__ bl, int64_t n, float *__restrict__ out16)
{
    __shared__ float pre[16][17];
    int c = threadIdx.x >> 4, j = threadIdx.x & 15;
    int64_t cand = (int64_t)blockIdx.x * 16 + c;
    int64_t cl = cand < n ? cand : n - 1;
    const float *w; const float *b; int idx, nh, K, KB; const float *src;
    if (j < 4)       { w = wb; b = bb; idx = j;      nh = 4; K = K4; KB = KB4; src = h4; }
    else if (j < 6)  { w = wz; b = bz; idx = j - 4;  nh = 2; K = K5; KB = KB5; src = h5; }
    else if (j < 10) { w = wt; b = bt; idx = j - 6;  nh = 4; K = K5; KB = KB5; src = h5; }
    else             { w = wl; b = bl; idx = j - 10; nh = 6; K = K5; KB = KB5; src = h5; }
    float acc = 0.0f;
    for (int k = 0; k < K; k++) {
        float xv = TM ? src[cv_tm_index(cl, k, KB)] : src[(size_t)cl * K + k];
        acc = __builtin_fmaf(xv, w[(size_t)k * nh + idx], acc);
    }
    pre[c][j] = acc + b[idx];
    __syncthreads();
    if (cand >= n) return;
    float *o = out16 + (size_t)cand * 16;
    if (j == 0) {
        for (int k = 0; k < 4; k++) o[k] = cvm::sigmoid(pre[c][k]);
    } else if (j == 1) {
        float l[2], p[2];
        for (int k = 0; k < 2; k++) l[k] = cvm::selu(pre[c][4 + k]) + 1e-10f;
        cvm::softmax<2>(l, p);
        for (int k = 0; k < 2; k++) o[4 + k] = p[k];
    } else if (j == 2) {
        float l[4], p[4];
        for (int k = 0; k < 4; k++) l[k] = cvm::selu(pre[c][6 + k]) + 1e-10f;
        cvm::softmax<4>(l, p);
        for (int k = 0; k < 4; k++) o[6 + k] = p[k];
    } else if (j == 3) {
        float l[6], p[6];
        for (int k = 0; k < 6; k++) l[k] = cvm::selu(pre[c][10 + k]) + 1e-10f;
        cvm::softmax<6>(l, p);
        for (int k = 0; k < 6; k++) o[10 + k] = p[k];
    }
}

// TM -> natural copy: feature k = pos*FPP + f (f < FP real, FPP padded)
__global__ void tm_to_natural(const float *__restrict__ tm, int KB, int FPP, int FP, int npos, int64_t n,
                              float *__restrict__ dst)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n * npos * FP;
    if (t >= total) return;
    int f = (int)(t % FP);
    int64_t r = t / FP;
    int pos = (int)(r % npos);
    int64_t i = r / npos;
    dst[t] = tm[cv_tm_index(i, pos * FPP + f, KB)];
}

inline unsigned nblk(int64_t total, int bs) { return (unsigned)((total + bs - 1) / bs); }

}  // namespace

int cv_launch_heads(cv_model *m, const float *h4, const float *h5, int tm, int64_t n, float *out16,
                    hipStream_t st)
{
    if (n <= 0) return 0;
    const float *P = m->params;
    const int64_t *o = m->poff;
    unsigned g = (unsigned)((n + 15) / 16);
    if (tm)
        heads_kernel<true><<<g, 256, 0, st>>>(h4, h5, m->arch.fc4, m->arch.fc5, m->sh.nb4, m->sh.nb5,
                                             P + o[10], P + o[11], P + o[12], P + o[13], P + o[14],
                                             P + o[15], P + o[16], P + o[17], n, out16);
    else
        heads_kernel<false><<<g, 256, 0, st>>>(h4, h5, m->arch.fc4, m->arch.fc5, 0, 0, P + o[10],
                                              P + o[11], P + o[12], P + o[13], P + o[14], P + o[15],
                                              P + o[16], P + o[17], n, out16);
    CV_HIP(hipGetLastError());
    return 0;
}

int cv_tm_to_natural(const float *tm, int KB, int FPP, int FP, int npos, int64_t n, float *dst,
                     hipStream_t st)
{
    if (n <= 0) return 0;
    int64_t total = n * npos * FP;
    tm_to_natural<<<nblk(total, 256), 256, 0, st>>>(tm, KB, FPP, FP, npos, n, dst);
    CV_HIP(hipGetLastError());
    return 0;
}

static int ref_alloc(cv_model *m, int64_t cap)
{
    if (m->ref_cap >= cap) return 0;
    for (int l = 0; l < 3; l++) {
        if (m->r_a[l]) hipFree(m->r_a[l]);
        if (m->r_p[l]) hipFree(m->r_p[l]);
        m->r_a[l] = m->r_p[l] = nullptr;
    }
    if (m->r_h4) hipFree(m->r_h4);
    if (m->r_h5) hipFree(m->r_h5);
    m->r_h4 = m->r_h5 = nullptr;
    m->ref_cap = 0;
    for (int l = 0; l < 3; l++) {
        CV_HIP(hipMalloc(&m->r_a[l], sizeof(float) * cap * m->sh.hc[l] * 4 * m->arch.cout[l]));
        CV_HIP(hipMalloc(&m->r_p[l], sizeof(float) * cap * m->sh.hp[l] * 4 * m->arch.cout[l]));
    }
    CV_HIP(hipMalloc(&m->r_h4, sizeof(float) * cap * m->arch.fc4));
    CV_HIP(hipMalloc(&m->r_h5, sizeof(float) * cap * m->arch.fc5));
    m->ref_cap = cap;
    return 0;
}

// one chunk (n <= ref_cap) through the plain kernels
int cv_ref_forward(cv_model *m, const float *x, int64_t n, float *out16, hipStream_t st)
{
    if (n <= 0) return 0;
    if (ref_alloc(m, n)) return 1;
    const float *P = m->params;
    const int64_t *o = m->poff;
    const float *in = x;
    for (int l = 0; l < 3; l++) {
        int H = m->sh.hc[l], C = m->arch.cout[l];
        int64_t tot = n * H * 4 * C;
        ref_conv_selu<<<nblk(tot, 256), 256, 0, st>>>(in, P + o[2 * l], P + o[2 * l + 1], m->r_a[l], n, H,
                                                      m->sh.cin[l], m->arch.kh[l], C);
        int64_t totp = n * m->sh.hp[l] * 4 * C;
        ref_pool<<<nblk(totp, 256), 256, 0, st>>>(m->r_a[l], m->r_p[l], n, H, C, m->arch.pool[l]);
        in = m->r_p[l];
    }
    ref_dense_selu<<<nblk(n * m->arch.fc4, 256), 256, 0, st>>>(in, P + o[6], P + o[7], m->r_h4, n,
                                                               m->sh.flat, m->arch.fc4);
    ref_dense_selu<<<nblk(n * m->arch.fc5, 256), 256, 0, st>>>(m->r_h4, P + o[8], P + o[9], m->r_h5, n,
                                                               m->arch.fc4, m->arch.fc5);
    CV_HIP(hipGetLastError());
    m->last_n = n;
    m->last_impl = 0;
    return cv_launch_heads(m, m->r_h4, m->r_h5, 0, n, out16, st);
}
