/*
 * cv_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the arithmetic of the Clairvoyante v3 / v3-slim
 * pileup CNN, used ONLY by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg to check (and to be timed beside) the HIP path.  Nothing
 * in clairvoyante_amd/ may import, link or call this file.
 *
 * PARITY STATUS: *parity unpinned by the reference*.  The arithmetic of the
 * reference lives in third-party TensorFlow (pinned tensorflow==1.12.0,
 * /root/reference/requirements.txt:1) which is not vendored and cannot be
 * imported in this image, and the reference ships no tests / golden vectors
 * for this path.  The restatement is pinned instead
 *   (1) against an independently written torch-CPU formulation of the same
 *       graph (tests/torch_ref.py; fixtures tests/golden/forward_*.npz written by
 *       tests/golden/make_golden_forward.py), and
 *   (2) for the host logic (VCF Output, GetTensor, DecompressArray) against
 *       the reference's own Python imported in the build container.
 *
 * What each function follows (file:line in /root/reference):
 *   conv/pool/fc/head graph ......... clairvoyante/clairvoyante_v3.py:54-138
 *   slim topology ................... clairvoyante/clairvoyante_v3_slim.py:53-118
 *   selu ............................ clairvoyante/selu.py:21-25
 *   alpha-dropout ................... clairvoyante/selu.py:34-69
 *   loss ............................ clairvoyante/clairvoyante_v3.py:140-152
 *   Adam (TF1 AdamOptimizer) ........ clairvoyante/clairvoyante_v3.py:174
 *
 * TF op semantics restated: conv2d = cross-correlation, NHWC / HWIO, SAME
 * padding (total = k-1, before = total/2, after = total-before), stride 1;
 * max_pooling2d VALID, stride 1, window (p,1); dense = x.W + b; softmax with
 * max subtraction; sigmoid = 1/(1+exp(-x)); l2_loss = sum(w^2)/2.
 *
 * CANONICAL SUMMATION ORDER (TensorFlow's Eigen order is unspecified, so the
 * oracle fixes one and the HIP kernels are written to reproduce it bit for
 * bit):  every contraction is ONE fp32 fused-multiply-add chain
 *      acc = 0;  for k ascending: acc = fmaf(x[k], w[k], acc);  y = acc + bias
 * with k running over (kh, kw, ci) for convolutions -- taps that fall on SAME
 * padding are skipped (they would add an exact 0) -- and over the input index
 * for dense layers.  exp() is the fixed polynomial cvo_expf below (Cephes
 * style, fmaf-only), so that CPU and GPU agree bitwise; it is within 2 ulp of
 * the true exponential, i.e. indistinguishable from TF's at the 1e-4 level.
 *
 * Build: see oracle/Makefile  (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CVO_H 33
#define CVO_W 4
#define CVO_CIN 4
#define CVO_KW 4
#define CVO_NOUT 16
#define CVO_NPARAM 18

/* Architecture descriptor.  v3 full: kh={1,2,3} cout={16,32,48} pool={5,4,3}
 * fc4=336 fc5=168 (clairvoyante_v3.py:9-12).  v3 slim: kh={1,3,5}
 * cout={8,16,32} pool={1,1,1} fc4=36 fc5=18 (clairvoyante_v3_slim.py:9-11). */
typedef struct {
    int kh[3];
    int cout[3];
    int pool[3];
    int fc4, fc5;
} cvo_arch;

/* Parameter order (TF variable names, TF layouts):
 *  0 conv1/kernel [kh,4,4,c1]   1 conv1/bias
 *  2 conv2/kernel [kh,4,c1,c2]  3 conv2/bias
 *  4 conv3/kernel [kh,4,c2,c3]  5 conv3/bias
 *  6 fc4/kernel [flat,fc4]      7 fc4/bias
 *  8 fc5/kernel [fc4,fc5]       9 fc5/bias
 * 10 YBaseChangeSigmoid/kernel [fc4,4]  11 bias      (reads fc4: v3.py:125)
 * 12 YZygosityFC/kernel [fc5,2]         13 bias
 * 14 YVarTypeFC/kernel [fc5,4]          15 bias
 * 16 YIndelLengthFC/kernel [fc5,6]      17 bias                              */

static const float SELU_ALPHA = 1.6732632423543772848170429916717f;
static const float SELU_SCALE = 1.0507009873554804934193349852946f;

/* ---- canonical scalar math --------------------------------------------- */

static inline float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Fixed-sequence expf: identical IEEE operations on CPU and GPU. */
float cvo_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -87.33654475055310f) return 0.0f;          /* flush below FLT_MIN */
    float z = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(z, -0.693359375f, x);
    r = __builtin_fmaf(z, 2.12194440e-4f, r);
    float r2 = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    int n = (int)z;
    int n1 = n >> 1, n2 = n - n1;
    y = y * as_float((uint32_t)(n1 + 127) << 23);
    y = y * as_float((uint32_t)(n2 + 127) << 23);
    return y;
}

/* selu.py:21-25: scale * where(x>=0, x, alpha*elu(x)), elu(x)=exp(x)-1.
 * The negative branch has its OWN fixed operation sequence (the device evaluates it ~40 times per candidate and
 * position, and on gfx950 every fp32 VALU operation is issue time next to the MFMAs): one-constant range reduction
 * (|z| <= 126: the error of fl(ln 2) stays below 2.4e-7 of exp, and only where exp(x) << 1), the Cephes polynomial of
 * cvo_expf, and scale*alpha*(y - 1) as ONE fused multiply-add with the product constant -- 3 operations fewer than
 * scale*(alpha*(cvo_expf(x) - 1)) and a little more accurate (max |err| 1.2e-7 against 2.0e-7 over all negative
 * floats).  Monotone over every negative float (cvo_selu_sweep), which the kernels rely on. */
static const float SELU_SA = (float)(1.0507009873554804934193349852946 * 1.6732632423543772848170429916717);
static inline float cvo_selu(float x)
{
    if (x >= 0.0f) return SELU_SCALE * x;
    if (x != x) return SELU_SCALE * x;                          /* NaN stays NaN (the device's select does the same) */
    float xc = x < -87.33654475055310f ? -87.33654475055310f : x;   /* exp(x) - 1 is -1 below the flush threshold */
    float z = __builtin_rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(z, -0.69314718055994530942f, xc);
    float r2 = r * r;
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    y = ldexpf(y, (int)z);                                      /* z in [-126, 0], y in [0.70, 1.42): a normal number */
    /* scale*alpha*(y - 1) as -(SA - SA*y): the same bits as fma(y, SA, -SA) except where y == 1 (-2^-25 < x < 0), which
     * gives -0.0 -- the output keeps the sign of a negative input, and the device's backward pass reads selu' off the
     * output's sign bit (csrc/cv_unpool.hpp).  The sign is set on the bit pattern: a compiler may turn -fma(a, b, c)
     * into a fused negate-multiply-subtract, whose exact zero is +0. */
    float mag = __builtin_fmaf(y, -SELU_SA, SELU_SA);           /* >= +0 */
    uint32_t mb; memcpy(&mb, &mag, 4); mb |= 0x80000000u; memcpy(&mag, &mb, 4);
    return mag;
}
float cvo_selu_scalar(float x) { return cvo_selu(x); }

static inline float cvo_selu_grad_from_pre(float pre)
{
    if (pre >= 0.0f) return SELU_SCALE;
    return SELU_SCALE * (SELU_ALPHA * cvo_expf(pre));
}

static inline float cvo_sigmoid(float x) { return 1.0f / (1.0f + cvo_expf(-x)); }
float cvo_sigmoid_scalar(float x) { return cvo_sigmoid(x); }

static void cvo_softmax(const float *l, int n, float *p)
{
    float m = l[0];
    for (int i = 1; i < n; i++) m = fmaxf(m, l[i]);
    float s = 0.0f;
    for (int i = 0; i < n; i++) { p[i] = cvo_expf(l[i] - m); s = (i == 0) ? p[0] : s + p[i]; }
    for (int i = 0; i < n; i++) p[i] = p[i] / s;
}

/* ---- shapes -------------------------------------------------------------- */

typedef struct {
    int hc[3];   /* conv output heights (== input heights, SAME) */
    int hp[3];   /* pooled heights */
    int cin[3];
    int flat;
} cvo_shape;

static void cvo_shapes(const cvo_arch *a, cvo_shape *s)
{
    int h = CVO_H, c = CVO_CIN;
    for (int l = 0; l < 3; l++) {
        s->cin[l] = c;
        s->hc[l] = h;
        h = h - (a->pool[l] - 1);
        s->hp[l] = h;
        c = a->cout[l];
    }
    s->flat = h * CVO_W * c;
}

int cvo_flat_size(const cvo_arch *a) { cvo_shape s; cvo_shapes(a, &s); return s.flat; }

/* sizes of the 18 parameters, in floats */
void cvo_param_sizes(const cvo_arch *a, int64_t *sz)
{
    cvo_shape s; cvo_shapes(a, &s);
    for (int l = 0; l < 3; l++) {
        sz[2 * l] = (int64_t)a->kh[l] * CVO_KW * s.cin[l] * a->cout[l];
        sz[2 * l + 1] = a->cout[l];
    }
    sz[6] = (int64_t)s.flat * a->fc4; sz[7] = a->fc4;
    sz[8] = (int64_t)a->fc4 * a->fc5; sz[9] = a->fc5;
    sz[10] = (int64_t)a->fc4 * 4; sz[11] = 4;
    sz[12] = (int64_t)a->fc5 * 2; sz[13] = 2;
    sz[14] = (int64_t)a->fc5 * 4; sz[15] = 4;
    sz[16] = (int64_t)a->fc5 * 6; sz[17] = 6;
}

/* ---- layers (one candidate) --------------------------------------------- */

/* conv2d SAME + bias (pre-activation) ; in [H][4][cin], out [H][4][cout] */
static void conv_pre(const float *in, int H, int cin, const float *wt, const float *bias,
                     int kh, int cout, float *out)
{
    const int padt = (kh - 1) / 2;   /* SAME: before = total/2 */
    const int padl = 1;              /* kw=4: total 3 -> before 1, after 2 */
    float acc[64];
    for (int h = 0; h < H; h++)
        for (int w = 0; w < CVO_W; w++) {
            for (int co = 0; co < cout; co++) acc[co] = 0.0f;
            for (int a = 0; a < kh; a++) {
                int hi = h + a - padt;
                if (hi < 0 || hi >= H) continue;
                for (int b = 0; b < CVO_KW; b++) {
                    int wi = w + b - padl;
                    if (wi < 0 || wi >= CVO_W) continue;
                    const float *xr = in + ((size_t)hi * CVO_W + wi) * cin;
                    const float *wr = wt + ((size_t)(a * CVO_KW + b) * cin) * cout;
                    for (int ci = 0; ci < cin; ci++) {
                        const float xv = xr[ci];
                        const float *wc = wr + (size_t)ci * cout;
                        for (int co = 0; co < cout; co++)
                            acc[co] = __builtin_fmaf(xv, wc[co], acc[co]);
                    }
                }
            }
            float *o = out + ((size_t)h * CVO_W + w) * cout;
            for (int co = 0; co < cout; co++) o[co] = acc[co] + bias[co];
        }
}

static void selu_inplace_copy(const float *pre, float *act, int n)
{
    for (int i = 0; i < n; i++) act[i] = cvo_selu(pre[i]);
}

/* max_pooling2d (p,1) stride 1 VALID over h; in [H][4][c] -> out [H-p+1][4][c] */
static void pool_h(const float *in, int H, int c, int p, float *out)
{
    int Ho = H - p + 1, row = CVO_W * c;
    for (int h = 0; h < Ho; h++)
        for (int i = 0; i < row; i++) {
            float m = in[(size_t)h * row + i];
            for (int d = 1; d < p; d++) m = fmaxf(m, in[(size_t)(h + d) * row + i]);
            out[(size_t)h * row + i] = m;
        }
}

/* dense pre-activation: y[n] = (sum_k x[k] w[k][n]) + b[n] */
static void dense_pre(const float *x, int K, const float *wt, const float *bias, int N, float *y)
{
    float acc[512];
    for (int n = 0; n < N; n++) acc[n] = 0.0f;
    for (int k = 0; k < K; k++) {
        const float xv = x[k];
        const float *wr = wt + (size_t)k * N;
        for (int n = 0; n < N; n++) acc[n] = __builtin_fmaf(xv, wr[n], acc[n]);
    }
    for (int n = 0; n < N; n++) y[n] = acc[n] + bias[n];
}

/* Per-candidate activation record offsets (floats) for cvo_forward_all */
typedef struct {
    size_t pre[3], act[3], pooled[3]; /* conv pre-activation, selu, pooled */
    size_t fc4pre, fc4, d4, fc5pre, fc5;
    size_t hpre[4];                   /* head pre-activations (4,2,4,6) */
    size_t out;                       /* 16 outputs */
    size_t total;
} cvo_layout;

static void cvo_make_layout(const cvo_arch *a, cvo_layout *L)
{
    cvo_shape s; cvo_shapes(a, &s);
    size_t o = 0;
    for (int l = 0; l < 3; l++) {
        size_t nc = (size_t)s.hc[l] * CVO_W * a->cout[l];
        size_t np = (size_t)s.hp[l] * CVO_W * a->cout[l];
        L->pre[l] = o; o += nc;
        L->act[l] = o; o += nc;
        L->pooled[l] = o; o += np;
    }
    L->fc4pre = o; o += a->fc4;
    L->fc4 = o; o += a->fc4;
    L->d4 = o; o += a->fc4;
    L->fc5pre = o; o += a->fc5;
    L->fc5 = o; o += a->fc5;
    static const int hn[4] = {4, 2, 4, 6};
    for (int i = 0; i < 4; i++) { L->hpre[i] = o; o += hn[i]; }
    L->out = o; o += CVO_NOUT;
    L->total = o;
}

int64_t cvo_record_size(const cvo_arch *a) { cvo_layout L; cvo_make_layout(a, &L); return (int64_t)L.total; }

/* offsets exported for Python: order pre1,act1,pool1,pre2,act2,pool2,pre3,act3,pool3,
 * fc4pre,fc4,d4,fc5pre,fc5,hpre0..3,out,total  (19 values) */
void cvo_record_offsets(const cvo_arch *a, int64_t *o)
{
    cvo_layout L; cvo_make_layout(a, &L);
    int j = 0;
    for (int l = 0; l < 3; l++) { o[j++] = L.pre[l]; o[j++] = L.act[l]; o[j++] = L.pooled[l]; }
    o[j++] = L.fc4pre; o[j++] = L.fc4; o[j++] = L.d4; o[j++] = L.fc5pre; o[j++] = L.fc5;
    for (int i = 0; i < 4; i++) o[j++] = L.hpre[i];
    o[j++] = L.out; o[j++] = L.total;
}

/* Forward of ONE candidate into a record.  mask4: NULL (inference: dropout is
 * the identity, selu.py:66-69) or fc4-long keep mask (0/1) with rate4. */
static void forward_one(const cvo_arch *a, const cvo_layout *L, const float *const *P,
                        const float *x, float *rec, const float *mask4, float rate4)
{
    cvo_shape s; cvo_shapes(a, &s);
    const float *in = x;
    for (int l = 0; l < 3; l++) {
        int n = s.hc[l] * CVO_W * a->cout[l];
        conv_pre(in, s.hc[l], s.cin[l], P[2 * l], P[2 * l + 1], a->kh[l], a->cout[l], rec + L->pre[l]);
        selu_inplace_copy(rec + L->pre[l], rec + L->act[l], n);
        pool_h(rec + L->act[l], s.hc[l], a->cout[l], a->pool[l], rec + L->pooled[l]);
        in = rec + L->pooled[l];
    }
    /* flatten = row-major NHWC reshape (v3.py:99-102): identity on memory */
    dense_pre(in, s.flat, P[6], P[7], a->fc4, rec + L->fc4pre);
    selu_inplace_copy(rec + L->fc4pre, rec + L->fc4, a->fc4);
    if (mask4) {
        /* selu.py:38-64 with alpha' = -1.7580993408473766, fixedPointMean 0, Var 1 */
        const float ap = -1.7580993408473766f;
        float q = 1.0f - rate4;
        float aa = sqrtf(1.0f / (q * ((1.0f - q) * (ap * ap) + 1.0f)));
        float bb = 0.0f - aa * (q * 0.0f + (1.0f - q) * ap);
        for (int i = 0; i < a->fc4; i++) {
            float m = mask4[i];
            float r = rec[L->fc4 + i] * m + ap * (1.0f - m);
            rec[L->d4 + i] = aa * r + bb;
        }
    } else {
        memcpy(rec + L->d4, rec + L->fc4, sizeof(float) * a->fc4);
    }
    dense_pre(rec + L->d4, a->fc4, P[8], P[9], a->fc5, rec + L->fc5pre);
    selu_inplace_copy(rec + L->fc5pre, rec + L->fc5, a->fc5);
    /* heads (v3.py:124-138).  dropout5 rate is 0.0 -> identity (selu.py:54-62) */
    float *o = rec + L->out;
    dense_pre(rec + L->d4, a->fc4, P[10], P[11], 4, rec + L->hpre[0]);
    for (int i = 0; i < 4; i++) o[i] = cvo_sigmoid(rec[L->hpre[0] + i]);
    static const int hn[4] = {4, 2, 4, 6};
    static const int ho[4] = {0, 4, 6, 10};
    for (int hd = 1; hd < 4; hd++) {
        float lg[6];
        dense_pre(rec + L->fc5, a->fc5, P[10 + 2 * hd], P[11 + 2 * hd], hn[hd], rec + L->hpre[hd]);
        for (int i = 0; i < hn[hd]; i++) lg[i] = cvo_selu(rec[L->hpre[hd] + i]) + 1e-10f;
        cvo_softmax(lg, hn[hd], o + ho[hd]);
    }
}

/* ---- public: inference --------------------------------------------------- */

/* dense pre-activation for a block of candidates: the weight row w[k][:] is loaded once per
 * block instead of once per candidate (fc4 is 6.2 MB); each accumulator is still the canonical
 * ascending-k fmaf chain, so the result is bit-identical to dense_pre. */
#define CVO_BLK 8
static void dense_pre_block(const float *const *xs, int nb, int K, const float *wt, const float *bias, int N,
                            float *const *ys)
{
    float acc[CVO_BLK][512];
    for (int b = 0; b < nb; b++)
        for (int n = 0; n < N; n++) acc[b][n] = 0.0f;
    for (int k = 0; k < K; k++) {
        const float *wr = wt + (size_t)k * N;
        for (int b = 0; b < nb; b++) {
            const float xv = xs[b][k];
            float *ab = acc[b];
            for (int n = 0; n < N; n++) ab[n] = __builtin_fmaf(xv, wr[n], ab[n]);
        }
    }
    for (int b = 0; b < nb; b++)
        for (int n = 0; n < N; n++) ys[b][n] = acc[b][n] + bias[n];
}

/* out16[n][16] = base(4) | zygosity(2) | type(4) | length(6)  (clairvoyante_v3.py:257-266) */
void cvo_predict(const cvo_arch *a, const float *const *P, const float *x, int64_t n,
                 float *out16, int nthreads)
{
    cvo_layout L; cvo_make_layout(a, &L);
    cvo_shape s; cvo_shapes(a, &s);
    static const int hn[4] = {4, 2, 4, 6};
    static const int ho[4] = {0, 4, 6, 10};
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    const int64_t nblk = (n + CVO_BLK - 1) / CVO_BLK;
#pragma omp parallel
    {
        float *recs = (float *)malloc(sizeof(float) * L.total * CVO_BLK);
#pragma omp for schedule(dynamic, 4)
        for (int64_t bi = 0; bi < nblk; bi++) {
            const int64_t i0 = bi * CVO_BLK;
            const int nb = (int)(n - i0 < CVO_BLK ? n - i0 : CVO_BLK);
            const float *xs[CVO_BLK]; float *ys[CVO_BLK];
            for (int b = 0; b < nb; b++) {          /* conv stack per candidate */
                float *rec = recs + (size_t)b * L.total;
                const float *in = x + (size_t)(i0 + b) * CVO_H * CVO_W * CVO_CIN;
                for (int l = 0; l < 3; l++) {
                    int cnt = s.hc[l] * CVO_W * a->cout[l];
                    conv_pre(in, s.hc[l], s.cin[l], P[2 * l], P[2 * l + 1], a->kh[l], a->cout[l], rec + L.pre[l]);
                    selu_inplace_copy(rec + L.pre[l], rec + L.act[l], cnt);
                    pool_h(rec + L.act[l], s.hc[l], a->cout[l], a->pool[l], rec + L.pooled[l]);
                    in = rec + L.pooled[l];
                }
            }
            for (int b = 0; b < nb; b++) { xs[b] = recs + (size_t)b * L.total + L.pooled[2]; ys[b] = recs + (size_t)b * L.total + L.fc4pre; }
            dense_pre_block(xs, nb, s.flat, P[6], P[7], a->fc4, ys);
            for (int b = 0; b < nb; b++) {
                float *rec = recs + (size_t)b * L.total;
                selu_inplace_copy(rec + L.fc4pre, rec + L.fc4, a->fc4);
                memcpy(rec + L.d4, rec + L.fc4, sizeof(float) * a->fc4);     /* inference: dropout = identity */
                xs[b] = rec + L.d4; ys[b] = rec + L.fc5pre;
            }
            dense_pre_block(xs, nb, a->fc4, P[8], P[9], a->fc5, ys);
            for (int b = 0; b < nb; b++) {
                float *rec = recs + (size_t)b * L.total;
                float *o = out16 + (size_t)(i0 + b) * CVO_NOUT;
                selu_inplace_copy(rec + L.fc5pre, rec + L.fc5, a->fc5);
                dense_pre(rec + L.d4, a->fc4, P[10], P[11], 4, rec + L.hpre[0]);
                for (int k = 0; k < 4; k++) o[k] = cvo_sigmoid(rec[L.hpre[0] + k]);
                for (int hd = 1; hd < 4; hd++) {
                    float lg[6];
                    dense_pre(rec + L.fc5, a->fc5, P[10 + 2 * hd], P[11 + 2 * hd], hn[hd], rec + L.hpre[hd]);
                    for (int k = 0; k < hn[hd]; k++) lg[k] = cvo_selu(rec[L.hpre[hd] + k]) + 1e-10f;
                    cvo_softmax(lg, hn[hd], o + ho[hd]);
                }
            }
        }
        free(recs);
    }
}

/* all intermediates: recs[n][record_size] */
void cvo_forward_all(const cvo_arch *a, const float *const *P, const float *x, int64_t n,
                     float *recs, const float *mask4, float rate4)
{
    cvo_layout L; cvo_make_layout(a, &L);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        forward_one(a, &L, P, x + (size_t)i * CVO_H * CVO_W * CVO_CIN, recs + (size_t)i * L.total,
                    mask4 ? mask4 + (size_t)i * a->fc4 : NULL, rate4);
}

/* ---- public: loss + gradients (training oracle) -------------------------- */

/* loss (v3.py:140-151), SUMS over the batch:
 *   sum (sigmoid - y[0:4])^2  +  sum -y*log_softmax(logits) for the 3 heads
 *   + lambda * sum_{non-bias kernels} sum(w^2)/2
 * losses[0..4] = loss1..loss4, lossL2 ; returns total.
 * grads: NULL or 18 buffers (same sizes as P) that RECEIVE d loss / d param.
 * mask4 / rate4: alpha-dropout keep mask on fc4 (NULL = phase False).        */
double cvo_loss_grad(const cvo_arch *a, const float *const *P, const float *x, const float *y,
                     int64_t n, float lambda, const float *mask4, float rate4,
                     double *losses, float *const *grads)
{
    cvo_layout L; cvo_make_layout(a, &L);
    cvo_shape s; cvo_shapes(a, &s);
    int64_t psz[CVO_NPARAM]; cvo_param_sizes(a, psz);
    static const int hn[4] = {4, 2, 4, 6};
    static const int ho[4] = {0, 4, 6, 10};
    const float ap = -1.7580993408473766f;
    float q = 1.0f - rate4;
    float aa = mask4 ? sqrtf(1.0f / (q * ((1.0f - q) * (ap * ap) + 1.0f))) : 1.0f;

    /* Candidates are independent: every thread walks ONE contiguous range of the batch with its own double
     * accumulators (losses and all 18 gradients); the ranges are added in thread order afterwards, so a result
     * depends on the thread count only in the last bits of a double (far below the float it is rounded to).   */
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
    if ((int64_t)nthreads > n) nthreads = n > 0 ? (int)n : 1;
#endif
    double (*lsum_t)[4] = (double (*)[4])calloc(nthreads, sizeof(double[4]));
    double **gacc_t = (double **)calloc((size_t)nthreads * CVO_NPARAM, sizeof(double *));
#pragma omp parallel num_threads(nthreads)
    {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    double *lsum = lsum_t[tid];
    double **gacc = gacc_t + (size_t)tid * CVO_NPARAM;
    for (int p = 0; p < CVO_NPARAM; p++) gacc[p] = grads ? (double *)calloc(psz[p], sizeof(double)) : NULL;
    float *rec = (float *)malloc(sizeof(float) * L.total);
    float *g = (float *)calloc(L.total, sizeof(float));   /* gradient record, same layout */
    const int64_t i_lo = n * tid / nthreads, i_hi = n * (tid + 1) / nthreads;
    for (int64_t i = i_lo; i < i_hi; i++) {
        const float *xi = x + (size_t)i * CVO_H * CVO_W * CVO_CIN;
        const float *yi = y + (size_t)i * CVO_NOUT;
        const float *mi = mask4 ? mask4 + (size_t)i * a->fc4 : NULL;
        forward_one(a, &L, P, xi, rec, mi, rate4);
        const float *o = rec + L.out;
        /* losses */
        for (int k = 0; k < 4; k++) { double d = (double)o[k] - yi[k]; lsum[0] += d * d; }
        for (int hd = 1; hd < 4; hd++) {
            float lg[6]; float m = -INFINITY;
            for (int k = 0; k < hn[hd]; k++) { lg[k] = cvo_selu(rec[L.hpre[hd] + k]) + 1e-10f; m = fmaxf(m, lg[k]); }
            double se = 0; for (int k = 0; k < hn[hd]; k++) se += exp((double)lg[k] - m);
            double lse = m + log(se);
            for (int k = 0; k < hn[hd]; k++) lsum[hd] += -(double)yi[ho[hd] + k] * ((double)lg[k] - lse);
        }
        if (!grads) continue;
        memset(g, 0, sizeof(float) * L.total);
        /* head pre-activation grads */
        for (int k = 0; k < 4; k++) {
            float sg = o[k];
            g[L.hpre[0] + k] = 2.0f * (sg - yi[k]) * sg * (1.0f - sg);
        }
        for (int hd = 1; hd < 4; hd++) {
            float ysum = 0; for (int k = 0; k < hn[hd]; k++) ysum += yi[ho[hd] + k];
            for (int k = 0; k < hn[hd]; k++) {
                float dl = o[ho[hd] + k] * ysum - yi[ho[hd] + k];
                g[L.hpre[hd] + k] = dl * cvo_selu_grad_from_pre(rec[L.hpre[hd] + k]);
            }
        }
        /* heads -> d(d4) and d(fc5) */
        for (int k = 0; k < a->fc4; k++) {
            float acc = 0;
            for (int j = 0; j < 4; j++) {
                acc += g[L.hpre[0] + j] * P[10][(size_t)k * 4 + j];
                gacc[10][(size_t)k * 4 + j] += (double)rec[L.d4 + k] * g[L.hpre[0] + j];
            }
            g[L.d4 + k] = acc;
        }
        for (int j = 0; j < 4; j++) gacc[11][j] += g[L.hpre[0] + j];
        for (int hd = 1; hd < 4; hd++) {
            int N = hn[hd];
            for (int k = 0; k < a->fc5; k++) {
                float acc = 0;
                for (int j = 0; j < N; j++) {
                    acc += g[L.hpre[hd] + j] * P[10 + 2 * hd][(size_t)k * N + j];
                    gacc[10 + 2 * hd][(size_t)k * N + j] += (double)rec[L.fc5 + k] * g[L.hpre[hd] + j];
                }
                g[L.fc5 + k] += acc;
            }
            for (int j = 0; j < N; j++) gacc[11 + 2 * hd][j] += g[L.hpre[hd] + j];
        }
        /* fc5 */
        for (int j = 0; j < a->fc5; j++) g[L.fc5pre + j] = g[L.fc5 + j] * cvo_selu_grad_from_pre(rec[L.fc5pre + j]);
        for (int k = 0; k < a->fc4; k++) {
            float acc = 0;
            const float *wr = P[8] + (size_t)k * a->fc5;
            double *gr = gacc[8] + (size_t)k * a->fc5;
            float dk = rec[L.d4 + k];
            for (int j = 0; j < a->fc5; j++) { acc += g[L.fc5pre + j] * wr[j]; gr[j] += (double)dk * g[L.fc5pre + j]; }
            g[L.d4 + k] += acc;
        }
        for (int j = 0; j < a->fc5; j++) gacc[9][j] += g[L.fc5pre + j];
        /* dropout4 backward: d(fc4) = a * m * d(d4) */
        for (int k = 0; k < a->fc4; k++) {
            float dd = g[L.d4 + k];
            if (mi) dd = aa * mi[k] * dd;
            g[L.fc4pre + k] = dd * cvo_selu_grad_from_pre(rec[L.fc4pre + k]);
        }
        /* fc4 */
        {
            const float *fin = rec + L.pooled[2];
            float *gin = g + L.pooled[2];
            for (int k = 0; k < s.flat; k++) {
                float acc = 0;
                const float *wr = P[6] + (size_t)k * a->fc4;
                double *gr = gacc[6] + (size_t)k * a->fc4;
                float fk = fin[k];
                for (int j = 0; j < a->fc4; j++) { acc += g[L.fc4pre + j] * wr[j]; gr[j] += (double)fk * g[L.fc4pre + j]; }
                gin[k] = acc;
            }
            for (int j = 0; j < a->fc4; j++) gacc[7][j] += g[L.fc4pre + j];
        }
        /* conv stack backward */
        for (int l = 2; l >= 0; l--) {
            int H = s.hc[l], C = a->cout[l], cin = s.cin[l], kh = a->kh[l], p = a->pool[l];
            int row = CVO_W * C, Ho = s.hp[l];
            /* pool backward: route to the first maximum in the window */
            for (int h = 0; h < Ho; h++)
                for (int e = 0; e < row; e++) {
                    int best = 0; float m = rec[L.act[l] + (size_t)h * row + e];
                    for (int d = 1; d < p; d++) {
                        float v = rec[L.act[l] + (size_t)(h + d) * row + e];
                        if (v > m) { m = v; best = d; }
                    }
                    g[L.act[l] + (size_t)(h + best) * row + e] += g[L.pooled[l] + (size_t)h * row + e];
                }
            for (int e = 0; e < H * row; e++)
                g[L.pre[l] + e] = g[L.act[l] + e] * cvo_selu_grad_from_pre(rec[L.pre[l] + e]);
            const float *in = (l == 0) ? xi : rec + L.pooled[l - 1];
            float *gin = (l == 0) ? NULL : g + L.pooled[l - 1];
            int padt = (kh - 1) / 2;
            for (int h = 0; h < H; h++)
                for (int w = 0; w < CVO_W; w++) {
                    const float *go = g + L.pre[l] + ((size_t)h * CVO_W + w) * C;
                    for (int co = 0; co < C; co++) gacc[2 * l + 1][co] += go[co];
                    for (int ka = 0; ka < kh; ka++) {
                        int hi = h + ka - padt; if (hi < 0 || hi >= H) continue;
                        for (int kb = 0; kb < CVO_KW; kb++) {
                            int wi = w + kb - 1; if (wi < 0 || wi >= CVO_W) continue;
                            const float *xr = in + ((size_t)hi * CVO_W + wi) * cin;
                            for (int ci = 0; ci < cin; ci++) {
                                size_t wo = ((size_t)(ka * CVO_KW + kb) * cin + ci) * C;
                                const float *wc = P[2 * l] + wo;
                                double *gw = gacc[2 * l] + wo;
                                float acc = 0, xv = xr[ci];
                                for (int co = 0; co < C; co++) { acc += go[co] * wc[co]; gw[co] += (double)xv * go[co]; }
                                if (gin) gin[((size_t)hi * CVO_W + wi) * cin + ci] += acc;
                            }
                        }
                    }
                }
        }
    }
    free(rec); free(g);
    }   /* omp parallel */
    double lsum[4] = {0, 0, 0, 0};
    double **gacc = gacc_t;                      /* thread 0's buffers receive the sum, threads ascending */
    for (int t = 0; t < nthreads; t++)
        for (int k = 0; k < 4; k++) lsum[k] += lsum_t[t][k];
    if (grads)
        for (int p = 0; p < CVO_NPARAM; p++) {
            double *dst = gacc[p];
#pragma omp parallel for schedule(static)
            for (int64_t k = 0; k < psz[p]; k++) {
                double v = dst[k];
                for (int t = 1; t < nthreads; t++) v += gacc_t[(size_t)t * CVO_NPARAM + p][k];
                dst[k] = v;
            }
            for (int t = 1; t < nthreads; t++) free(gacc_t[(size_t)t * CVO_NPARAM + p]);
        }
    /* L2 (v3.py:150): lambda * sum over non-bias variables of sum(w^2)/2 */
    double l2 = 0;
    for (int p = 0; p < CVO_NPARAM; p += 2) {
        double sq = 0;
        for (int64_t k = 0; k < psz[p]; k++) sq += (double)P[p][k] * P[p][k];
        l2 += sq / 2.0;
    }
    l2 *= lambda;
    if (grads) {
        for (int p = 0; p < CVO_NPARAM; p++) {
            for (int64_t k = 0; k < psz[p]; k++) {
                double v = gacc[p][k];
                if ((p & 1) == 0) v += (double)lambda * P[p][k];
                grads[p][k] = (float)v;
            }
            free(gacc[p]);
        }
    }
    if (losses) { for (int k = 0; k < 4; k++) losses[k] = lsum[k]; losses[4] = l2; }
    double total = lsum[0] + lsum[1] + lsum[2] + lsum[3] + l2;
    free(lsum_t); free(gacc_t);
    return total;
}

/* TF1 AdamOptimizer step (beta1 .9, beta2 .999, eps 1e-8, "epsilon hat" form):
 *   lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 *   w -= lr_t * m / (sqrt(v) + eps)                                           */
void cvo_adam_step(float *w, float *m, float *v, const float *g, int64_t n, float lr, int t)
{
    const double b1 = 0.9, b2 = 0.999;
    float lr_t = (float)(lr * sqrt(1.0 - pow(b2, t)) / (1.0 - pow(b1, t)));
    for (int64_t i = 0; i < n; i++) {
        m[i] = 0.9f * m[i] + (1.0f - 0.9f) * g[i];
        v[i] = 0.999f * v[i] + (1.0f - 0.999f) * (g[i] * g[i]);
        w[i] = w[i] - lr_t * m[i] / (sqrtf(v[i]) + 1e-8f);
    }
}

/* Sweep of cvo_selu over the bit patterns [lo, hi] taken as negative floats (ascending pattern = descending value):
 * returns the number of adjacent pairs whose output INCREASES (0 over 0x80000000..0xff800000 = SELU is monotone
 * on the whole negative axis) and the wrapping sum of the output patterns of all but the first input -- the same
 * quantities the device reports through cv_selu_sweep. */
uint64_t cvo_selu_sweep(uint32_t lo, uint32_t hi, uint64_t *checksum)
{
    uint64_t viol = 0, chk = 0;
    const uint64_t chunk = 1u << 20;
#pragma omp parallel for schedule(dynamic) reduction(+ : viol, chk)
    for (uint64_t c = lo; c < (uint64_t)hi; c += chunk) {
        const uint64_t end = c + chunk < (uint64_t)hi ? c + chunk : (uint64_t)hi;
        float prev = cvo_selu(as_float((uint32_t)c));
        for (uint64_t u = c + 1; u <= end; u++) {
            const float cur = cvo_selu(as_float((uint32_t)u));
            uint32_t b; memcpy(&b, &cur, 4);
            viol += cur > prev;
            chk += b;
            prev = cur;
        }
    }
    if (checksum) *checksum = chk;
    return viol;
}
