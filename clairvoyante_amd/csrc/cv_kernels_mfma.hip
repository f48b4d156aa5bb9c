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
#include "cv_internal.hpp"
#include "cv_math.hpp"

typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f4 selu4(f4 v)
{
    f4 r;
    r[0] = cvm::selu(v[0]); r[1] = cvm::selu(v[1]); r[2] = cvm::selu(v[2]); r[3] = cvm::selu(v[3]);
    return r;
}

__device__ __forceinline__ f4 max4(f4 a, f4 b)
{
    f4 r;
    r[0] = fmaxf(a[0], b[0]); r[1] = fmaxf(a[1], b[1]); r[2] = fmaxf(a[2], b[2]); r[3] = fmaxf(a[3], b[3]);
    return r;
}

// lane (c = lane&15, q = lane>>4) register r of a D tile holds output feature
// 16*ob + 4*r + q  (sigma-permuted weight rows): its bias
__device__ __forceinline__ f4 load_bias4(const float *__restrict__ bias, int ob, int q, int nout)
{
    f4 b;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int f = 16 * ob + 4 * r + q;
        b[r] = f < nout ? bias[f] : 0.0f;
    }
    return b;
}

// ---------------------------------------------------------------------------
// weight packing (runs once per parameter change)
// ---------------------------------------------------------------------------
__global__ void pack_conv1(const float *__restrict__ w, float *__restrict__ wp, int cout)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;   // [kw][lane]
    if (t >= 4 * 64) return;
    int lane = t & 63, kw = t >> 6;
    int i = lane & 15, kq = lane >> 4;
    int co = cv_sigma(i);
    wp[t] = co < cout ? w[((size_t)kw * 4 + kq) * cout + co] : 0.0f;
}

__global__ void pack_conv(const float *__restrict__ w, float *__restrict__ wp, int KH, int cin, int cout,
                          int CINB, int NT)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
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

// dense [K][N] -> [kb][ob][lane][s];  input feature k = 16*kb + 4*s + kq
__global__ void pack_dense(const float *__restrict__ w, float *__restrict__ wp, int K, int N, int KB, int NB)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)KB * NB * 256;
    if (t >= total) return;
    int s = (int)(t & 3), lane = (int)((t >> 2) & 63);
    int64_t frag = t >> 8;
    int ob = (int)(frag % NB);
    int kb = (int)(frag / NB);
    int i = lane & 15, kq = lane >> 4;
    int k = 16 * kb + 4 * s + kq, o = 16 * ob + cv_sigma(i);
    wp[t] = (k < K && o < N) ? w[(size_t)k * N + o] : 0.0f;
}

// ---------------------------------------------------------------------------
// conv1 (k(1,4), cin 4) + SELU + max-pool(POOL,1): raw X [n,33,4,4] -> TM
// One wave per group of 16 candidates; per position 12 MFMA steps (K = 4 each).
// ---------------------------------------------------------------------------
template <int POOL>
__global__ __launch_bounds__(256) void conv1_tm(const float *__restrict__ x, int64_t n,
                                                 const float *__restrict__ wp1,
                                                 const float *__restrict__ bias, int cout,
                                                 f4 *__restrict__ out_tm, int G)
{
    constexpr int HIN = CV_INPUT_H, HOUT = HIN - POOL + 1;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
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
    for (int w = 0; w < 4; w++) xw[w] = xp[w * 4];
    f4 *op = out_tm + (size_t)g * HOUT * 4 * 64 + lane;
#pragma unroll 1
    for (int h = 0; h < HIN; h++) {
        if (h + 1 < HIN) {
#pragma unroll
            for (int w = 0; w < 4; w++) xn[w] = xp[(h + 1) * 16 + w * 4];
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
#pragma unroll
            for (int j = 0; j + 1 < POOL - 1; j++)
#pragma unroll
                for (int w = 0; w < 4; w++) pw[j][w] = pw[j + 1][w];
#pragma unroll
            for (int w = 0; w < 4; w++) pw[POOL - 2][w] = v[w];
            if (h >= POOL - 1) {
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
// generic conv (k(KH,4), CINB*16 -> NT*16 channels) + SELU + max-pool(POOL,1),
// TM -> TM.  One wave per (group, output tile nt); weights of the whole layer
// sit in LDS (loaded once per workgroup).  Per position and wave:
// KH*4*CINB ds_read_b128 feed KH*12*CINB*4 MFMA steps.
// ---------------------------------------------------------------------------
template <int KH, int CINB, int NT, int POOL, int HIN>
__global__ __launch_bounds__(256) void conv_tm(const f4 *__restrict__ in_tm, const f4 *__restrict__ wp,
                                                const float *__restrict__ bias, int cout,
                                                f4 *__restrict__ out_tm, int G)
{
    extern __shared__ __attribute__((aligned(16))) f4 ldsw[];
    constexpr int PADT = (KH - 1) / 2;
    constexpr int HOUT = HIN - POOL + 1;
    constexpr int NFRAG = NT * KH * 4 * CINB;
    for (int i = threadIdx.x; i < NFRAG * 64; i += 256) ldsw[i] = wp[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = wv / NT, nt = wv % NT;
    if (g >= G) return;
    const int q = lane >> 4;
    const f4 b4 = load_bias4(bias, nt, q, cout);
    const f4 *inp = in_tm + (size_t)g * (HIN * 4 * CINB * 64) + lane;
    const f4 *wl = ldsw + (size_t)nt * (KH * 4 * CINB * 64) + lane;
    f4 *op = out_tm + (size_t)g * (HOUT * 4 * NT * 64) + (size_t)nt * 64 + lane;

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
        const int hr = j - PADT;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) {
                f4 v = zero;
                if (hr >= 0 && hr < HIN) v = inp[(size_t)((hr * 4 + w) * CINB + cb) * 64];
                if (j < KH - 1) win[j][w][cb] = v; else nxt[w][cb] = v;
            }
    }
#pragma unroll 1
    for (int h = 0; h < HIN; h++) {
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) win[KH - 1][w][cb] = nxt[w][cb];
        {   // prefetch the row the next position needs
            const int hr = h + 1 + (KH - 1) - PADT;
            if (hr < HIN) {
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int cb = 0; cb < CINB; cb++) nxt[w][cb] = inp[(size_t)((hr * 4 + w) * CINB + cb) * 64];
            }
        }
        f4 acc[4];
#pragma unroll
        for (int w = 0; w < 4; w++) acc[w] = zero;
#pragma unroll
        for (int kh = 0; kh < KH; kh++) {
            const int hr = h + kh - PADT;
            if (hr >= 0 && hr < HIN) {     // wave-uniform; SAME padding rows are skipped
#pragma unroll
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
                                acc[wo] = mfma4(A[s], win[kh][wi][cb][s], acc[wo]);
                            }
                    }
            }
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
#pragma unroll
            for (int j = 0; j + 1 < POOL - 1; j++)
#pragma unroll
                for (int w = 0; w < 4; w++) pw[j][w] = pw[j + 1][w];
#pragma unroll
            for (int w = 0; w < 4; w++) pw[POOL - 2][w] = v[w];
            if (h >= POOL - 1) {
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
}

// ---------------------------------------------------------------------------
// dense (KB*16 -> NB*16) + bias + SELU, TM -> TM.  One wave per group of 16
// candidates holds all NB accumulator tiles; the workgroup streams the packed
// weight matrix through a 3-stage LDS ring (one barrier per 16-deep k step),
// each wave streams its own activation fragments straight from HBM/L2.
// ---------------------------------------------------------------------------
template <int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dense_tm(const f4 *__restrict__ in_tm, int KB,
                                                        const f4 *__restrict__ wp,
                                                        const float *__restrict__ bias, int nout,
                                                        f4 *__restrict__ out_tm, int G)
{
    extern __shared__ __attribute__((aligned(16))) f4 ring[];
    constexpr int T = WAVES * 64;
    constexpr int STAGE = NB * 64;               // f4 per stage
    constexpr int PER = (STAGE + T - 1) / T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = blockIdx.x * WAVES + (tid >> 6);
    const int gl = g < G ? g : G - 1;
    const f4 *bp = in_tm + (size_t)gl * KB * 64 + lane;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[NB];
#pragma unroll
    for (int ob = 0; ob < NB; ob++) acc[ob] = zero;
    f4 st[PER];
    auto load_stage = [&](int kb) {
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int idx = tid + p * T;
            if (idx < STAGE) st[p] = wp[(size_t)kb * STAGE + idx];
        }
    };
    auto write_stage = [&](int slot) {
#pragma unroll
        for (int p = 0; p < PER; p++) {
            const int idx = tid + p * T;
            if (idx < STAGE) ring[slot * STAGE + idx] = st[p];
        }
    };
    load_stage(0);
    write_stage(0);
    if (KB > 1) { load_stage(1); write_stage(1); }
    __syncthreads();
    f4 B = bp[0];
    int slot = 0;
#pragma unroll 1
    for (int kb = 0; kb < KB; kb++) {
        if (kb + 2 < KB) load_stage(kb + 2);
        f4 Bn = zero;
        if (kb + 1 < KB) Bn = bp[(size_t)(kb + 1) * 64];
        const f4 *wl = ring + slot * STAGE + lane;
#pragma unroll
        for (int ob = 0; ob < NB; ob += 3) {
            f4 A[3];
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (ob + j < NB) A[j] = wl[(ob + j) * 64];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (ob + j < NB) acc[ob + j] = mfma4(A[j][s], B[s], acc[ob + j]);
        }
        int wslot = slot + 2; if (wslot >= 3) wslot -= 3;
        if (kb + 2 < KB) write_stage(wslot);
        __syncthreads();
        B = Bn;
        slot = slot + 1 == 3 ? 0 : slot + 1;
    }
    if (g >= G) return;
    const int q = lane >> 4;
    f4 *op = out_tm + (size_t)g * NB * 64 + lane;
#pragma unroll
    for (int ob = 0; ob < NB; ob++) {
        const f4 b4 = load_bias4(bias, ob, q, nout);
        op[ob * 64] = selu4(acc[ob] + b4);
    }
}

inline unsigned nblk(int64_t total, int bs) { return (unsigned)((total + bs - 1) / bs); }

template <typename K>
int set_lds(K kernel, size_t bytes)
{
    CV_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

template <int KH, int CINB, int NT, int POOL, int HIN>
int launch_conv(const float *in, const float *wp, const float *bias, int cout, float *out, int G,
                hipStream_t st)
{
    auto k = conv_tm<KH, CINB, NT, POOL, HIN>;
    size_t lds = (size_t)NT * KH * 4 * CINB * 1024;
    if (set_lds(k, lds)) return 1;
    unsigned grid = nblk((int64_t)G * NT, 4);
    k<<<grid, 256, lds, st>>>((const f4 *)in, (const f4 *)wp, bias, cout, (f4 *)out, G);
    CV_HIP(hipGetLastError());
    return 0;
}

template <int NB>
int launch_dense(const float *in, int KB, const float *wp, const float *bias, int nout, float *out, int G,
                 hipStream_t st)
{
    constexpr int WAVES = 4;
    auto k = dense_tm<NB, WAVES>;
    size_t lds = (size_t)3 * NB * 1024;
    if (set_lds(k, lds)) return 1;
    k<<<nblk(G, WAVES), WAVES * 64, lds, st>>>((const f4 *)in, KB, (const f4 *)wp, bias, nout, (f4 *)out, G);
    CV_HIP(hipGetLastError());
    return 0;
}

bool arch_is(const cv_arch &a, int k0, int k1, int k2, int c0, int c1, int c2, int p0, int p1, int p2,
             int f4_, int f5_)
{
    return a.kh[0] == k0 && a.kh[1] == k1 && a.kh[2] == k2 && a.cout[0] == c0 && a.cout[1] == c1 &&
           a.cout[2] == c2 && a.pool[0] == p0 && a.pool[1] == p1 && a.pool[2] == p2 && a.fc4 == f4_ &&
           a.fc5 == f5_;
}

}  // namespace

int cv_pack_weights(cv_model *m, hipStream_t st)
{
    const float *P = m->params;
    const int64_t *o = m->poff;
    const cv_shapes &s = m->sh;
    pack_conv1<<<1, 256, 0, st>>>(P + o[0], m->wp_conv1, m->arch.cout[0]);
    for (int l = 1; l < 3; l++) {
        int64_t tot = (int64_t)s.ntile[l] * m->arch.kh[l] * 4 * s.cinb[l] * 256;
        pack_conv<<<nblk(tot, 256), 256, 0, st>>>(P + o[2 * l], m->wp_conv[l], m->arch.kh[l], s.cin[l],
                                                  m->arch.cout[l], s.cinb[l], s.ntile[l]);
    }
    {
        int64_t tot = (int64_t)s.kb4 * s.nb4 * 256;
        pack_dense<<<nblk(tot, 256), 256, 0, st>>>(P + o[6], m->wp_fc4, s.flat, m->arch.fc4, s.kb4, s.nb4);
        tot = (int64_t)s.nb4 * s.nb5 * 256;
        pack_dense<<<nblk(tot, 256), 256, 0, st>>>(P + o[8], m->wp_fc5, m->arch.fc4, m->arch.fc5, s.nb4, s.nb5);
    }
    CV_HIP(hipGetLastError());
    m->packed_dirty = false;
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

// one chunk (n <= chunk) through the tile kernels
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
    if (m->packed_dirty && cv_pack_weights(m, st)) return 1;
    const float *P = m->params;
    const int64_t *o = m->poff;
    const int G = (int)((n + 15) / 16);
    const cv_shapes &s = m->sh;
    int rc = 0;
    if (full) {
        cv_prof_begin(m, 0, st);
        conv1_tm<5><<<nblk(G, 4), 256, 0, st>>>(x, n, m->wp_conv1, P + o[1], a.cout[0], (f4 *)m->tm_p1, G);
        cv_prof_end(m, 0, st);
        cv_prof_begin(m, 1, st);
        rc |= launch_conv<2, 1, 2, 4, 29>(m->tm_p1, m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
        cv_prof_end(m, 1, st);
        cv_prof_begin(m, 2, st);
        rc |= launch_conv<3, 2, 3, 3, 26>(m->tm_p2, m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st);
        cv_prof_end(m, 2, st);
        cv_prof_begin(m, 3, st);
        rc |= launch_dense<21>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st);
        cv_prof_end(m, 3, st);
        cv_prof_begin(m, 4, st);
        rc |= launch_dense<11>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st);
        cv_prof_end(m, 4, st);
    } else {
        cv_prof_begin(m, 0, st);
        conv1_tm<1><<<nblk(G, 4), 256, 0, st>>>(x, n, m->wp_conv1, P + o[1], a.cout[0], (f4 *)m->tm_p1, G);
        cv_prof_end(m, 0, st);
        cv_prof_begin(m, 1, st);
        rc |= launch_conv<3, 1, 1, 1, 33>(m->tm_p1, m->wp_conv[1], P + o[3], a.cout[1], m->tm_p2, G, st);
        cv_prof_end(m, 1, st);
        cv_prof_begin(m, 2, st);
        rc |= launch_conv<5, 1, 2, 1, 33>(m->tm_p2, m->wp_conv[2], P + o[5], a.cout[2], m->tm_p3, G, st);
        cv_prof_end(m, 2, st);
        cv_prof_begin(m, 3, st);
        rc |= launch_dense<3>(m->tm_p3, s.kb4, m->wp_fc4, P + o[7], a.fc4, m->tm_h4, G, st);
        cv_prof_end(m, 3, st);
        cv_prof_begin(m, 4, st);
        rc |= launch_dense<2>(m->tm_h4, s.nb4, m->wp_fc5, P + o[9], a.fc5, m->tm_h5, G, st);
        cv_prof_end(m, 4, st);
    }
    if (rc) return 1;
    CV_HIP(hipGetLastError());
    m->last_n = n;
    m->last_impl = 1;
    cv_prof_begin(m, 5, st);
    rc = cv_launch_heads(m, m->tm_h4, m->tm_h5, 1, n, out16, st);
    cv_prof_end(m, 5, st);
    return rc;
}
