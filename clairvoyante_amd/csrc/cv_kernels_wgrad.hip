// cv_kernels_wgrad.hip -- gfx950 weight-gradient kernels of the training step and their launch sites (split from
// cv_kernels_mfma.hip in round 6; clairvoyante_v3.py:174 minimises the loss of v3.py:140-151 over the 18 variables).
#include "cv_tile.hpp"

// ---------------------------------------------------------------------------
// weight gradients: contraction over CANDIDATES on the matrix cores.
//
// A TM fragment has the candidate as the tile column.  For dW = X^T . G both operands need
// the candidate as the MFMA k index instead, i.e. the 16x16 transpose of the fragment
// ("CM" fragment: lane (f, rg) register t = value(feature f, candidate 4*rg + t)).  CM fragments of the
// layer input (A operand) and of the pre-activation gradient (B operand) then give
// dW[i][j] += sum_c X[c][i] G[c][j]  with four MFMA steps per 16 candidates.
//
// The transpose rides on the global -> LDS DMA that brings the fragment in (`cm_stage`): every lane of the
// DMA instruction fetches the 16 bytes of ANOTHER lane of the fragment (position p of the LDS slot receives
// source lane 16 (p & 3) + 4 ((p >> 2) & 3) + (p >> 4)), which places the four values a CM lane needs 64
// dwords apart and all 64 lanes of one such read in 64 different banks.  No candidate-major copy of any
// tensor exists in HBM, the matrix pipe does none of the data movement, and the DMA of the next step runs
// under the MFMAs of the current one.  Completion of a DMA is the explicit vmcnt wait; a slot is re-filled
// only after the reads of its previous contents have returned (lgkmcnt wait).
// ---------------------------------------------------------------------------
struct cm_stage {
    float *slots;           // LDS, 256 floats per fragment slot
    unsigned base;          // LDS byte address of slots
    int src_lane;           // fragment lane whose 16 bytes this lane's DMA piece fetches
    int ridx;               // first dword this lane reads of a slot
    __device__ __forceinline__ cm_stage(float *lds, int lane)
        : slots(lds), base((unsigned)(size_t)(__attribute__((address_space(3))) float *)lds),
          src_lane(16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (lane >> 4)),
          ridx(16 * (lane >> 4) + 4 * (lane & 3) + ((lane & 15) >> 2)) {}
    // frag: first f4 of a TM fragment (wave-uniform); slot: wave-uniform
    __device__ __forceinline__ void fetch(const f4 *frag, int slot) const
    {
        const f4 *gp = frag + src_lane;
        const unsigned ldst = __builtin_amdgcn_readfirstlane(base + (unsigned)slot * 1024u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
    }
    // the same with the fragment address as a SCALAR base (the caller's pointer must be provably wave-uniform) + this
    // lane's byte offset: no 64-bit vector address per piece
    __device__ __forceinline__ void fetch_s(const f4 *frag, int slot) const
    {
        const unsigned off = (unsigned)src_lane * 16u;
        const unsigned ldst = __builtin_amdgcn_readfirstlane(base + (unsigned)slot * 1024u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(off), "s"(ldst), "s"(frag) : "memory");
    }
    __device__ __forceinline__ f4 read(int slot) const
    {
        const float *q = slots + slot * 256 + ridx;
        return (f4){q[0], q[64], q[128], q[192]};
    }
    // the same for a fragment that lies in NATURAL order in memory -- 16 candidates x 16 consecutive floats,
    // candidate stride `cstride` floats: the lane feeding LDS position p fetches quarter (p & 3) of candidate
    // 4 ((p >> 2) & 3) + (p >> 4); read_nat then finds value(f, 4 rg + t) at dword 64 t + lane.
    // cand0 = first candidate of the group; candidates >= n are clamped to n - 1 (their gradients are zero).
    __device__ __forceinline__ void fetch_nat(const float *base, int64_t cand0, int64_t n, size_t cstride, int slot,
                                              int lane) const
    {
        int64_t cand = cand0 + 4 * ((lane >> 2) & 3) + (lane >> 4);
        if (cand >= n) cand = n - 1;
        const float *gp = base + (size_t)cand * cstride + 4 * (lane & 3);
        const unsigned ldst = __builtin_amdgcn_readfirstlane(this->base + (unsigned)slot * 1024u);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gp), "s"(ldst) : "memory");
    }
    __device__ __forceinline__ f4 read_nat(int slot, int lane) const
    {
        const float *q = slots + slot * 256 + lane;
        return (f4){q[0], q[64], q[128], q[192]};
    }
    template <int N> static __device__ __forceinline__ void landed()        // all but the newest N pieces
    {
        static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
    }
    static __device__ __forceinline__ void reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }
};

// dense layer: dW[k][j] += sum_cand X[cand][k] G[cand][j], db[j] += sum_cand G[cand][j].
// Workgroup = 8 waves = 8 XF input fragments (XF per wave) x all NJB output fragments.  Per group the
// workgroup stages the NJB gradient fragments (shared) and every wave its XF input fragments, double
// buffered: the pieces of group g+1 are in flight while group g is multiplied; one barrier per group.
// grid = (ceil(KB / (8 XF)), group splits); dynamic LDS = 2 * (NJB + 8 XF) KiB.
// XF = 1 (few groups): twice the workgroups along k, so half the candidate ranges fill the same CUs -- and the
// per-range tiles, which the second pass has to read back, are half as many bytes.
// GNAT (heads, NJB == 1): the gradient is the natural [n][16] array of the 16 head pre-activation gradients.
template <int NJB, bool GNAT = false, int XF = 2>
__global__ __launch_bounds__(512) void wgrad_dense_cm(const f4 *__restrict__ x_tm, int KB,
                                                       const f4 *__restrict__ g_tm, int G, int64_t n,
                                                       f4 *__restrict__ part)
{
    static_assert(!GNAT || NJB == 1, "natural gradients: one fragment per group");
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    constexpr int NSLOT = NJB + 8 * XF;              // per buffer: NJB gradient fragments, then XF per wave
    constexpr int PERG = (NJB + 7) / 8;              // gradient fragments each wave fetches (clamped: duplicates
                                                     // of the last one land on identical bytes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    CV_STAMP_BEGIN
    const cm_stage S(wg_lds, lane);
    const int kb0 = blockIdx.x * (8 * XF) + wid * XF;
    const int per = (G + gridDim.y - 1) / gridDim.y;
    const int g0 = blockIdx.y * per, g1 = g0 + per < G ? g0 + per : G;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[XF][NJB];
#pragma unroll
    for (int a = 0; a < XF; a++)
#pragma unroll
        for (int jb = 0; jb < NJB; jb++) acc[a][jb] = zero;
    const bool v0 = kb0 < KB, v1 = XF > 1 && kb0 + 1 < KB;
    const int kc0 = v0 ? kb0 : KB - 1, kc1 = v1 ? kb0 + 1 : KB - 1;       // clamped: fetched, never multiplied
    constexpr int NBS = (NJB + 7) / 8;               // bias: wave w of column 0 sums fragments w, w+8, ..
    f4 bs[NBS];
#pragma unroll
    for (int i = 0; i < NBS; i++) bs[i] = zero;
    const bool do_bias = blockIdx.x == 0;
    // this wave's DMA pieces of group g: i < PERG gradient fragments, then its two input fragments
    constexpr int NP = PERG + XF;
    auto piece = [&](int g, int buf, int i) {
        if (i < PERG) {
            if constexpr (GNAT) {
                if (wid == 0) S.fetch_nat((const float *)g_tm, (int64_t)g * 16, n, 16, buf * NSLOT, lane);
            } else {
                const int jb = wid + 8 * i < NJB ? wid + 8 * i : NJB - 1;
                S.fetch_s(g_tm + ((size_t)g * NJB + jb) * 64, buf * NSLOT + jb);
            }
        } else {
            const int a = i - PERG;
            S.fetch_s(x_tm + ((size_t)g * KB + (a ? kc1 : kc0)) * 64, buf * NSLOT + NJB + XF * wid + a);
        }
    };
    // In the loop the pieces of group g+1 go out one at a time between the multiplications of group g (every
    // PSTEP-th gradient fragment): as a burst behind the barrier all 8 waves queue on the CU's vector-memory port at
    // once and none of them multiplies meanwhile (see wgrad_conv_cm).
    constexpr bool SPREAD = NJB >= NP;
    constexpr int PSTEP = SPREAD ? NJB / NP : 1;       // (every block instead, so that all pieces are out early: no difference)
    if (g0 < g1) {
#pragma unroll
        for (int i = 0; i < NP; i++) piece(g0, 0, i);
    }
    int buf = 0;
#pragma unroll 1
    for (int g = g0; g < g1; g++) {
        cm_stage::landed<0>();
        __syncthreads();                     // group g is in LDS for everyone; buffer buf^1 is no longer read
        // (past the last group the pieces re-fetch it into the idle buffer: no branch around every piece -- with
        // branches the loop body falls into blocks and hipcc spills accumulators across them)
        const int gn = g + 1 < g1 ? g + 1 : g;
        if (!SPREAD) {
#pragma unroll
            for (int i = 0; i < NP; i++) piece(gn, buf ^ 1, i);
        }
        f4 X0 = S.read(buf * NSLOT + NJB + XF * wid), X1 = zero;
        if constexpr (XF > 1) X1 = S.read(buf * NSLOT + NJB + XF * wid + 1);
        if (!v0) X0 = zero;
        if (!v1) X1 = zero;
        f4 Bnext = zero;                     // gradient fragment jb + 1, read while fragment jb is multiplied
        if constexpr (!GNAT) Bnext = S.read(buf * NSLOT);
#pragma unroll
        for (int jb = 0; jb < NJB; jb++) {
            f4 B;
            if constexpr (GNAT) {
                B = S.read_nat(buf * NSLOT, lane);
                if (g == G - 1) {            // candidates past n were fetched clamped: they carry no gradient
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        if ((int64_t)g * 16 + 4 * (lane >> 4) + t >= n) B[t] = 0.0f;
                }
            } else {
                B = Bnext;
                if (jb + 1 < NJB) Bnext = S.read(buf * NSLOT + jb + 1);
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                acc[0][jb] = mfma4(X0[t], B[t], acc[0][jb]);
                if constexpr (XF > 1) acc[1][jb] = mfma4(X1[t], B[t], acc[1][jb]);
            }
            if (do_bias && (jb & 7) == wid) bs[jb >> 3] += B;
            if (SPREAD && jb % PSTEP == 0 && jb / PSTEP < NP) piece(gn, buf ^ 1, jb / PSTEP);
        }
        cm_stage::reads_done();
        buf ^= 1;
    }
    cm_stage::landed<0>();                   // the surplus pieces of the last group
    // this split's tiles as whole fragments, then its bias sums: combined by wgrad_dense_reduce
    if (do_bias) {
        // CM fragment: lane (f, rg) register t = G(feature f, candidate 4 rg + t): sum registers, then lanes rg
        float *bpart = (float *)(part + (size_t)gridDim.y * KB * NJB * 64) + (size_t)blockIdx.y * NJB * 16;
#pragma unroll
        for (int i = 0; i < NBS; i++) {
            const int jb = i * 8 + wid;
            float v = (bs[i][0] + bs[i][1]) + (bs[i][2] + bs[i][3]);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (jb < NJB && lane < 16) bpart[16 * jb + lane] = v;
        }
    }
#pragma unroll
    for (int a = 0; a < XF; a++) {
        const int kb = kb0 + a;
        if (kb >= KB) continue;
        f4 *pp = part + (((size_t)blockIdx.y * KB + kb) * NJB) * 64 + lane;
#pragma unroll
        for (int jb = 0; jb < NJB; jb++) pp[jb * 64] = acc[a][jb];
    }
    CV_STAMP_END(NJB == 21, 5);
}

// second pass of the dense weight gradient: dW += sum over splits (ascending: a fixed summation order),
// one thread per (kb, jb, lane) fragment element quadruple; the threads behind those sum the bias parts
// acc (all four second passes): 1 = add to what the gradient holds (a later slice of the step), 0 = the first slice:
// 0 + sum is STORED -- the bits a zeroed buffer would end up with, without the 6.5 MB memset at the head of every step.
__global__ void wgrad_dense_reduce(const f4 *__restrict__ part, int splits, int KB, int NJB, int K, int N,
                                   float *__restrict__ dw, float *__restrict__ db, int acc)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)KB * NJB * 64;
    if (t >= per) {
        const int j = (int)(t - per);
        if (j >= N || j >= NJB * 16) return;
        const float *bpart = (const float *)(part + (size_t)splits * per);
        float b = bpart[j];
        for (int sidx = 1; sidx < splits; sidx++) b += bpart[(size_t)sidx * NJB * 16 + j];
        db[j] = (acc ? db[j] : 0.0f) + b;
        return;
    }
    f4 v = part[t];
    int sidx = 1;
    for (; sidx + 8 <= splits; sidx += 8) {      // eight loads in flight, added in split order
        f4 w[8];
#pragma unroll
        for (int u = 0; u < 8; u++) w[u] = part[(size_t)(sidx + u) * per + t];
#pragma unroll
        for (int u = 0; u < 8; u++) v += w[u];
    }
    for (; sidx + 4 <= splits; sidx += 4) {
        f4 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = part[(size_t)(sidx + u) * per + t];
#pragma unroll
        for (int u = 0; u < 4; u++) v += w[u];
    }
    for (; sidx < splits; sidx++) v += part[(size_t)sidx * per + t];
    const int lane = (int)(t & 63);
    const int64_t frag = t >> 6;
    const int jb = (int)(frag % NJB), kb = (int)(frag / NJB);
    const int j = 16 * jb + (lane & 15), q = lane >> 4;
    if (j >= N) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int k = 16 * kb + 4 * q + r;
        if (k < K) dw[(size_t)k * N + j] = (acc ? dw[(size_t)k * N + j] : 0.0f) + v[r];
    }
}

// conv layer: dW[kh][kw][ci][co] += sum_{cand,h,wo} In[cand][h+kh-PT][wo+kw-1][ci] G[cand][h][wo][co].
// One wave per (output fragment cob, split = a range of the flat (group, row) sequence); it keeps all KH*4*CINB tiles
// of that cob in registers and streams over groups and rows with a KH-row window of input CM fragments.
// Step s of a group brings in input row s and gradient row s - PRE (PRE = KH-1-PADT rows of lead; rows
// outside the map are fetched clamped and never multiplied) through two private LDS slots, one per operand,
// software-pipelined so that neither the DMA nor the LDS reads wait in front of the matrix pipe:
//   taps kh < KA (older rows)  |  read input row s  |  taps KA..KH-2  |  DMA input row s+1, read G row s+1
//   | tap KH-1 (the new row)   -- the DMA of G row s+1 goes out at the top of the step.
// grid = 8 * NT * ceil(splits / 8) one-wave workgroups, dynamic LDS = (4*CINB + 4) KiB.
// (measured, round 3: the same waves as 4-wave workgroups -- four independent waves, no barrier, own LDS each, so
// that the dispatcher has 469 instead of 1 875 workgroups to place -- are slower inside the step, conv3 280 -> 289 us
// and conv2 88 -> 103 us, step 2.25 -> 2.32 ms: a 4-wave workgroup needs four free slots on ONE CU while the dgrad
// chain's kernels hold slots on every CU, one-wave workgroups fill whatever is free.)
template <int KH, int CINB, int NT, int HIN>
__global__ __launch_bounds__(64) void wgrad_conv_cm(const f4 *__restrict__ in_tm, const f4 *__restrict__ g_tm,
                                                     int G, int splits, int rows_per, f4 *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    constexpr int PADT = (KH - 1) / 2;
    constexpr int PRE = KH - 1 - PADT;
    constexpr int NI = 4 * CINB;                     // input fragments per step (slots 0..NI-1), then 4 of G
    constexpr int STEPS = HIN + PRE;                 // steps per group
    constexpr int KA = KH >= 3 ? KH - 2 : KH - 1;    // taps multiplied before the new input row is read
    const int lane = threadIdx.x;
    CV_STAMP_BEGIN
    const cm_stage S(wg_lds, lane);
    // workgroups go round-robin over the 8 XCDs: the NT waves of one split (same input rows) take ids 8 apart, so
    // they share one XCD's L2 and start back to back
    const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
    const int cob = rest % NT, split = (rest / NT) * 8 + xcd;
    // A split owns the gradient rows [r0, r1) of the flat (group, row) sequence -- rows_per of them, whatever the
    // group boundaries: every wave of the launch has the same work (with whole groups per wave train.py's batch gave
    // 1 875 waves for 2 048 slots: 173 SIMDs held one wave instead of two and idled for the second half), and a small
    // batch still fills the chip.  Before its first row a wave streams the KH - 1 input rows above it (`warm` steps
    // that fetch but do not multiply), as every group start does.
    const int R = G * HIN;
    const int r0 = split * rows_per, r1 = split >= splits ? r0 : (r0 + rows_per < R ? r0 + rows_per : R);
    if (r0 >= r1) return;
    const int gF = r0 / HIN, hF = r0 - gF * HIN, sF = hF > PADT ? hF - PADT : 0;
    const int gL = (r1 - 1) / HIN, hL = (r1 - 1) - gL * HIN;
    const int total = (gL * STEPS + hL + PRE + 1) - (gF * STEPS + sF);
    const int warm = hF + PRE - sF;
    const int g1 = gL + 1;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[KH][4][CINB];
#pragma unroll
    for (int a = 0; a < KH; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
            for (int c = 0; c < CINB; c++) acc[a][b][c] = zero;
    f4 bsum = zero;
    // fetch cursors (group, step) of the two operands; past the end they re-read valid data nobody uses
    int ig = gF, is = sF, gg_ = gF, gs = sF;
    auto next_in = [&]() -> const f4 * {
        const int gc = ig < g1 ? ig : g1 - 1;
        const int hr = is < HIN ? is : HIN - 1;
        if (++is == STEPS) { is = 0; ig++; }
        return in_tm + ((size_t)gc * HIN + hr) * (NI * 64);
    };
    auto next_g = [&]() -> const f4 * {
        const int gc = gg_ < g1 ? gg_ : g1 - 1;
        const int hg = gs - PRE < 0 ? 0 : gs - PRE;
        if (++gs == STEPS) { gs = 0; gg_++; }
        return g_tm + (((size_t)gc * HIN + hg) * 4 * NT + cob) * 64;
    };
    // window of the KH newest input rows, rotating: flat step j keeps its row in win[j % KH], so tap kh of the row
    // being accumulated (input row s - (KH-1-kh)) sits in win[(j - (KH-1-kh)) % KH] -- no register moves.  Rows
    // outside the map are never multiplied (the hr test), so stale or clamped contents are harmless.
    f4 win[KH][4][CINB];
    f4 Gr[4];
    {
        const f4 *gp = next_g(), *ip = next_in();
#pragma unroll
        for (int w = 0; w < 4; w++) S.fetch_s(gp + (size_t)w * (NT * 64), NI + w);
#pragma unroll
        for (int f = 0; f < NI; f++) S.fetch_s(ip + f * 64, f);
    }
    cm_stage::landed<NI>();                          // G row of step 0
#pragma unroll
    for (int w = 0; w < 4; w++) Gr[w] = S.read(NI + w);
    cm_stage::reads_done();
    // The DMA pieces of the next rows go out ONE AT A TIME between the MFMA blocks of a tap, not as a burst: the
    // vector-memory issue port is shared by the CU's waves, a burst of 8 pieces holds its wave for ~1 k cycles without
    // an MFMA, and with 8 waves per CU the bursts queue behind each other (measured with per-wave time stamps: the
    // same wave takes 108 us alone on its CU, 118 us with 3 neighbours, 156 us with 5, 178 us with 7).
    constexpr bool SPREAD_G = KH >= 3;               // KH == 2: the G row is needed one tap later -- no room
    int s = sF;
#pragma unroll 1
    for (int i = 0; i < total; i += KH) {
#pragma unroll
        for (int u = 0; u < KH; u++) {
            if (i + u >= total) break;
            const int h = i + u >= warm ? s - PRE : -1;      // -1: no gradient row is accumulated in this step
            auto taps = [&](int kh, auto piece) {     // kh is a constant after unrolling; piece(b) after block b < 12
                const int hr = h + kh - PADT;
                const int ws = (u + kh + 1) % KH;     // = (u - (KH-1-kh)) mod KH
                if (h >= 0 && hr >= 0 && hr < HIN) {  // (one branch per tap, not per block)
                    int b = 0;
#pragma unroll
                    for (int kw = 0; kw < 4; kw++)
#pragma unroll
                        for (int wo = 0; wo < 4; wo++) {
                            const int wi = wo + kw - 1;
                            if (wi < 0 || wi > 3) continue;
#pragma unroll
                            for (int t = 0; t < 4; t++)
#pragma unroll
                                for (int cb = 0; cb < CINB; cb++)
                                    acc[kh][kw][cb] = mfma4(win[ws][wi][cb][t], Gr[wo][t], acc[kh][kw][cb]);
                            piece(b++);
                        }
                } else {                              // a row outside the map: nothing to hide the pieces under
#pragma unroll
                    for (int b = 0; b < 12; b++) piece(b);
                }
            };
            auto none = [](int) {};
            const f4 *gp = next_g();                  // G row of step j+1 (its slot was read one step ago)
            auto g_piece = [&](int b) { if (b < 4) S.fetch_s(gp + (size_t)b * (NT * 64), NI + b); };
            if (!SPREAD_G) {
#pragma unroll
                for (int w = 0; w < 4; w++) g_piece(w);
            }
            if (h >= 0) bsum += (Gr[0] + Gr[1]) + (Gr[2] + Gr[3]);
#pragma unroll
            for (int kh = 0; kh < KA; kh++) {
                if (SPREAD_G && kh == 0) taps(kh, g_piece); else taps(kh, none);
            }
            cm_stage::landed<4>();                    // input row of this step (the G pieces above may still fly)
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int cb = 0; cb < CINB; cb++) win[u][w][cb] = S.read(w * CINB + cb);
            // (measured: issuing the next DMAs earlier, so that both have a whole step to land, at the price of waiting for the
            // LDS reads instead of running MFMAs under them: 281 -> 322 us for conv3 -- the reads must stay hidden)
#pragma unroll
            for (int kh = KA; kh < KH - 1; kh++) taps(kh, none);
            cm_stage::reads_done();
            cm_stage::landed<0>();                    // G row of step j+1
            f4 Gn[4];
#pragma unroll
            for (int w = 0; w < 4; w++) Gn[w] = S.read(NI + w);
            const f4 *ip = next_in();                 // input row of step j+1, under the last tap
            taps(KH - 1, [&](int b) { if (b < NI) S.fetch_s(ip + b * 64, b); });
            cm_stage::reads_done();
#pragma unroll
            for (int w = 0; w < 4; w++) Gr[w] = Gn[w];
            if (++s == STEPS) s = 0;
        }
    }
    cm_stage::landed<0>();           // the surplus fetches of the last step
    // this split's tiles as whole fragments (+ the bias sums as one more), combined by wgrad_conv_reduce
    constexpr int TILES = KH * 4 * CINB;
    f4 *pp = part + ((size_t)split * NT + cob) * (TILES + 1) * 64 + lane;
#pragma unroll
    for (int kh = 0; kh < KH; kh++)
#pragma unroll
        for (int kw = 0; kw < 4; kw++)
#pragma unroll
            for (int cb = 0; cb < CINB; cb++) pp[((kh * 4 + kw) * CINB + cb) * 64] = acc[kh][kw][cb];
    pp[TILES * 64] = bsum;
    CV_STAMP_END(KH == 3 && CINB == 2, 0);
}

// second pass of the convolution weight gradients: fragment f = cob * (TILES + 1) + tile of every split, summed
// in a fixed order (16 strided partial sums, then those in ascending order) by one 1024-thread workgroup.
// tile < TILES: lane (c', q) register r  <->  dW[kh][kw][ci = 16 cb + 4q + r][co = 16 cob + c'];
// tile == TILES: CM bias sums, lane (f, rg) register t -> db[16 cob + f] (registers, then lanes rg).
__global__ __launch_bounds__(1024) void wgrad_conv_reduce(const f4 *__restrict__ part, int splits, int NT, int TILES,
                                                          int CINB, int cin, int cout, float *__restrict__ dw,
                                                          float *__restrict__ db, int acc)
{
    __shared__ f4 sh[16][64];
    const int lane = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int frag = blockIdx.x;
    const size_t stride = (size_t)NT * (TILES + 1) * 64;
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    int sp = j;
    for (; sp + 48 < splits; sp += 64) {         // four loads in flight, added in split order
        f4 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = part[(size_t)(sp + 16 * u) * stride + (size_t)frag * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; u++) v += w[u];
    }
    for (; sp < splits; sp += 16) v += part[(size_t)sp * stride + (size_t)frag * 64 + lane];
    sh[j][lane] = v;
    __syncthreads();
    if (j != 0) return;
#pragma unroll
    for (int k = 1; k < 16; k++) v += sh[k][lane];
    const int cob = frag / (TILES + 1), tile = frag % (TILES + 1);
    const int cq = lane & 15, q = lane >> 4;
    const int co = 16 * cob + cq;
    if (tile == TILES) {
        float b = (v[0] + v[1]) + (v[2] + v[3]);
        b += __shfl_xor(b, 16);
        b += __shfl_xor(b, 32);
        if (lane < 16 && co < cout) db[co] = (acc ? db[co] : 0.0f) + b;
        return;
    }
    const int cb = tile % CINB, kk = tile / CINB;        // kk = kh * 4 + kw
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int ci = 16 * cb + 4 * q + r;
        if (ci < cin && co < cout) dw[((size_t)kk * cin + ci) * cout + co] = (acc ? dw[((size_t)kk * cin + ci) * cout + co] : 0.0f) + v[r];
    }
}

// first layer (k(1,4), 4 input channels): the 16 (base, matrix) values of a position form ONE
// fragment, so  T_wo[(wi, ci)][co] += X[h][(wi, ci)] G[h][wo][co]  gives every tap at once:
// dW[kw][ci][co] = sum_wo T_wo[(wo + kw - 1, ci)][co].  One wave per candidate-range split; a step is one
// position (1 + 4 fragments, 16 MFMAs), so R = 6 steps are kept in flight (30 KiB of LDS per wave).
constexpr int CV_WG1_RING = 6;
// bias gradient of one base from a wave's candidate-major sum fragment: lane (co, q) register r -> the four registers, then
// the four q rows; lanes 0..15 hold channel co.  (Done by the PRODUCER: a split then hands 64 floats of bias sums to the
// second pass instead of four fragments -- at 625 splits the one workgroup that adds them was reading 2.5 MB.)
__device__ __forceinline__ float conv1_bias_lanes(f4 v)
{
    float b = (v[0] + v[1]) + (v[2] + v[3]);
    b += __shfl_xor(b, 16);
    b += __shfl_xor(b, 32);
    return b;
}
__global__ __launch_bounds__(64) void wgrad_conv1_cm(const float *__restrict__ x, int64_t n, const f4 *__restrict__ g_tm,
                                                      int G, f4 *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) float wg_lds[];
    constexpr int HIN = CV_INPUT_H;
    constexpr int R = CV_WG1_RING;
    const int lane = threadIdx.x;
    const cm_stage S(wg_lds, lane);
    const int per = (G + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per, g1 = g0 + per < G ? g0 + per : G;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[4] = {zero, zero, zero, zero};
    f4 bsum[4] = {zero, zero, zero, zero};        // per base: the second pass adds the four (the fused kernel below makes them on four waves)
    // positions of consecutive groups are consecutive in both buffers: flat position i of this split
    const int total = (g1 > g0 ? g1 - g0 : 0) * HIN;
    const f4 *gp = g_tm + (size_t)g0 * HIN * 4 * 64;
    auto fetch_pos = [&](int i, int slot) {
        const int ic = i < total ? i : total - 1;
        const int gi = ic / HIN, h = ic - gi * HIN;          // X is read where the caller left it: [n][33][16] floats
        S.fetch_nat(x + (size_t)h * 16, (int64_t)(g0 + gi) * 16, n, (size_t)HIN * 16, slot * 5, lane);
#pragma unroll
        for (int wo = 0; wo < 4; wo++) S.fetch(gp + ((size_t)ic * 4 + wo) * 64, slot * 5 + 1 + wo);
    };
    if (total > 0) {
#pragma unroll
        for (int r = 0; r < R; r++) fetch_pos(r, r);
    }
    int slot = 0;
#pragma unroll 1
    for (int i = 0; i < total; i++) {
        cm_stage::landed<5 * (R - 1)>();
        const f4 X = S.read_nat(slot * 5, lane);
        f4 Gf[4];
#pragma unroll
        for (int wo = 0; wo < 4; wo++) Gf[wo] = S.read(slot * 5 + 1 + wo);
        cm_stage::reads_done();
        fetch_pos(i + R, slot);
#pragma unroll
        for (int wo = 0; wo < 4; wo++) {
            bsum[wo] += Gf[wo];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[wo] = mfma4(X[t], Gf[wo][t], acc[wo]);
        }
        slot = slot + 1 == R ? 0 : slot + 1;
    }
    cm_stage::landed<0>();
    if (g0 >= g1) return;
    f4 *pp = part + (size_t)blockIdx.x * 8 * 64 + lane;       // T_0..T_3, then the bias sums of this split: [base][16 channels]
#pragma unroll
    for (int wo = 0; wo < 4; wo++) {
        pp[wo * 64] = acc[wo];
        const float b = conv1_bias_lanes(bsum[wo]);
        if (lane < 16) reinterpret_cast<float *>(part + ((size_t)blockIdx.x * 8 + 4) * 64)[wo * 16 + lane] = b;
    }
}

// The same FUSED with the max-pool backward + SELU' of the first layer (round 5).  A workgroup of four waves owns a
// group; wave w owns BASE w: it makes that base's pre-activation gradient row on the spot from the pooled-map gradient,
// the pooled output and the window-offset codes (cv_unpool.hpp: one unpool_col, rows in sequence -- b_unpool_tm's terms
// and order), writes it to LDS at the positions cm_stage's DMA would have put it (fragment lane F at 16-byte position
// (F >> 4) + 4 ((F >> 2) & 3) + 16 (F & 3)), reads it back candidate-major and contracts it with the position's X
// fragment: T_w and the bias sum of base w.  The first layer's pre-activation gradient (33 rows x 4 KiB per group,
// 74 MB at train.py's batch) has no other reader: it is never written, and the element-wise pass in front of the
// weight gradient is gone (42 + 27 us at the tail of the 10 000 step, 13 + 16 us at 79 groups).  Per weight the same sum
// over (group, position, candidate) in the same order, per bias the same four per-base sums: same bits as unpool +
// wgrad_conv1_cm.  Plain loads, a block of 11 positions' operands requested at once; no DMA, no barrier.
template <int P>
__global__ __launch_bounds__(256) void wgrad_conv1_unpool_cm(const float *__restrict__ x, int64_t n, const f4 *__restrict__ gpool,
                                                              const f4 *__restrict__ pooled, const u32x2 *__restrict__ codes,
                                                              int G, f4 *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float lds[4][2][256];      // per wave: the X fragment, the gradient fragment
    constexpr int HIN = CV_INPUT_H, HO = HIN - P + 1;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const cm_stage S(&lds[w][0][0], lane);
    const int per = (G + gridDim.x - 1) / gridDim.x;
    const int g0 = blockIdx.x * per, g1 = g0 + per < G ? g0 + per : G;
    if (g0 >= g1) return;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc = zero, bsum = zero;
    f4 *xslot = reinterpret_cast<f4 *>(&lds[w][0][0]) + lane;
    f4 *gslot = reinterpret_cast<f4 *>(&lds[w][1][0]) + ((lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3));
#pragma unroll 1
    for (int g = g0; g < g1; g++) {
        const f4 *gp = gpool + ((size_t)g * HO * 4 + w) * 64 + lane;         // row ho: + ho * 256
        const f4 *pp = pooled + ((size_t)g * HO * 4 + w) * 64 + lane;
        const u32x2 *cp = codes + (size_t)g * HO * 64 + lane;                // row ho: + ho * 64
        // X in natural order: LDS position `lane` holds quarter (lane & 3) of candidate 4 ((lane >> 2) & 3) + (lane >> 4)
        int64_t cand = (int64_t)g * 16 + 4 * ((lane >> 2) & 3) + (lane >> 4);
        if (cand >= n) cand = n - 1;
        const f4 *xp = reinterpret_cast<const f4 *>(x + (size_t)cand * (HIN * 16) + 4 * (lane & 3));      // row h: + 4 h
        unpool_col<P> U;
        U.init();
        // Positions in blocks of RB: a block's operands (14 dwords per position) are all requested before its first
        // position is worked on -- one round trip to memory per block instead of one per position (with the next
        // position's operands requested one position ahead the wave stalled on every one of them: 0.9 us per position)
        constexpr int RB = 11;
        static_assert(HIN % RB == 0, "whole blocks");
#pragma unroll 1
        for (int hb = 0; hb < HIN; hb += RB) {
            f4 gvb[RB], yvb[RB], xvb[RB]; u32x2 cvb[RB];
#pragma unroll
            for (int i = 0; i < RB; i++) {
                const int hn = hb + i < HO ? hb + i : HO - 1;            // (rows past the last window: re-read, unused)
                gvb[i] = gp[(size_t)hn * 256]; yvb[i] = pp[(size_t)hn * 256]; cvb[i] = cp[(size_t)hn * 64];
                xvb[i] = xp[(size_t)(hb + i) * 4];
            }
#pragma unroll
            for (int i = 0; i < RB; i++) {
                if (hb + i < HO) U.push(gvb[i], yvb[i], cv_code16(cvb[i][0], cvb[i][1], w));
                else U.push_none();
                *gslot = U.emit();
                *xslot = xvb[i];
                const f4 X = S.read_nat(0, lane);
                const f4 Gf = S.read(1);
                bsum += Gf;
#pragma unroll
                for (int t = 0; t < 4; t++) acc = mfma4(X[t], Gf[t], acc);
            }
        }
    }
    part[((size_t)blockIdx.x * 8 + w) * 64 + lane] = acc;
    const float b = conv1_bias_lanes(bsum);
    if (lane < 16) reinterpret_cast<float *>(part + ((size_t)blockIdx.x * 8 + 4) * 64)[w * 16 + lane] = b;
}

// second pass, first layer.  lane (co, q) register r of T_wo: row i = 4q + r = wi*4 + ci  =>  wi = q, ci = r, and
// dW[kw][ci][co] = sum_wo T_wo[wi = wo + kw - 1].  Workgroup kw < 4 gathers, for every wo, the 16 lanes of T_wo
// that belong to its tap; workgroup 4 sums the bias fragment.  Fixed order: 16 strided partial sums over the
// splits, those ascending, then wo ascending.
__global__ __launch_bounds__(1024) void wgrad_conv1_reduce(const f4 *__restrict__ part, int splits, int cout,
                                                           float *__restrict__ dw, float *__restrict__ db, int acc)
{
    __shared__ f4 sh[16][64];
    const int l = threadIdx.x & 63, j = threadIdx.x >> 6;
    const int kw = blockIdx.x;
    const int co = l & 15, wo = l >> 4;
    const int wi = wo + kw - 1;
    const bool bias = kw == 4;
    const bool valid = bias || (wi >= 0 && wi <= 3);
    f4 v = (f4){0.f, 0.f, 0.f, 0.f};
    // (a split holds T_0..T_3 and, in its fifth fragment, 64 floats of bias sums [base][channel] the producer already
    // added over its lanes; the bias workgroup adds them over the splits, then base after base)
    if (bias) {
        __shared__ float shb[16][64];
        float vb = 0.0f;
        for (int sp = j; sp < splits; sp += 16) vb += reinterpret_cast<const float *>(part + ((size_t)sp * 8 + 4) * 64)[l];
        shb[j][l] = vb;
        __syncthreads();
        if (j != 0) return;
#pragma unroll
        for (int k = 1; k < 16; k++) vb += shb[k][l];
        shb[0][l] = vb;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (l < 16 && l < cout) {
            const float b = ((shb[0][l] + shb[0][16 + l]) + shb[0][32 + l]) + shb[0][48 + l];
            db[l] = (acc ? db[l] : 0.0f) + b;
        }
        return;
    }
    if (valid) {
        const int src = wo * 64 + wi * 16 + co;
        int sp = j;
        for (; sp + 48 < splits; sp += 64) {     // four loads in flight, added in split order
            f4 w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = part[(size_t)(sp + 16 * u) * 8 * 64 + src];
#pragma unroll
            for (int u = 0; u < 4; u++) v += w[u];
        }
        for (; sp < splits; sp += 16) v += part[(size_t)sp * 8 * 64 + src];
    }
    sh[j][l] = v;
    __syncthreads();
    if (j != 0) return;
#pragma unroll
    for (int k = 1; k < 16; k++) v += sh[k][l];
    sh[0][l] = v;                   // wave 0 only from here on (its own earlier reads of sh are done)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (l < 16 && l < cout) {
        f4 t = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; w++) t += sh[0][w * 16 + l];     // invalid (wo, kw) pairs hold zeros
#pragma unroll
        for (int r = 0; r < 4; r++) dw[((size_t)kw * 4 + r) * cout + l] = (acc ? dw[((size_t)kw * 4 + r) * cout + l] : 0.0f) + t[r];
    }
}

// ---------------------------------------------------------------------------
// launch sites (cv_train.hip calls them on the side streams)
// ---------------------------------------------------------------------------
// enough workgroups to cover the chip once (one 8-wave workgroup per CU); the per-split tiles go to a scratch
// buffer and are summed in a fixed order by wgrad_dense_reduce (no float atomics on the weights).
// scratch for the per-split tiles of a weight gradient (one kernel pair at a time uses it, in stream order)
// Scratch of the per-split tiles: one REGION per weight-gradient launch site (0 heads, 1 fc5, 2 fc4, 3 conv3, 4 conv2,
// 5 conv1), so that the sites may run on different streams at the same time.  The regions are sized at their upper
// bounds (the split counts are capped, so they do not depend on the batch) and reserved before a step is enqueued
// (cv_wgrad_scratch_reserve): nothing inside the step synchronises or reallocates.
static int wg_region(cv_model *m, int region, size_t need_bytes, float **out)
{
    if (!m->wg_part || region < 0 || region >= CV_WG_REGIONS || need_bytes > m->wg_size[region] * sizeof(float)) {
        cv_set_error("weight-gradient scratch region %d not reserved (%zu bytes needed)", region, need_bytes);
        return 1;
    }
    *out = m->wg_part + m->wg_off[region];
    return 0;
}

int cv_wgrad_scratch_reserve(cv_model *m)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    size_t sz[CV_WG_REGIONS];
    const int njb4 = s.nb4, njb5 = s.nb5;
    sz[0] = (size_t)65 * ((size_t)s.nb4 * 256 + 16);
    sz[1] = (size_t)(256 / ((s.nb4 + 15) / 16) + 1) * ((size_t)s.nb4 * njb5 * 256 + njb5 * 16);
    sz[2] = (size_t)(256 / ((s.kb4 + 15) / 16) + 1) * ((size_t)s.kb4 * njb4 * 256 + njb4 * 16);
    for (int l = 2; l >= 1; l--) {
        const size_t NT = s.ntile[l], TILES = (size_t)a.kh[l] * 4 * s.cinb[l];
        sz[5 - l] = ((2048 + NT - 1) / NT) * NT * (TILES + 1) * 256;
    }
    sz[5] = (size_t)1024 * 8 * 256;
    size_t total = 0;
    for (int r = 0; r < CV_WG_REGIONS; r++) { sz[r] = (sz[r] + 63) / 64 * 64; total += sz[r]; }
    if (m->wg_part && m->wg_part_bytes >= total * sizeof(float)) return 0;
    CV_HIP(hipDeviceSynchronize());
    if (m->wg_part) CV_HIP(hipFree(m->wg_part));
    m->wg_part = nullptr; m->wg_part_bytes = 0;
    CV_HIP(hipMalloc(&m->wg_part, total * sizeof(float)));
    m->wg_part_bytes = total * sizeof(float);
    size_t off = 0;
    for (int r = 0; r < CV_WG_REGIONS; r++) { m->wg_off[r] = off; m->wg_size[r] = sz[r]; off += sz[r]; }
    return 0;
}

// ranges: candidate ranges (0 = enough for one workgroup per CU with two input fragments per wave)
template <int NJB, int XF = 2>
static int dense_wgrad_launch(cv_model *m, int region, const float *x_tm, int KB, const float *g_tm, int G, int K, int N, float *dw,
                              float *db, hipStream_t st, int ranges = 0)
{
    float *scratch = nullptr;
    const int kblocks = (KB + 8 * XF - 1) / (8 * XF);
    int splits = ranges > 0 ? ranges : 256 / ((KB + 15) / 16);
    if (splits > G) splits = G;
    if (splits < 1) splits = 1;
    if (wg_region(m, region, (size_t)splits * (KB * NJB * 256 + NJB * 16) * sizeof(float), &scratch)) return 1;
    const size_t lds = (size_t)2 * (NJB + 8 * XF) * 1024;
    if (set_lds(wgrad_dense_cm<NJB, false, XF>, lds)) return 1;
    wgrad_dense_cm<NJB, false, XF><<<dim3(kblocks, splits), 512, lds, st>>>((const f4 *)x_tm, KB, (const f4 *)g_tm, G, 0,
                                                                            (f4 *)scratch);
    const int64_t per = (int64_t)KB * NJB * 64;
    wgrad_dense_reduce<<<nblk(per + NJB * 16, 256), 256, 0, st>>>((const f4 *)scratch, splits, KB, NJB, K, N, dw, db, m->tr_accumulate);
    CV_HIP(hipGetLastError());
    return 0;
}

// fc4 of the full topology (6.2 MB of weights, 18 blocks of 16 input fragments): how many candidate ranges, and how the
// input fragments are dealt.  Above 256 groups: 14 ranges x 18 blocks, one workgroup per CU.  Below, the kernel runs on a
// side stream next to the data-gradient chain and what it costs the step is the CUs and the HBM bytes it takes from that
// chain -- the per-range tiles are 6.2 MB each, written here and read back by the second pass: 7 ranges (5 at 79 groups:
// at least 16 groups each) leave half the CUs to the main stream and halve those bytes; between 140 and 256 groups the
// same 7 ranges over 36 blocks of 8 fragments (one per wave) fill the CUs at that size.  Same-box sweep over ten batch
// sizes, profiles/r05/step_ab_session13_fc4_ranges.txt: -10 .. -38 us of a step from 960 to 4 000 candidates, nothing lost
// elsewhere.  (Development: dbg5 >= 16 sets the ranges, bit 3 deals one fragment per wave, bit 2 two.)
static void fc4_wgrad_shape(const cv_model *m, int G, int *ranges, int *xf)
{
    int r = 14, x = 2;
    if (G <= 140) { r = (G + 15) / 16; r = r < 4 ? 4 : r > 7 ? 7 : r; }
    else if (G <= 256) { r = 7; x = 1; }
    if (m->dbg[5] >= 16) r = m->dbg[5] >> 4;
    if (r > 14) r = 14;                                  // the scratch region holds 15
    if (m->dbg[5] & 8) x = 1;
    if (m->dbg[5] & 4) x = 2;
    *ranges = r; *xf = x;
}

int cv_tile_dense_wgrad(cv_model *m, int layer, const float *x_tm, const float *g_tm, int64_t n, hipStream_t st)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const int G = (int)((n + 15) / 16);
    float *Gd = m->grads; const int64_t *o = m->poff;
    if (layer == 4) {
        if (is_full(a)) {
            int ranges, xf;
            fc4_wgrad_shape(m, G, &ranges, &xf);
            if (xf == 1) return dense_wgrad_launch<21, 1>(m, 2, x_tm, s.kb4, g_tm, G, s.flat, a.fc4, Gd + o[6], Gd + o[7], st, ranges);
            return dense_wgrad_launch<21>(m, 2, x_tm, s.kb4, g_tm, G, s.flat, a.fc4, Gd + o[6], Gd + o[7], st, ranges);
        }
        return dense_wgrad_launch<3>(m, 2, x_tm, s.kb4, g_tm, G, s.flat, a.fc4, Gd + o[6], Gd + o[7], st);
    }
    if (is_full(a)) return dense_wgrad_launch<11>(m, 1, x_tm, s.nb4, g_tm, G, a.fc4, a.fc5, Gd + o[8], Gd + o[9], st);
    return dense_wgrad_launch<2>(m, 1, x_tm, s.nb4, g_tm, G, a.fc4, a.fc5, Gd + o[8], Gd + o[9], st);
}

// heads: dW16[k][j] = sum_c X[c][k] g[c][j] for the 16 head outputs j at once (wgrad_dense_cm with the natural
// gradient array), then this pass sums the splits in order and scatters the columns [j_lo, j_hi) into the head
// matrices: column j belongs to head q with offset j0[q] and width N[q]; bias sums likewise.
struct head_cols { float *dw[4], *db[4]; int j0[4], N[4]; };

__global__ void wgrad_heads_reduce(const f4 *__restrict__ part, int splits, int KB, int K, int j_lo, int j_hi,
                                   head_cols hc, int acc)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)KB * 64;
    const bool bias = t >= per;
    const int j = bias ? (int)(t - per) : (int)(t & 15);
    if (j >= 16 || j < j_lo || j >= j_hi) return;
    int q = 0;
#pragma unroll
    for (int i = 1; i < 4; i++)
        if (j >= hc.j0[i]) q = i;
    const int col = j - hc.j0[q], N = hc.N[q];
    if (bias) {
        const float *bpart = (const float *)(part + (size_t)splits * per);
        float b = bpart[j];
        int sidx = 1;
        for (; sidx + 8 <= splits; sidx += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = bpart[(size_t)(sidx + u) * 16 + j];
#pragma unroll
            for (int u = 0; u < 8; u++) b += w[u];
        }
        for (; sidx < splits; sidx++) b += bpart[(size_t)sidx * 16 + j];
        hc.db[q][col] = (acc ? hc.db[q][col] : 0.0f) + b;
        return;
    }
    f4 v = part[t];
    int sidx = 1;
    for (; sidx + 4 <= splits; sidx += 4) {
        f4 w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = part[(size_t)(sidx + u) * per + t];
#pragma unroll
        for (int u = 0; u < 4; u++) v += w[u];
    }
    for (; sidx < splits; sidx++) v += part[(size_t)sidx * per + t];
    const int lane = (int)(t & 63), kb = (int)(t >> 6), qq = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int k = 16 * kb + 4 * qq + r;
        if (k < K) hc.dw[q][(size_t)k * N + col] = (acc ? hc.dw[q][(size_t)k * N + col] : 0.0f) + v[r];
    }
}

// x_tm = dropped-out fc4 output (heads 0) or fc5 output (heads 1..3); g16 = [n][16] head gradients
int cv_tile_heads_wgrad(cv_model *m, const float *d4_tm, const float *h5_tm, const float *g16, int64_t n, hipStream_t st)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const int G = (int)((n + 15) / 16);
    if (G <= 0) return 0;
    float *Gd = m->grads; const int64_t *o = m->poff;
    head_cols hc;
    const int j0[4] = {0, 4, 6, 10}, N[4] = {4, 2, 4, 6};
    for (int q = 0; q < 4; q++) { hc.dw[q] = Gd + o[10 + 2 * q]; hc.db[q] = Gd + o[11 + 2 * q]; hc.j0[q] = j0[q]; hc.N[q] = N[q]; }
    const size_t lds = (size_t)2 * (1 + 16) * 1024;
    for (int pass = 0; pass < 2; pass++) {
        const float *x_tm = pass == 0 ? d4_tm : h5_tm;
        const int KB = pass == 0 ? s.nb4 : s.nb5, K = pass == 0 ? a.fc4 : a.fc5;
        const int kblocks = (KB + 15) / 16;
        int splits = 64 / kblocks;               // little work per group: few, longer ranges keep the second pass short
        if (splits > G) splits = G;
        float *scratch = nullptr;
        if (wg_region(m, 0, (size_t)splits * (KB * 256 + 16) * sizeof(float), &scratch)) return 1;
        wgrad_dense_cm<1, true><<<dim3(kblocks, splits), 512, lds, st>>>((const f4 *)x_tm, KB, (const f4 *)g16, G, n,
                                                                          (f4 *)scratch);
        wgrad_heads_reduce<<<nblk((int64_t)KB * 64 + 16, 256), 256, 0, st>>>((const f4 *)scratch, splits, KB, K,
                                                                             pass == 0 ? 0 : 4, pass == 0 ? 4 : 16, hc, m->tr_accumulate);
    }
    CV_HIP(hipGetLastError());
    return 0;
}

// layer 1 = conv2 (in = pool1 TM), 2 = conv3 (in = pool2 TM); g = pre-activation gradient TM.
// One wave per (output fragment, split): splits for 2 waves per SIMD (1024 SIMDs) with the same number of gradient
// rows each; the per-split tiles go to the scratch buffer and are summed in a fixed order by wgrad_conv_reduce (no
// float atomics).  The split boundaries depend on the number of groups only: the same batch gives the same bits.
template <int KH, int CINB, int NT, int HIN>
static int conv_wgrad_launch(cv_model *m, int region, const float *in_tm, const float *g_tm, int G, int cin, int cout, float *dw,
                             float *db, hipStream_t st)
{
    float *scratch = nullptr;
    constexpr int TILES = KH * 4 * CINB;
    // rows of the flat (group, row) sequence per wave: 2 048 / NT waves per output fragment (two per SIMD), but at
    // least 8 rows each (a wave re-streams KH - 1 rows above its range and leaves TILES + 1 fragments to the second pass)
    if (G <= 0) return 0;
    const int R = G * HIN, wmax = 2048 / NT;
    int rows_per = (R + wmax - 1) / wmax;
    if (rows_per < 8) rows_per = 8;
    const int used = (R + rows_per - 1) / rows_per;              // = splits: every one owns at least one row
    if (wg_region(m, region, (size_t)used * NT * (TILES + 1) * 256 * sizeof(float), &scratch)) return 1;
    wgrad_conv_cm<KH, CINB, NT, HIN><<<8 * NT * ((used + 7) / 8), 64, (4 * CINB + 4) * 1024, st>>>(
        (const f4 *)in_tm, (const f4 *)g_tm, G, used, rows_per, (f4 *)scratch);
    wgrad_conv_reduce<<<NT * (TILES + 1), 1024, 0, st>>>((const f4 *)scratch, used, NT, TILES, CINB, cin, cout, dw, db, m->tr_accumulate);
    CV_HIP(hipGetLastError());
    return 0;
}

int cv_tile_conv_wgrad(cv_model *m, int layer, const float *in_tm, const float *g_tm, int64_t n, hipStream_t st)
{
    const cv_shapes &s = m->sh; const cv_arch &a = m->arch;
    const int G = (int)((n + 15) / 16);
    float *dw = m->grads + m->poff[2 * layer];
    float *db = m->grads + m->poff[2 * layer + 1];
    const int cin = s.cin[layer], cout = a.cout[layer];
    if (is_full(a)) {
        if (layer == 2) return conv_wgrad_launch<3, 2, 3, 26>(m, 5 - layer, in_tm, g_tm, G, cin, cout, dw, db, st);
        return conv_wgrad_launch<2, 1, 2, 29>(m, 5 - layer, in_tm, g_tm, G, cin, cout, dw, db, st);
    }
    if (layer == 2) return conv_wgrad_launch<5, 1, 2, 33>(m, 5 - layer, in_tm, g_tm, G, cin, cout, dw, db, st);
    return conv_wgrad_launch<3, 1, 1, 33>(m, 5 - layer, in_tm, g_tm, G, cin, cout, dw, db, st);
}

// first layer: X as the caller holds it ([n][33 positions][16 = base*4 + matrix] floats, transposed by the
// fetch), g = TM of its pre-activation gradient ([33*4] fragments per group)
// first layer of a topology that pools it by 5 (full): max-pool backward + SELU' + weight gradient in one kernel
// (wgrad_conv1_unpool_cm); gpool / pooled / codes as for the unpool pass.  *done = false: not this topology, or
// dbg4 = 4 -- the caller unpools, then cv_tile_conv1_wgrad.
int cv_tile_conv1_wgrad_unpool(cv_model *m, const float *x, const float *gpool, const float *pooled, const float *codes, int64_t n,
                               hipStream_t st, bool *done)
{
    *done = false;
    const int G = (int)((n + 15) / 16);
    if (G <= 0 || m->arch.pool[0] != 5 || m->sh.ntile[0] != 1 || m->dbg[4] == 4) return 0;
    const int splits = G < 1024 ? G : 1024;
    const int per = (G + splits - 1) / splits;
    const int used = (G + per - 1) / per;
    float *scratch = nullptr;
    if (wg_region(m, 5, (size_t)splits * 8 * 256 * sizeof(float), &scratch)) return 1;
    wgrad_conv1_unpool_cm<5><<<splits, 256, 0, st>>>(x, n, (const f4 *)gpool, (const f4 *)pooled, (const u32x2 *)codes, G, (f4 *)scratch);
    wgrad_conv1_reduce<<<5, 1024, 0, st>>>((const f4 *)scratch, used, m->arch.cout[0], m->grads + m->poff[0],
                                          m->grads + m->poff[1], m->tr_accumulate);
    CV_HIP(hipGetLastError());
    *done = true;
    return 0;
}

int cv_tile_conv1_wgrad(cv_model *m, const float *x, const float *g_tm, int64_t n, hipStream_t st)
{
    const int G = (int)((n + 15) / 16);
    if (G <= 0) return 0;
    const int splits = G < 1024 ? G : 1024;
    const int per = (G + splits - 1) / splits;
    const int used = (G + per - 1) / per;
    float *scratch = nullptr;
    if (wg_region(m, 5, (size_t)splits * 8 * 256 * sizeof(float), &scratch)) return 1;
    wgrad_conv1_cm<<<splits, 64, CV_WG1_RING * 5 * 1024, st>>>(x, n, (const f4 *)g_tm, G, (f4 *)scratch);
    wgrad_conv1_reduce<<<5, 1024, 0, st>>>((const f4 *)scratch, used, m->arch.cout[0], m->grads + m->poff[0],
                                          m->grads + m->poff[1], m->tr_accumulate);
    CV_HIP(hipGetLastError());
    return 0;
}
