// cv_kernels_conv.hpp -- the convolution kernels of the tile path (conv1_tm, front_source, conv_tm, front2_tm, conv3_rot)
// and the pooling codes of the training forward pass.  A FRAGMENT of cv_kernels_mfma.hip: included there, once, inside its
// anonymous namespace behind the packing functions (it uses cv_tile.hpp's helpers and that file's wave-stamp macros); split
// off in round 6 so that no file of the tile path is longer than ~1 800 lines.  Launchers and the choice of kernel forms:
// cv_kernels_mfma.hip.
#pragma once
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

