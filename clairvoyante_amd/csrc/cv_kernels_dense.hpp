// cv_kernels_dense.hpp -- the dense-layer kernels of the tile path: heads_tm / heads_finish, conv3fc4_slim, dense_tm,
// dense_rag, dense_dgrad_unpool, dense_small, dense_ksum, heads_train_tm, train_tail_tm, infer_tail_tm.  A FRAGMENT of
// cv_kernels_mfma.hip: included there, once, inside its anonymous namespace behind cv_kernels_conv.hpp; split off in
// round 6.  Launchers and the choice of kernel forms: cv_kernels_mfma.hip.
#pragma once
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

// dense_small<1, D> with TWO groups per wave: the weight fragment of a step multiplies both groups' activation fragments
// -- 3 KB of loads per 8 MFMAs instead of 4 KB, half as many waves.  For passes where dense_small's waves are two to a SIMD
// and share a CU's 64 B per clock of vector loads (49 .. 80 groups of the full topology's fc4).  Same chain per value.
template <int D>
__global__ __launch_bounds__(256) void dense_small2(const f4 *__restrict__ in_tm, int KB, const f4 *__restrict__ wp_all,
                                                     const float *__restrict__ bias, int nout, f4 *__restrict__ out_tm,
                                                     int G, int NSLAB, int NBT)
{
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int GP = (G + 1) / 2;
    if (unit >= GP * NSLAB) return;
    const int g0 = 2 * (unit / NSLAB), slab = unit % NSLAB;
    const bool two = g0 + 1 < G;
    const f4 *bp0 = in_tm + (size_t)g0 * KB * 64 + lane;
    const f4 *bp1 = in_tm + (size_t)(two ? g0 + 1 : g0) * KB * 64 + lane;
    const f4 *wp = wp_all + (size_t)slab * KB * 64 + lane;
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc0 = zero, acc1 = zero;
    f4 A[D], B0[D], B1[D];
    auto fetch = [&](int d) {
        B0[d] = bp0[(size_t)d * 64];
        B1[d] = bp1[(size_t)d * 64];
        A[d] = wp[(size_t)d * 64];
    };
    auto step = [&](int d) {
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            acc0 = mfma4(A[d][s4], B0[d][s4], acc0);
            acc1 = mfma4(A[d][s4], B1[d][s4], acc1);
        }
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d);
#pragma unroll 1
    for (int kb0 = D; kb0 < KB; kb0 += D) {
        bp0 += (size_t)D * 64; bp1 += (size_t)D * 64; wp += (size_t)D * 64;
#pragma unroll
        for (int d = 0; d < D; d++) {
            step(d);
            fetch(d);
        }
    }
#pragma unroll
    for (int d = 0; d < D; d++) step(d);
    if (slab >= NBT) return;
    const int q = lane >> 4;
    const f4 b4 = load_bias4(bias, slab, q, nout);
    out_tm[((size_t)g0 * NBT + slab) * 64 + lane] = selu4(acc0 + b4);
    if (two) out_tm[((size_t)(g0 + 1) * NBT + slab) * 64 + lane] = selu4(acc1 + b4);
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

