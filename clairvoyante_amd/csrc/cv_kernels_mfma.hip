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


// the kernels themselves (fragments of this translation unit, inside this namespace)
#include "cv_kernels_conv.hpp"
#include "cv_kernels_dense.hpp"

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
            else if (m->dbg[1] != 4) {
                // two groups per wave (dense_small2): 3 KB of loads per 8 MFMAs instead of 4 KB -- 63 groups 55.2 -> 48.7 us, a
                // pass of 1 000 candidates 143.8 -> 137.3 us (ring depth 8; 4 / 6 / 12: 148.2 / 142.5 / 139.6); at 100 groups
                // it loses to the slab form (120.6 against 83.6 us).  dbg1 = 4: one group per wave, ring depth 4
                m->stage_kernel[3] = "dense_small2<8>";
                dense_small2<8><<<nblk((int64_t)((G + 1) / 2) * 21, 4), 256, 0, st>>>((const f4 *)m->tm_p3, s.kb4, (const f4 *)m->wps21_fc4,
                                                                                      P + o[7], a.fc4, (f4 *)m->tm_h4, G, 21, 21);
            }
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
    // (slim: 11 slabs of 24 output fragments over 3 k fragments -- a streaming kernel: it reads and writes the 16.9 KB-per-
    // candidate conv3 map once each.  Up to 128 groups four-wave workgroups, twice as many of them: the slim step at 320 / 640 /
    // 1 250 / 2 000 candidates 0.331 / 0.344 / 0.352 / 0.395 -> 0.322 / 0.327 / 0.340 / 0.385 ms; from 157 groups on the
    // eight-wave form is as good or better (two groups per wave: -20 us at 4 000 and 5 000 only, +10 .. +25 us at 2 500 and
    // 3 200: not used) -- profiles/r06/slim_fc4_dgrad_shapes_ab.txt; dbg4 = 9: eight waves at every size)
    if (G <= 128 && m->dbg[4] != 9)
        return launch_dense<24, 4, 1>(g_tm, s.nb4, m->wpd_fc4, nullptr, 0, gin_tm, G, st, s.kb4 / 24, 1, nullptr, hd);
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