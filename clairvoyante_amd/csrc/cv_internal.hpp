// cv_internal.hpp -- model object and data layouts shared by the kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <stdint.h>
#include <stddef.h>
#include "../../include/clairvoyante_amd.h"

// ---------------------------------------------------------------------------
// Tile-major ("TM") activation layout -- the HBM layout of every intermediate.
//
// Candidates are processed in GROUPS of 16 (one MFMA tile column each).  A
// per-candidate feature vector of K = 16*KB values is stored as KB fragments
// of 1 KiB:   TM[group][kb][lane][s],  lane = kq*16 + c,  (c = candidate in
// group, feature k = 16*kb + 4*s + kq).  One wave reads a fragment with ONE
// fully coalesced 16-byte-per-lane load and the four dwords it gets are
// exactly the B operands (B[k][j]: lane = j + 16*k) of the four
// v_mfma_f32_16x16x4_f32 steps that cover features 16*kb .. 16*kb+15 in
// ascending order.  Kernels compute D = W^T * act^T (rows = output features,
// columns = candidates); with the weight rows permuted by sigma(i) = 4*(i%4) +
// i/4 the D registers of a lane are again one TM fragment dword-for-dword, so a
// layer's output is written with one coalesced 16-byte store per lane and feeds
// the next layer without any transpose.
// ---------------------------------------------------------------------------
__host__ __device__ inline int cv_sigma(int i) { return 4 * (i & 3) + (i >> 2); }

// float offset of (candidate, feature k) in a TM buffer with KB fragments per group
__host__ __device__ inline size_t cv_tm_index(int64_t cand, int k, int KB)
{
    int64_t g = cand >> 4;
    int c = (int)(cand & 15);
    int kb = k >> 4, j = k & 15, s = j >> 2, kq = j & 3;
    return ((size_t)(g * KB + kb) * 64 + (size_t)(kq * 16 + c)) * 4 + s;
}

struct cv_shapes {
    int hc[3];    // conv output heights (SAME: == input heights)
    int hp[3];    // pooled heights
    int cin[3];   // input channels
    int ntile[3]; // ceil(cout/16)
    int cinb[3];  // ceil(cin/16) (layer 0: unused)
    int flat;     // hp[2]*4*cout[2]
    int kb4;      // fc4 K fragments = hp[2]*4*ntile[2]
    int nb4, nb5; // fc4 / fc5 output fragments
};

// up to this many groups the convolutions of a training step run as position parts of whole groups and the side
// streams are chained before the main stream's one wait (the rest of the small-batch kernel set goes by cv_model::tiny_g).
// 160 in rounds 2-4; swept in round 5 (profiles/r05/step_ab_session14_small_batch_regime.txt)
constexpr int CV_TINY_PARTS_MAX_G = 80;

struct cv_model {
    cv_arch arch;
    cv_shapes sh;
    int device;
    int impl;          // 0 plain kernels, 1 MFMA kernels
    int64_t chunk;     // candidates per internal pass
    // parameters: flat, TF layouts, table order
    int64_t psize[CV_NUM_PARAMS];
    int64_t poff[CV_NUM_PARAMS + 1];
    int pndim[CV_NUM_PARAMS];
    int64_t pdims[CV_NUM_PARAMS][4];
    float *params, *grads, *adam_m, *adam_v;
    // packed (fragment-major) weights for the MFMA kernels
    float *wp_conv1;     // [kw][64]
    float *wp_conv[3];   // [1],[2]: [nt][kh][kw][cb][64][4]
    float *wp_fc4;       // [kb][nb4][64][4]
    float *wp_fc5;       // [nb4][nb5][64][4]
    float *wpd_conv[3];  // data-gradient weights of conv2 / conv3 (pack_conv_dgrad)
    float *wpd_fc4;      // data-gradient weights of fc4 [slab][jb][24][64][4]
    float *wpr_fc4;      // data-gradient weights of fc4 by column and pooled row [col][row][24][64][4] (full topology)
    float *wg_part;      // per-split tiles of the dense weight gradients (two-pass combine), owned
    size_t wg_part_bytes;
    size_t wg_off[6], wg_size[6];   // regions of wg_part in floats (CV_WG_REGIONS), one per weight-gradient launch site
    float *wps_fc4;      // forward weights of fc4 in 3 slabs [slab][kb][8][64][4] (full topology, small batches)
    float *wps7_fc4;     // ... and in 7 slabs [slab][kb][3][64][4] (dense_small: one wave per group and slab; slim: 3 slabs of one)
    float *wps21_fc4;    // full topology: 21 slabs of one fragment [slab][kb][1][64][4] (dense_small, the smallest inference passes)
    float *wps3_fc5;     // fc5 in 3 slabs [slab][kb][4][64][4] (dense_small)
    void *tail_dev;      // device copy of the tail arguments of the fused fc4 + fc5 + heads kernel (dense_tm EPI 3)
    unsigned char tail_host[256];   // what tail_dev holds
    float *wp5p_fc5;     // fc5 in k pairs [kp][24][64][4] (tail of the large-pass fc4 kernel, dense_tm EPI 3; full topology)
    float *wpd_fc5;      // data-gradient weights of fc5 [jb][24 | 4][64][4]
    float *wp_heads0;    // [nb4][64][4]  base head (rows 0..3)
    float *wp_heads1;    // [nb5][64][4]  zygosity / type / length heads
    float *wp_heads12;   // [nb5*16][12]  the same three heads' weights of an fc5 unit side by side (training heads: data gradient)
    int variant;         // kernel selection bits, see include/clairvoyante_amd.h (cv_set_option "variant")
    // Packed (MFMA fragment) copies of the weights that are CURRENT, one bit per layout (CVL_*).  A pass packs what its
    // kernels will read and is not valid (a training step of G groups reads two of the four fc4 layouts: the others stay
    // stale until a pass that reads them); every change of the weights clears the mask (cv_layouts_stale).  Consumers
    // check the bit where they take the pointer (cv_layout_current): a mismatch is an error, never stale arithmetic.
    unsigned packed_valid;
    // workspaces (allocated lazily for `ws_cap` candidates)
    int64_t ws_cap;      // MFMA path capacity (multiple of 16)
    float *tm_p1, *tm_p2, *tm_p3, *tm_h4, *tm_h5;
    int64_t ref_cap;     // plain path capacity
    float *r_a[3], *r_p[3], *r_h4, *r_h5;
    int64_t last_n;      // candidates of the last chunk (for cv_get_activation)
    int last_impl;
    int last_variant;
    int keep_act;        // option "keep_activations" (default 0): a pass whose fc5 + heads ride on the fc4 kernel also stores the
                         // fc4 / fc5 maps that only cv_get_activation reads (68 % of that kernel's writes)
    int last_maps;       // the last pass left the fc4 / fc5 maps in memory
    // training workspaces
    int64_t tr_cap;
    float *t_buf;        // one slab, carved by the training code
    size_t t_bytes;
    double *loss_dev;    // 8 doubles: losses of the current pass (inside the grads_own allocation, behind the gradients)
    double *loss_acc;    // 8 doubles: losses accumulated over steps (cv_loss_accumulate / cv_loss_read)
    // fixed-order loss sums of the tile path: per-block rows of the heads kernel (4 doubles each, slice after slice) and
    // the per-block sums of the L2 kernel ([kernel][256]); t_loss_finish adds them into loss_dev in a fixed order
    double *loss_rows; int64_t loss_rows_cap, loss_rows_used;
    double *l2_rows;
    float *grads_own;    // the library's own gradient bucket (CV_GRAD_HEADER + count floats); `grads` points
                         // CV_GRAD_HEADER floats into it, or into the caller's bucket (cv_bind_grad_bucket)
    // training step: side stream of the weight-gradient kernels, fork / join events, "dense gradients final"
    hipStream_t tr_side;
    hipStream_t tr_side_more[2];   // further side streams: weight gradients of different layers are independent
    int train_sides;     // option "train_side_streams": side streams of the weight gradients at tiny batches, 1..3 (default 3)
    hipEvent_t tr_ev[16];
    hipEvent_t tr_dense_ready;
    hipEvent_t tr_l2_done;         // recorded on the side stream behind the L2 kernel of a step
    hipEvent_t tr_pack_fork, tr_pack_done;   // weight packing on the side stream (cv_pack_for_training)
    int train_overlap;   // option: weight gradients on the side stream (default 1)
    int train_ksplit;    // option: k-split fc4 forward at tiny batches (default 1)
    // cv_forward picks kernels by the number of groups (options "infer_small_groups", "infer_fc4_small_groups",
    // "infer_slab_groups"): up to inf_small_g the convolutions unfused with their positions over four waves, up to
    // inf_fc4_small_g fc4 / fc5 as one wave per (group, slab), up to inf_slab_g fc4 as three output slabs per group block
    int inf_small_g, inf_fc4_small_g, inf_slab_g;
    int inf_fc4_one_g;         // ... up to this many groups fc4 as one wave per (group, fragment) (21 slabs)
    // option "dense_rag": fc4's three-slab form on ragged waves (dense_rag, round 6): 0 = shape by formula (default), 4..14 = that many
    // tile-units per SIMD and workgroup (A/B, calibration), -1 = the round-5 kernel (one group x 7 tiles per wave)
    int inf_rag_s;
    // option "infer_flat": the per-group convolution kernels of an inference pass on flat (group, row) ranges when the model says a
    // whole-group launch would take longer (1, default); 0 = always whole groups (rounds 1-5), 2 = always flat ranges (tests)
    int inf_flat;
    // option "slim_waves": groups per workgroup of the slim topology's conv3 + fc4 kernel: 0 = from the number of groups (default), 2 / 4 / 8
    int inf_slim_waves;
    int inf_slim_small_g;      // slim: passes of up to this many groups run the small-pass kernel set (cv_mfma_forward)
    int tiny_g;          // option "train_tiny_groups": batches of up to this many groups take the latency-oriented
                         // kernel variants of the training step (default 400; 0 = never)
    // fc4 dropout output / keep mask (a*keep) of the LAST training slice, for cv_get_activation 6 / 7
    const float *last_tr_d4, *last_tr_mask;
    const float *last_tr_pool[3], *last_tr_gpre[3];   // cv_get_activation 11..13 / 21..23: maps of the last training slice (nullptr: not materialised)
    int64_t last_tr_n;
    int last_tr_tile;    // 1: tile-major buffers, 0: natural [n, fc4]
    // option keep_activations + a step of several slices: the two maps of EVERY slice, copied here slice after slice
    // (mask first, dropout output behind it), so that cv_get_activation 6 / 7 covers the whole batch
    float *tr_keep; size_t tr_keep_floats;
    int tr_accumulate;   // the weight-gradient second passes ADD to the gradient (1) or store 0 + sum (0: first slice of a step)
    // optional per-kernel timing (option "profile")
    // options "dbg0".."dbg7": development switches of the training step (A/B runs and variant tests; 0 = shipped path).
    //   dbg0 = n: position parts of the convolution data gradients      dbg1 = n: ... of the training-forward convolutions
    //        (n = 9: the batch-dependent number of parts instead of flat row ranges; n = 7: flat ranges for a small batch too;
    //         dbg1 = 8: conv2 forward on flat ranges too)
    //   dbg2 = 1 / 2: unpool always thread-per-row / always streaming, 3: fc4's alpha-dropout as its own pass, 4: thread-per-row at tiny batches (default there: row segments), 5: the tail of the tiny-batch forward as three kernels, 6: row segments at every size
    //   dbg3 = 1: fc4 data gradient and conv3 unpool as two kernels
    //   dbg4 = 3: slim selu' as its own pass, 4: conv1's unpool and weight gradient as two kernels
    //   dbg5 = 1: all weight packing in one launch in stream order (>= 16: dbg5 >> 4 candidate ranges of fc4's weight gradient, bit 3 / bit 2: one / two input fragments per wave there)      dbg6 = n: row parts of dense_dgrad_unpool (few groups; + 100: 4-wave workgroups; 0 = chosen by the number of groups)
    //   dbg7 = 1: training-forward conv3 on conv_tm instead of conv3_rot
    int dbg[8];
    // option "train_sched": the round-5 re-cut of the step's schedule, one bit per change (default 3839 = all but bit 8; A/B runs and
    // the variant tests switch them off one by one -- same arithmetic either way):
    //   1  loss header behind the heads kernel on a side stream (tiny batches: the first, larger ones: the second, idle one)
    //      instead of at the tail of the step
    //   2  conv1's weight gradient on the main stream at tiny batches instead of a side stream
    //   4  ONE marker on the main stream for the L2 term and the weight packing instead of one each
    //   8  launch sites at the same point of the main stream share a marker (not for the full topology above 512 groups:
    //      at train.py's 625 the step is 39 us SLOWER with it, at 313 groups 50 us faster -- profiles/r05/step_ab_session4_sched_bits.txt)
    //   16 a pass packs only the weight layouts its kernels read instead of every forward layout
    //   32 the base head's data gradient, the dropout factor and selu'(fc4) on the store of fc5's data-gradient kernel
    //      instead of a pass of their own
    //   64 no memset of the gradient at the head of a step: the second passes of the first slice store instead of adding
    //   128 tiny batches: the side streams chained before the ONE wait of the main stream at the end of the step
    //   512 conv1's weight gradient on the main stream at EVERY batch size (the chain's tail: -11 us at 5 000, -12 us at 10 000)
    //   2048 full topology up to 512 groups: the side stream's L2 term and weight packing start BEHIND conv1's forward kernel
    //      instead of beside it (-20 us at 79 groups, -5 at 313; +8 at 625 and +25 us for slim at 79, hence the bounds)
    //   1024 batches above the tiny range (up to 2 048 groups): fc5 + heads + losses + head gradients as one kernel (-12 us at
    //      5 000, -10 us at 10 000)
    int sched;
    int profile;
    void *prof;          // cv_prof*, owned
    const char *stage_kernel[CV_NUM_STAGES];   // kernel (template instance) each stage of the last cv_forward chunk ran
};

// brackets one kernel launch with events when profiling is on (no-ops otherwise)
void cv_prof_begin(cv_model *m, int stage, hipStream_t st);
void cv_prof_end(cv_model *m, int stage, hipStream_t st);
void cv_prof_free(cv_model *m);

void cv_set_error(const char *fmt, ...);

// layouts of cv_model::packed_valid
enum : unsigned {
    CVL_CONV = 1u,      // wp_conv1, wp_conv[1..2]
    CVL_FC4 = 2u,       // wp_fc4
    CVL_FC5 = 4u,       // wp_fc5
    CVL_FC5P = 8u,      // wp5p_fc5
    CVL_FC4S3 = 16u,    // wps_fc4
    CVL_FC5S3 = 32u,    // wps3_fc5
    CVL_FC4S7 = 64u,    // wps7_fc4
    CVL_HEADS = 128u,   // wp_heads0 / wp_heads1
    CVL_DCONV = 256u,   // wpd_conv[1..2]
    CVL_DFC4 = 512u,    // wpr_fc4 or wpd_fc4 (by dbg3)
    CVL_DFC5 = 1024u,   // wpd_fc5
    CVL_FC4S21 = 2048u, // wps21_fc4
    CVL_FORWARD = CVL_CONV | CVL_FC4 | CVL_FC5 | CVL_FC5P | CVL_FC4S3 | CVL_FC5S3 | CVL_FC4S7 | CVL_FC4S21 | CVL_HEADS,
    CVL_BACKWARD = CVL_DCONV | CVL_DFC4 | CVL_DFC5,
};
inline void cv_layouts_stale(cv_model *m, unsigned which = ~0u) { m->packed_valid &= ~which; }
int cv_layout_current(const cv_model *m, unsigned layout, const char *who);    // 0 = current; else sets the error text

#define CV_TR_EVENTS 16
#define CV_WG_REGIONS 6
#define CV_TR_SIDES 3          // side streams of the weight-gradient kernels (tr_side + tr_side_more)
#define CV_GRAD_HEADER 16     // floats in front of the flat gradient: the loss header (cv_train.hip)

#define CV_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            cv_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

// kernel launchers (defined in the .hip files)
int cv_ref_forward(cv_model *m, const float *x, int64_t n, float *out16, hipStream_t st);
int cv_mfma_forward(cv_model *m, const float *x, int64_t n, float *out16, hipStream_t st);
int cv_pack_weights(cv_model *m, hipStream_t st);
int cv_launch_heads(cv_model *m, const float *h4, const float *h5, int tm, int64_t n, float *out16,
                    hipStream_t st);
bool cv_tile_supported(const cv_model *m);
int cv_pack_train_weights(cv_model *m, hipStream_t st);
// what a training pass over G groups will read and is stale; sw_ordered: sw already runs behind st (the caller forked)
int cv_pack_for_training(cv_model *m, hipStream_t st, bool backward, int G, hipStream_t sw = nullptr, hipEvent_t fork = nullptr,
                         hipEvent_t done = nullptr, bool *wait_before_dense = nullptr, bool sw_ordered = false, int phase = 0);
int cv_tile_train_convs(cv_model *m, const float *x, int64_t n, float *p1, float *a1, float *p2, float *a2,
                        float *p3, float *a3, hipStream_t st, const std::function<int()> *after_conv1 = nullptr);
#define CV_DENSE_KSPLIT 8      // k ranges of the fc4 training forward at tiny batches (cv_tile_dense_fwd)
// part: scratch of CV_DENSE_KSPLIT * groups * nb4 fragments, or NULL = always the single ascending-k chain
// drop / drop_done (fc4 of a training pass): where the kernel set allows it the alpha-dropout is applied by the layer's
// last kernel (*drop_done = true); otherwise the caller runs cv_dropout_tm
struct cv_train_dropout { float *d4, *amask; float rate; uint64_t seed, step; int64_t cand0; };
int cv_tile_dense_fwd(cv_model *m, int layer, const float *in_tm, float *out_tm, int64_t n, hipStream_t st,
                      float *part = nullptr, const cv_train_dropout *drop = nullptr, bool *drop_done = nullptr);
int cv_tile_train_tail(cv_model *m, const float *p3_tm, float *h4_tm, float *h5_tm, const float *y, int64_t n, int want_grad,
                       float *g16, float *g5pre_tm, float *part, const cv_train_dropout *drop, hipStream_t st, bool *done);
int cv_tile_train_fc5_heads(cv_model *m, float *d4_tm, float *h5_tm, const float *y, int64_t n, int want_grad, float *g16,
                            float *g5pre_tm, hipStream_t st, bool *done);
int cv_tile_fc5_dgrad(cv_model *m, const float *g_tm, float *gin_tm, int64_t n, hipStream_t st, const float *g16 = nullptr,
                      const float *mask_tm = nullptr, const float *act_tm = nullptr);
// act_below (layers without pooling, slim): the layer-below output; the result is then times selu' = its pre-activation gradient
int cv_tile_fc4_dgrad(cv_model *m, const float *g_tm, float *gin_tm, int64_t n, hipStream_t st, const float *act_below = nullptr);
int cv_tile_fc4_dgrad_unpool(cv_model *m, const float *g_tm, const float *pooled, const float *codes, float *gpre, int64_t n,
                             hipStream_t st);
int cv_tile_conv_dgrad(cv_model *m, int layer, const float *g_tm, float *gin_tm, int64_t n, hipStream_t st,
                       const float *act_below = nullptr);
int cv_tile_dense_wgrad(cv_model *m, int layer, const float *x_tm, const float *g_tm, int64_t n, hipStream_t st);
int cv_tile_conv_wgrad(cv_model *m, int layer, const float *in_tm, const float *g_tm, int64_t n, hipStream_t st);
int cv_tile_conv1_wgrad(cv_model *m, const float *x, const float *g_tm, int64_t n, hipStream_t st);
int cv_tile_conv1_wgrad_unpool(cv_model *m, const float *x, const float *gpool, const float *pooled, const float *codes, int64_t n,
                               hipStream_t st, bool *done);
int cv_wgrad_scratch_reserve(cv_model *m);      // scratch of the weight-gradient kernels at its upper bound
int cv_tile_heads_wgrad(cv_model *m, const float *d4_tm, const float *h5_tm, const float *g16, int64_t n, hipStream_t st);
int cv_tile_heads_pre(cv_model *m, const float *d4_tm, const float *h5_tm, int64_t n, float *pre16, hipStream_t st);
int cv_tile_heads_train(cv_model *m, const float *d4_tm, const float *h5_tm, const float *y, int64_t n, int want_grad,
                        float *g16, float *g5pre_tm, hipStream_t st);
int cv_dropout_tm(cv_model *m, const float *h4, float *d4, float *amask, int64_t n, float rate, uint64_t seed,
                  uint64_t step, int64_t cand0, hipStream_t st);
int cv_tm_to_natural(const float *tm, int KB, int feat_per_pos_padded, int feat_per_pos, int npos,
                     int64_t n, float *dst, hipStream_t st);
