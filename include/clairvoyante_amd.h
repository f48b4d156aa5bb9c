/*
 * clairvoyante_amd.h -- C ABI of the MI355X-native Clairvoyante v3 pileup-CNN path.
 *
 * The reference has no FFI layer: its boundary is the duck-typed Python class
 * `Clairvoyante` (clairvoyante/clairvoyante_v3.py:5-284, same surface in
 * clairvoyante_v3_slim.py) whose methods each wrap ONE `tf.Session.run`.  This
 * header is what a binding for that class would bind instead of TensorFlow: every
 * entry point below names the reference method / session.run it replaces.
 * INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain C; pointers + sizes only.  `*_dev` pointers are DEVICE (HBM) pointers
 *     (e.g. torch `tensor.data_ptr()`), `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  Work is enqueued on `stream`; calls do not
 *     synchronise unless stated.
 *   - tensors use the reference's layouts: X [n,33,4,4] fp32 NHWC
 *     (position, base ACGT, matrix), already with matrices 1..3 minus matrix 0
 *     (clairvoyante/utils_v2.py:46); Y [n,16] fp32; parameters in TensorFlow
 *     layouts (conv HWIO, dense [in,out]) keyed by their checkpoint variable names.
 *   - every function returns 0 on success, non-zero on error; cv_last_error()
 *     returns a thread-local message.  No ownership crosses the boundary except
 *     the opaque handle.
 *   - a handle is bound to one GPU and may be used from any host thread, one call
 *     at a time (the reference drives predictNoRT/trainNoRT from a worker thread:
 *     callVar.py:197-204, train.py:87-109).
 */
#ifndef CLAIRVOYANTE_AMD_H
#define CLAIRVOYANTE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CV_INPUT_H 33      /* 2*flankingBaseNum+1, clairvoyante/param.py:6 */
#define CV_INPUT_W 4       /* A C G T */
#define CV_INPUT_C 4       /* matrixNum, clairvoyante/param.py:7 */
#define CV_NUM_OUT 16      /* base4 | zygosity2 | varType4 | indelLength6 */
#define CV_NUM_PARAMS 18

/* Constructor arguments of the reference class (clairvoyante_v3.py:7-16).
 * v3 full: kh {1,2,3} cout {16,32,48} pool {5,4,3} fc4 336 fc5 168
 * v3 slim: kh {1,3,5} cout {8,16,32}  pool {1,1,1} fc4 36  fc5 18
 * (clairvoyante_v3_slim.py:9-11; no pooling layers = window 1).              */
typedef struct cv_arch {
    int32_t kh[3];      /* kernelSize{1,2,3}[0]; the width is always 4        */
    int32_t cout[3];    /* numFeature{1,2,3}                                   */
    int32_t pool[3];    /* pollSize{1,2,3}[0]; 1 = layer absent                */
    int32_t fc4, fc5;   /* hiddenLayerUnits{4,5}                               */
} cv_arch;

typedef struct cv_model cv_model;

/* thread-local text of the last failure on the calling thread */
const char *cv_last_error(void);

/* replaces Clairvoyante.__init__ + _buildGraph + tf.Session (v3.py:7-29):
 * allocates weights, optimizer slots and workspaces on GPU `device`.          */
int cv_create(const cv_arch *arch, int device, cv_model **out);
/* replaces Clairvoyante.close / __del__ (v3.py:180,283) */
int cv_destroy(cv_model *m);

/* Variable table = tf.trainable_variables() of the reference graph
 * (jupyter_nb/visualization.ipynb:103-120).  idx in [0, CV_NUM_PARAMS).       */
int cv_param_info(const cv_model *m, int idx, const char **tf_name, int *ndim, int64_t dims[4]);
/* Flat fp32 device buffer holding all 18 variables back to back in table order
 * (TF layouts): what restoreParameters fills and saveParameters reads
 * (v3.py:243-251), what a data-parallel host all-reduces / broadcasts.
 * `offsets` (optional) receives CV_NUM_PARAMS+1 float offsets.                 */
int cv_param_buffer(cv_model *m, float **flat_dev, int64_t *count, int64_t *offsets);
/* copy one variable in (tf.train.Saver.restore, v3.py:248-251) / out (save).
 * `src`/`dst` are HOST pointers; synchronous with respect to `stream`.        */
int cv_set_param(cv_model *m, const char *tf_name, const float *src, int64_t count, void *stream);
int cv_get_param(cv_model *m, const char *tf_name, float *dst, int64_t count, void *stream);
/* tell the model the flat buffer was modified externally (repack on next use) */
int cv_params_changed(cv_model *m);

/* replaces session.run((YBaseChangeSigmoid, YZygositySoftmax, YVarTypeSoftmax,
 * YIndelLengthSoftmax), phase False) of predict / predictNoRT (v3.py:257-280).
 * out16_dev [n,16]: columns 0..3 base sigmoid, 4..5 zygosity softmax,
 * 6..9 variant-type softmax, 10..15 indel-length softmax.  n may be 0.          */
int cv_forward(cv_model *m, const float *x_dev, int64_t n, float *out16_dev, void *stream);

/* Device-side part of callVar.Output (callVar.py:59-72,81-87): per candidate
 * argmax of the three softmax heads (lowest index wins ties, np.argmax), the two
 * best bases of the sigmoid head (highest index wins ties, argsort()[::-1]),
 * the genotype-quality operands (top-2 products, fp32) and the depth sum `dp`.
 * call_dev [n,8] int32: varType, zygosity, indelLength, base1, base2, 0,0,0
 * qual_dev [n,4] fp32 : top1 product, top2 product, dp, 0                       */
int cv_call_postproc(cv_model *m, const float *x_dev, const float *out16_dev, int64_t n,
                     int32_t *call_dev, float *qual_dev, void *stream);

/* Host half of callVar.Output (callVar.py:72-153): the VCF records of n candidates from the decisions of
 * cv_call_postproc -- quality int(-4.343*log((p2+1e-300)/(p1+1e-300))), SNP / REF allele, inserted bases and
 * indel-length guess from the tensor, <INS>/<DEL> + SVTYPE, LENGUESS, GT, FILTER, "%.4f" allele fraction -- as text,
 * one '\n'-terminated line per record, in candidate order.  All pointers are HOST pointers.
 *   call [n,8] int32, qual [n,4] fp32: as cv_call_postproc wrote them;
 *   x: tensors [rows,33,4,4] fp32 (matrices 1..3 minus matrix 0); candidate i uses row xrow[i] (xrow NULL: row i);
 *   pos_buf / pos_meta [rows,6] int64: byte offset and length of contig, position and 33-base reference sequence of
 *   a candidate inside pos_buf (what cv_parse_tensor_text emits); candidate i uses row pos_row[i] (NULL: row i);
 *   show_ref: also candidates called REF (--showRef); has_qual / qual_min: --qual (FILTER PASS / LowQual, else ".").
 * Candidates with depth 0 give no record.  Runs on the cv_set_host_threads() threads.
 * Returns 0, or 2 when out_cap is too small (*out_len = bytes needed, nothing written), or 1 (cv_last_error).      */
int cv_format_vcf(const int32_t *call, const float *qual, int64_t n, const float *x, const int64_t *xrow,
                  const char *pos_buf, const int64_t *pos_meta, const int64_t *pos_row, int show_ref,
                  int has_qual, int qual_min, char *out, int64_t out_cap, int64_t *out_len, int64_t *nrecords);

/* debug / parity: copy one intermediate of the LAST cv_forward chunk to
 * dst_dev in the reference's natural layout ([n,h,4,c] NHWC or [n,units]).
 * layer: 1..3 = pool1..pool3 outputs (for slim: conv outputs), 4 = fc4, 5 = fc5.
 * Serves what getTensorAndLayerPNG.py:30-37 reaches into m.conv1.. for.
 * layer 6 / 7 refer to the last cv_grad / cv_loss slice instead: 6 = the alpha-dropout
 * keep mask of fc4 times its affine factor a (selu.py:53-62; 0 where a unit was
 * dropped, a where kept, 1 everywhere at rate 0), 7 = dropout4, the layer's output
 * [n, fc4] -- what the parity tests feed to / compare with the oracle.
 * layers 11..13 / 21..23: the pooled maps (slim: conv outputs) and the pre-activation gradients of conv1..conv3 of the
 * last single-slice training step, natural layout (21 of the full topology is not materialised: the first layer's
 * unpool rides in its weight-gradient kernel) -- what tools/gpu_train_map_diff.py lays beside the plain kernels'.
 * Layers 4 / 5 of a pass that ran fc5 and the heads on the tail of the fc4 kernel exist only with option
 * "keep_activations" set before the pass (error otherwise).                        */
int cv_get_activation(cv_model *m, int layer, float *dst_dev, int64_t n, void *stream);

/* debug / parity: the device's SELU (csrc/cv_math.hpp, selu.py:21-25) evaluated on every fp32 bit pattern in
 * [lo_bits, hi_bits] in ascending pattern order; *violations = adjacent pairs, taken as NEGATIVE floats (pattern
 * ascending = value descending), where the output increases; *checksum = wrapping sum of the output bit patterns of
 * all but the first input (the oracle's sweep must give the same).  0x80000000 .. 0xff800000 is the whole negative
 * axis: zero violations there prove SELU monotone, which is what lets the convolution kernels apply max-pooling
 * to the raw accumulators and the activation once per pooled row (max and a monotone map commute).  Synchronous. */
int cv_selu_sweep(int device, uint32_t lo_bits, uint32_t hi_bits, uint64_t *violations, uint64_t *checksum);

/* knobs: "impl" (0 = plain one-thread-per-output kernels, 1 = MFMA tile kernels),
 * "chunk" (candidates per internal pass), "profile" (0/1, see cv_kernel_times), "keep_activations" (0/1, default 0:
 * a pass whose fc5 + heads ride on the tail of the fc4 kernel -- variant bit 10 -- also stores the fc4 / fc5 maps,
 * which then only cv_get_activation layers 4 / 5 read; off, those layers report an error after such a pass and the
 * kernel writes a third of the bytes), "train_overlap" (0/1: weight
 * gradients of the training step on a side stream next to the data-gradient chain; default 1, same bits),
 * "infer_small_groups" / "infer_fc4_small_groups" / "infer_slab_groups" (defaults 256 / 288 / -1: cv_forward picks its
 * kernels by the number of groups of 16 candidates in the pass -- up to the first the convolutions unfused with their
 * positions over 8, 4 or 2 waves; up to the second fc4 / fc5 may run as one wave per (group, slab) -- up to
 * "infer_fc4_one_groups" (default 80) fc4 as one wave per (group, output fragment): 1 000 candidates 194 -> 148 us with fc5 + the heads as one launch --; fc4 otherwise as three
 * output slabs on ragged waves (dense_rag) or, with fc5 and the heads on its tail, all 21 tiles per wave: -1 = whichever an
 * estimate of the launch's time says is shorter at this size, >= 0 = the slab form up to that many groups; the same bits
 * whichever runs), "dense_rag" (0 default: the shape of the three-slab fc4 launch -- tile-units per SIMD and workgroup --
 * from the number of groups, so that the time of a pass is proportional to its size; 4..14 = that shape, -1 = the round-5
 * kernel with one group x 7 tiles per wave; A/B and calibration, same bits), "infer_flat" (1 default: the per-group
 * convolution kernel of an inference pass runs on equal ranges of the flat (group, row) sequence when a whole-group launch
 * would leave SIMDs with a wave more than others; 0 = always whole groups, 2 = always flat ranges; same bits),
 * "slim_waves" (0 default: groups per workgroup of the slim topology's conv3 + fc4 kernel -- 8 or 4 -- from the number of
 * groups, so that a small pass spreads over all CUs; 4 / 8 = that many; same bits), "slim_small_groups" (-1 default: a pass
 * of the slim topology runs its layers unfused -- positions split over several waves or equal ranges of the flat (group,
 * row) sequence, fc4 as one wave per (group, output fragment); time linear in the pass -- wherever an estimate says the
 * fused kernels' rounds of workgroups would take longer: a predict() call of 1 000 candidates takes 84 us instead of 400,
 * 32 784 candidates 0.98 ms instead of 1.18; 0..65536 = up to that many groups, fused beyond; same bits),
 * "train_tiny_groups" (0..4096, default 400: training batches of up to that many groups of 16 candidates split the
 * serial loops of their layers over more waves -- same bits; the position parts of the convolutions stop at 80 groups
 * whatever the value), "train_ksplit" (0/1, default 1: at such batches
 * the fc4 forward of the TRAINING pass adds eight partial sums over k ranges instead of one ascending-k chain; fixed
 * order, reproducible run to run, within the gradient tolerance of the single chain, 4 % faster at 1 250 candidates;
 * the slim topology's fc4 -- 396 dependent k steps -- does so at every batch; never used by cv_forward) and
 * "train_side_streams" (1..3, default 3: at such batches the weight gradients of different layers -- independent of
 * each other -- run on up to that many side streams; larger batches use two when the value is >= 2, the second one
 * for the last layers only; same bits),
 * "train_sched" (bits, default 3839: the step's schedule as re-cut in round 5 -- 1 loss header behind the heads kernel on
 * the side stream at tiny batches, 2 the first layer's weight gradient on the main stream at tiny batches, 4 one
 * marker for L2 term + weight packing, 8 launch sites at one point of the main stream share a marker, 16 a pass packs
 * only the weight layouts its kernels read, 32 the base head's data gradient + dropout / selu' factor on the store of
 * fc5's data-gradient kernel, 64 no memset of the gradient at the head of a step (the weight-gradient second passes of
 * its first slice store instead of adding), 128 tiny batches: the side streams chained before the one wait of the main
 * stream, 512 the first layer's weight gradient on the main stream at every batch size, 1024 fc5 + heads + losses + head
 * gradients of a training pass above the tiny range as one kernel, 2048 up to 512 groups the side stream's L2 term and
 * packing start behind conv1's forward kernel (bits 256 and 4096 named two schedules that lost their A/B runs in round 5
 * and are gone); same bits with any value -- the reported loss to its last bits),
 * "dbg0".."dbg7" (development A/B switches of the training step, 0 = shipped path; see cv_internal.hpp),
 * "variant" (bit 0: first layer fused into the conv2 kernel, bit 1: MFMA heads kernel,
 * bit 2: 8-wave fc4 workgroups, bit 3: rotating-window conv3 kernel, bit 5: fc4 with two groups of
 * 16 candidates per wave, bit 6: fused conv1+conv2 kernel whose two waves per group share the first layer through
 * LDS (full topology), bit 7: fc4 of passes of up to 256 groups on a kernel without barriers (one wave per group
 * and slab of 3 output fragments, operands prefetched from L2 through a register ring), bit 8: slim topology, conv3
 * and fc4 as one kernel (the conv3 map stays in registers), bit 9: the four heads ride on the fc5 kernel (passes of
 * more than 256 groups whose fc5 is a kernel of its own: 12-16 us of a pass between 5 000 and 50 000 candidates, on by
 * default since round 6), bit 10: fc5 and the four heads on the TAIL
 * of the large-pass fc4 kernel (passes of more than 2 048 groups, full topology: the fc4 output never leaves the
 * registers it was accumulated in; 1.533 -> 1.516 ms for the three layers); default 2031; the alternatives give bit-identical results and exist for A/B
 * timing).                                                                                          */
int cv_set_option(cv_model *m, const char *key, int64_t value);
int cv_get_option(const cv_model *m, const char *key, int64_t *value);

/* Per-kernel device timing of cv_forward (option "profile" = 1): every kernel launch
 * is bracketed by hipEventRecord on the launch stream.  cv_kernel_times synchronises,
 * returns for stage s = 0..CV_NUM_STAGES-1 (conv1, conv2, conv3, fc4, fc5, heads)
 * the summed milliseconds and launch counts since the last call, and resets them.  */
#define CV_NUM_STAGES 6
int cv_kernel_times(cv_model *m, double ms[CV_NUM_STAGES], int64_t launches[CV_NUM_STAGES]);
/* The kernel (template instance as rocprofv3 prints it, without the argument list) stage s of the last cv_forward
 * chunk ran, or NULL if the stage was fused away / the plain kernels ran: lets a measurement taken in another
 * process (rocprofv3 --pmc passes, profiles/pmc_traffic.json) be matched to the binary that is running.          */
int cv_kernel_name(const cv_model *m, int stage, const char **name);

/* ---- training (replaces the session.run calls of train / trainNoRT /
 * getLoss / getLossNoRT, v3.py:183-227, and AdamOptimizer, v3.py:174) ---------- */

/* forward loss with phase False, dropout 0, lambda 0 (getLoss, v3.py:207-216).
 * losses_host[6]: loss1..loss4, lossL2, total -- SUMS over the batch
 * (v3.py:140-151).  Synchronises `stream`.                                      */
int cv_loss(cv_model *m, const float *x_dev, const float *y_dev, int64_t n, double *losses_host,
            void *stream);
/* forward (phase True: alpha-dropout rate `drop4` on fc4, selu.py:34-69; rate on
 * fc5 is 0.0 = identity) + backward into the flat gradient buffer (data terms
 * only, no lambda term).  seed/step select the counter-based dropout stream.
 * losses_host as above with lossL2 = lambda*sum(w^2)/2.  Synchronises.          */
int cv_grad(cv_model *m, const float *x_dev, const float *y_dev, int64_t n, float drop4,
            float lambda, uint64_t seed, uint64_t step, double *losses_host, void *stream);
/* flat gradient buffer, same order/size as cv_param_buffer (for RCCL all-reduce) */
int cv_grad_buffer(cv_model *m, float **flat_dev, int64_t *count);

/* ---- the optimizer step without host round trips (what train.py's loop and a data-parallel host use) ----
 * The gradient BUCKET is `header` (= 16) floats followed by the flat gradient.  The header carries the losses of
 * the step so that ONE all-reduce(SUM) of the bucket exchanges gradients and losses together: floats 2k, 2k+1 =
 * loss k (base, zygosity, type, length; v3.py:140-148) as a (hi, lo) float pair whose sum is the double the
 * kernels accumulated, floats 8, 9 = lambda * sum(w^2)/2, float 10 = 1 per contributing rank, rest 0.
 * dense_begin = first float of the fc4 / fc5 / head gradients: [dense_begin, count) is 95 % of the bucket and is
 * final early (before the convolution backward pass), [0, dense_begin) at the end of the step.
 * cv_bind_grad_bucket makes the step write into a caller-owned device buffer of `count` floats (16-byte aligned;
 * e.g. a torch tensor that torch.distributed reduces in place -- no staging copies); NULL returns to the
 * library's own.  The caller keeps the buffer alive while it is bound.                                          */
int cv_grad_bucket_info(const cv_model *m, int64_t *count, int64_t *header, int64_t *dense_begin);
int cv_bind_grad_bucket(cv_model *m, float *bucket_dev, int64_t count);
/* cv_grad without the host synchronisation: enqueues forward + backward on `stream` (the weight-gradient kernels
 * on an internal side stream that forks from / joins `stream`; option "train_overlap" = 0 keeps one stream) and
 * the loss header.  comm_stream (may be NULL): made to wait, through an event, for the moment the dense part of
 * the bucket is final, so that an exchange enqueued on it overlaps the rest of the backward pass; the part
 * before dense_begin is final in `stream` order.  The caller orders cv_apply_adam behind its exchange.         */
int cv_grad_async(cv_model *m, const float *x_dev, const float *y_dev, int64_t n, float drop4, float lambda,
                  uint64_t seed, uint64_t step, void *stream, void *comm_stream);
/* add the (exchanged) loss header of the bucket to a device-side accumulator (the L2 term divided by the rank
 * count of float 10) -- train.py only needs the SUM of the batch losses of an epoch (train.py:113-114,123).     */
int cv_loss_accumulate(cv_model *m, void *stream);
/* read the accumulator: losses_host[6] = loss1..loss4, lossL2, total summed over the accumulated steps, *steps =
 * how many (may be NULL); reset != 0 zeroes it.  Synchronises `stream`.                                         */
int cv_loss_read(cv_model *m, double losses_host[6], int64_t *steps, int reset, void *stream);
/* TF1 Adam (beta1 .9, beta2 .999, eps 1e-8, lr_t = lr*sqrt(1-b2^t)/(1-b1^t)) on
 * grad + lambda*w for non-bias variables (l2 term of v3.py:150); t = 1,2,...     */
int cv_apply_adam(cv_model *m, float lr, float lambda, int64_t t, void *stream);
/* cv_apply_adam followed by cv_loss_accumulate in ONE launch (the step of train.run_epoch: a launch less at the tail
 * of every step); same arithmetic as the two calls.                                                            */
int cv_apply_adam_accumulate(cv_model *m, float lr, float lambda, int64_t t, void *stream);
/* optimizer slots m / v (checkpoint variables "<name>/Adam", "<name>/Adam_1")   */
int cv_adam_buffers(cv_model *m, float **m_dev, float **v_dev, int64_t *count);
/* device-to-device copy between a caller buffer (e.g. a torch tensor handed to
 * torch.distributed) and one of the flat buffers: which 0 = parameters,
 * 1 = gradients, 2 = Adam m, 3 = Adam v; to_model != 0 copies caller -> model.  */
int cv_flat_copy(cv_model *m, int which, float *caller_dev, int to_model, void *stream);

/* ---- host data plane (no GPU work) ------------------------------------------------ */

/* Text-tensor reader = the per-row work of utils_v2.GetTensor (utils_v2.py:20-21,33-46;
 * writer dataPrepScripts/CreateTensor.py:24,56).  Parses whole lines
 * "<ctg> <pos> <refSeq33> <528 numbers>" from buf[0,len): rows whose centre base is not
 * A/C/G/T are dropped (utils_v2.py:38-40); x_out[row][528] receives the values with
 * matrices 1..3 minus matrix 0 (utils_v2.py:45-46); meta_out[row][6] = byte offset and
 * length of ctg, pos, seq inside buf.  Stops after max_rows rows or the last complete
 * line; *consumed = bytes eaten, *nrows = rows written, *nbad = malformed rows skipped.  */
int cv_parse_tensor_text(const char *buf, int64_t len, int64_t max_rows, float *x_out,
                         int64_t *meta_out, int64_t *consumed, int64_t *nrows, int64_t *nbad);

/* Host threads cv_parse_tensor_text may use (process-wide, default 1); the rows do not depend on it. */
int cv_set_host_threads(int n);

/* c-blosc 1.x chunk codec for the 500-item blocks of the `.bin` training file
 * (utils_v2.py:159-186 blosc.pack_array(cname='lz4hc'), :189-207 blosc.unpack_array;
 * tensor2Bin.py:24-28).  Decoder: LZ4/LZ4HC streams, byte shuffle, split blocks, memcpy'd
 * chunks.  Encoder: one LZ4 block with byte shuffle (readable by c-blosc).              */
int64_t cv_blosc_nbytes(const uint8_t *chunk, int64_t clen);
int cv_blosc_decompress(const uint8_t *chunk, int64_t clen, uint8_t *dst, int64_t dstcap);
int cv_blosc_compress_lz4(const uint8_t *src, int64_t n, int typesize, uint8_t *dst, int64_t dstcap,
                          int64_t *clen);
/* n chunks at once on the host threads of cv_set_host_threads (the blocks of one DecompressArray call,
 * utils_v2.py:196-203); status[i] != 0 marks a chunk that failed.                                  */
int cv_blosc_decompress_many(const uint8_t *const *chunks, const int64_t *clens, uint8_t *const *dsts,
                             const int64_t *dstcaps, int64_t n, int32_t *status);
/* The same blocks straight into ONE array: each chunk holds one pickled ndarray (blosc.pack_array); its raw data is
 * located inside the decompressed pickle and copied to dst + i*block_bytes (every block but the last must hold
 * exactly block_bytes).  lens[i] = data bytes of block i; status[i] = 0 ok / 1 corrupt / 2 layout not recognised (the
 * caller then un-pickles instead).  Returns 0 only if all blocks are ok.                                */
int cv_blosc_unpack_blocks(const uint8_t *const *chunks, const int64_t *clens, int64_t n, uint8_t *dst,
                           int64_t block_bytes, int64_t *lens, int32_t *status);

/* CRC32C (Castagnoli) of the tensor bytes / table blocks of the TensorFlow V2 checkpoint
 * bundle written by saveParameters and read by restoreParameters (v3.py:243-251).        */
uint32_t cv_crc32c(uint32_t crc, const void *data, int64_t n);

/* ---- pileup front end: alignments -> count tensors -------------------------
 * Replaces the body of dataPrepScripts/CreateTensor.py: the per-read CIGAR walk of
 * OutputAlnTensor (:140-246) and the per-candidate accumulation of GenerateTensor (:23-54).
 * The host parses SAM text (what `samtools view -F 2308` prints, :128-130) into alignment
 * segments; a scatter kernel adds every alignment column to the counters of the candidates whose
 * 33-position window holds it; a finalize kernel writes the [n,33,4,4] tensors in HBM, either raw
 * (what CreateTensor.py prints with "%0.1f", :52) or already with matrices 1..3 minus matrix 0
 * (what utils_v2.GetTensor hands to the network, utils_v2.py:46) so they can feed cv_forward
 * without the text round trip.  Not modelled: the reference's 10 000 000-column buffering cap
 * (`availableSlots`, :96) which silently drops columns in extremely deep regions.             */
typedef struct cv_pileup cv_pileup;

/* min_mq: --minMQ (:153); dcov: --dcov, reads beyond that many sharing one POS are skipped
 * (:165-172); consider_left_edge: --considerleftedge (:63-71).                                 */
int cv_pileup_create(int device, int min_mq, int dcov, int consider_left_edge, cv_pileup **out);
void cv_pileup_destroy(cv_pileup *p);

/* Reference bases as `samtools faidx` printed them (case and N kept: only upper-case ACGT count,
 * :28-31).  seq[0] is the 0-based contig position `first_pos0` (refStart-1, :99-104).  Copied.  */
int cv_pileup_set_reference(cv_pileup *p, const char *seq, int64_t len, int64_t first_pos0);

/* Candidate centres: 1-based positions, strictly ascending (GetCandidate :56-62 after its contig /
 * range filter).  Allocates and zeroes the device counters.  Copied.                           */
int cv_pileup_set_candidates(cv_pileup *p, const int64_t *centers, int64_t n);

/* Parse whole SAM lines from text[0..nbytes); *consumed = bytes up to the last complete line (feed
 * the rest again with the next chunk; pass final != 0 to take an unterminated last line too).
 * Header lines, reads below min_mq and reads beyond dcov are dropped exactly like :146-172; the
 * POS / depth-cap state carries over between calls.  *kept = reads queued by this call.          */
int cv_pileup_add_sam(cv_pileup *p, const char *text, int64_t nbytes, int final, int64_t *consumed,
                      int64_t *kept);

/* The same reads straight from BAM records (cv_bam_view_records): every record is taken as the line
 * `samtools view` prints for it (SEQ "*" for an absent sequence, no CIGAR operations for "*"; the RNAME test
 * of the candidate pass is the flag contig_ok: the view's contig is the one of cv_pileup_set_contig).
 * Same filters, same running state, same result as cv_pileup_add_sam on that text.                 */
int cv_pileup_add_bam(cv_pileup *p, const uint8_t *base, const uint32_t *offs, int64_t n, int contig_ok,
                      int64_t *kept);

/* Bases queued on the host and not yet scattered (callers flush when this gets large).          */
int64_t cv_pileup_pending(const cv_pileup *p);

/* Upload the queued segments and run the scatter kernel on `stream`; returns after the launch.   */
int cv_pileup_flush(cv_pileup *p, void *stream);

/* Flush, then write per candidate i: tensors_dev[i] = [33,4,4] fp32 counts (subtract != 0: matrices
 * 1..3 minus matrix 0), depth_dev[i] = aligned depth at the centre column (the --minCoverage test,
 * :51), touched_dev[i] = 1 iff some read was activated for it (only those get a row, :232-246).
 * Any output pointer may be NULL.  Counters stay valid: more reads may be added afterwards.     */
int cv_pileup_finish(cv_pileup *p, float *tensors_dev, int32_t *depth_dev, uint8_t *touched_dev,
                     int subtract, void *stream);

/* HIP-event time of the launches since creation: ms[0] scatter total, ms[1] finalize total, ms[2]
 * candidate pass (count + select); counts[0] alignment columns queued, counts[1] segments,
 * counts[2] scatter launches.                                                                   */
int cv_pileup_stats(cv_pileup *p, float ms[3], int64_t counts[3]);

/* ---- candidate extraction on the same alignments ------------------------------------------------
 * Replaces the body of dataPrepScripts/ExtractVariantCandidates.py (MakeCandidates :118-246,
 * OutputCandidate :22-42).  Options (before the first read): "evc" = 1 also books every read that
 * passes the candidate filters (contig, "evc_min_mq", >= 55 % aligned, :137-160) into per-position
 * counters A,C,G,T,I,D,N over the reference window; "retain" = 1 keeps the uploaded alignments in
 * HBM so that cv_pileup_adopt_candidates can run the tensor scatter over them again -- the SAM text
 * is parsed once for both steps (the reference pipes `samtools view` twice, callVarBam.py:116-131).
 * "threads" = host threads cv_pileup_add_sam may use on chunks of >= 1 MiB (the result does not depend
 * on it: records are parsed independently, the POS-run state is applied afterwards in read order).   */
int cv_pileup_set_option(cv_pileup *p, const char *key, int64_t value);
int cv_pileup_set_contig(cv_pileup *p, const char *name);        /* RNAME test of the candidate pass */

/* Flush, then select the positions OutputCandidate would print: total count >= min_coverage and
 * (top fraction <= 1 - threshold and second fraction >= threshold, or top symbol != reference base);
 * has_region: 0-based position in [ctg_start, ctg_end] as the reference compares them (:181-183, i.e.
 * ctg_start already +1); nbed >= 0: position inside one of the half-open BED intervals (:90-103), -1 =
 * no BED file.  Ties keep the order A,C,G,T,I,D,N (insertion order; see tests/golden/make_golden_evc.py).
 * *n_out = entries selected (a position may have a second, "late" entry, :215-241).               */
int cv_pileup_extract_candidates(cv_pileup *p, double threshold, double min_coverage, int has_region,
                                 int64_t ctg_start, int64_t ctg_end, const int64_t *bed_begin,
                                 const int64_t *bed_end, int64_t nbed, void *stream, int64_t *n_out);

/* The selected entries (host arrays of n_out elements; any may be NULL): 0-based position, 1 for a
 * late entry, the seven counts in A,C,G,T,I,D,N order; info[0] = reads the pass took (processedReads,
 * :150), info[1] = 0-based POS of the last of them (positions from there on are reported by the
 * reference's final loop, :215-241, together with the late entries).                            */
int cv_pileup_get_extracted(cv_pileup *p, int64_t *pos0, int32_t *late, int32_t *counts7, int64_t info[2]);

/* Make the selected positions (+1, optionally restricted to [lo1, hi1] like CreateTensor.py:60-61)
 * the candidate centres and scatter the retained alignments into their counters.               */
int cv_pileup_adopt_candidates(cv_pileup *p, int has_range, int64_t lo1, int64_t hi1, void *stream,
                               int64_t *n_out);

/* Zero the per-position counters and run the candidate-pass count kernel again over the retained
 * alignments (measurement: repeats the device pass without re-parsing; same result).            */
int cv_pileup_recount(cv_pileup *p, void *stream);

/* Current candidate centres (1-based); centers may be NULL to query the count.                  */
int cv_pileup_get_candidates(cv_pileup *p, int64_t *centers, int64_t cap, int64_t *n_out);

/* One text row of CreateTensor.py (:52): "<ctg> <center> <seq33> " + 528 x "%0.1f" (no newline).
 * counts: [33,4,4] fp32 raw counts (host).  Returns the length written, or -1 if cap is small.  */
int64_t cv_format_tensor_row(const char *ctg, int64_t center, const char *seq, int64_t seqlen,
                             const float *counts, char *dst, int64_t cap);

/* ---- native `samtools view` (optional producer, host only) ------------------------------------
 * The text of `samtools view -F <exclude_flags> BAM CTG[:S-E]` (CreateTensor.py:128-130,
 * ExtractVariantCandidates.py:112-114) without the external process: BGZF blocks inflated by `threads`
 * host threads, records of the region printed as SAM lines (11 columns, QUAL '*' unless with_qual;
 * no auxiliary tags).  Uses BAM.bai / .bai (linear index) when present, otherwise scans from the start.
 * Formats per the SAM/BAM specification; validated against files written by tests/bam_writer.py.      */
typedef struct cv_bam cv_bam;
int cv_bam_open(const char *path, int threads, cv_bam **out);
void cv_bam_close(cv_bam *b);
int cv_bam_nref(const cv_bam *b);
int cv_bam_ref(const cv_bam *b, int i, const char **name, int64_t *len);
int cv_bam_has_index(const cv_bam *b);
/* beg1 / end1: 1-based inclusive region, both <= 0 for the whole contig.                              */
int cv_bam_view_begin(cv_bam *b, const char *ref, int64_t beg1, int64_t end1, int exclude_flags, int with_qual);
/* Whole SAM lines into buf[0, cap); returns bytes written (0 and *done = 1 at the end), -1 on error.   */
int64_t cv_bam_view_read(cv_bam *b, char *buf, int64_t cap, int *done);
/* The same selection without the text: the next run of selected records (about max_bytes of inflated BAM);
 * returns their count, record i starts at *base + (*offs)[i] with its refID field (SAM/BAM specification 4.2).
 * The pointers stay valid until the next call on the handle; 0 and *done = 1 at the end, -1 on error.
 * Feeds cv_pileup_add_bam.                                                                             */
int64_t cv_bam_view_records(cv_bam *b, int64_t max_bytes, const uint8_t **base, const uint32_t **offs, int *done);
/* CIGAR words of a record handed out by cv_bam_view_records (rec at refID): the inline ones, or the CG:B,I array
 * behind the long-read placeholder <l_seq>S<span>N (more than 65535 operations, SAMv1 4.2.2).  0 ok.      */
int cv_bam_record_cigar(const uint8_t *rec, const uint8_t **ops, int64_t *n);
/* The block decoder behind the BGZF reader (raw DEFLATE, RFC 1951, whole block in memory): src[0, n) must be
 * followed by 8 readable bytes, the stream must produce exactly cap bytes; returns cap or -1.  And the CRC-32 of
 * the gzip trailer (start with crc = 0).                                                                 */
int64_t cv_inflate_raw(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap);
/* The same decoder over a gzip member too large to inflate at once (text-tensor files, utils_v2.GetTensor -- the
 * reference pipes them through `gzip -fdc`, utils_v2.py:25): src[0, n) = the raw-DEFLATE data behind the gzip header,
 * followed by >= 8 readable bytes; *bitpos = bit offset to go on from (0 first); dst[0, have) = the output produced
 * last (>= its last 32 768 bytes), new output is appended at dst + have up to dst + cap.  Returns at the first block
 * boundary with >= want new bytes or at the end of the stream (*final = 1; the CRC-32 / ISIZE trailer starts at the
 * byte boundary behind *bitpos): the number of new bytes, or -1 (malformed / a block that does not fit).          */
int64_t cv_inflate_stream(const uint8_t *src, int64_t n, int64_t *bitpos, uint8_t *dst, int64_t have, int64_t cap,
                          int64_t want, int32_t *final);
uint32_t cv_crc32_ieee(uint32_t crc, const uint8_t *p, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* CLAIRVOYANTE_AMD_H */
