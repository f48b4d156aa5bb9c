// cv_post.hip -- device side of callVar.Output, optimizer step and flat-buffer plumbing.
#include "cv_internal.hpp"
#include "cv_math.hpp"

namespace {

// Device part of callVar.Output (/root/reference/clairvoyante/callVar.py:59-87).
// One thread per candidate.
__global__ void call_postproc(const float *__restrict__ x, const float *__restrict__ out16, int64_t n,
                              int32_t *__restrict__ call, float *__restrict__ qual)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *o = out16 + (size_t)i * 16;
    // np.argmax: first maximum; np.sort()[::-1]: descending values
    auto top2 = [](const float *p, int cnt, int &am, float &v1, float &v2) {
        am = 0; v1 = p[0]; v2 = -__builtin_inff();
        for (int k = 1; k < cnt; k++) {
            float v = p[k];
            if (v > v1) { v2 = v1; v1 = v; am = k; }
            else if (v > v2) v2 = v;
        }
    };
    int at, az, al; float t1, t2, z1, z2, l1, l2;
    top2(o + 6, 4, at, t1, t2);
    top2(o + 4, 2, az, z1, z2);
    top2(o + 10, 6, al, l1, l2);
    // base[j].argsort()[::-1] : descending by (value, index) -- the HIGHER index
    // wins ties (callVar.py:81-83)
    int b1 = 0, b2 = -1;
    for (int k = 1; k < 4; k++)
        if (o[k] >= o[b1]) b1 = k;
    for (int k = 0; k < 4; k++) {
        if (k == b1) continue;
        if (b2 < 0 || o[k] >= o[b2]) b2 = k;
    }
    int32_t *c = call + (size_t)i * 8;
    c[0] = at; c[1] = az; c[2] = al; c[3] = b1; c[4] = b2; c[5] = 0; c[6] = 0; c[7] = 0;
    // qual operands: fp32 products of the sorted probabilities (callVar.py:69-72)
    float *q = qual + (size_t)i * 4;
    q[0] = (t1 * z1) * l1;
    q[1] = (t2 * z2) * l2;
    // dp = sum X[16,:,0] + sum X[17,:,1] + sum X[17,:,2] + sum X[16,:,3]  (callVar.py:86-87)
    const float *xi = x + (size_t)i * (CV_INPUT_H * 16);
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int b = 0; b < 4; b++) {
        s0 += xi[16 * 16 + b * 4 + 0];
        s1 += xi[17 * 16 + b * 4 + 1];
        s2 += xi[17 * 16 + b * 4 + 2];
        s3 += xi[16 * 16 + b * 4 + 3];
    }
    q[2] = ((s0 + s1) + s2) + s3;
    q[3] = 0.0f;
}

struct offs_t { int64_t o[CV_NUM_PARAMS + 1]; };

// TF1 AdamOptimizer update (/root/reference/clairvoyante/clairvoyante_v3.py:174) on
// g + lambda*w for kernels (the l2 term of v3.py:150), g for biases.
// acc != NULL (cv_apply_adam_accumulate): the first threads also add the loss header in front of the gradients to the
// accumulator -- cv_loss_accumulate's arithmetic (cv_train.hip t_loss_accumulate) without its launch
__device__ __forceinline__ void adam_one(float &wi, float &mi, float &vi, float gi, bool kernel, float lr_t, float lambda)
{
    if (kernel) gi = gi + lambda * wi;
    const float m1 = mi + (gi - mi) * (1.0f - 0.9f);
    const float v1 = vi + (gi * gi - vi) * (1.0f - 0.999f);
    mi = m1; vi = v1;
    wi = wi - (m1 * lr_t) / (sqrtf(v1) + 1e-8f);
}

// Four consecutive elements per thread through 16-byte loads / stores (the update moves 7 floats per element and is the
// last kernel of every step: 13 us as one element per thread, HBM-bound at ~46 MB); whether an element belongs to a
// kernel (lambda term) or a bias is decided per ELEMENT from the offsets, so the quadruples may straddle tensors.
__global__ void adam_kernel(float *__restrict__ w, float *__restrict__ mm, float *__restrict__ vv,
                            const float *__restrict__ g, offs_t offs, float lr_t, float lambda, double *__restrict__ acc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (acc && t < 7) {
        const float *hdr = g - CV_GRAD_HEADER;
        const int k = (int)t;
        if (k < 4) acc[k] += (double)hdr[2 * k] + (double)hdr[2 * k + 1];
        else if (k == 4) acc[4] += ((double)hdr[8] + (double)hdr[9]) / (double)(hdr[10] > 0.5f ? hdr[10] : 1.0f);
        else if (k == 6) acc[6] += 1.0;
    }
    const int64_t n = offs.o[CV_NUM_PARAMS];
    const int64_t i = t * 4;
    if (i >= n) return;
    int p = 0;
    while (i >= offs.o[p + 1]) p++;
    typedef float v4 __attribute__((ext_vector_type(4)));
    if (i + 3 < n) {
        v4 W = *reinterpret_cast<const v4 *>(w + i), M = *reinterpret_cast<const v4 *>(mm + i), V = *reinterpret_cast<const v4 *>(vv + i);
        const v4 G = *reinterpret_cast<const v4 *>(g + i);
#pragma unroll
        for (int e = 0; e < 4; e++) {
            while (i + e >= offs.o[p + 1]) p++;
            float a = W[e], b = M[e], c = V[e];
            adam_one(a, b, c, G[e], (p & 1) == 0, lr_t, lambda);
            W[e] = a; M[e] = b; V[e] = c;
        }
        *reinterpret_cast<v4 *>(w + i) = W; *reinterpret_cast<v4 *>(mm + i) = M; *reinterpret_cast<v4 *>(vv + i) = V;
        return;
    }
    for (int64_t j = i; j < n; j++) {
        while (j >= offs.o[p + 1]) p++;
        adam_one(w[j], mm[j], vv[j], g[j], (p & 1) == 0, lr_t, lambda);
    }
}

// Exhaustive monotonicity sweep of the canonical SELU over the negative floats (bit patterns 0x80000000 .. -inf):
// each thread walks RUN consecutive patterns (magnitude ascending = value descending) plus the first one of the next
// run and counts pairs with selu(next) > selu(prev); chk accumulates a checksum of the outputs so that the same
// sweep on the oracle can be compared.
__global__ void selu_sweep(uint32_t lo, uint32_t hi, uint32_t run, unsigned long long *__restrict__ out)
{
    const uint64_t first = (uint64_t)lo + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * run;
    if (first > hi) return;
    uint64_t last = first + run;
    if (last > hi) last = hi;
    float prev = cvm::selu(cvm::bits2f((uint32_t)first));
    unsigned long long viol = 0, chk = 0;
    for (uint64_t u = first + 1; u <= last; u++) {
        const float cur = cvm::selu(cvm::bits2f((uint32_t)u));
        viol += cur > prev ? 1ull : 0ull;
        chk += (unsigned long long)__builtin_bit_cast(uint32_t, cur);
        prev = cur;
    }
    if (viol) atomicAdd(&out[0], viol);
    atomicAdd(&out[1], chk);
}

}  // namespace

extern "C" int cv_selu_sweep(int device, uint32_t lo_bits, uint32_t hi_bits, uint64_t *violations, uint64_t *checksum)
{
    if (lo_bits > hi_bits || !violations) { cv_set_error("cv_selu_sweep: bad range"); return 1; }
    CV_HIP(hipSetDevice(device));
    unsigned long long *d = nullptr, h[2] = {0, 0};
    CV_HIP(hipMalloc(&d, sizeof(h)));
    CV_HIP(hipMemset(d, 0, sizeof(h)));
    const uint32_t run = 4096;
    const uint64_t threads = ((uint64_t)hi_bits - lo_bits) / run + 1;
    selu_sweep<<<(unsigned)((threads + 255) / 256), 256>>>(lo_bits, hi_bits, run, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { cv_set_error("cv_selu_sweep: %s", hipGetErrorString(e)); return 1; }
    *violations = h[0];
    if (checksum) *checksum = h[1];
    return 0;
}

extern "C" int cv_call_postproc(cv_model *m, const float *x_dev, const float *out16_dev, int64_t n,
                                int32_t *call_dev, float *qual_dev, void *stream)
{
    if (!m) { cv_set_error("cv_call_postproc: null model"); return 1; }
    if (n <= 0) return 0;
    if (!x_dev || !out16_dev || !call_dev || !qual_dev) { cv_set_error("cv_call_postproc: null buffer"); return 1; }
    CV_HIP(hipSetDevice(m->device));
    call_postproc<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x_dev, out16_dev, n, call_dev,
                                                                               qual_dev);
    CV_HIP(hipGetLastError());
    return 0;
}

extern "C" int cv_grad_buffer(cv_model *m, float **flat_dev, int64_t *count)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (flat_dev) *flat_dev = m->grads;
    if (count) *count = m->poff[CV_NUM_PARAMS];
    return 0;
}

extern "C" int cv_grad_bucket_info(const cv_model *m, int64_t *count, int64_t *header, int64_t *dense_begin)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (count) *count = CV_GRAD_HEADER + m->poff[CV_NUM_PARAMS];
    if (header) *header = CV_GRAD_HEADER;
    if (dense_begin) *dense_begin = CV_GRAD_HEADER + m->poff[6];
    return 0;
}

extern "C" int cv_bind_grad_bucket(cv_model *m, float *bucket_dev, int64_t count)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (!bucket_dev) { m->grads = m->grads_own + CV_GRAD_HEADER; return 0; }
    if (count != CV_GRAD_HEADER + m->poff[CV_NUM_PARAMS]) {
        cv_set_error("cv_bind_grad_bucket: %lld floats, the bucket holds %lld", (long long)count,
                     (long long)(CV_GRAD_HEADER + m->poff[CV_NUM_PARAMS]));
        return 1;
    }
    if (((uintptr_t)bucket_dev & 15) != 0) { cv_set_error("cv_bind_grad_bucket: the bucket must be 16-byte aligned"); return 1; }
    m->grads = bucket_dev + CV_GRAD_HEADER;
    return 0;
}

extern "C" int cv_adam_buffers(cv_model *m, float **m_dev, float **v_dev, int64_t *count)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (m_dev) *m_dev = m->adam_m;
    if (v_dev) *v_dev = m->adam_v;
    if (count) *count = m->poff[CV_NUM_PARAMS];
    return 0;
}

extern "C" int cv_flat_copy(cv_model *m, int which, float *caller_dev, int to_model, void *stream)
{
    if (!m || !caller_dev) { cv_set_error("cv_flat_copy: null argument"); return 1; }
    float *bufs[4] = {m->params, m->grads, m->adam_m, m->adam_v};
    if (which < 0 || which > 3) { cv_set_error("cv_flat_copy: which=%d not in 0..3", which); return 1; }
    CV_HIP(hipSetDevice(m->device));
    size_t bytes = sizeof(float) * m->poff[CV_NUM_PARAMS];
    if (to_model) {
        CV_HIP(hipMemcpyAsync(bufs[which], caller_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        if (which == 0) { cv_layouts_stale(m); }
    } else {
        CV_HIP(hipMemcpyAsync(caller_dev, bufs[which], bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return 0;
}

static int apply_adam(cv_model *m, float lr, float lambda, int64_t t, void *stream, bool accumulate)
{
    if (!m) { cv_set_error("null model"); return 1; }
    if (t < 1) { cv_set_error("cv_apply_adam: step count t must be >= 1"); return 1; }
    CV_HIP(hipSetDevice(m->device));
    offs_t offs;
    for (int i = 0; i <= CV_NUM_PARAMS; i++) offs.o[i] = m->poff[i];
    double lr_t = (double)lr * sqrt(1.0 - pow(0.999, (double)t)) / (1.0 - pow(0.9, (double)t));
    int64_t n = m->poff[CV_NUM_PARAMS];
    adam_kernel<<<(unsigned)((n + 1023) / 1024), 256, 0, (hipStream_t)stream>>>(m->params, m->adam_m, m->adam_v,
                                                                             m->grads, offs, (float)lr_t, lambda,
                                                                             accumulate ? m->loss_acc : nullptr);
    CV_HIP(hipGetLastError());
    cv_layouts_stale(m);
    return 0;
}

extern "C" int cv_apply_adam(cv_model *m, float lr, float lambda, int64_t t, void *stream)
{
    return apply_adam(m, lr, lambda, t, stream, false);
}

extern "C" int cv_apply_adam_accumulate(cv_model *m, float lr, float lambda, int64_t t, void *stream)
{
    return apply_adam(m, lr, lambda, t, stream, true);
}
