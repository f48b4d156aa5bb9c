// cv_math.hpp -- device scalar math of the pileup-CNN path (gfx950).
//
// Every contraction in this library is one fp32 fused-multiply-add chain in
// ascending (kh, kw, ci) / ascending-k order, bias added after the chain
// (v_mfma_f32_16x16x4_f32 is bit-for-bit such a chain), and exp() is the fixed
// fmaf-only sequence below, so results are a pure function of the inputs --
// independent of tiling, wave count or GPU count.  Compile with
// -ffp-contract=off: the only fused operations are the explicit ones.
//
// selu follows /root/reference/clairvoyante/selu.py:21-25
//   scale * where(x >= 0, x, alpha * elu(x)),  elu(x) = exp(x) - 1
// (its negative branch is an own fixed sequence, see selu() below; exp() elsewhere -- sigmoid, softmax, selu' -- is
// expf_fixed)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvm {

__device__ __forceinline__ float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// Cephes-style expf with a fixed operation sequence (<= 2 ulp).
__device__ __forceinline__ float expf_fixed(float x)
{
    if (x != x) return x;
    if (x > 88.72283905206835f) return __builtin_inff();
    if (x < -87.33654475055310f) return 0.0f;
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
    y = y * bits2f((uint32_t)(n1 + 127) << 23);
    y = y * bits2f((uint32_t)(n2 + 127) << 23);
    return y;
}

constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;

constexpr float SELU_SA = (float)(1.0507009873554804934193349852946 * 1.6732632423543772848170429916717);

// SELU, branch-free (a wave never diverges on the sign of an activation).  The negative branch is its own fixed
// operation sequence (the CPU checker of the test suite states the same one): clamp at the flush threshold (below it exp(x) - 1
// rounds to -1 either way), ONE-constant range reduction r = x - z*fl(ln 2) (|z| <= 126), the Cephes polynomial of
// expf_fixed, y*2^z by one v_ldexp_f32 (the product is a normal number), and scale*alpha*(y - 1) as ONE fused
// multiply-add with the product constant -- 18 element operations per value where scale*(alpha*(expf_fixed(x)-1))
// took 21, a little MORE accurate (max |err| 1.2e-7 vs 2.0e-7 over all negative floats) and monotone over every
// fp32 input (cv_selu_sweep).  The select is `x < 0 ? neg : pos`, which routes NaN (and -0) to pos.
// The negative branch carries the SIGN of its input even where its value rounds to zero (-2^-25 < x < 0: exp(x) rounds to 1
// and the result is -0.0, computed as -(SA - SA*y)): the backward pass takes selu' from the layer OUTPUT (cv_unpool.hpp:
// sign bit set -> y + SA, else SCALE), and an output of +0 there would read as the x >= 0 branch -- one element in ~3e7,
// a flip of selu' from 1.758 to 1.051 that the reference (selu.py:21-25 through tf.where's gradient) does not have.
// (Development build flag CV_FAST_SELU: the negative branch through the hardware exponential, v_exp_f32(x * log2 e) and
// one fma -- 5 element operations instead of 18, ~1 ulp of 2^t instead of the fixed sequence: NOT the canonical
// arithmetic, no bitwise parity with the CPU checker; measured against the exact path by tools/gpu_fast_selu_ab.sh.)
__device__ __forceinline__ float selu(float x)
{
#ifdef CV_FAST_SELU
    {
        const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f);       // x <= -104: 0 (the fma then gives -SA)
        const float neg = __builtin_fmaf(e, SELU_SA, -SELU_SA);
        const float pos = SELU_SCALE * x;
        return x < 0.0f ? neg : pos;
    }
#endif
    const float xc = __builtin_amdgcn_fmed3f(x, -87.33654475055310f, 0.0f);
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
    y = __builtin_ldexpf(y, (int)z);
    const float mag = __builtin_fmaf(y, -SELU_SA, SELU_SA);      // scale*alpha*(1 - y) >= +0
    const float pos = SELU_SCALE * x;
    return x < 0.0f ? -mag : pos;                                // (the negation is a source modifier of the select)
}

// Two SELUs at once on the packed fp32 pipe (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two IEEE operations per
// instruction).  The operation sequence per element is selu()'s and each packed instruction rounds its two lanes
// exactly like the scalar one: bit-identical.  12 of the 18 instructions per pair are packed; measured, a packed
// instruction occupies the pipe like the two scalar ones it replaces (slim's kernels did not move when this went
// in), so what it buys is shorter code, not time -- the cost of exact fp32 VALU work is per element operation.
typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2v selu2(f2v x)
{
#ifdef CV_FAST_SELU
    {
        f2v o;
        o[0] = selu(x[0]); o[1] = selu(x[1]);
        return o;
    }
#endif
    f2v xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -87.33654475055310f, 0.0f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -87.33654475055310f, 0.0f);
    const f2v t = xc * 1.44269504088896341f;
    f2v z;
    z[0] = __builtin_rintf(t[0]); z[1] = __builtin_rintf(t[1]);
    const f2v r = __builtin_elementwise_fma(z, (f2v)(-0.69314718055994530942f), xc);
    const f2v r2 = r * r;
    f2v p = (f2v)(1.9875691500e-4f);
    p = __builtin_elementwise_fma(p, r, (f2v)(1.3981999507e-3f));
    p = __builtin_elementwise_fma(p, r, (f2v)(8.3334519073e-3f));
    p = __builtin_elementwise_fma(p, r, (f2v)(4.1665795894e-2f));
    p = __builtin_elementwise_fma(p, r, (f2v)(1.6666665459e-1f));
    p = __builtin_elementwise_fma(p, r, (f2v)(5.0000001201e-1f));
    f2v y = __builtin_elementwise_fma(p, r2, r);
    y = y + 1.0f;
    y[0] = __builtin_ldexpf(y[0], (int)z[0]);
    y[1] = __builtin_ldexpf(y[1], (int)z[1]);
    const f2v mag = __builtin_elementwise_fma(y, (f2v)(-SELU_SA), (f2v)(SELU_SA));
    const f2v pos = SELU_SCALE * x;
    f2v o;
    o[0] = x[0] < 0.0f ? -mag[0] : pos[0];
    o[1] = x[1] < 0.0f ? -mag[1] : pos[1];
    return o;
}

// d selu / d pre-activation
__device__ __forceinline__ float selu_grad(float pre)
{
    if (pre >= 0.0f) return SELU_SCALE;
    return SELU_SCALE * (SELU_ALPHA * expf_fixed(pre));
}

// tf.nn.sigmoid = 1 / (1 + exp(-x))
__device__ __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf_fixed(-x)); }

// tf.nn.softmax over n <= 6 logits: exp(l - max) / sum, sum in index order
template <int N>
__device__ __forceinline__ void softmax(const float (&l)[N], float (&p)[N])
{
    float m = l[0];
#pragma unroll
    for (int i = 1; i < N; i++) m = fmaxf(m, l[i]);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < N; i++) {
        p[i] = expf_fixed(l[i] - m);
        s = (i == 0) ? p[0] : s + p[i];
    }
#pragma unroll
    for (int i = 0; i < N; i++) p[i] = p[i] / s;
}

}  // namespace cvm
