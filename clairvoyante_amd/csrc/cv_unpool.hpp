// cv_unpool.hpp -- max-pool backward + SELU' of the training step on tile-major maps, from the window-offset codes
// the forward pass recorded (cv_kernels_mfma.hip, pool_code4).
//
// Layer with pooling window P over positions (clairvoyante_v3.py:59-67,74-82,89-97: SELU, then max_pooling2d (P,1)
// stride 1 VALID):  pooled[ho] = max_d act[ho + d],  act = selu(pre).  Backward, per channel / base / candidate:
//     gpre[h] = selu'(pre[h]) * sum over the windows ho = h-d (d = 0..P-1) whose FIRST maximum sits at offset d
//               of gpool[ho]
// A window's maximum IS the activation of the row it selects, so selu'(pre[h]) = selu'-from-output(pooled[ho]) for
// every window that selects h: the factor is applied to each window's gradient as it arrives,
//     gs[ho] = gpool[ho] * selu'-from-output(pooled[ho]),   gpre[h] = sum_{d = P-1 .. 0} [code[h-d] == d] gs[h-d]
// (windows in ascending order).  No pre-pool activation is read.  Differs from "sum first, multiply once" by one
// rounding where two or three windows select the same row.
#pragma once
#include "cv_math.hpp"

typedef float unp_f4 __attribute__((ext_vector_type(4)));

// d selu / d pre-activation expressed through the layer OUTPUT y = selu(pre):
//   pre >= 0  <=>  sign bit of y clear : SCALE ;   pre < 0 : SCALE*ALPHA*exp(pre) = y + SCALE*ALPHA
// (cvm::selu keeps the sign of a negative input on an output that rounds to zero: -0.0 is the x < 0 branch, +0.0 is x = 0.)
__device__ __forceinline__ float cv_selu_grad_from_out(float y)
{
    return (int32_t)__builtin_bit_cast(uint32_t, y) >= 0 ? cvm::SELU_SCALE : y + cvm::SELU_SCALE * cvm::SELU_ALPHA;
}

// one fragment column (fixed base w and tile): the last P window gradients and their 16-bit code groups
template <int P>
struct unpool_col {
    unp_f4 gs[P];          // gs[j]: window (newest - j), already times selu'
    unsigned cd[P];        // its codes: value r at bits 4r..4r+3 (15 = matches no offset)
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int j = 0; j < P; j++) { gs[j] = (unp_f4){0.f, 0.f, 0.f, 0.f}; cd[j] = 0xFFFFu; }
    }
    // window `ho` arrives: its pooled-output gradient g, pooled output y, codes c16
    __device__ __forceinline__ void push(unp_f4 g, unp_f4 y, unsigned c16)
    {
#pragma unroll
        for (int j = P - 1; j > 0; j--) { gs[j] = gs[j - 1]; cd[j] = cd[j - 1]; }
#pragma unroll
        for (int r = 0; r < 4; r++) gs[0][r] = g[r] * cv_selu_grad_from_out(y[r]);
        cd[0] = c16;
    }
    __device__ __forceinline__ void push_none()          // past the last window
    {
#pragma unroll
        for (int j = P - 1; j > 0; j--) { gs[j] = gs[j - 1]; cd[j] = cd[j - 1]; }
        gs[0] = (unp_f4){0.f, 0.f, 0.f, 0.f};
        cd[0] = 0xFFFFu;
    }
    // pre-activation gradient of the row the newest window starts at (row index = newest window index)
    __device__ __forceinline__ unp_f4 emit() const
    {
        unp_f4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float a = 0.0f;
#pragma unroll
            for (int d = P - 1; d >= 0; d--) a += ((cd[d] >> (4 * r)) & 15u) == (unsigned)d ? gs[d][r] : 0.0f;
            o[r] = a;
        }
        return o;
    }
};

// the 16 code bits of base w out of a lane's 64-bit code word
__device__ __forceinline__ unsigned cv_code16(unsigned lo, unsigned hi, int w)
{
    const unsigned v = w < 2 ? lo : hi;
    return (v >> (16 * (w & 1))) & 0xFFFFu;
}
