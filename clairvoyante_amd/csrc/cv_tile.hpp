// cv_tile.hpp -- what the tile-kernel translation units share: vector types, the MFMA / SELU / max helpers, the
// development wave stamps, launch helpers.  Included by cv_kernels_mfma.hip (forward path, data gradients, training
// forward) and cv_kernels_wgrad.hip (weight gradients); everything here is per translation unit (anonymous namespace).
#pragma once
#include "cv_internal.hpp"
#include "cv_math.hpp"
#include "cv_unpool.hpp"
#include <type_traits>
#include <functional>
#include <atomic>
#include <string.h>

typedef float f4 __attribute__((ext_vector_type(4)));

#if defined(CV_WG_STAMP) && defined(CV_TILE_STAMPS)
// Development build (tools/gpu_wave_stamps.sh): when and where the waves of the instrumented kernels ran -- 100 MHz
// wall clock at entry / exit, shader cycles in between, HW_ID, XCC_ID; one ring of 4 096 records per kernel id (the
// reader takes the newest launch).  ids: 0 wgrad_conv_cm (conv3), 1 conv3_rot (training), 2 conv_tm (conv3 data
// gradient), 3 dense_tm (training fc4), 4 dense_dgrad_unpool, 5 wgrad_dense_cm (fc4), 6 conv_tm (conv2 forward)
constexpr int CV_STAMP_KERNELS = 8;
__device__ unsigned long long cv_wg_stamp[CV_STAMP_KERNELS * 4096 * 4];
__device__ unsigned cv_wg_stamp_n[CV_STAMP_KERNELS];
extern "C" int cv_debug_wg_stamps(unsigned long long *out, unsigned *counts)
{
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(cv_wg_stamp_n), sizeof(cv_wg_stamp_n)) != hipSuccess) return 1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(cv_wg_stamp), sizeof(cv_wg_stamp)) != hipSuccess;
}
struct cv_stamp {
    unsigned long long t0, c0;
    __device__ __forceinline__ cv_stamp() : t0(__builtin_amdgcn_s_memrealtime()), c0(__builtin_amdgcn_s_memtime()) {}
    __device__ __forceinline__ void end(int kid) const
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime(), c1 = __builtin_amdgcn_s_memtime();
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
        if ((threadIdx.x & 63) == 0) {
            const unsigned i = atomicAdd(&cv_wg_stamp_n[kid], 1u) & 4095u;
            unsigned long long *o = cv_wg_stamp + ((size_t)kid * 4096 + i) * 4;
            o[0] = t0; o[1] = t1; o[2] = c1 - c0; o[3] = ((unsigned long long)xcc << 32) | hw;
        }
    }
};
#define CV_STAMP_BEGIN const cv_stamp cv_st;
#define CV_STAMP_END(cond, kid) do { if (cond) cv_st.end(kid); } while (0)
// -DCV_ROW_PHASES on top: where the cycles of a barrier-ring loop go, per wave (record id 7: cycles up to the end of the
// iteration's instruction issue / waiting for its own memory operations / waiting at the barrier, summed over the loop;
// the fourth word is the wave's index in its workgroup in its low byte and a fourth phase above it).  front2_tm (an
// inference pass never runs dense_dgrad_unpool, so the record id is free): producer half / conv2 over the chunk /
// both barriers / waiting for the raw X rows.  tools/gpu_row_phases.py prints the shares.
#ifdef CV_ROW_PHASES
#define CV_PHASE_BEGIN unsigned long long cv_ph[4] = {0, 0, 0, 0}; unsigned long long cv_pt = __builtin_amdgcn_s_memtime();
#define CV_PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); cv_ph[i] += t_ - cv_pt; cv_pt = t_; } while (0)
#define CV_PHASE_END(cond, w) do { if ((cond) && (threadIdx.x & 63) == 0) { const unsigned i_ = atomicAdd(&cv_wg_stamp_n[7], 1u) & 4095u; \
    unsigned long long *o_ = cv_wg_stamp + ((size_t)7 * 4096 + i_) * 4; o_[0] = cv_ph[0]; o_[1] = cv_ph[1]; o_[2] = cv_ph[2]; o_[3] = (unsigned long long)(w) | (cv_ph[3] << 8); } } while (0)
#define CV_PHASE_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#else
#define CV_STAMP_BEGIN
#define CV_STAMP_END(cond, kid) do { } while (0)
#endif
#ifndef CV_PHASE_BEGIN
#define CV_PHASE_BEGIN
#define CV_PHASE(i) do { } while (0)
#define CV_PHASE_END(cond, w) do { } while (0)
#endif
#ifndef CV_PHASE_DRAIN
#define CV_PHASE_DRAIN() do { } while (0)
#endif

namespace {

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f4 selu4(f4 v)
{
    const cvm::f2v a = cvm::selu2((cvm::f2v){v[0], v[1]}), b = cvm::selu2((cvm::f2v){v[2], v[3]});
    return (f4){a[0], a[1], b[0], b[1]};
}

// One v_max_f32.  fmaxf() costs two when the compiler cannot prove an operand canonical (a loop-carried running
// maximum, an MFMA result): in IEEE mode it first quiets signalling NaNs with v_max_f32 x, x, x.  The instruction
// itself already returns the non-NaN operand, which is all max-pooling needs.  Operands and results only travel
// between ordinary VALU instructions (no MFMA / memory hazard windows around the asm).
__device__ __forceinline__ float vmaxf(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max of three in one v_max3_f32 (same reasoning as vmaxf)
__device__ __forceinline__ f4 max3_4(f4 a, f4 b, f4 c)
{
    f4 r;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float v;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(v) : "v"(a[k]), "v"(b[k]), "v"(c[k]));
        r[k] = v;
    }
    return r;
}

__device__ __forceinline__ f4 max4(f4 a, f4 b)
{
    f4 r;
    r[0] = vmaxf(a[0], b[0]); r[1] = vmaxf(a[1], b[1]); r[2] = vmaxf(a[2], b[2]); r[3] = vmaxf(a[3], b[3]);
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

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// alpha-dropout of one fc4 value (selu.py:34-69): d4 = a*(h4*keep + alpha'*(1-keep)) + b; mk = a*keep is kept for the
// backward pass.  Counter-based stream of (seed, step, candidate, unit).
struct cv_dropout_args { float *d4, *amask; int nunits; float rate; uint64_t seed, step; int64_t cand0; };

__device__ __forceinline__ void dropout_value(float &v, float &mk, int unit, int nunits, int64_t cand, float rate, uint64_t seed,
                                              uint64_t step)
{
    mk = 1.0f;
    if (unit >= nunits) { v = 0.0f; mk = 0.0f; }
    else if (rate > 0.0f) {
        const float ap = -1.7580993408473766f;
        float q = 1.0f - rate;
        float a = sqrtf(1.0f / (q * ((1.0f - q) * (ap * ap) + 1.0f)));
        float b = 0.0f - a * ((1.0f - q) * ap);
        uint64_t ctr = (seed * 0x9E3779B97F4A7C15ull) ^ (step << 40) ^ (uint64_t)(cand * nunits + unit);
        ctr += 0x9E3779B97F4A7C15ull;
        ctr = (ctr ^ (ctr >> 30)) * 0xBF58476D1CE4E5B9ull;
        ctr = (ctr ^ (ctr >> 27)) * 0x94D2049BB133111Bull;
        ctr = ctr ^ (ctr >> 31);
        float u = (float)((uint32_t)(ctr >> 32) >> 8) * (1.0f / 16777216.0f);
        float keep = floorf(q + u);
        v = a * (v * keep + ap * (1.0f - keep)) + b;
        mk = a * keep;
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

bool arch_is(const cv_arch &a, int k0, int k1, int k2, int c0, int c1, int c2, int p0, int p1, int p2,
             int f4_, int f5_)
{
    return a.kh[0] == k0 && a.kh[1] == k1 && a.kh[2] == k2 && a.cout[0] == c0 && a.cout[1] == c1 &&
           a.cout[2] == c2 && a.pool[0] == p0 && a.pool[1] == p1 && a.pool[2] == p2 && a.fc4 == f4_ &&
           a.fc5 == f5_;
}

static inline bool is_full(const cv_arch &a) { return arch_is(a, 1, 2, 3, 16, 32, 48, 5, 4, 3, 336, 168); }
static inline bool is_slim(const cv_arch &a) { return arch_is(a, 1, 3, 5, 8, 16, 32, 1, 1, 1, 36, 18); }

static int device_cus(int *out)
{
    static std::atomic<int> cus_by_dev[64];
    int dev = 0;
    CV_HIP(hipGetDevice(&dev));
    int cus = cus_by_dev[dev & 63].load(std::memory_order_relaxed);
    if (cus == 0) {
        CV_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (cus <= 0) cus = 256;
        cus_by_dev[dev & 63].store(cus, std::memory_order_relaxed);
    }
    *out = cus;
    return 0;
}

}  // namespace
