// The inner loop of dense_tm<21,8,*,2> (fc4: 4 608 -> 336, two groups of 16 candidates per wave, weights through a
// three-slot LDS-DMA ring, one barrier per 16-deep k step) in miniature, ONCE per fp32 MFMA shape (development tool):
//   SHAPE 16: v_mfma_f32_16x16x4_f32   -- 21 output tiles x 2 groups, 168 MFMAs per k step (what the library ships)
//   SHAPE 32: v_mfma_f32_32x32x2_f32   -- 10 tiles of 32 outputs x 32 candidates (80 MFMAs) + one 16-wide remainder
//             (8 MFMAs of the 16x16x4 shape); the B operand (2 k x 32 candidates per register) is assembled from the
//             two groups' tile-major fragments with v_permlane16_swap / v_permlane32_swap (PERM 1) or ds_bpermute (PERM 0)
// Reports: kernel time, TFLOP/s, shader cycles of the k loop per wave and per 2 048 FLOP of SIMD time (ideal 32), and
// whether every output equals the ascending-k fmaf chain BIT FOR BIT (the property the whole build rests on).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_feed.hip -o mfma_feed && ./mfma_feed
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int NB = 21, WAVES = 8, NBP = 24, STAGE = NBP * 64, PER = NBP / WAVES, NOUT = 336;

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f16v mfma32(float a, float b, f16v c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// Weight layouts (one k step = 16 k values = NBP fragments of 64 lanes x 4 dwords; fragments 21..23 are padding):
//   SHAPE 16: fragment ob, lane (kq = lane >> 4, i = lane & 15), dword s  = W[16 kb + 4 s + kq][16 ob + i]
//   SHAPE 32: tile T < 10 -> fragments 2T, 2T + 1: fragment 2T + j, lane (kk = lane >> 5, i = lane & 31), dword s
//             = W[16 kb + 2 (4 j + s) + kk][32 T + i]; fragment 20 = the 16x16x4 layout of outputs 320..335
// Activations: tile-major as in the library, fragment (g, kb): lane (kq, c), dword s = act[16 g + c][16 kb + 4 s + kq]
template <int SHAPE, int PERM>
__global__ __launch_bounds__(WAVES * 64, 2) void feed(const f4 *__restrict__ in_tm, int KB, const f4 *__restrict__ wp,
                                                      float *__restrict__ out, int G, long long *__restrict__ cyc)
{
    extern __shared__ __attribute__((aligned(16))) f4 ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = (blockIdx.x * WAVES + wid) * 2;
    unsigned bo[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int gl = g + r < G ? g + r : G - 1;
        bo[r] = (unsigned)((((size_t)gl * KB) * 64 + lane) * sizeof(f4));
    }
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[2][NB];            // SHAPE 16: all of them; SHAPE 32: acc[.][20] only (the 16-wide remainder)
    f16v big[10];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int ob = 0; ob < NB; ob++) acc[r][ob] = zero;
#pragma unroll
    for (int t = 0; t < 10; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) big[t][i] = 0.f;
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
    auto load_frag_off = [&](unsigned byte_off) {
        f4 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(in_tm) : "memory");
        return v;
    };
    stage_async(0, 0);
    stage_async(KB > 1 ? 1 : 0, 1);
    f4 B[2];
#pragma unroll
    for (int r = 0; r < 2; r++) B[r] = load_frag_off(bo[r]);
#pragma unroll
    for (int r = 0; r < 2; r++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(B[r]) : : "memory");
    __syncthreads();
    // ds_bpermute source addresses of the B assembly (PERM 0): destination lane (kk, grp, c) of step t reads lane
    // (kq = 2 (t & 1) + kk, c) of group grp's fragment
    const int kk = lane >> 5, grp = (lane >> 4) & 1, c = lane & 15;
    const int src_even = ((0 + kk) * 16 + c) * 4, src_odd = ((2 + kk) * 16 + c) * 4;
    int slot = 0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int kb = 0; kb < KB; kb++) {
        const int ks = kb + 2 < KB ? kb + 2 : KB - 1;
        const int kn = kb + 1 < KB ? kb + 1 : KB - 1;
        int wslot = slot + 2; if (wslot >= 3) wslot -= 3;
        f4 Bn[2];
#pragma unroll
        for (int r = 0; r < 2; r++) Bn[r] = load_frag_off(bo[r] + (unsigned)kn * 1024u);
        stage_async(ks, wslot);
        const f4 *wl = ring + slot * STAGE + lane;
        if constexpr (SHAPE == 16) {
            constexpr int AB = 3;
#pragma unroll
            for (int ob = 0; ob < NB; ob += AB) {
                f4 A[AB];
#pragma unroll
                for (int j = 0; j < AB; j++) A[j] = wl[(ob + j) * 64];
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int j = 0; j < AB; j++)
#pragma unroll
                        for (int r = 0; r < 2; r++) acc[r][ob + j] = mfma4(A[j][s], B[r][s], acc[r][ob + j]);
            }
        } else {
            // B operands of the eight 32x32x2 steps: Bt[2 s] = rows (g0 kq0, g1 kq0, g0 kq1, g1 kq1) of dword s,
            // Bt[2 s + 1] = rows (g0 kq2, g1 kq2, g0 kq3, g1 kq3)
            float Bt[8];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if constexpr (PERM == 1) {
                    float x = B[0][s], y = B[1][s];
                    // odd 16-lane rows of x <-> even rows of y: x = (a0 b0 a2 b2), y = (a1 b1 a3 b3)
                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                    // upper half of x <-> lower half of y: x = (a0 b0 a1 b1), y = (a2 b2 a3 b3)
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                    Bt[2 * s] = x; Bt[2 * s + 1] = y;
                } else {
                    const int b0 = __float_as_int(B[0][s]), b1 = __float_as_int(B[1][s]);
                    const int e0 = __builtin_amdgcn_ds_bpermute(src_even, b0), e1 = __builtin_amdgcn_ds_bpermute(src_even, b1);
                    const int o0 = __builtin_amdgcn_ds_bpermute(src_odd, b0), o1 = __builtin_amdgcn_ds_bpermute(src_odd, b1);
                    Bt[2 * s] = __int_as_float(grp ? e1 : e0); Bt[2 * s + 1] = __int_as_float(grp ? o1 : o0);
                }
            }
#pragma unroll
            for (int T = 0; T < 10; T += 2) {            // two tiles in rotation: four fragments read ahead
                f4 A[2][2];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int j = 0; j < 2; j++) A[u][j] = wl[(2 * (T + u) + j) * 64];
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int u = 0; u < 2; u++) big[T + u] = mfma32(A[u][j][s], Bt[4 * j + s], big[T + u]);
            }
            const f4 A = wl[20 * 64];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int r = 0; r < 2; r++) acc[r][20] = mfma4(A[s], B[r][s], acc[r][20]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(3)" : "+v"(Bn[0]) : : "memory");
        asm volatile("" : "+v"(Bn[1]) : : "memory");
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 2; r++) B[r] = Bn[r];
        slot = slot + 1 == 3 ? 0 : slot + 1;
    }
    const long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) cyc[blockIdx.x * WAVES + wid] = t1 - t0;
    // raw accumulators to out[cand][336] (natural layout: the probe checks values, not the store)
    if constexpr (SHAPE == 16) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (g + r >= G) break;
#pragma unroll
            for (int ob = 0; ob < NB; ob++)
#pragma unroll
                for (int s = 0; s < 4; s++)          // D of 16x16x4: lane (q, cand), dword s = row 4 q + s
                    out[((size_t)(g + r) * 16 + (lane & 15)) * NOUT + 16 * ob + 4 * (lane >> 4) + s] = acc[r][ob][s];
        }
    } else {
        const int col = lane & 31;                   // candidate (grp, c) of the pair of groups
        if (g + (col >> 4) < G) {
            float *o = out + ((size_t)(g + (col >> 4)) * 16 + (col & 15)) * NOUT;
#pragma unroll
            for (int T = 0; T < 10; T++)
#pragma unroll
                for (int i = 0; i < 16; i++)         // D of 32x32x2: dword i = row 8 (i / 4) + 4 (lane / 32) + i % 4
                    o[32 * T + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)] = big[T][i];
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (g + r >= G) break;
#pragma unroll
            for (int s = 0; s < 4; s++)
                out[((size_t)(g + r) * 16 + (lane & 15)) * NOUT + 320 + 4 * (lane >> 4) + s] = acc[r][20][s];
        }
    }
}

static float rnd(unsigned &h) { h = h * 1664525u + 1013904223u; return ((h >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

template <int SHAPE, int PERM>
static void run(const char *name, int G, int KB, const std::vector<float> &act, const std::vector<float> &W, int reps)
{
    const int K = 16 * KB;
    // pack
    std::vector<float> in((size_t)G * KB * 256), wp((size_t)KB * NBP * 256, 0.f);
    for (int g = 0; g < G; g++)
        for (int kb = 0; kb < KB; kb++)
            for (int l = 0; l < 64; l++)
                for (int s = 0; s < 4; s++)
                    in[(((size_t)g * KB + kb) * 64 + l) * 4 + s] = act[((size_t)g * 16 + (l & 15)) * K + 16 * kb + 4 * s + (l >> 4)];
    for (int kb = 0; kb < KB; kb++)
        for (int f = 0; f < NB; f++)
            for (int l = 0; l < 64; l++)
                for (int s = 0; s < 4; s++) {
                    float v;
                    if (SHAPE == 16 || f == 20) v = W[(size_t)(16 * kb + 4 * s + (l >> 4)) * NOUT + 16 * f + (l & 15)];
                    else { const int T = f >> 1, j = f & 1; v = W[(size_t)(16 * kb + 2 * (4 * j + s) + (l >> 5)) * NOUT + 32 * T + (l & 31)]; }
                    wp[(((size_t)kb * NBP + f) * 64 + l) * 4 + s] = v;
                }
    float *din, *dwp, *dout; long long *dcyc;
    const int wgs = (G + 2 * WAVES - 1) / (2 * WAVES);
    hipMalloc(&din, in.size() * 4); hipMalloc(&dwp, wp.size() * 4); hipMalloc(&dout, (size_t)G * 16 * NOUT * 4);
    hipMalloc(&dcyc, sizeof(long long) * wgs * WAVES);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dwp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, (size_t)G * 16 * NOUT * 4);
    auto k = feed<SHAPE, PERM>;
    const size_t lds = 3 * STAGE * sizeof(f4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<<<wgs, WAVES * 64, lds>>>((const f4 *)din, KB, (const f4 *)dwp, dout, G, dcyc);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); return; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) k<<<wgs, WAVES * 64, lds>>>((const f4 *)din, KB, (const f4 *)dwp, dout, G, dcyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> cy(wgs * WAVES);
    hipMemcpy(cy.data(), dcyc, cy.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (auto v : cy) cs += (double)v; cs /= cy.size();
    // bitwise check of sampled outputs against the ascending-k fmaf chain
    std::vector<float> o((size_t)G * 16 * NOUT);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    long long bad = 0, checked = 0; double maxd = 0;
    unsigned h = 99;
    for (int t = 0; t < 40000; t++) {
        h = h * 1664525u + 1013904223u; const int cand = (h >> 8) % (G * 16);
        h = h * 1664525u + 1013904223u; const int j = t < 336 ? t : (h >> 8) % NOUT;
        float a = 0.f;
        for (int kx = 0; kx < K; kx++) a = fmaf(act[(size_t)cand * K + kx], W[(size_t)kx * NOUT + j], a);
        const float got = o[(size_t)cand * NOUT + j];
        checked++;
        if (memcmp(&a, &got, 4) != 0) { bad++; const double d = fabs((double)a - got); if (d > maxd || d != d) maxd = d; }
    }
    const double flop = 2.0 * G * 16 * (double)K * NOUT;
    printf("%-34s G=%d KB=%d  %.3f ms  %6.1f TFLOP/s  loop %.0f cyc/wave = %.1f cyc per k step = %.2f cyc of SIMD time per 2048 FLOP (ideal 32)  "
           "fmaf-chain mismatches %lld / %lld (max |d| %.3g)\n",
           name, G, KB, ms, flop / ms / 1e9, cs, cs / KB, cs / KB / (168.0 * 2.0), bad, checked, maxd);
    hipFree(din); hipFree(dwp); hipFree(dout); hipFree(dcyc);
}

int main(int argc, char **argv)
{
    // bitwise check on a small problem (host fmaf chains are slow), timing on the inference launch (4 096 groups, KB 288)
    for (int pass = 0; pass < 2; pass++) {
        const int G = pass == 0 ? 64 : 4096, KB = pass == 0 ? 288 : 288, reps = pass == 0 ? 1 : 8;
        if (pass == 1 && argc > 1 && !strcmp(argv[1], "small")) break;
        const int K = 16 * KB;
        std::vector<float> act((size_t)G * 16 * K), W((size_t)K * NOUT);
        unsigned h = 1234567u;
        for (auto &v : act) v = rnd(h);
        for (auto &v : W) v = rnd(h) * 0.05f;
        printf("== %s\n", pass == 0 ? "small launch (every wave checked)" : "inference-size launch (65 536 candidates)");
        run<16, 0>("16x16x4 (shipped loop)", G, KB, act, W, reps);
        run<32, 1>("32x32x2, B via permlane swaps", G, KB, act, W, reps);
        run<32, 0>("32x32x2, B via ds_bpermute", G, KB, act, W, reps);
        run<16, 0>("16x16x4 (again)", G, KB, act, W, reps);
    }
    return 0;
}
