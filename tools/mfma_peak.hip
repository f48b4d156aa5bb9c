// Practical fp32-MFMA ceiling and shader clock of the GPU box under load (development tool):
// independent v_mfma_f32_16x16x4_f32 chains, no memory traffic.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
// RANDOM: operands with random mantissas, different per lane / accumulator and changing sign every iteration (the
// accumulators stay bounded) -- the chip clocks to its power budget, and constant operands toggle few wires.
template <int NACC, bool RANDOM>
__global__ __launch_bounds__(256) void peak(float *out, long long *cyc, int iters)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a[NACC], b[NACC];
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < NACC; i++) {
        if (RANDOM) {
            h = h * 1664525u + 1013904223u; a[i] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
            h = h * 1664525u + 1013904223u; b[i] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
        } else { a[i] = threadIdx.x * 1e-3f; b[i] = 1.0f + blockIdx.x * 1e-6f; }
    }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[(i + r) % NACC], acc[i], 0, 0, 0);
        if (RANDOM)
#pragma unroll
            for (int i = 0; i < NACC; i++) a[i] = -a[i];
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, bool RANDOM>
void run(int wgs, int iters)
{
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * wgs * 256); hipMalloc(&cyc, sizeof(long long) * wgs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    peak<NACC, RANDOM><<<wgs, 256>>>(out, cyc, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    peak<NACC, RANDOM><<<wgs, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
    double flops = (double)wgs * 4 * iters * 4 * NACC * 2048.0;
    printf("%s operands nacc=%d wgs=%d iters=%d  %.3f ms  %.1f TFLOP/s  wave cycles(readcyclecounter)=%lld  => %.3f GHz-equivalent, %.2f cyc/MFMA\n",
           RANDOM ? "random  " : "constant", NACC, wgs, iters, ms, flops / ms / 1e9, c0, c0 / (ms * 1e6), (double)c0 / (iters * 4.0 * NACC));
    hipFree(out); hipFree(cyc);
}
// dependent-chain distance: NACC accumulators in rotation = NACC - 1 other MFMAs between two on the same accumulator,
// at 1, 2 and 4 waves per SIMD (256 CUs x 4 SIMDs: 256 / 512 / 1024 workgroups of 4 waves)
void chains()
{
    run<1, true>(256, 40000); run<2, true>(256, 40000); run<3, true>(256, 30000); run<4, true>(256, 20000); run<8, true>(256, 10000);
    run<1, true>(512, 40000); run<2, true>(512, 40000); run<3, true>(512, 30000); run<4, true>(512, 20000); run<8, true>(512, 10000);
    run<1, true>(1024, 40000); run<2, true>(1024, 40000);
}
int main(int argc, char **argv)
{
    if (argc > 1) { chains(); return 0; }
    run<8, false>(1024, 10000); run<8, true>(1024, 10000); run<8, true>(2048, 10000); run<8, true>(1024, 100000);
    run<4, true>(1024, 20000); run<8, false>(1024, 100000);
    return 0;
}
