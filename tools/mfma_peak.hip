// Practical fp32-MFMA ceiling and shader clock of the GPU box under load (development tool):
// independent v_mfma_f32_16x16x4_f32 chains, no memory traffic.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void peak(float *out, long long *cyc, int iters)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int wgs, int iters)
{
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * wgs * 256); hipMalloc(&cyc, sizeof(long long) * wgs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    peak<NACC><<<wgs, 256>>>(out, cyc, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    peak<NACC><<<wgs, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
    double flops = (double)wgs * 4 * iters * 4 * NACC * 2048.0;
    printf("nacc=%d wgs=%d iters=%d  %.3f ms  %.1f TFLOP/s  wave cycles(readcyclecounter)=%lld  => %.3f GHz-equivalent, %.2f cyc/MFMA\n",
           NACC, wgs, iters, ms, flops / ms / 1e9, c0, c0 / (ms * 1e6), (double)c0 / (iters * 4.0 * NACC));
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<4>(1024, 20000); run<8>(1024, 10000); run<8>(2048, 10000); run<2>(1024, 40000); run<8>(512, 10000);
    return 0;
}
