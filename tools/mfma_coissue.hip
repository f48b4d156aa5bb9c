// Does the SIMD issue plain VALU work under a running v_mfma_f32_16x16x4_f32 (development probe)?
// K independent v_fma_f32 per MFMA in one instruction stream (inline asm keeps the order), 1 or 2
// waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/mfma_coissue.hip -o tools/mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K, int TRANS>
__global__ __launch_bounds__(256) void mix(float *out, int iters)
{
    f4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = a + i;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pv[4], pa = {a, b}, pb = {b, a};
    for (int i = 0; i < 4; i++) pv[i] = (f2){a + i, b + i};
    int iv = threadIdx.x & 3;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (TRANS == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[k & 3]) : "v"(pb), "v"(pa));
                    else if (TRANS == 3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(b), "v"(a));
                    else if (TRANS == 4) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[k & 7]) : "v"(iv));
                    else if (TRANS == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k & 7]) : "v"(b));
                    else if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k & 7]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(b), "v"(a));
                }
            }
    }
    float s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i];
    for (int i = 0; i < 4; i++) s += pv[i][0] + pv[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int K, int TRANS>
void run(int wgs, int iters)
{
    float *out;
    hipMalloc(&out, sizeof(float) * wgs * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<K, TRANS><<<wgs, 256>>>(out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix<K, TRANS><<<wgs, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nm = (double)iters * 32;                 // MFMAs per wave
    double waves_per_simd = wgs * 4.0 / 1024.0;
    printf("K=%d trans=%d wgs=%d (%.0f waves/SIMD): %.3f ms, %.1f ns per MFMA slot per SIMD (MFMA alone = 32 clk), MFMA rate %.1f TFLOP/s\n",
           K, TRANS, wgs, waves_per_simd, ms, ms * 1e6 / (nm * waves_per_simd), (double)wgs * 4 * nm * 2048.0 / ms / 1e9);
    hipFree(out);
}

int main()
{
    const int it = 4000;
    run<0, 0>(256, it); run<2, 0>(256, it); run<4, 0>(256, it); run<6, 0>(256, it); run<7, 0>(256, it); run<8, 0>(256, it);
    run<12, 0>(256, it);
    run<0, 0>(512, it); run<4, 0>(512, it); run<7, 0>(512, it); run<8, 0>(512, it); run<12, 0>(512, it);
    run<2, 1>(256, it); run<4, 1>(256, it);
    run<4, 2>(256, it); run<8, 2>(256, it); run<8, 2>(512, it);      // packed fp32 fma
    run<8, 3>(256, it); run<8, 4>(256, it); run<8, 5>(256, it);      // med3, ldexp, mov
    return 0;
}
