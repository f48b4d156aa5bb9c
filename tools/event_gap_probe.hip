// Development probe: what does a fork (hipEventRecord on the main stream + hipStreamWaitEvent on a side stream) or a
// join cost the MAIN stream's chain of dependent kernels?  Each kernel spins ~20 us and stamps its start / end with the
// wall clock (s_memrealtime, 100 MHz), so the gaps between consecutive kernels of the chain are measured on the device.
//   hipcc --offload-arch=gfx950 -O2 -o tools/event_gap_probe tools/event_gap_probe.hip && ./tools/event_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void spin(unsigned long long *stamp, int slot, int us, int blocks_stamp)
{
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long w0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[2 * slot] = w0;
    while (wall_clock64() - w0 < (unsigned long long)us * 100ull) { }
    if (blockIdx.x == 0 && threadIdx.x == 0) stamp[2 * slot + 1] = wall_clock64();
    (void)t0;
}

int main()
{
    const int N = 12, REP = 20;
    unsigned long long *stamp;
    CK(hipMalloc(&stamp, sizeof(unsigned long long) * 2 * 64));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(64);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const char *names[] = {"plain chain (same stream)", "event record between kernels (nobody waits)",
                           "fork: record + side stream waits and runs a kernel", "fork every kernel + join of the side kernel before the next",
                           "join only: side stream kernel (no fork), main waits for it before each kernel",
                           "fork: two side streams wait on the same event"};
    for (int mode = 0; mode < 6; mode++) {
        std::vector<double> gaps;
        for (int rep = 0; rep < REP; rep++) {
            int e = 0;
            for (int i = 0; i < N; i++) {
                if (mode == 3 && i > 0) CK(hipStreamWaitEvent(s1, ev[e - 1], 0));
                if (mode == 4) {
                    spin<<<64, 64, 0, s2>>>(stamp, 32 + (i & 15), 5, 0);
                    CK(hipEventRecord(ev[e], s2));
                    CK(hipStreamWaitEvent(s1, ev[e], 0));
                    e++;
                }
                spin<<<256, 64, 0, s1>>>(stamp, i, 20, 0);
                if (mode == 1) { CK(hipEventRecord(ev[e++], s1)); }
                if (mode == 2 || mode == 3 || mode == 5) {
                    CK(hipEventRecord(ev[e], s1));
                    CK(hipStreamWaitEvent(s2, ev[e], 0));
                    spin<<<64, 64, 0, s2>>>(stamp, 32 + (i & 15), 5, 0);
                    if (mode == 5) { CK(hipStreamWaitEvent(s3, ev[e], 0)); spin<<<64, 64, 0, s3>>>(stamp, 48 + (i & 15), 5, 0); }
                    e++;
                    if (mode == 3) { CK(hipEventRecord(ev[e], s2)); e++; }
                }
            }
            CK(hipDeviceSynchronize());
            unsigned long long h[2 * 64];
            CK(hipMemcpy(h, stamp, sizeof(h), hipMemcpyDeviceToHost));
            if (rep < 2) continue;
            for (int i = 1; i < N; i++) gaps.push_back((double)(h[2 * i] - h[2 * (i - 1) + 1]) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        printf("%-80s gap between chain kernels: median %5.2f us  p10 %5.2f  p90 %5.2f\n", names[mode], gaps[gaps.size() / 2],
               gaps[gaps.size() / 10], gaps[gaps.size() * 9 / 10]);
    }
    return 0;
}
