// cu_mask_probe.hip -- which compute units a stream made with hipExtStreamCreateWithCUMask dispatches to on this device.
// Every workgroup of a 8 192-workgroup launch records (XCC_ID, HW_ID) of its wave; the program prints, per mask, the
// number of distinct (xcc, se, sh, cu) places and the workgroups per XCC.
//   hipcc --offload-arch=gfx950 -O2 -o tools/cu_mask_probe tools/cu_mask_probe.hip && tools/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void where(unsigned long long *out, int spin)
{
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = ((unsigned long long)xcc << 32) | hw | (a == 12345.f ? 1u << 31 : 0u);
}
int main()
{
    const int NB = 8192;
    unsigned long long *d; CK(hipMalloc(&d, sizeof(*d) * NB));
    std::vector<unsigned long long> h(NB);
    struct { const char *name; uint32_t m[8]; } masks[] = {
        {"all 256 bits", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"bits 0..127", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
        {"bits 0..63", {~0u, ~0u, 0, 0, 0, 0, 0, 0}},
        {"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}},
        {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
        {"bits 128..255", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
        {"low half of every 32", {0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu, 0xffffu}},
    };
    for (auto &mk : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mk.m);
        if (e != hipSuccess) { printf("%-22s: %s\n", mk.name, hipGetErrorString(e)); continue; }
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        where<<<NB, 64, 0, st>>>(d, 10);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        where<<<NB, 64, 0, st>>>(d, 20000);
        CK(hipEventRecord(b, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), d, sizeof(*d) * NB, hipMemcpyDeviceToHost));
        std::set<unsigned long long> places; int per_xcc[16] = {0};
        for (int i = 0; i < NB; i++) {
            const unsigned hw = (unsigned)h[i], xcc = (unsigned)(h[i] >> 32) & 15u;
            const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            places.insert(((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu);
            per_xcc[xcc]++;
        }
        printf("%-22s: %3zu distinct (xcc, se, sh, cu), %.2f ms for the spin launch; workgroups per xcc:", mk.name, places.size(), ms);
        for (int x = 0; x < 8; x++) printf(" %d", per_xcc[x]);
        printf("\n");
        CK(hipStreamDestroy(st));
    }
    return 0;
}
