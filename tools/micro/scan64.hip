// Microbenchmark: scanning a table of 64-byte slots (window-close extraction) - per-lane slot reads vs coalesced reads.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint4* tab, uint64_t nslots, unsigned long long* out) {
    unsigned long long acc = 0;
    const uint64_t nthr = (uint64_t)gridDim.x * blockDim.x, t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (MODE == 0) {  // lane = slot: four 16-byte loads at stride 64 B
        for (uint64_t i = t; i < nslots; i += nthr) {
            const uint4* q = tab + i * 4;
            const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
            acc += a.x + b.y + c.z + d.w;
        }
    } else if (MODE == 1) {  // coalesced: consecutive lanes read consecutive 16-byte pieces
        for (uint64_t i = t; i < nslots * 4; i += nthr) { const uint4 a = tab[i]; acc += a.x + a.w; }
    } else if (MODE == 2) {  // coalesced, 4 loads in flight per lane
        for (uint64_t i = t; i < nslots * 4; i += 4 * nthr) {
            const uint4 a = tab[i], b = tab[min(i + nthr, nslots * 4 - 1)], c = tab[min(i + 2 * nthr, nslots * 4 - 1)], d = tab[min(i + 3 * nthr, nslots * 4 - 1)];
            acc += a.x + b.y + c.z + d.w;
        }
    } else if (MODE == 3) {  // lane = slot, only the 16 bytes that hold w2,w3
        for (uint64_t i = t; i < nslots; i += nthr) { const uint4 b = tab[i * 4 + 1]; acc += b.y + b.w; }
    }
    if (acc == 0x123456789ull) *out = acc;
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const uint64_t n = 1ull << lg;
    uint4* tab; unsigned long long* out;
    if (hipMalloc(&tab, n * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 8);
    (void)hipMemset(tab, 0, n * 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"lane = slot, 4 x 16 B", "coalesced 16 B", "coalesced, 4 in flight", "lane = slot, 16 B of 64"};
#define RUN(M, G)                                                                                         \
    {                                                                                                     \
        float best = 1e9;                                                                                 \
        for (int r = 0; r < 3; r++) {                                                                     \
            (void)hipEventRecord(e0);                                                                     \
            hipLaunchKernelGGL(k<M>, dim3(G), dim3(256), 0, 0, tab, n, out);                              \
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);                                      \
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;                   \
        }                                                                                                 \
        printf("2^%d slots  grid %5d  %-26s %8.3f ms  %7.1f GB/s\n", lg, G, names[M], best, n * 64 / best / 1e6); \
    }
    RUN(0, 1024) RUN(0, 4096) RUN(1, 1024) RUN(1, 4096) RUN(2, 2048) RUN(3, 4096)
    return 0;
}
