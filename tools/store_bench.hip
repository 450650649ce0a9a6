// store_bench.hip - what does the tuple scatter of the tile kernel cost, by write granularity?
//
// The tile kernel appends one 16-byte tuple per record to the (key partition, workgroup) segment
// (ingest.cuh, lane_work).  Ablation on MI355X: those stores are 0.145 ms of a 0.475 ms launch
// (16.67 M records).  This microbenchmark replays only the store pattern: G adjacent lanes append
// G*16 contiguous bytes to the same segment (G = 1 is today's pattern; G = 4 / 8 is what LDS-binned
// flushing of full 64 / 128-byte lines would emit), P partitions, persistent 256-thread workgroups.
//   hipcc --offload-arch=gfx950 -O3 -o store_bench store_bench.hip && ./store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31; return z;
}

template <int G>
__global__ __launch_bounds__(256) void ka(uint4* seg, uint32_t plog2, uint32_t capq, size_t region, uint64_t n) {
    __shared__ uint32_t cnt[1024];
    const uint32_t P = 1u << plog2, W = gridDim.x, w = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < P; i += 256) cnt[i] = 0;
    __syncthreads();
    const uint64_t ntiles = (n + 255) / 256;
    for (uint64_t t = w; t < ntiles; t += W) {
        const uint64_t i = t * 256 + threadIdx.x;
        const uint64_t r = mix64((i / G) * 0x9E3779B97F4A7C15ull + 777);
        const uint32_t p = (uint32_t)r >> (32 - plog2);
        uint32_t q = 0;
        if ((threadIdx.x % G) == 0) q = atomicAdd(&cnt[p], (uint32_t)G);
        q = __shfl(q, (threadIdx.x & 63) / G * G) + threadIdx.x % G;
        if (q < capq) seg[(size_t)p * region + (size_t)w * capq + q] = make_uint4((uint32_t)i, (uint32_t)(r >> 32), q, p);
    }
}

template <int G>
static void run(uint4* seg, size_t seg_tuples, uint64_t n, uint32_t plog2, int wgpc) {
    const uint32_t W = 256 * wgpc, P = 1u << plog2;
    const uint32_t capq = (uint32_t)((2 * (n / ((uint64_t)W * P)) + 32 + 7) & ~7ull);
    const size_t region = (size_t)W * capq + 24;
    if (region * P > seg_tuples) { printf("skip\n"); return; }
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 5; it++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(ka<G>, dim3(W), dim3(256), 0, 0, seg, plog2, capq, region, n);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("G=%2d (%4d B runs) P=%4u wg/cu=%d capq=%u : %.3f ms  %.1f GB/s  %.2f G tuples/s\n", G, G * 16, P, wgpc, capq, best,
           n * 16.0 / best / 1e6, n / best / 1e6);
}

int main() {
    const uint64_t n = 16666667;
    const size_t seg_tuples = (size_t)96 << 20;  // 1.5 GiB
    uint4* seg;
    CHK(hipMalloc(&seg, seg_tuples * 16));
    CHK(hipMemset(seg, 0, seg_tuples * 16));
    for (int plog2 : {6, 7, 8}) {
        for (int wgpc : {4, 6}) {
            run<1>(seg, seg_tuples, n, plog2, wgpc);
            run<2>(seg, seg_tuples, n, plog2, wgpc);
            run<4>(seg, seg_tuples, n, plog2, wgpc);
            run<8>(seg, seg_tuples, n, plog2, wgpc);
            run<16>(seg, seg_tuples, n, plog2, wgpc);
        }
    }
    return 0;
}
