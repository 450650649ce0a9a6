// Microbenchmark: what does one random 64-byte-slot upsert cost on gfx950, step by step?  (tools/, not product code)
//   hipcc -O3 --offload-arch=gfx950 -o wide_rmw wide_rmw.hip && ./wide_rmw [log2 slots] [records]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
struct __attribute__((aligned(64))) S { unsigned long long w[4], v0, v1, v2, pad; };
__device__ __forceinline__ uint64_t mix64(uint64_t z) { z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31; return z; }
template <int MODE>
__global__ __launch_bounds__(256) void k(S* tab, uint64_t mask, uint64_t n, uint64_t seed, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = mix64(i + seed), h = mix64(key) & mask;
        S* s = &tab[h];
        const unsigned long long kw[4] = {key | 1ull << 63, ~key | 1ull << 63, (key >> 3) | 1ull << 63, (key >> 7) | 1ull << 63};
        if (MODE == 0) {  // two 16-byte loads
            const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&s->w[0]);
            const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(&s->w[2]);
            acc += a.x + a.y + b.x + b.y;
        } else if (MODE == 1) {  // loads + one CAS
            const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&s->w[0]);
            acc += a.y + atomicCAS(&s->w[0], a.x & 0ull, kw[0]);
        } else if (MODE == 2 || MODE == 3 || MODE == 6) {  // loads + four dependent CAS (+ three adds)
            const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(&s->w[0]);
            unsigned long long c = a.x & 0ull;
            for (int j = 0; j < 4; j++) c = atomicCAS(&s->w[j], c & 0ull, kw[j]);
            acc += c;
            if (MODE == 3) { atomicAdd(&s->v0, key & 1023); atomicAdd(&s->v1, 1 + (key & 7)); atomicAdd(&s->v2, 1ull); }
            if (MODE == 6) { __hip_atomic_fetch_add(&s->v0, key & 1023, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(&s->v1, 1 + (key & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(&s->v2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        } else if (MODE == 4) {  // one CAS, then plain stores of the rest of the key and the values (exclusive owner)
            const unsigned long long c = atomicCAS(&s->w[0], 0ull, kw[0]);
            if (c == 0) { s->w[1] = kw[1]; s->w[2] = kw[2]; s->w[3] = kw[3]; s->v0 = key & 1023; s->v1 = 1 + (key & 7); s->v2 = 1; }
            acc += c;
        } else if (MODE == 5) {  // plain 64-byte store only
            uint4* q = reinterpret_cast<uint4*>(s);
            q[0] = make_uint4((uint32_t)key, 1, 2, 3); q[1] = q[0]; q[2] = q[0]; q[3] = q[0];
        } else if (MODE == 7) {  // three adds only
            atomicAdd(&s->v0, key & 1023); atomicAdd(&s->v1, 1 + (key & 7)); atomicAdd(&s->v2, 1ull);
        }
    }
    if (acc == 0x1234567ull) *sink = acc;
}
int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    const uint64_t n = argc > 2 ? strtoull(argv[2], 0, 10) : 16666667ull;
    S* tab; unsigned long long* sink;
    if (hipMalloc(&tab, sizeof(S) << lg) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"2 loads", "load + 1 CAS", "load + 4 dependent CAS", "load + 4 CAS + 3 adds", "1 CAS + plain stores (new row)", "plain 64 B store", "load + 4 CAS + 3 agent-scope adds", "3 adds"};
#define RUN(M)                                                                                              \
    {                                                                                                       \
        hipMemset(tab, 0, sizeof(S) << lg);                                                                 \
        float best = 1e9;                                                                                   \
        for (int r = 0; r < 3; r++) {                                                                       \
            hipEventRecord(e0);                                                                             \
            hipLaunchKernelGGL(k<M>, dim3(256 * 8), dim3(256), 0, 0, tab, (1ull << lg) - 1, n, 1000ull * r, sink); \
            hipEventRecord(e1); hipEventSynchronize(e1);                                                    \
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;                           \
        }                                                                                                   \
        printf("2^%d slots, %llu records  %-36s %8.3f ms  %7.2f G rec/s\n", lg, (unsigned long long)n, names[M], best, n / best / 1e6); \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(6) RUN(4) RUN(5) RUN(7)
    return 0;
}
