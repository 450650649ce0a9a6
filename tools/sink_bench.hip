// sink_bench.hip - micro-benchmark behind the flows_5m group-by sink design.
// Question: at what rate can random (uniform, high-cardinality) slots of a hash table be
// updated with three 64-bit sums per record, as a function of
//   - atomic scope: agent (memory-side on a multi-XCD part) vs workgroup (executed in the
//     issuing XCD's L2; needs one table replica per XCD, selected with HW_REG_XCC_ID),
//   - replica size (L2 = 4 MiB per XCD, MALL = 256 MiB),
//   - slot stride (64 B or 32 B),
//   - quad grouping (3 lanes of one instruction add the 3 words of one line).
//   hipcc --offload-arch=gfx950 -O3 -o sink_bench sink_bench.hip && ./sink_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31; return z;
}
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

template <int SCOPE>
__device__ __forceinline__ void add64(unsigned long long* p, unsigned long long v) {
    if (SCOPE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (SCOPE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// One record per lane; GROUPED: lanes 0..2 of each quad add the three words of the record of quad-lane q
// (4 rounds), i.e. one instruction touches each line once with 3 lanes.
template <int SCOPE, int SLOTW, bool GROUPED, bool READKEY>
__global__ __launch_bounds__(256) void upd(unsigned long long* tab, uint32_t slots_mask, uint64_t replica_words,
                                           uint64_t n, int replicas, unsigned long long* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long* base = tab;
    if (replicas > 1) base = tab + (uint64_t)(xcc_id() % replicas) * replica_words;
    unsigned long long acc = 0;
    const uint32_t w = threadIdx.x & 3;
    for (; i < n + (stride - n % stride) % stride; i += stride) {  // whole waves stay together
        const bool live = i < n;
        const uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
        unsigned long long* s = base + (uint64_t)((uint32_t)h & slots_mask) * SLOTW;
        if (READKEY && live) acc += s[0];
        const unsigned long long v0 = (h >> 8) & 1023, v1 = (h >> 9) & 1023, v2 = (h >> 10) & 1023;
        if (!GROUPED) {
            if (live) {
                add64<SCOPE>(&s[SLOTW - 3], v0);
                add64<SCOPE>(&s[SLOTW - 2], v1);
                add64<SCOPE>(&s[SLOTW - 1], v2);
            }
        } else {
            const uint64_t ptr = live ? (uint64_t)s : 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int src = (threadIdx.x & 60) | q;
                const uint64_t qp = (uint64_t)(uint32_t)__shfl((int)(uint32_t)ptr, src) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(ptr >> 32), src) << 32;
                const uint32_t a = (uint32_t)__shfl((int)(uint32_t)v0, src), b = (uint32_t)__shfl((int)(uint32_t)v1, src), c = (uint32_t)__shfl((int)(uint32_t)v2, src);
                const unsigned long long v = w == 0 ? a : w == 1 ? b : c;
                if (qp && w < 3) add64<SCOPE>((unsigned long long*)qp + (SLOTW - 3) + w, v);
            }
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}

__global__ void total(const unsigned long long* tab, uint64_t words, int slotw, unsigned long long* out) {
    unsigned long long a = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
        if ((int)(i % slotw) >= slotw - 3) a += tab[i];
    atomicAdd(out, a);
}

static unsigned long long g_want = 0;

template <int SCOPE, int SLOTW, bool GROUPED, bool READKEY>
static void run(const char* name, unsigned long long* tab, int slots_log2, int replicas, uint64_t n, unsigned long long* d_tmp, int blocks_per_cu) {
    const uint64_t words = ((uint64_t)SLOTW << slots_log2);
    CHK(hipMemset(tab, 0, words * 8 * replicas));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    dim3 g(256 * blocks_per_cu), b(256);
    hipLaunchKernelGGL((upd<SCOPE, SLOTW, GROUPED, READKEY>), g, b, 0, 0, tab, (1u << slots_log2) - 1, words, n / 8, replicas, d_tmp);
    CHK(hipDeviceSynchronize());
    CHK(hipMemset(tab, 0, words * 8 * replicas));
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((upd<SCOPE, SLOTW, GROUPED, READKEY>), g, b, 0, 0, tab, (1u << slots_log2) - 1, words, n, replicas, d_tmp);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipMemset(d_tmp + 1, 0, 8));
    hipLaunchKernelGGL(total, dim3(1024), dim3(256), 0, 0, tab, words * replicas, SLOTW, d_tmp + 1);
    unsigned long long got; CHK(hipMemcpy(&got, d_tmp + 1, 8, hipMemcpyDeviceToHost));
    printf("%-40s slot=%2dB slots=2^%-2d (%6.1f MB x%d) wg/cu=%d %8.3f ms %7.2f G rec/s  %s\n", name, SLOTW * 8, slots_log2,
           words * 8 / 1e6, replicas, blocks_per_cu, ms, n / ms / 1e6, got == g_want ? "sum ok" : "SUM MISMATCH");
    fflush(stdout);
}

__global__ void xcc_census(unsigned int* counts) { if (threadIdx.x == 0) atomicAdd(&counts[xcc_id() & 15], 1u); }

int main() {
    unsigned int* d_counts; CHK(hipMalloc(&d_counts, 64)); CHK(hipMemset(d_counts, 0, 64));
    hipLaunchKernelGGL(xcc_census, dim3(2048), dim3(64), 0, 0, d_counts);
    unsigned int hc[16]; CHK(hipMemcpy(hc, d_counts, 64, hipMemcpyDeviceToHost));
    printf("xcc census of 2048 blocks:"); for (int i = 0; i < 16; i++) printf(" %u", hc[i]); printf("\n");
    unsigned long long* tab; CHK(hipMalloc(&tab, (8ull << 21) * 8 * 8));  // 1 GiB
    unsigned long long* d_tmp; CHK(hipMalloc(&d_tmp, 64));
    const uint64_t n = 50000000;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t z = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
        g_want += ((z >> 8) & 1023) + ((z >> 9) & 1023) + ((z >> 10) & 1023);
    }
    printf("== agent scope (memory-side), one table ==\n");
    for (int sl : {14, 17, 19, 21}) {
        run<0, 8, false, false>("agent, 3 separate atomics", tab, sl, 1, n, d_tmp, 8);
        run<0, 8, true, false>("agent, quad-grouped (1 line txn/rec)", tab, sl, 1, n, d_tmp, 8);
    }
    run<0, 8, true, true>("agent, quad-grouped + key read", tab, 19, 1, n, d_tmp, 8);
    printf("== workgroup scope (XCD L2), 8 replicas selected by XCC_ID ==\n");
    for (int sl : {12, 14, 15, 16, 17, 18, 19, 20}) {
        run<1, 8, false, false>("wg x8, 3 separate atomics", tab, sl, 8, n, d_tmp, 8);
        run<1, 8, true, false>("wg x8, quad-grouped", tab, sl, 8, n, d_tmp, 8);
    }
    for (int sl : {15, 16, 17, 18, 19, 20}) {
        run<1, 4, true, false>("wg x8, quad-grouped, 32B slots", tab, sl, 8, n, d_tmp, 8);
    }
    for (int sl : {16, 18}) {
        run<1, 8, true, true>("wg x8, quad-grouped + key read", tab, sl, 8, n, d_tmp, 8);
        run<2, 8, true, false>("wavefront scope x8, quad-grouped", tab, sl, 8, n, d_tmp, 8);
        run<1, 8, true, false>("wg x8, quad-grouped, 4 wg/cu", tab, sl, 8, n, d_tmp, 4);
    }
    return 0;
}
