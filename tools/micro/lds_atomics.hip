// LDS atomic add throughput on random addresses of a 128 KiB array (what cms_agg_kernel does per sketch tuple):
// ds_add_u64 vs ds_add_u32 vs ds_add_rtn_u32 (+ rare carry) per CU, 16 waves per CU, one workgroup per CU.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mixu(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, int iters) {
    __shared__ unsigned long long arr[16384];
    uint32_t* arr32 = reinterpret_cast<uint32_t*>(arr);
    for (int i = threadIdx.x; i < 16384; i += 1024) arr[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
        x = mixu(x + it);
        const uint32_t w = (x >> 11) | 1u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t col = (r << 12) + ((x + r * 0x9E3779B1u) >> 20);
            if (MODE == 0) atomicAdd(&arr[col], (unsigned long long)w);
            else if (MODE == 1) atomicAdd(&arr32[2 * col], w);
            else if (MODE == 2) {
                const uint32_t old = atomicAdd(&arr32[2 * col], w);
                if (old + w < old) atomicAdd(&arr32[2 * col + 1], 1u);
            } else if (MODE == 3) {  // packed: two rows' 32-bit lows in one 64-bit word? (not equivalent; rate probe only)
                atomicAdd(&arr[col & ~1u], ((unsigned long long)w << 32) | w);
            }
        }
    }
    __syncthreads();
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < 16384; i += 1024) s += arr[i];
    if (s == 0x1234567ull) out[blockIdx.x] = s;
}
template <int MODE>
static void run(const char* name) {
    unsigned long long* out;
    hipMalloc(&out, 8 * 1024);
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double adds = 256.0 * 1024 * iters * 4;
    printf("%-28s %.3f ms  %.1f G lane-adds/s  (%.2f per CU per ns)\n", name, ms, adds / ms / 1e6, adds / ms / 1e6 / 256);
    hipFree(out);
}
int main() {
    run<0>("ds_add_u64");
    run<1>("ds_add_u32");
    run<2>("ds_add_rtn_u32 + carry");
    run<3>("ds_add_u64 (packed probe)");
    return 0;
}
