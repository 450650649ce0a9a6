// Does gfx950 LDS serve unaligned ds_read_b64 / ds_read_b32 correctly (ROCm sets
// SH_MEM_CONFIG.alignment_mode = unaligned)?  Prints per-offset verdicts and a
// throughput comparison against the aligned 3-dword + v_alignbyte window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void test(uint64_t* out, uint32_t* bad) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    uint32_t addr = (uint32_t)(uintptr_t)buf + threadIdx.x * 13;  // odd strides: every alignment
    uint64_t v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    uint64_t want = 0;
    for (int k = 0; k < 8; k++) want |= (uint64_t)buf[threadIdx.x * 13 + k] << (8 * k);
    out[threadIdx.x] = v;
    if (v != want) atomicAdd(bad, 1u);
}
template <int MODE>
__global__ void bw(uint32_t* sink, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i * 2654435761u;
    __syncthreads();
    uint32_t pos = threadIdx.x * 83, acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t w;
        if (MODE == 0) {
            uint32_t i = pos >> 2, sh = pos & 3;
            uint32_t d0 = buf[i], d1 = buf[i + 1], d2 = buf[i + 2];
            w = (uint64_t)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32 | __builtin_amdgcn_alignbyte(d1, d0, sh);
        } else {
            uint32_t addr = (uint32_t)(uintptr_t)buf + pos;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr) : "memory");
        }
        acc += (uint32_t)w ^ (uint32_t)(w >> 32);
        pos = (pos + 1 + ((uint32_t)w & 7)) & 16383;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    uint64_t* out; uint32_t* bad; uint32_t* sink;
    hipMalloc(&out, 256 * 8); hipMalloc(&bad, 4); hipMemset(bad, 0, 4); hipMalloc(&sink, 4 * 256 * 2048);
    hipLaunchKernelGGL(test, dim3(1), dim3(256), 0, 0, out, bad);
    uint32_t hb = 99; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("unaligned ds_read_b64: %u of 256 lanes wrong (%s)\n", hb, hipGetErrorString(hipGetLastError()));
    for (int mode = 0; mode < 2; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        if (mode == 0) hipLaunchKernelGGL(bw<0>, dim3(2048), dim3(256), 0, 0, sink, 200);
        else hipLaunchKernelGGL(bw<1>, dim3(2048), dim3(256), 0, 0, sink, 200);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(bw<0>, dim3(2048), dim3(256), 0, 0, sink, 2000);
        else hipLaunchKernelGGL(bw<1>, dim3(2048), dim3(256), 0, 0, sink, 2000);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s window: %.3f ms for %.1f G windows -> %.1f G windows/s\n", mode ? "unaligned ds_read_b64" : "aligned+alignbyte", ms,
               2048.0 * 256 * 2000 / 1e9, 2048.0 * 256 * 2000 / ms / 1e6);
    }
    return 0;
}
