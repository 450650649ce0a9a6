// read_bench.hip - what does this MI355X deliver for a pure streaming read, by access style?
// (calibration for DESIGN.md's roofline discussion: the tile kernel's stage-only ablation reads at ~4.9 TB/s)
//   hipcc --offload-arch=gfx950 -O3 -o read_bench read_bench.hip && ./read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// persistent grid, 16 B per lane, UNROLL independent loads in flight per lane
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k_regs(const uint4* __restrict__ p, size_t n16, uint32_t* out) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (NT) {
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p + i + u * stride));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else {
                v[u] = p[i + u * stride];
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// DMA into LDS (global_load_lds_dwordx4), PIECES KiB per wave in flight, then wait; no barrier
template <int PIECES, int AUX>
__global__ __launch_bounds__(256) void k_lds(const uint8_t* __restrict__ p, size_t bytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[4 * PIECES * 256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* mine = buf + wave * PIECES * 256;
    const size_t chunk = (size_t)PIECES * 1024;
    const size_t nchunks = bytes / chunk;
    uint32_t acc = 0;
    for (size_t c = (size_t)blockIdx.x * 4 + wave; c < nchunks; c += (size_t)gridDim.x * 4) {
#pragma unroll
        for (int k = 0; k < PIECES; k++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + c * chunk + k * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(mine + k * 256), 16, 0, AUX);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        acc ^= mine[lane];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class F>
static void run(const char* name, size_t bytes, F launch) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 6; it++) {
        CHK(hipEventRecord(e0));
        launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
}

int main() {
    const size_t bytes = (size_t)1200 << 20;  // ~ one launch of config 2
    uint8_t* d; uint32_t* out;
    CHK(hipMalloc(&d, bytes)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(d, 1, bytes));
    const size_t n16 = bytes / 16;
    for (int wgpc : {4, 8}) {
        const int g = 256 * wgpc;
        char nm[96];
        snprintf(nm, sizeof nm, "regs x4        wg/cu=%d", wgpc); run(nm, bytes, [&] { hipLaunchKernelGGL((k_regs<4, false>), dim3(g), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "regs x8        wg/cu=%d", wgpc); run(nm, bytes, [&] { hipLaunchKernelGGL((k_regs<8, false>), dim3(g), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "regs x8 nt     wg/cu=%d", wgpc); run(nm, bytes, [&] { hipLaunchKernelGGL((k_regs<8, true>), dim3(g), dim3(256), 0, 0, (const uint4*)d, n16, out); });
        snprintf(nm, sizeof nm, "lds-dma 4 KiB  wg/cu=%d", wgpc); run(nm, bytes, [&] { hipLaunchKernelGGL((k_lds<4, 0>), dim3(g), dim3(256), 0, 0, d, bytes, out); });
        snprintf(nm, sizeof nm, "lds-dma 4 KiB nt wg/cu=%d", wgpc); run(nm, bytes, [&] { hipLaunchKernelGGL((k_lds<4, 2>), dim3(g), dim3(256), 0, 0, d, bytes, out); });
        if (wgpc == 4) { snprintf(nm, sizeof nm, "lds-dma 8 KiB nt wg/cu=%d", wgpc); run(nm, bytes, [&] { hipLaunchKernelGGL((k_lds<8, 2>), dim3(g), dim3(256), 0, 0, d, bytes, out); }); }
    }
    return 0;
}
