// read_bench2.hip - which ingredient of the ingest kernel's staging loop costs read bandwidth?
// Emulates the wave-tile loop (flow-pipeline_amd/csrc/ingest.cuh, wtile_kernel) piece by piece:
//   F_DESC   tile bounds from an offsets array (2 uniform loads per tile, prefetched one tile ahead)
//   F_LANE   one offset load per lane per tile (prefetched one tile ahead)
//   F_UNAL   tiles start at the record's 16-byte-aligned address (not 1 KiB aligned), 72-byte records
//   F_READ   two per-lane LDS reads of the staged tile after the wait
//   hipcc --offload-arch=gfx950 -O3 -o read_bench2 read_bench2.hip && ./read_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
enum { F_DESC = 1, F_LANE = 2, F_UNAL = 4, F_READ = 8 };
constexpr int STRIDE = 5472;

template <int F>
__global__ __launch_bounds__(512) void k(const uint8_t* __restrict__ buf, const uint32_t* __restrict__ off, uint32_t ntiles, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint32_t tiles[8 * STRIDE / 4];
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t* tile = tiles + wave * (STRIDE / 4);
    const uint32_t stride = gridDim.x * 8;
    uint32_t acc = 0;
    uint32_t t = blockIdx.x * 8 + wave;
    auto bounds = [&](uint32_t tt, uint32_t& lo, uint32_t& hi) {
        if (tt >= ntiles) { lo = hi = 0; return; }
        if (F & F_DESC) { lo = off[tt * 64]; hi = off[tt * 64 + 64]; }
        else if (F & F_UNAL) { lo = tt * 4608u + ((tt * 40u) & 0x3f0u); hi = lo + 4608u; }
        else { lo = tt * 4608u; hi = lo + 4608u; }
    };
    auto dma = [&](uint32_t lo, uint32_t hi) {
        const uint32_t cbase = lo & ~15u, nbytes = hi - cbase;
        for (uint32_t o = lane * 16u; o < nbytes; o += 1024u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + cbase + o),
                                             (__attribute__((address_space(3))) void*)(tile + (o - lane * 16u) / 4u), 16, 0, 2);
    };
    uint32_t lo, hi, nlo, nhi, o0 = 0, n0 = 0;
    bounds(t, lo, hi);
    if ((F & F_LANE) && t < ntiles) o0 = off[t * 64 + lane];
    dma(lo, hi);
    bounds(t + stride, nlo, nhi);
    if ((F & F_LANE) && t + stride < ntiles) n0 = off[(t + stride) * 64 + lane];
    for (; t < ntiles; t += stride) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (F & F_READ) {
            const uint32_t pos = (F & F_LANE) ? (o0 - (lo & ~15u)) : lane * 72u;
            const uint32_t i = (pos >> 2) % (STRIDE / 4 - 2);
            acc ^= tile[i] ^ tile[i + 1];
        } else {
            acc ^= o0;
        }
        lo = nlo; hi = nhi; o0 = n0;
        dma(lo, hi);
        bounds(t + 2 * stride, nlo, nhi);
        n0 = 0;
        if ((F & F_LANE) && t + 2 * stride < ntiles) n0 = off[(t + 2 * stride) * 64 + lane];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int F>
static void run(const char* name, const uint8_t* d, const uint32_t* off, uint32_t ntiles, size_t bytes, uint32_t* out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float best = 1e9;
    for (int it = 0; it < 6; it++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<F>), dim3(512), dim3(512), 0, 0, d, off, ntiles, out);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-48s %.3f ms  %.0f GB/s\n", name, best, bytes / best / 1e6);
}

int main() {
    const uint32_t nrec = 16666624;  // multiple of 64
    const uint32_t ntiles = nrec / 64;
    std::vector<uint32_t> h(nrec + 1);
    uint32_t pos = 0;
    uint64_t s = 88172645463325252ull;
    for (uint32_t i = 0; i <= nrec; i++) { h[i] = pos; s ^= s << 13; s ^= s >> 7; s ^= s << 17; pos += (s & 1) ? 84 : 60; }
    const size_t bytes = h[nrec];
    uint8_t* d; uint32_t *off, *out;
    CHK(hipMalloc(&d, bytes + 65536)); CHK(hipMalloc(&off, (nrec + 1) * 4)); CHK(hipMalloc(&out, 4));
    CHK(hipMemset(d, 1, bytes + 65536));
    CHK(hipMemcpy(off, h.data(), (nrec + 1) * 4, hipMemcpyHostToDevice));
    const size_t synth = (size_t)ntiles * 4608;
    run<0>("fixed 4.5 KiB tiles", d, off, ntiles, synth, out);
    run<F_READ>("fixed + lds reads", d, off, ntiles, synth, out);
    run<F_UNAL>("unaligned fixed", d, off, ntiles, synth, out);
    run<F_UNAL | F_READ>("unaligned fixed + lds reads", d, off, ntiles, synth, out);
    run<F_DESC>("bounds from offsets", d, off, ntiles, bytes, out);
    run<F_DESC | F_READ>("bounds from offsets + lds reads", d, off, ntiles, bytes, out);
    run<F_DESC | F_LANE>("bounds + lane offsets", d, off, ntiles, bytes, out);
    run<F_DESC | F_LANE | F_READ>("bounds + lane offsets + lds reads", d, off, ntiles, bytes, out);
    return 0;
}
