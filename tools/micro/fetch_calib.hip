// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of this library's kernels.
// MI355X_MICROARCH.md ("HBM"): FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane);
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".
// Every kernel here touches a KNOWN number of bytes (printed as JSON on stdout: name -> bytes requested / distinct bytes
// touched per sector size); tools/fetch_calib_summary.py divides the counter by it.
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o fetch -- ./fetch_calib      (and a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31;
    return z;
}
#define SINK(acc) if ((acc) == 0x123456789abcdefull) *out = (acc)

// coalesced streaming reads, W bytes per lane and load
__global__ __launch_bounds__(256) void stream16(const uint4* p, uint64_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.w; }
    SINK(acc);
}
__global__ __launch_bounds__(256) void stream8(const uint2* p, uint64_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { const uint2 v = p[i]; acc += v.x ^ v.y; }
    SINK(acc);
}
__global__ __launch_bounds__(256) void stream4(const uint32_t* p, uint64_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) acc += p[i];
    SINK(acc);
}
// the distinct-address set's probe: 16 B + 8 B of a random 32-byte slot (sinks.cuh keyset_probe), table of `slots` slots
__global__ __launch_bounds__(256) void rand_slot32(const uint4* tab, uint64_t slots, uint64_t probes, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < probes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = mix64(i + 1) & (slots - 1);
        const uint4 a = tab[s * 2];
        const uint2 b = *reinterpret_cast<const uint2*>(&tab[s * 2 + 1]);
        acc += a.x ^ a.w ^ b.y;
    }
    SINK(acc);
}
// a whole random 64-byte line (the wide table's slot; keyset_step looks at two neighbouring 32-byte slots)
__global__ __launch_bounds__(256) void rand_line64(const uint4* tab, uint64_t lines, uint64_t probes, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < probes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = mix64(i + 7) & (lines - 1);
        const uint4 a = tab[s * 4], b = tab[s * 4 + 1], c = tab[s * 4 + 2], d = tab[s * 4 + 3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    SINK(acc);
}
// random 8-byte reads (Count-Min counters: fa_topk's estimates, the atomic paths' reads)
__global__ __launch_bounds__(256) void rand8(const unsigned long long* tab, uint64_t words, uint64_t probes, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < probes; i += (uint64_t)gridDim.x * blockDim.x) acc += tab[mix64(i + 3) & (words - 1)];
    SINK(acc);
}
// per-lane reads of consecutive 8-byte pairs (the offsets array: every lane loads off[r], off[r + 1] with one 8-byte load at a 4-byte stride)
__global__ __launch_bounds__(256) void stream_off8(const uint32_t* p, uint64_t n, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i + 1 < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint2 v;
        __builtin_memcpy(&v, p + i, 8);
        acc += v.x ^ v.y;
    }
    SINK(acc);
}
// writes: coalesced 16 B / lane; 64-byte store units at random places (the tuple segments); single 8-byte stores at random places
__global__ __launch_bounds__(256) void write16(uint4* p, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void write_unit64(uint4* p, uint64_t units, uint64_t stores) {  // 4 neighbouring lanes = one 64-byte unit at a random place
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < stores * 4; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t u = mix64((i >> 2) + 11) & (units - 1);
        p[u * 4 + (i & 3)] = make_uint4((uint32_t)i, 1, 2, 3);
    }
}
__global__ __launch_bounds__(256) void write_single8(uint2* p, uint64_t words, uint64_t stores) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < stores; i += (uint64_t)gridDim.x * blockDim.x) p[mix64(i + 13) & (words - 1)] = make_uint2((uint32_t)i, 5);
}

int main() {
    const uint64_t big = 4ull << 30, small = 32ull << 20;  // 4 GiB (far beyond the 256 MiB Infinity Cache), 32 MiB (a sketch: inside it)
    uint4* buf;
    unsigned long long* out;
    if (hipMalloc(&buf, big) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 1, big);
    (void)hipDeviceSynchronize();
    const dim3 g(4096), b(256);
    const uint64_t stream_bytes = 2ull << 30, probes = 64ull << 20;
    printf("{\n");
    hipLaunchKernelGGL(stream16, g, b, 0, 0, buf, stream_bytes / 16, out);
    printf(" \"stream16\": {\"bytes\": %llu, \"what\": \"coalesced 16 B/lane over 2 GiB\"},\n", (unsigned long long)stream_bytes);
    hipLaunchKernelGGL(stream8, g, b, 0, 0, (const uint2*)buf, stream_bytes / 8, out);
    printf(" \"stream8\": {\"bytes\": %llu, \"what\": \"coalesced 8 B/lane over 2 GiB\"},\n", (unsigned long long)stream_bytes);
    hipLaunchKernelGGL(stream4, g, b, 0, 0, (const uint32_t*)buf, stream_bytes / 4, out);
    printf(" \"stream4\": {\"bytes\": %llu, \"what\": \"coalesced 4 B/lane over 2 GiB\"},\n", (unsigned long long)stream_bytes);
    hipLaunchKernelGGL(stream_off8, g, b, 0, 0, (const uint32_t*)buf, stream_bytes / 4, out);
    printf(" \"stream_off8\": {\"bytes\": %llu, \"what\": \"8-byte loads at a 4-byte stride (the offsets array) over 2 GiB: every byte requested twice\"},\n", (unsigned long long)stream_bytes);
    hipLaunchKernelGGL(rand_slot32, g, b, 0, 0, buf, big / 32, probes, out);
    printf(" \"rand_slot32\": {\"probes\": %llu, \"bytes\": %llu, \"bytes_if_64B_lines\": %llu, \"what\": \"16 + 8 B of a random 32-byte slot in 4 GiB\"},\n",
           (unsigned long long)probes, (unsigned long long)(probes * 32), (unsigned long long)(probes * 64));
    hipLaunchKernelGGL(rand_line64, g, b, 0, 0, buf, big / 64, probes, out);
    printf(" \"rand_line64\": {\"probes\": %llu, \"bytes\": %llu, \"what\": \"a whole random 64-byte line in 4 GiB\"},\n", (unsigned long long)probes, (unsigned long long)(probes * 64));
    hipLaunchKernelGGL(rand8, g, b, 0, 0, (const unsigned long long*)buf, big / 8, probes, out);
    printf(" \"rand8_4GiB\": {\"probes\": %llu, \"bytes\": %llu, \"bytes_if_32B_sectors\": %llu, \"bytes_if_64B_lines\": %llu, \"what\": \"random 8-byte reads in 4 GiB\"},\n",
           (unsigned long long)probes, (unsigned long long)(probes * 8), (unsigned long long)(probes * 32), (unsigned long long)(probes * 64));
    hipLaunchKernelGGL(rand8, dim3(4095), b, 0, 0, (const unsigned long long*)buf, small / 8, probes, out);
    printf(" \"rand8_32MiB\": {\"probes\": %llu, \"bytes\": %llu, \"table_bytes\": %llu, \"what\": \"random 8-byte reads in 32 MiB (a sketch; grid 4095 tells it apart)\"},\n",
           (unsigned long long)probes, (unsigned long long)(probes * 8), (unsigned long long)small);
    hipLaunchKernelGGL(write16, g, b, 0, 0, buf, stream_bytes / 16);
    printf(" \"write16\": {\"bytes\": %llu, \"what\": \"coalesced 16 B/lane stores over 2 GiB\"},\n", (unsigned long long)stream_bytes);
    hipLaunchKernelGGL(write_unit64, g, b, 0, 0, buf, big / 64, probes / 4);
    printf(" \"write_unit64\": {\"bytes\": %llu, \"what\": \"64-byte store units (4 lanes x 16 B) at random places in 4 GiB\"},\n", (unsigned long long)(probes / 4 * 64));
    hipLaunchKernelGGL(write_single8, g, b, 0, 0, (uint2*)buf, big / 8, probes / 4);
    printf(" \"write_single8\": {\"bytes\": %llu, \"bytes_if_32B_sectors\": %llu, \"what\": \"single 8-byte stores at random places in 4 GiB\"}\n}\n",
           (unsigned long long)(probes / 4 * 8), (unsigned long long)(probes / 4 * 32));
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    return 0;
}
