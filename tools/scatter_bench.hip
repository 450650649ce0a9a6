// scatter_bench.hip - prototype of the key-partitioned flows_5m sink, on synthetic tuples.
//
// Global 64-bit atomics retire at ~23.7 G line requests/s on MI355X whatever the scope, table size
// or slot size (tools/sink_bench.hip), i.e. at most 23.7 G records/s for a one-request-per-record
// sink.  This prototype measures the alternative:
//   KA: every workgroup appends a 16-byte tuple per record to a PRIVATE segment per key partition
//       (p = top bits of the key hash; position from an LDS counter; no global atomics),
//   KB: partition p's segments are read back (coalesced) by one workgroup that aggregates them in an
//       LDS hash table and flushes the table to the device-wide table once.
//   hipcc --offload-arch=gfx950 -O3 -o scatter_bench scatter_bench.hip && ./scatter_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31; return z;
}
// cheap 32-bit key hash (2 quarter-rate multiplies + a few full-rate ops)
__host__ __device__ __forceinline__ uint32_t hash32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u + b;
    h ^= h >> 15; h = (h ^ c) * 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

struct Slot { unsigned long long k0, k1, bytes, packets, count, pad[3]; };

// synthetic record i -> (src_as, dst_as, meta, bytes, packets); G groups
__host__ __device__ __forceinline__ void synth(uint64_t i, uint32_t gmask, uint32_t& sa, uint32_t& da, uint32_t& et, uint32_t& tb,
                                               uint32_t& by, uint32_t& pk) {
    uint64_t r = mix64(i * 0x9E3779B97F4A7C15ull + 777);
    uint32_t g = (uint32_t)r & gmask;
    sa = 64512 + (g & 255); da = 64512 + ((g >> 8) & 255); et = (g >> 16) & 1 ? 0x86dd : 0x0800; tb = (g >> 17);
    by = (uint32_t)(r >> 32) % 1500; pk = (uint32_t)(r >> 48) % 100;
}

template <int MODE>  // 0: private segments (no global atomics)   1: direct global atomics (baseline, quad-grouped omitted)
__global__ __launch_bounds__(256) void ka(uint4* seg, uint32_t* counts, uint32_t plog2, uint32_t capq, size_t region, uint64_t n, uint32_t gmask,
                                          unsigned long long* ovf) {
    extern __shared__ uint32_t cnt[];  // P counters
    const uint32_t P = 1u << plog2, W = gridDim.x, w = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < P; i += 256) cnt[i] = 0;
    __syncthreads();
    const uint64_t ntiles = (n + 255) / 256;
    uint32_t lost = 0;
    for (uint64_t t = w; t < ntiles; t += W) {
        const uint64_t i = t * 256 + threadIdx.x;
        if (i < n) {
            uint32_t sa, da, et, tb, by, pk;
            synth(i, gmask, sa, da, et, tb, by, pk);
            const uint32_t h = hash32(sa, da, et | tb << 16);
            const uint32_t p = h >> (32 - plog2);
            const uint32_t q = atomicAdd(&cnt[p], 1u);
            if (q < capq) seg[(size_t)p * region + (size_t)w * capq + q] = make_uint4(sa, da, by | tb << 28, pk | et << 16);
            else lost++;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < P; i += 256) counts[(size_t)i * W + w] = min(cnt[i], capq);
    if (lost) atomicAdd(ovf, (unsigned long long)lost);
}

template <int SLOTS>
struct LdsTab { unsigned long long k0[SLOTS], k1[SLOTS], bytes[SLOTS], packets[SLOTS], count[SLOTS]; };

__device__ __forceinline__ void gl_add(Slot* tab, uint32_t mask, uint64_t k0, uint64_t k1, uint32_t h, uint64_t b, uint64_t p, uint64_t c) {
    uint32_t i = h & mask;
    for (int probe = 0; probe < 256; probe++, i = (i + 1) & mask) {
        Slot* s = &tab[i];
        unsigned long long c0 = s->k0;
        if (c0 == 0) c0 = atomicCAS(&s->k0, 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        unsigned long long c1 = s->k1;
        if (c1 == 0) c1 = atomicCAS(&s->k1, 0ull, (unsigned long long)k1);
        if (c1 != 0 && c1 != k1) continue;
        if (b) atomicAdd(&s->bytes, (unsigned long long)b);
        if (p) atomicAdd(&s->packets, (unsigned long long)p);
        atomicAdd(&s->count, (unsigned long long)c);
        return;
    }
}

// One 1024-thread workgroup per partition (16 waves share one LDS table).  Each wave takes SU consecutive
// segments at a time (counts via scalar loads) and keeps SU 16-byte loads per lane in flight.
template <int SLOTS, int SU>
__global__ __launch_bounds__(1024) void kb(const uint4* seg, const uint32_t* counts, uint32_t capq, size_t region, uint32_t W,
                                           Slot* tab, uint32_t mask, unsigned long long* fallback) {
    __shared__ LdsTab<SLOTS> lt;
    const uint32_t p = blockIdx.x;
    for (int i = threadIdx.x; i < SLOTS; i += 1024) { lt.k0[i] = 0; lt.k1[i] = 0; lt.bytes[i] = 0; lt.packets[i] = 0; lt.count[i] = 0; }
    __syncthreads();
    uint32_t nfb = 0;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* pbase = seg + (size_t)p * region;
    const uint32_t* pc = counts + (size_t)p * W;
    for (uint32_t w0 = wave * SU; w0 < W; w0 += 16 * SU) {
        uint32_t c[SU], cmax = 0;
#pragma unroll
        for (int s = 0; s < SU; s++) { c[s] = w0 + s < W ? pc[w0 + s] : 0; cmax = max(cmax, c[s]); }
        for (uint32_t k = 0; k * 64 < cmax; k++) {
            const uint32_t q = lane + 64 * k;
            uint4 t[SU];
#pragma unroll
            for (int s = 0; s < SU; s++) if (q < c[s]) t[s] = pbase[(size_t)(w0 + s) * capq + q];
#pragma unroll
            for (int s = 0; s < SU; s++) {
                if (q >= c[s]) continue;
                const uint32_t sa = t[s].x, da = t[s].y, by = t[s].z & 0x0fffffffu, tb = t[s].z >> 28, pk = t[s].w & 0xffffu, et = t[s].w >> 16;
                const uint64_t k0 = (1ull << 63) | ((uint64_t)(da & 0x7fffffffu) << 32) | sa;
                const uint64_t k1 = (1ull << 63) | ((uint64_t)(da >> 31) << 59) | ((uint64_t)tb << 32) | et;
                const uint32_t h = hash32(sa, da, et | tb << 16);
                uint32_t i = h & (SLOTS - 1);
                bool done = false;
#pragma unroll 1
                for (int probe = 0; probe < 8 && !done; probe++, i = (i + 1) & (SLOTS - 1)) {
                    unsigned long long c0 = lt.k0[i];
                    if (c0 == 0) c0 = atomicCAS(&lt.k0[i], 0ull, (unsigned long long)k0);
                    if (c0 != 0 && c0 != k0) continue;
                    unsigned long long c1 = lt.k1[i];
                    if (c1 == 0) c1 = atomicCAS(&lt.k1[i], 0ull, (unsigned long long)k1);
                    if (c1 != 0 && c1 != k1) continue;
                    if (by) atomicAdd(&lt.bytes[i], (unsigned long long)by);
                    if (pk) atomicAdd(&lt.packets[i], (unsigned long long)pk);
                    atomicAdd(&lt.count[i], 1ull);
                    done = true;
                }
                if (!done) { gl_add(tab, mask, k0, k1, h, by, pk, 1); nfb++; }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SLOTS; i += 1024) {
        unsigned long long k0 = lt.k0[i], k1 = lt.k1[i], c = lt.count[i];
        if (k0 && k1 && c) {
            uint32_t sa = (uint32_t)k0, da = (uint32_t)((k0 >> 32) & 0x7fffffffu) | (uint32_t)((k1 >> 59) & 1) << 31;
            uint32_t et = (uint32_t)k1 & 0xffff, tb = (uint32_t)(k1 >> 32) & 0xf;
            gl_add(tab, mask, k0, k1, hash32(sa, da, et | tb << 16), lt.bytes[i], lt.packets[i], c);
        }
    }
    if (nfb) atomicAdd(fallback, (unsigned long long)nfb);
}

__global__ void tab_total(const Slot* tab, uint32_t nslots, unsigned long long* out) {
    unsigned long long b = 0, p = 0, c = 0, g = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        if (tab[i].count) { b += tab[i].bytes; p += tab[i].packets; c += tab[i].count; g++; }
    }
    atomicAdd(&out[0], b); atomicAdd(&out[1], p); atomicAdd(&out[2], c); atomicAdd(&out[3], g);
}

// scattered plain stores / loads of SZ bytes per lane to random 64-byte slots (request-rate probe)
template <int SZ, int OP>
__global__ __launch_bounds__(256) void scatter_plain(uint32_t* tab, uint32_t slots_mask, uint64_t n, uint32_t* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) {
        uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
        uint32_t* s = tab + (uint64_t)((uint32_t)h & slots_mask) * 16;
        if (OP == 0) {
            if (SZ == 4) acc += *s;
            if (SZ == 8) { uint2 v = *(uint2*)s; acc += v.x ^ v.y; }
            if (SZ == 16) { uint4 v = *(uint4*)s; acc += v.x ^ v.w; }
        } else {
            if (SZ == 4) *s = (uint32_t)h;
            if (SZ == 8) *(uint2*)s = make_uint2((uint32_t)h, 1);
            if (SZ == 16) *(uint4*)s = make_uint4((uint32_t)h, 1, 2, 3);
        }
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// one atomic per lane to a random slot: 32-bit / 64-bit, returning or not
template <int BITS, bool RET>
__global__ __launch_bounds__(256) void scatter_atomic(uint32_t* tab, uint32_t slots_mask, uint64_t n, uint32_t* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (; i < n; i += stride) {
        uint64_t h = mix64(i * 0x9E3779B97F4A7C15ull + 12345);
        uint32_t* s = tab + (uint64_t)((uint32_t)h & slots_mask) * 16;
        if (BITS == 32) { uint32_t r = atomicAdd(s, 1u); if (RET) acc += r; }
        else { unsigned long long r = atomicAdd((unsigned long long*)s, 1ull); if (RET) acc += r; }
    }
    if (RET && acc == 0x1234567) sink[0] = (uint32_t)acc;
}

template <class F>
static float timeit(F f) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    f(); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); f(); CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 0) : 25000000ull;
    uint32_t* rtab; CHK(hipMalloc(&rtab, 64ull << 21));
    uint32_t* sink; CHK(hipMalloc(&sink, 64));
    const uint32_t rmask = (1u << 21) - 1;
    printf("== request-rate probes: %llu lane-ops to random 64 B slots of a 128 MB table ==\n", (unsigned long long)n * 2);
    const uint64_t m = n * 2;
#define PROBE(name, ...) { float ms = timeit([&] { hipLaunchKernelGGL(__VA_ARGS__); }); printf("%-44s %8.3f ms %8.2f G ops/s\n", name, ms, m / ms / 1e6); fflush(stdout); }
    PROBE("plain store 4 B", (scatter_plain<4, 1>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("plain store 8 B", (scatter_plain<8, 1>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("plain store 16 B", (scatter_plain<16, 1>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("plain load 4 B", (scatter_plain<4, 0>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("plain load 16 B", (scatter_plain<16, 0>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u32, no return", (scatter_atomic<32, false>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u64, no return", (scatter_atomic<64, false>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u32, returning", (scatter_atomic<32, true>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u64, returning", (scatter_atomic<64, true>), dim3(2048), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u64, 1024 blocks", (scatter_atomic<64, false>), dim3(1024), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u64, 256 blocks (1/CU)", (scatter_atomic<64, false>), dim3(256), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u64, 128 blocks (half the CUs)", (scatter_atomic<64, false>), dim3(128), dim3(256), 0, 0, rtab, rmask, m, sink);
    PROBE("atomic add u64, 64 blocks (quarter)", (scatter_atomic<64, false>), dim3(64), dim3(256), 0, 0, rtab, rmask, m, sink);
    {
        const uint32_t small = (1u << 12) - 1;  // 256 KB table: L2 resident in every XCD
        PROBE("atomic add u64, 256 KB table", (scatter_atomic<64, false>), dim3(2048), dim3(256), 0, 0, rtab, small, m, sink);
        PROBE("plain store 16 B, 256 KB table", (scatter_plain<16, 1>), dim3(2048), dim3(256), 0, 0, rtab, small, m, sink);
    }

    printf("== key-partitioned sink prototype: n = %llu records per launch ==\n", (unsigned long long)n);
    Slot* tab; const uint32_t tlog2 = 20; CHK(hipMalloc(&tab, sizeof(Slot) << tlog2));
    unsigned long long* d_u64; CHK(hipMalloc(&d_u64, 64));
    for (uint32_t glog2 : {17u, 18u}) {
        const uint32_t gmask = (1u << glog2) - 1;
        unsigned long long wb = 0, wp = 0;
        for (uint64_t i = 0; i < n; i++) { uint32_t a, b, c, d, by, pk; synth(i, gmask, a, b, c, d, by, pk); wb += by; wp += pk; }
        for (uint32_t wgpc : {4u, 5u, 6u}) {
            const uint32_t W = 256 * wgpc;
            for (uint32_t plog2 : {7u, 8u, 9u}) {
                const uint32_t P = 1u << plog2;
                uint32_t avg = (uint32_t)(n / ((uint64_t)W * P));
                uint32_t capq = (2 * avg + 32 + 3) & ~3u;
                for (uint32_t skew : {0u, 24u}) {
                const size_t region = (size_t)W * capq + skew;  // in tuples
                uint4* seg; uint32_t* counts;
                const size_t seg_bytes = (size_t)P * region * 16;
                CHK(hipMalloc(&seg, seg_bytes)); CHK(hipMalloc(&counts, (size_t)P * W * 4));
                CHK(hipMemset(d_u64, 0, 64));
                float ms_a = timeit([&] { hipLaunchKernelGGL(ka<0>, dim3(W), dim3(256), P * 4, 0, seg, counts, plog2, capq, region, n, gmask, d_u64 + 4); });
                unsigned long long ovf_a = 0; CHK(hipMemcpy(&ovf_a, d_u64 + 4, 8, hipMemcpyDeviceToHost)); ovf_a /= 2;
                for (int variant = 0; variant < 4; variant++) {
                    CHK(hipMemset(tab, 0, sizeof(Slot) << tlog2));
                    CHK(hipMemset(d_u64, 0, 64));
                    CHK(hipDeviceSynchronize());
                    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
                    CHK(hipEventRecord(e0));
                    const char* vn = "";
#define KB(SL, UU) hipLaunchKernelGGL((kb<SL, UU>), dim3(P), dim3(1024), 0, 0, seg, counts, capq, region, W, tab, (1u << tlog2) - 1, d_u64 + 5); vn = #SL "/SU" #UU
                    if (variant == 0) { KB(2048, 2); } else if (variant == 1) { KB(2048, 4); } else if (variant == 2) { KB(4096, 4); } else { KB(1024, 4); }
                    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
                    float ms_b; CHK(hipEventElapsedTime(&ms_b, e0, e1));
                    hipLaunchKernelGGL(tab_total, dim3(1024), dim3(256), 0, 0, tab, 1u << tlog2, d_u64);
                    unsigned long long r[8]; CHK(hipMemcpy(r, d_u64, 64, hipMemcpyDeviceToHost));
                    const bool ok = ovf_a == 0 ? (r[0] == wb && r[1] == wp && r[2] == n) : (r[2] + ovf_a == n);
                    printf("G=2^%u wg/cu=%u P=%4u capq=%3u skew=%2u kb<%s> | KA %6.3f ms  KB %6.3f ms  A+B %6.3f ms = %5.1f G rec/s | ovf=%llu lds-fallback=%llu %s\n",
                           glog2, wgpc, P, capq, skew, vn, ms_a, ms_b, ms_a + ms_b, n / (ms_a + ms_b) / 1e6,
                           ovf_a, r[5], ok ? "totals ok" : "TOTALS MISMATCH");
                    fflush(stdout);
                }
                CHK(hipFree(seg)); CHK(hipFree(counts));
                }
            }
        }
    }
    return 0;
}
