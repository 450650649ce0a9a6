// wagg.cuh - aggregation of the scattered (SrcAddr,DstPort,Proto) tuples (gfx950).
//
// Config 5's second key set has almost as many rows as records (random ports: nothing to fold), and every row costs the
// atomic path four dependent 64-bit CAS (the key, word by word) plus three adds - seven memory-side atomics at the
// ~25 G/s the memory side sustains, 1.2-1.4 G records/s in the round-2 measurements (tools/micro/wide_rmw.hip:
// 3.5 G/s for the bare sequence, 17 G/s for one CAS + plain stores).  Here the wave-tile kernel only scatters one
// 32-byte tuple per record into the segment of the key's table REGION (wide.cuh: probe sequences never leave a region),
// and this kernel runs one workgroup per region, which therefore owns its slots for the duration of the launch:
//   * a chunk of 1024 tuples is staged in LDS and deduplicated there (an LDS table of representatives: the first lane
//     that claims a slot for a key represents it, later lanes with the same key add their sums to its LDS accumulators);
//   * each representative then upserts its key with PLAIN loads and stores: keys of a chunk are unique, chunks are
//     separated by workgroup barriers, and no other workgroup touches the region.  The only atomic left is one
//     workgroup-scope CAS on the first key word when a slot is claimed - two different keys of a chunk may want the same
//     empty slot.
// All paths that use atomics on this table (ingest fallbacks, second-chance kernel, merges, rebuilds) are separate
// dispatches on the same stream: never concurrent with this kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sinks.cuh"

namespace fa {

constexpr int WAGG_BLOCK = 1024;
constexpr int WAGG_SLOTS = 2048;  // LDS table of representatives: load <= 0.5
constexpr int WAGG_MAX_NWG = 1536;

struct WAggLds {
    uint32_t prefix[WAGG_MAX_NWG + 1];
    unsigned long long k[4][WAGG_BLOCK];
    unsigned long long accb[WAGG_BLOCK], accp[WAGG_BLOCK];
    uint32_t accc[WAGG_BLOCK];
    uint32_t rep[WAGG_SLOTS];
    uint32_t created;
};

// Upsert of one key into the region this workgroup owns.  false: probe limit (the caller parks the update).
__device__ __forceinline__ bool wagg_upsert(const WArgs& t, const WKey& k, uint32_t h, uint64_t b, uint64_t p, uint64_t c, uint32_t& created) {
    uint32_t i = h & t.mask;
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = (i & ~t.rmask) | ((i + 1) & t.rmask)) {
        WSlot* s = &t.tab[i];
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&s->w[0]);
        const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&s->w[2]);
        if (k01.x == k.w[0] && k01.y == k.w[1] && k23.x == k.w[2] && k23.y == k.w[3]) {
            const ulonglong2 v01 = *reinterpret_cast<const ulonglong2*>(&s->v0);
            const unsigned long long v2 = s->v2;
            *reinterpret_cast<ulonglong2*>(&s->v0) = make_ulonglong2(v01.x + b, v01.y + p);
            s->v2 = v2 + c;
            return true;
        }
        if (k01.x == 0) {
            unsigned long long expect = 0ull;
            if (__hip_atomic_compare_exchange_strong(&s->w[0], &expect, k.w[0], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                s->w[1] = k.w[1];
                *reinterpret_cast<ulonglong2*>(&s->w[2]) = make_ulonglong2(k.w[2], k.w[3]);
                *reinterpret_cast<ulonglong2*>(&s->v0) = make_ulonglong2(b, p);
                s->v2 = c;
                created++;
                return true;
            }
            // lost to another key of this chunk (keys of a chunk are unique, so it is not this one - its words may still
            // be on their way): next slot
        }
    }
    return false;
}

__global__ __launch_bounds__(WAGG_BLOCK) void wagg_kernel(KArgs a) {
    __shared__ WAggLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t part = blockIdx.x;
    const uint32_t nwg = min(a.nwg, (uint32_t)WAGG_MAX_NWG);
    const WArgs t = wargs(a);
    const uint32_t tb_base = a.ctr->tb_base;
    for (int i = tid; i < WAGG_SLOTS; i += WAGG_BLOCK) L.rep[i] = 0;
    if (tid == 0) L.created = 0;
    if (wave == 0) {  // exclusive prefix sums of the segments' tuple counts
        uint32_t running = 0;
        for (uint32_t base = 0; base < nwg; base += 64) {
            const uint32_t j = base + lane;
            const uint32_t c = j < nwg ? min(a.wseg_counts[(size_t)part * a.nwg + j], a.wcapq) : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if ((int)lane >= d) incl += up;
            }
            if (j < nwg) L.prefix[j] = running + incl - c;
            running += (uint32_t)__shfl((int)incl, 63);
        }
        if (lane == 0) L.prefix[nwg] = running;
    }
    __syncthreads();
    const uint32_t total = L.prefix[nwg];
    const uint4* pseg = a.wseg + 2u * (size_t)part * a.wregion;
    uint32_t created = 0;
    for (uint32_t base = 0; base < total; base += WAGG_BLOCK) {
        const uint32_t idx = base + tid;
        const bool have = idx < total;
        WKey k{{0, 0, 0, 0}};
        uint64_t b = 0, p = 0;
        uint32_t h = 0;
        if (have) {
            uint32_t lo = 0, hi = nwg;  // the segment that holds tuple idx: largest j with prefix[j] <= idx
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (L.prefix[mid] <= idx) lo = mid;
                else hi = mid;
            }
            const uint4* q = pseg + 2u * ((size_t)lo * a.wcapq + (idx - L.prefix[lo]));
            const uint4 q0 = q[0], q1 = q[1];
            wtup_unpack(q0, q1, tb_base, k, b, p);
            h = wkey_hash(k);
        }
        L.k[0][tid] = k.w[0];
        L.k[1][tid] = k.w[1];
        L.k[2][tid] = k.w[2];
        L.k[3][tid] = k.w[3];
        L.accb[tid] = b;
        L.accp[tid] = p;
        L.accc[tid] = have ? 1u : 0u;
        __syncthreads();
        bool is_rep = false;
        uint32_t myslot = 0;
        if (have) {
            uint32_t s = (h ^ (h >> 13)) & (WAGG_SLOTS - 1);
            for (;;) {
                uint32_t r = L.rep[s];
                if (r == 0) r = atomicCAS(&L.rep[s], 0u, tid + 1u);
                if (r == 0) {
                    is_rep = true;
                    myslot = s;
                    break;
                }
                r -= 1u;
                if (L.k[0][r] == k.w[0] && L.k[1][r] == k.w[1] && L.k[2][r] == k.w[2] && L.k[3][r] == k.w[3]) {
                    if (b) atomicAdd(&L.accb[r], (unsigned long long)b);
                    if (p) atomicAdd(&L.accp[r], (unsigned long long)p);
                    atomicAdd(&L.accc[r], 1u);
                    break;
                }
                s = (s + 1) & (WAGG_SLOTS - 1);
            }
        }
        __syncthreads();
        if (is_rep) {
            const uint64_t sb = L.accb[tid], sp = L.accp[tid], sc = L.accc[tid];
            if (!wagg_upsert(t, k, h, sb, sp, sc, created)) wspill_park(t, k, sb, sp, sc);
            L.rep[myslot] = 0;
        }
        __syncthreads();  // (workgroup-scope release / acquire: the next chunk sees this chunk's rows)
    }
    const uint32_t cw = (uint32_t)wave_sum_u64(created);
    if (lane == 0 && cw) atomicAdd(&L.created, cw);
    __syncthreads();
    if (tid == 0 && L.created) atomicAdd(t.used, (unsigned long long)L.created);
}

}  // namespace fa
