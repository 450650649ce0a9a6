// wagg.cuh - aggregation of the scattered (SrcAddr,DstPort,Proto) tuples (gfx950).
//
// Config 5's second key set has almost as many rows as records (random ports: nothing to fold), and every row costs the
// atomic path four dependent 64-bit CAS (the key, word by word) plus three adds - seven memory-side atomics at the
// ~25 G/s the memory side sustains, 1.2-1.4 G records/s in the round-2 measurements (tools/micro/wide_rmw.hip:
// 3.5 G/s for the bare sequence, 17 G/s for one CAS + plain stores).  Here the wave-tile kernel only scatters one
// 32-byte tuple per record into the segment of the key's table REGION (wide.cuh: probe sequences never leave a region),
// and this kernel runs one workgroup per region, which therefore owns its slots for the duration of the launch:
//   * a chunk of WAGG_CHUNK tuples is staged in LDS and deduplicated there (an LDS table of representatives: the first lane
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

#ifndef FA_WAGG_BLOCK
#define FA_WAGG_BLOCK 512
#endif
#ifndef FA_WAGG_U
#define FA_WAGG_U 2
#endif
#ifndef FA_WAGG_EARLY
#define FA_WAGG_EARLY 1
#endif
// Two workgroups per CU (512 regions, 66 KiB of LDS each): a chunk's LDS phases (staging, dedup) and its table phase
// (random 64-byte lines: latency) alternate, and with one workgroup per CU nothing was in flight during the former.
constexpr int WAGG_BLOCK = FA_WAGG_BLOCK;
constexpr int WAGG_U = FA_WAGG_U;                      // tuples per thread and chunk
constexpr int WAGG_CHUNK = WAGG_BLOCK * WAGG_U;
constexpr int WAGG_SLOTS = 2 * WAGG_CHUNK;             // LDS table of representatives: load <= 0.5
constexpr int WAGG_MAX_NWG = 1536;

struct WAggLds {
    uint32_t prefix[WAGG_MAX_NWG + 1];
    unsigned long long k[4][WAGG_CHUNK];
    unsigned long long accb[WAGG_CHUNK], accp[WAGG_CHUNK];
    uint32_t accc[WAGG_CHUNK];
    uint32_t rep[WAGG_SLOTS];
    uint32_t created;
};
static_assert(sizeof(WAggLds) <= 160 * 1024, "wagg_kernel LDS");

__device__ __forceinline__ uint32_t wagg_next(const WArgs& t, uint32_t i) { return (i & ~t.rmask) | ((i + 1) & t.rmask); }
__device__ __forceinline__ void wagg_fill(WSlot* s, const WKey& k, uint64_t b, uint64_t p, uint64_t c) {
    s->w[1] = k.w[1];
    *reinterpret_cast<ulonglong2*>(&s->w[2]) = make_ulonglong2(k.w[2], k.w[3]);
    *reinterpret_cast<ulonglong2*>(&s->v0) = make_ulonglong2(b, p);
    s->v2 = c;
}
__device__ __forceinline__ bool wagg_claim(WSlot* s, const WKey& k) {
    unsigned long long expect = 0ull;
    return __hip_atomic_compare_exchange_strong(&s->w[0], &expect, k.w[0], __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Upsert of one key into the region this workgroup owns, probing from slot i.  false: probe limit (the caller parks
// the update).
__device__ __forceinline__ bool wagg_upsert(const WArgs& t, const WKey& k, uint32_t i, uint64_t b, uint64_t p, uint64_t c, uint32_t& created) {
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = wagg_next(t, i)) {
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
        if (k01.x == 0 && wagg_claim(s, k)) {
            wagg_fill(s, k, b, p, c);
            created++;
            return true;
        }
        // (an empty slot lost to another key of this chunk - keys of a chunk are unique, so it is not this one; its
        // other words may still be on their way): next slot
    }
    return false;
}

// tb_base_ptr / wm: a launch's tuples may be folded LATER (log mode, flowagg.hip "wide log"): the time base their relative
// buckets count from was saved with them, and tuples of buckets below the chunk's watermark - windows dropped in the
// meantime - are skipped.  nullptr / 0: the launch that has just scattered them.
__global__ __launch_bounds__(WAGG_BLOCK) void wagg_kernel(KArgs a, const uint32_t* tb_base_ptr, uint32_t wm) {
    __shared__ WAggLds L;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t part = blockIdx.x;
    const uint32_t nwg = min(a.nwg, (uint32_t)WAGG_MAX_NWG);
    const WArgs t = wargs(a);
    const uint32_t tb_base = tb_base_ptr ? *tb_base_ptr : a.ctr->tb_base;
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
    // the tuples of the NEXT chunk are loaded while this one is folded (two 16-byte loads per tuple; a tuple past
    // the end re-reads tuple 0 of the region's first segment)
    uint4 n0[WAGG_U], n1[WAGG_U];
    auto fetch = [&](uint32_t base) {
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            const uint32_t idx = base + (uint32_t)u * WAGG_BLOCK + tid;
            size_t at = 0;
            if (idx < total) {
                uint32_t lo = 0, hi = nwg;  // the segment that holds tuple idx: largest j with prefix[j] <= idx
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (L.prefix[mid] <= idx) lo = mid;
                    else hi = mid;
                }
                at = (size_t)lo * a.wcapq + (idx - L.prefix[lo]);
            }
            n0[u] = pseg[2u * at];
            n1[u] = pseg[2u * at + 1u];
        }
    };
    if (total) fetch(0);
    for (uint32_t base = 0; base < total; base += WAGG_CHUNK) {
        WKey k[WAGG_U];
        uint64_t b[WAGG_U], p[WAGG_U];
        uint32_t h[WAGG_U];
        bool have[WAGG_U];
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            const uint32_t e = (uint32_t)u * WAGG_BLOCK + tid;
            have[u] = base + e < total && tb_base + (n1[u].w >> 24) >= wm;
            wtup_unpack(n0[u], n1[u], tb_base, k[u], b[u], p[u]);
            h[u] = wkey_hash(k[u]);
            L.k[0][e] = k[u].w[0];
            L.k[1][e] = k[u].w[1];
            L.k[2][e] = k[u].w[2];
            L.k[3][e] = k[u].w[3];
            L.accb[e] = b[u];
            L.accp[e] = p[u];
            L.accc[e] = have[u] ? 1u : 0u;
        }
        // the home slots' key words: issued now, looked at behind the dedup (the previous chunk's rows are visible - the
        // barrier that ended it; what this chunk changes in between is caught by the claim's compare-and-swap)
        WSlot* hs[WAGG_U];
        ulonglong2 k01[WAGG_U], k23[WAGG_U];
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            hs[u] = &t.tab[h[u] & t.mask];
#if FA_WAGG_EARLY
            k01[u] = *reinterpret_cast<const ulonglong2*>(&hs[u]->w[0]);
            k23[u] = *reinterpret_cast<const ulonglong2*>(&hs[u]->w[2]);
#endif
        }
        if (base + WAGG_CHUNK < total) fetch(base + WAGG_CHUNK);
        __syncthreads();
        bool is_rep[WAGG_U];
        uint32_t myslot[WAGG_U];
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            const uint32_t e = (uint32_t)u * WAGG_BLOCK + tid;
            is_rep[u] = false;
            myslot[u] = 0;
            if (have[u]) {
                uint32_t s = (h[u] ^ (h[u] >> 13)) & (WAGG_SLOTS - 1);
                for (;;) {
                    uint32_t r = L.rep[s];
                    if (r == 0) r = atomicCAS(&L.rep[s], 0u, e + 1u);
                    if (r == 0) {
                        is_rep[u] = true;
                        myslot[u] = s;
                        break;
                    }
                    r -= 1u;
                    if (L.k[0][r] == k[u].w[0] && L.k[1][r] == k[u].w[1] && L.k[2][r] == k[u].w[2] && L.k[3][r] == k[u].w[3]) {
                        if (b[u]) atomicAdd(&L.accb[r], (unsigned long long)b[u]);
                        if (p[u]) atomicAdd(&L.accp[r], (unsigned long long)p[u]);
                        atomicAdd(&L.accc[r], 1u);
                        break;
                    }
                    s = (s + 1) & (WAGG_SLOTS - 1);
                }
            }
        }
        __syncthreads();
        // the representatives' upserts: the home-slot step of all of this thread's keys together (loads, then the
        // claims / the sums' loads, then the stores), what is left - the home slot holds another key - one by one
        ulonglong2 v01[WAGG_U];
        unsigned long long v2[WAGG_U], sb[WAGG_U], sp[WAGG_U], sc[WAGG_U];
        bool match[WAGG_U], won[WAGG_U];
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            const uint32_t e = (uint32_t)u * WAGG_BLOCK + tid;
#if !FA_WAGG_EARLY
            k01[u] = *reinterpret_cast<const ulonglong2*>(&hs[u]->w[0]);
            k23[u] = *reinterpret_cast<const ulonglong2*>(&hs[u]->w[2]);
#endif
            sb[u] = L.accb[e];
            sp[u] = L.accp[e];
            sc[u] = L.accc[e];
        }
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            match[u] = is_rep[u] && k01[u].x == k[u].w[0] && k01[u].y == k[u].w[1] && k23[u].x == k[u].w[2] && k23[u].y == k[u].w[3];
            won[u] = false;
            v01[u] = make_ulonglong2(0, 0);
            v2[u] = 0;
            if (match[u]) {
                v01[u] = *reinterpret_cast<const ulonglong2*>(&hs[u]->v0);
                v2[u] = hs[u]->v2;
            } else if (is_rep[u] && k01[u].x == 0) {
                won[u] = wagg_claim(hs[u], k[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            if (match[u]) {
                *reinterpret_cast<ulonglong2*>(&hs[u]->v0) = make_ulonglong2(v01[u].x + sb[u], v01[u].y + sp[u]);
                hs[u]->v2 = v2[u] + sc[u];
            } else if (won[u]) {
                wagg_fill(hs[u], k[u], sb[u], sp[u], sc[u]);
                created++;
            }
        }
#pragma unroll
        for (int u = 0; u < WAGG_U; u++) {
            if (is_rep[u]) {
                if (!match[u] && !won[u] && !wagg_upsert(t, k[u], wagg_next(t, h[u] & t.mask), sb[u], sp[u], sc[u], created))
                    wspill_park(t, k[u], sb[u], sp[u], sc[u]);
                L.rep[myslot[u]] = 0;
            }
        }
        __syncthreads();  // (workgroup-scope release / acquire: the next chunk sees this chunk's rows)
    }
    const uint32_t cw = (uint32_t)wave_sum_u64(created);
    if (lane == 0 && cw) atomicAdd(&L.created, cw);
    __syncthreads();
    if (tid == 0 && L.created) atomicAdd(t.used, (unsigned long long)L.created);
    if (tid == 0 && tb_base_ptr && total) atomicAdd(&a.ctr->wfold_n, (unsigned long long)total);  // (a log chunk's fold: feedback for the host)
}

}  // namespace fa
