// wide.cuh - exact group-by state for the wide keys (gfx950).
//
// One open-addressed table serves every exact key set whose key does not fit the two words of
// table.cuh:
//   WK_APP      (Timeslot, SrcAddr FixedString(16), DstPort, Proto) -> sum(Bytes), sum(Packets), count()
//               - the second concurrent key set of BASELINE.json config 5 (SURVEY.md 8(d) cfg 5), with the
//               same Date/Timeslot rule as flows_5m (compose/clickhouse/create.sh:92-110);
//   WK_SRCPORT / WK_DSTPORT  ports >= 65536 of the dashboards' `GROUP BY SrcPort / DstPort`
//               (viz-ch.json:358,604; the column is UInt32, create.sh:20-21 - ports < 65536 live in the
//               dense histograms);
//   WK_MINUTE   toStartOfMinute(TimeFlowStart) -> sum(Bytes*SamplingRate) (viz-ch.json:74).
// Slot = one 64-byte line: 4 key words + 3 sums.  Every key word has bit 63 set, so 0 is EMPTY for each;
// a slot is claimed word by word with 64-bit CAS in the order w0..w3.  A word is written once and never
// changes; a contender that loses word j leaves the slot, so the lane that wins w3 matched w0..w2 and the
// slot always ends up holding one real key.  Stale plain reads (per-XCD L2s are not coherent) can only
// show EMPTY or the final value - never a false match.  u64 wrap-around adds commute: any update order
// gives bit-identical sums.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "table.cuh"

namespace fa {

enum { WK_APP = 1, WK_SRCPORT = 2, WK_DSTPORT = 3, WK_MINUTE = 4 };

struct __attribute__((aligned(64))) WSlot {
    unsigned long long w[4];
    unsigned long long v0, v1, v2, pad;
};
static_assert(sizeof(WSlot) == 64, "one wide slot per 64-byte line");

struct WKey {
    unsigned long long w[4];
};

struct WSpillEntry {
    unsigned long long w[4], v0, v1, v2;
};

// addr = FixedString(16) as two little-endian u64 (lo = bytes 0..7); tb < 2^27; port, proto: UInt32 columns
__host__ __device__ __forceinline__ void wkey_pack(uint32_t kind, uint32_t tb, uint64_t lo, uint64_t hi, uint32_t port,
                                                   uint32_t proto, WKey& k) {
    constexpr unsigned long long B63 = 1ull << 63, M63 = B63 - 1;
    k.w[0] = B63 | (lo & M63);
    k.w[1] = B63 | (((lo >> 63) | (hi << 1)) & M63);
    k.w[2] = B63 | ((hi >> 62) << 59) | ((unsigned long long)(tb & 0x7ffffffu) << 32) | port;
    k.w[3] = B63 | ((unsigned long long)(kind & 0xffu) << 32) | proto;
}
__host__ __device__ __forceinline__ void wkey_unpack(const unsigned long long w[4], uint32_t& kind, uint32_t& tb, uint64_t& lo,
                                                     uint64_t& hi, uint32_t& port, uint32_t& proto) {
    constexpr unsigned long long M63 = (1ull << 63) - 1;
    const unsigned long long a = w[0] & M63, b = w[1] & M63;
    lo = a | (b << 63);
    hi = (b >> 1) | (((w[2] >> 59) & 3ull) << 62);
    tb = (uint32_t)((w[2] >> 32) & 0x7ffffffu);
    port = (uint32_t)w[2];
    kind = (uint32_t)((w[3] >> 32) & 0xffu);
    proto = (uint32_t)w[3];
}
__host__ __device__ __forceinline__ uint32_t wkey_hash(const WKey& k) {
    const uint32_t h0 = key_hash(k.w[0], k.w[1]);
    const uint32_t h1 = key_hash(k.w[2] ^ ((unsigned long long)h0 << 32), k.w[3]);
    return h1 ^ (h0 >> 7);
}

// The table is cut into 2^wide_plog2 aligned regions and a probe sequence never leaves the region of the key's home
// slot (it wraps inside it).  Nothing changes for the atomic paths; it is what lets wagg_kernel (wagg.cuh) hand each
// region to ONE workgroup that updates it with plain loads and stores.  Regions hold at least 64 slots.
#ifndef FA_WIDE_PLOG2_MAX
#define FA_WIDE_PLOG2_MAX 9
#endif
constexpr uint32_t WIDE_PLOG2_MAX = FA_WIDE_PLOG2_MAX;  // 512 regions: two wagg_kernel workgroups per CU
__host__ __device__ constexpr uint32_t wide_plog2(uint32_t cap_log2) {
    return cap_log2 >= WIDE_PLOG2_MAX + 6u ? WIDE_PLOG2_MAX : cap_log2 > 6u ? cap_log2 - 6u : 0u;
}

struct WArgs {
    WSlot* tab;
    uint32_t mask;
    uint32_t rmask;  // slots per region - 1
    WSpillEntry* spill;
    uint32_t spill_cap;
    unsigned int* spill_count;        // Counters::wspill_count
    unsigned long long* spill_lost;   // Counters::wspill_lost
    unsigned long long* used;         // Counters::wused
};

// ---- scattered form of a (SrcAddr,DstPort,Proto) update: 32 bytes = two uint4 -------------------------------
//   q0 = SrcAddr (16 bytes)      q1 = { Bytes lo, Bytes hi, Packets (32 bits), DstPort | Proto << 16 | tbr << 24 }
// tbr = time bucket relative to the launch's tb_base.  count() is 1 per tuple.  What does not fit (ports >= 2^16 and
// protocol numbers >= 2^8 - the columns are UInt32, create.sh:19-22 -, Packets >= 2^32, a record 256 buckets away
// from the batch) takes the atomic path.
__host__ __device__ __forceinline__ bool wtup_fits(uint32_t tbr, uint64_t packets, uint32_t port, uint32_t proto) {
    return tbr < 256u && (packets >> 32) == 0 && port < 65536u && proto < 256u;
}
__host__ __device__ __forceinline__ void wtup_pack(const uint32_t addr[4], uint32_t tbr, uint64_t bytes, uint64_t packets, uint32_t port,
                                                   uint32_t proto, uint4& q0, uint4& q1) {
    q0 = make_uint4(addr[0], addr[1], addr[2], addr[3]);
    q1 = make_uint4((uint32_t)bytes, (uint32_t)(bytes >> 32), (uint32_t)packets, port | proto << 16 | tbr << 24);
}
__host__ __device__ __forceinline__ void wtup_unpack(const uint4& q0, const uint4& q1, uint32_t tb_base, WKey& k, uint64_t& bytes, uint64_t& packets) {
    wkey_pack(WK_APP, tb_base + (q1.w >> 24), (uint64_t)q0.y << 32 | q0.x, (uint64_t)q0.w << 32 | q0.z, q1.w & 0xffffu, (q1.w >> 16) & 0xffu, k);
    bytes = (uint64_t)q1.y << 32 | q1.x;
    packets = q1.z;
}

// Finds or claims the slot of k; nullptr when the probe limit is hit (the caller parks the update).
__device__ __forceinline__ WSlot* wtable_find_or_claim(const WArgs& t, const WKey& k, uint32_t h) {
    uint32_t i = h & t.mask;
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = (i & ~t.rmask) | ((i + 1) & t.rmask)) {
        WSlot* s = &t.tab[i];
        const ulonglong2 k01 = *reinterpret_cast<const ulonglong2*>(&s->w[0]);
        const ulonglong2 k23 = *reinterpret_cast<const ulonglong2*>(&s->w[2]);
        unsigned long long c[4] = {k01.x, k01.y, k23.x, k23.y};
        if (c[0] == k.w[0] && c[1] == k.w[1] && c[2] == k.w[2] && c[3] == k.w[3]) return s;  // common case: no atomics
        bool mine = true;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!mine) break;
            if (c[j] == 0) {
                c[j] = atomicCAS(&s->w[j], 0ull, k.w[j]);
                if (c[j] == 0) {
                    c[j] = k.w[j];
                    if (j == 3) count_created(t.used);  // this lane created the group
                }
            }
            mine = c[j] == k.w[j];
        }
        if (mine) return s;
    }
    return nullptr;
}

__device__ __forceinline__ void wspill_park(const WArgs& t, const WKey& k, uint64_t v0, uint64_t v1, uint64_t v2) {
    const unsigned int j = atomicAdd(t.spill_count, 1u);
    if (j < t.spill_cap)
        t.spill[j] = WSpillEntry{{k.w[0], k.w[1], k.w[2], k.w[3]}, v0, v1, v2};
    else
        atomicAdd(t.spill_lost, 1ull);
}

// Upsert with plain per-lane atomics (slow paths: deferred records, merges, rebuilds, flushes).
__device__ __forceinline__ void wagg_global(const WArgs& t, const WKey& k, uint64_t v0, uint64_t v1, uint64_t v2) {
    WSlot* s = wtable_find_or_claim(t, k, wkey_hash(k));
    if (!s) {
        wspill_park(t, k, v0, v1, v2);
        return;
    }
    if (v0) atomicAdd(&s->v0, (unsigned long long)v0);
    if (v1) atomicAdd(&s->v1, (unsigned long long)v1);
    if (v2) atomicAdd(&s->v2, (unsigned long long)v2);
}

// ---- per-workgroup pre-aggregation of the per-minute series --------------------------------------------
// Every record of a batch falls into a handful of minutes: without this stage all waves of the chip would
// hammer the same few memory-side atomics.  key = minute index + 1 (0 = empty).
constexpr int LDS_MINUTES = 16;
struct LdsMinutes {
    unsigned int key[LDS_MINUTES];
    unsigned long long w[LDS_MINUTES], c[LDS_MINUTES];
};
__device__ __forceinline__ void lds_minutes_clear(LdsMinutes& m) {
    for (int i = threadIdx.x; i < LDS_MINUTES; i += blockDim.x) {
        m.key[i] = 0;
        m.w[i] = 0;
        m.c[i] = 0;
    }
}
// true = absorbed
__device__ __forceinline__ bool lds_minutes_add(LdsMinutes& m, uint32_t minute, uint64_t w, uint64_t c) {
    const unsigned int key = minute + 1u;  // minute < 2^32/60
    uint32_t i = (minute * 0x9E3779B1u) >> 28;
#pragma unroll 1
    for (int probe = 0; probe < LDS_MINUTES; probe++, i = (i + 1) & (LDS_MINUTES - 1)) {
        unsigned int cur = m.key[i];
        if (cur == 0) cur = atomicCAS(&m.key[i], 0u, key);
        if (cur != 0 && cur != key) continue;
        if (w) atomicAdd(&m.w[i], (unsigned long long)w);
        atomicAdd(&m.c[i], (unsigned long long)c);
        return true;
    }
    return false;
}

}  // namespace fa
