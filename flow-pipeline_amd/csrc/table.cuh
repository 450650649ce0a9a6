// table.cuh - flows_5m group-by state: open-addressed hash tables (gfx950).
//
// Replaces the ClickHouse aggregation of flows_5m_view
// (compose/clickhouse/create.sh:92-110: GROUP BY Date, Timeslot, SrcAS, DstAS,
// EType -> sum(Bytes), sum(Packets), count()) and the SummingMergeTree collapse
// of partial rows (create.sh:70-90).  u64 wrap-around adds commute, so any
// update order gives bit-identical sums.
//
// Key packing (lock-free two-word claim, no sentinel collisions):
//   tb   = TimeReceived(u32) / granule      (granule >= 60 -> tb < 2^27)
//   k0   = 1<<63 | (DstAS & 0x7fffffff) << 32 | SrcAS
//   k1   = 1<<63 | (DstAS >> 31) << 59 | tb << 32 | EType
// Both words always have bit 63 set, so 0 is a safe EMPTY for each.  A slot is
// claimed word by word with 64-bit CAS (k0 then k1); a word is written once and
// never changes, so a slot always ends up holding a real key, probing is
// lock-free, and a (possibly stale) plain read can only ever show EMPTY or the
// final value - never a false match.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

// One more group in a table: called by every lane that has just created one, from divergent code.  The lanes that
// are active here are counted with one atomic by the first of them (a per-lane atomicAdd on the one counter word
// serialises the whole chip when most records open a new group - config 5's (SrcAddr,DstPort,Proto) rows).
__device__ __forceinline__ void count_created(unsigned long long* used) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    if (__lane_id() == (uint32_t)__builtin_ctzll(m)) atomicAdd(used, (unsigned long long)__builtin_popcountll(m));
}


__device__ __forceinline__ uint32_t lds_add_rtn_u32(uint32_t* p, uint32_t v) { return atomicAdd(p, v); }
__device__ __forceinline__ void lds_add_u32(uint32_t* p, uint32_t v) { atomicAdd(p, v); }
__device__ __forceinline__ void lds_add_u64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
__device__ __forceinline__ unsigned long long lds_cas_u64(unsigned long long* p, unsigned long long cmp, unsigned long long val) {
    return atomicCAS(p, cmp, val);
}

struct __attribute__((aligned(64))) Slot {
    unsigned long long k0, k1;
    unsigned long long bytes, packets, count;
    unsigned long long pad[3];
};
static_assert(sizeof(Slot) == 64, "one slot per 64-byte line");

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}

__host__ __device__ __forceinline__ void pack_key(uint32_t tb, uint32_t src_as, uint32_t dst_as,
                                                  uint32_t etype, uint64_t& k0, uint64_t& k1) {
    k0 = (1ull << 63) | ((uint64_t)(dst_as & 0x7fffffffu) << 32) | src_as;
    k1 = (1ull << 63) | ((uint64_t)(dst_as >> 31) << 59) | ((uint64_t)(tb & 0x7ffffffu) << 32) | etype;
}
__host__ __device__ __forceinline__ void unpack_key(uint64_t k0, uint64_t k1, uint32_t& tb,
                                                    uint32_t& src_as, uint32_t& dst_as,
                                                    uint32_t& etype) {
    src_as = (uint32_t)k0;
    dst_as = (uint32_t)((k0 >> 32) & 0x7fffffffu) | ((uint32_t)((k1 >> 59) & 1) << 31);
    tb = (uint32_t)((k1 >> 32) & 0x7ffffffu);
    etype = (uint32_t)k1;
}
// 32-bit key hash: three quarter-rate multiplies and a few full-rate ops (the splitmix64 cascade it
// replaces cost four 64-bit multiplies, ~100 issue slots per record).  Bit usage: the device-wide
// table and the aggregation kernel's LDS table index with the LOW bits, the key partition of the
// scatter sink is the TOP bits, the per-workgroup hot-key table uses bits 8.., so the three are
// independent enough for linear probing.
__host__ __device__ __forceinline__ uint32_t key_hash(uint64_t k0, uint64_t k1) {
    const uint32_t a = (uint32_t)k0, b = (uint32_t)(k0 >> 32), c = (uint32_t)k1, d = (uint32_t)(k1 >> 32);
    uint32_t h = a * 0x9E3779B1u + b;
    h ^= h >> 15;
    h = (h ^ c ^ ((d << 13) | (d >> 19))) * 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// ---- scatter-sink tuples (ingest.cuh -> agg.cuh) -----------------------------------------------------
// A record that is not absorbed by the workgroup's hot-key table leaves the ingest kernel as a tuple in the
// workgroup's private segment of its key partition.  Two formats; the host picks one per launch (flowagg.hip,
// adaptive: compact while (almost) every record fits it).  A record that does not fit the launch's format takes
// the direct device-wide-table path - formats only ever decide speed, never results.
//   wide    (16 B): {SrcAS, DstAS, Bytes[27:0] | tbr << 28, Packets[14:0] | EType << 15}; partition = key_hash >> 24
//   compact ( 8 B): SrcAS, DstAS < 2^20 (every publicly routed ASN), Bytes < 2^17, Packets < 2^9, EType one of
//                   {0, 0x0800, 0x86dd, 0x0806}.  The low 8 bits of SrcAS are NOT stored: the partition is
//                   part = SrcAS[7:0] ^ mix8(everything else of the key), a bijection for fixed "everything else",
//                   so the aggregation workgroup of partition `part` recovers them - 72 bits of record in 64.
//                   (Balanced whenever either SrcAS[7:0] or the rest of the key varies.)
//                   lo = DstAS | SrcAS[19:8] << 20;  hi = Bytes | Packets << 17 | (tb & 15) << 26 | etcode << 30
// tbr = time bucket relative to the launch's tb_base (< 16); the compact tuple stores tb & 15 - the ABSOLUTE bucket's low
// bits, from which tb_base recovers the bucket - so that the partition is a function of the key alone: the same group
// lands in the same partition in every launch, and that partition's workgroup owns the group's region of the device
// table (as_region below).
constexpr uint32_t TUPLE_TB_SPAN = 16;
constexpr uint32_t TUPLE_MAX_BYTES = 1u << 28, TUPLE_MAX_PACKETS = 1u << 15, TUPLE_MAX_ETYPE = 1u << 16;
constexpr uint32_t T8_MAX_AS = 1u << 20, T8_MAX_BYTES = 1u << 17, T8_MAX_PACKETS = 1u << 9;
struct TupleVals {
    uint32_t src_as, dst_as, bytes, packets, tbr, etype;
};
__host__ __device__ __forceinline__ bool tup16_fits(uint32_t tbr, uint64_t b, uint64_t p, uint32_t etype) {
    return tbr < TUPLE_TB_SPAN && b < TUPLE_MAX_BYTES && p < TUPLE_MAX_PACKETS && etype < TUPLE_MAX_ETYPE;
}
__host__ __device__ __forceinline__ uint4 tup16_pack(uint32_t src_as, uint32_t dst_as, uint32_t b, uint32_t p, uint32_t tbr, uint32_t etype) {
    return make_uint4(src_as, dst_as, b | (tbr << 28), p | (etype << 15));
}
__host__ __device__ __forceinline__ void tup16_unpack(const uint4& t, TupleVals& v) {
    v.src_as = t.x;
    v.dst_as = t.y;
    v.bytes = t.z & 0x0fffffffu;
    v.tbr = t.z >> 28;
    v.packets = t.w & 0x7fffu;
    v.etype = t.w >> 15;
}
// 2-bit EType dictionary of the compact format: 0 -> 0 (absent), 1 -> 0x0800, 2 -> 0x86dd, 3 -> 0x0806
// (branch-free: as a chain of compares the compiler built a decision tree with an exec-mask branch per level into the ingest
// kernel's sink - on a stream that mixes IPv4 and IPv6 every level is taken by some lane.  Bits 2..1 of the three known
// values are 00, 10, 11: a candidate code without a compare, then ONE compare against the value that code stands for.)
__host__ __device__ __forceinline__ uint32_t t8_etcode(uint32_t etype) {
    uint32_t c = (etype >> 1) & 3u;  // 0x0800 -> 0, 0x86dd -> 2, 0x0806 -> 3
    c = c ? c : 1u;
    const uint32_t stands_for = (uint32_t)(0x080686dd08000000ull >> (16u * c)) & 0xffffu;
    return stands_for == etype ? c : 0u;
}
__host__ __device__ __forceinline__ uint32_t t8_etype(uint32_t code) {
    return (uint32_t)(0x080686dd08000000ull >> (16u * (code & 3u))) & 0xffffu;
}
// 8 well-mixed bits of the stored part of the key (lo = DstAS | SrcAS[19:8] << 20, kh = (tb & 15) | etcode << 4)
__host__ __device__ __forceinline__ uint32_t t8_mix8(uint32_t lo, uint32_t kh) {
    // one multiply (this runs once per record in the ingest kernel): Fibonacci hashing carries lo's low bits into the
    // top byte, kh (6 bits) enters at bits 26.. and lands there directly
    return ((lo ^ (kh << 26)) * 0x9E3779B1u) >> 24;
}
__host__ __device__ __forceinline__ bool t8_fits(uint32_t src_as, uint32_t dst_as, uint32_t tbr, uint64_t b, uint64_t p, uint32_t etype) {
    // (& and |, not && and ||: five tests side by side; as short circuits they were nested exec-mask regions in the ingest kernel)
    return (tbr < TUPLE_TB_SPAN) & ((src_as | dst_as) < T8_MAX_AS) & (b < T8_MAX_BYTES) & (p < T8_MAX_PACKETS) &
           ((etype == 0u) | (t8_etcode(etype) != 0u));
}
// the partition of a compact-eligible key: a function of the key alone (tb = absolute time bucket)
__host__ __device__ __forceinline__ uint32_t t8_part(uint32_t src_as, uint32_t dst_as, uint32_t tb, uint32_t etype) {
    return (src_as & 0xffu) ^ t8_mix8(dst_as | ((src_as >> 8) << 20), (tb & 15u) | (t8_etcode(etype) << 4));
}
__host__ __device__ __forceinline__ uint2 t8_pack(uint32_t src_as, uint32_t dst_as, uint32_t b, uint32_t p, uint32_t tbr, uint32_t etype,
                                                  uint32_t tb_base, uint32_t& part) {
    const uint32_t lo = dst_as | ((src_as >> 8) << 20);
    const uint32_t kh = ((tb_base + tbr) & 15u) | (t8_etcode(etype) << 4);
    part = (src_as & 0xffu) ^ t8_mix8(lo, kh);
    return make_uint2(lo, b | (p << 17) | (kh << 26));
}
__host__ __device__ __forceinline__ void t8_unpack(const uint2& t, uint32_t part, uint32_t tb_base, TupleVals& v) {
    const uint32_t kh = t.y >> 26;
    v.dst_as = t.x & 0xfffffu;
    v.src_as = ((t.x >> 20) << 8) | ((part ^ t8_mix8(t.x, kh)) & 0xffu);
    v.bytes = t.y & 0x1ffffu;
    v.packets = (t.y >> 17) & 0x1ffu;
    v.tbr = (kh - tb_base) & 15u;
    v.etype = t8_etype(kh >> 4);
}

// ---- regions of the device-wide table ---------------------------------------------------------------------
// The table is cut into 2^rlog2 aligned regions (256 once it has 2^14 slots; never fewer than 64 slots each) and a key
// only ever lives in ITS region: home slot = region << (log2 slots - rlog2) | hash bits, and probing wraps inside the
// region.  The region of a compact-eligible key is its compact-tuple partition (t8_part), of any other key the top bits
// of its hash.  Every path agrees on this through as_home / as_next; what it buys: in a compact-tuple launch the
// aggregation workgroup of partition p is the only writer of region p for the duration of its kernel and adds its
// groups with plain loads and stores (agg.cuh) instead of three memory-side atomics per group.
constexpr uint32_t AS_RLOG2_MAX = 8;
__host__ __device__ constexpr uint32_t as_rlog2(uint32_t cap_log2) {
    return cap_log2 >= AS_RLOG2_MAX + 6u ? AS_RLOG2_MAX : cap_log2 > 6u ? cap_log2 - 6u : 0u;
}
__host__ __device__ __forceinline__ uint32_t as_region8(uint64_t k0, uint64_t k1, uint32_t h) {
    uint32_t tb, sa, da, et;
    unpack_key(k0, k1, tb, sa, da, et);
    const bool eligible = (sa | da) < T8_MAX_AS && (et == 0u || t8_etcode(et) != 0u);
    return eligible ? t8_part(sa, da, tb, et) : h >> 24;
}
// home slot of a key (h = key_hash(k0, k1); mask = slots - 1; rlog2 = as_rlog2(log2 slots))
__host__ __device__ __forceinline__ uint32_t as_home(uint64_t k0, uint64_t k1, uint32_t h, uint32_t mask, uint32_t rlog2) {
    const uint32_t rmask = mask >> rlog2;
    return ((as_region8(k0, k1, h) >> (AS_RLOG2_MAX - rlog2)) * (rmask + 1u)) | (h & rmask);
}
__host__ __device__ __forceinline__ uint32_t as_next(uint32_t i, uint32_t mask, uint32_t rlog2) {
    const uint32_t rmask = mask >> rlog2;
    return (i & ~rmask) | ((i + 1u) & rmask);
}

#define FA_MAX_PROBES 128

// ---- per-workgroup LDS pre-aggregation table --------------------------------
// Absorbs hot keys (mocker mode has 9 groups, mocker.go:61-62) before they reach
// the device-wide table.  SoA in LDS; same two-word claim protocol with ds CAS.
template <int SLOTS>
struct LdsTable {
    unsigned long long k0[SLOTS], k1[SLOTS], bytes[SLOTS], packets[SLOTS], count[SLOTS];
};

template <int SLOTS>
__device__ __forceinline__ void lds_table_clear(LdsTable<SLOTS>& t) {
    for (int i = threadIdx.x; i < SLOTS; i += blockDim.x) {
        t.k0[i] = 0;
        t.k1[i] = 0;
        t.bytes[i] = 0;
        t.packets[i] = 0;
        t.count[i] = 0;
    }
}

// Returns true if absorbed; false -> caller goes to the device-wide table.
template <int SLOTS, int PROBES>
__device__ __forceinline__ bool lds_table_add(LdsTable<SLOTS>& t, uint64_t k0, uint64_t k1, uint32_t h,
                                              uint64_t bytes, uint64_t packets, uint64_t count) {
    uint32_t i = (h >> 8) & (SLOTS - 1);
#pragma unroll 1
    for (int probe = 0; probe < PROBES; probe++, i = (i + 1) & (SLOTS - 1)) {
        unsigned long long c0 = t.k0[i];
        if (c0 == 0) c0 = lds_cas_u64(&t.k0[i], 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        unsigned long long c1 = t.k1[i];
        if (c1 == 0) c1 = lds_cas_u64(&t.k1[i], 0ull, (unsigned long long)k1);
        if (c1 != 0 && c1 != k1) continue;
        if (bytes) lds_add_u64(&t.bytes[i], (unsigned long long)bytes);
        if (packets) lds_add_u64(&t.packets[i], (unsigned long long)packets);
        lds_add_u64(&t.count[i], (unsigned long long)count);
        return true;
    }
    return false;
}


// ---- wavefront helpers -------------------------------------------------------
// 64-lane sum of a u64 with DPP row shifts inside each 16-lane row and scalar
// readlane across the four rows (wave64; no LDS traffic).
template <int CTRL>
__device__ __forceinline__ uint64_t dpp_shr_u64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    // bound_ctrl = true: lanes shifted in from outside the row read 0
    lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);
    hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
    return (uint64_t)hi << 32 | lo;
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
    v += dpp_shr_u64<0x111>(v);  // row_shr:1
    v += dpp_shr_u64<0x112>(v);  // row_shr:2
    v += dpp_shr_u64<0x114>(v);  // row_shr:4
    v += dpp_shr_u64<0x118>(v);  // row_shr:8  -> lane 15 of every row holds the row sum
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    uint64_t tot = 0;
#pragma unroll
    for (int row = 0; row < 4; row++) {
        uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)lo, row * 16 + 15);
        uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hi, row * 16 + 15);
        tot += (uint64_t)h << 32 | l;
    }
    return tot;  // wave-uniform
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {  // (sums that fit 32 bits: half the DPP steps of wave_sum_u64)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    uint32_t tot = 0;
#pragma unroll
    for (int row = 0; row < 4; row++) tot += (uint32_t)__builtin_amdgcn_readlane((int)v, row * 16 + 15);
    return tot;  // wave-uniform
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    uint32_t l = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)h << 32 | l;
}

// Wave-level duplicate combining before the tables are touched: lanes holding
// the same key are folded into their lowest lane (leader).  Stops as soon as two
// consecutive leaders have fewer than MIN_GROUP followers, so high-cardinality
// batches pay ~2 rounds and 9-group mocker batches collapse to 9 updates/wave.
template <int MAX_ROUNDS, int MIN_GROUP>
__device__ __forceinline__ void wave_combine(bool& valid, uint64_t k0, uint64_t k1, uint64_t& bytes,
                                             uint64_t& packets, uint64_t& count) {
    uint64_t todo = __ballot(valid);
    int small = 0;
    const int lane = __lane_id();
#pragma unroll 1
    for (int round = 0; round < MAX_ROUNDS && todo != 0 && small < 2; round++) {
        int leader = __builtin_ctzll(todo);
        uint64_t l0 = readlane_u64(k0, leader), l1 = readlane_u64(k1, leader);
        bool match = valid && ((todo >> lane) & 1) && k0 == l0 && k1 == l1;
        uint64_t mm = __ballot(match);
        todo &= ~mm;
        int n = __builtin_popcountll(mm);
        if (n < MIN_GROUP) {
            small++;
            continue;
        }
        small = 0;
        uint64_t sb = wave_sum_u64(match ? bytes : 0);
        uint64_t sp = wave_sum_u64(match ? packets : 0);
        uint64_t sc = wave_sum_u64(match ? count : 0);
        if (match) {
            if (lane == leader) {
                bytes = sb;
                packets = sp;
                count = sc;
            } else {
                valid = false;
            }
        }
    }
}

}  // namespace fa
