// agg.cuh - aggregation of the scattered tuples: one workgroup per key partition, LDS hash table, one device-table
// update per group.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sinks.cuh"

namespace fa {

// ---- aggregation of the scattered tuples -----------------------------------------------------
// One 1024-thread workgroup per key partition.  LDS table slot = key (2 words, same claim protocol
// as the other tables) + two packed sums:  s1 = sum(bytes) (< 2^28 * 2^24),
// s2 = sum(packets) << 25 | count  (packets < 2^15, count <= 2^24: no carry between the fields).
struct AggTable {
    unsigned long long k0[AGG_SLOTS], k1[AGG_SLOTS], s1[AGG_SLOTS], s2[AGG_SLOTS];
};
static_assert(sizeof(AggTable) == 32 * AGG_SLOTS, "agg_kernel LDS table");

// slow path of the LDS upsert: claim / probe; false = the table is full around this hash
__device__ __forceinline__ bool agg_lds_upsert(AggTable& lt, uint64_t k0, uint64_t k1, uint32_t h, uint32_t by,
                                               unsigned long long v2) {
    uint32_t i = h & (AGG_SLOTS - 1);
#pragma unroll 1
    for (int probe = 0; probe < AGG_PROBES; probe++, i = (i + 1) & (AGG_SLOTS - 1)) {
        unsigned long long c0 = lt.k0[i];
        if (c0 == 0) c0 = atomicCAS(&lt.k0[i], 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        unsigned long long c1 = lt.k1[i];
        if (c1 == 0) c1 = atomicCAS(&lt.k1[i], 0ull, (unsigned long long)k1);
        if (c1 != 0 && c1 != k1) continue;
        if (by) atomicAdd(&lt.s1[i], (unsigned long long)by);
        atomicAdd(&lt.s2[i], v2);
        return true;
    }
    return false;
}

constexpr int AGG_MAX_NWG = 1536;  // ingest-kernel workgroups: 256 CUs x 6 (workgroup-tile kernel) or x 2 (wave-tile kernel)
#ifndef FA_AGG_SU
#define FA_AGG_SU 4
#endif
constexpr int AGG_SU = FA_AGG_SU;  // segments a wave reads at a time (16-byte loads in flight per lane, x2 buffers)
constexpr int AGG_PAD = AGG_SU * 8;  // zero counts behind the last segment (the back pass reads 8 segments per load)
constexpr int AGG_CH = 4;  // tuples of a batch that are hashed / probed together

// 16-byte loads a lane has in flight per batch: AGG_SU wide tuples, or half as many pieces of compact tuples (the
// same number of tuples per batch - and of registers per chunk: 1024 threads leave 128 VGPRs per lane)
template <bool T8>
constexpr int agg_su() { return T8 ? (AGG_SU + 1) / 2 : AGG_SU; }
struct AggBatch {
    uint4 t[AGG_SU];  // 16 bytes per load: one wide tuple or two compact ones
    uint32_t v;       // wide: bit s = t[s] is a tuple of this lane (not a dummy load); compact: bits 2s, 2s+1 = its two halves
};

// issue the loads of lanes [0,64) of AGG_SU consecutive segments starting at w0.  The segment counts
// come from LDS (pc, zero padded): a count read from global memory would put a full vmcnt drain between
// consecutive tuple loads.
// Front parts: chunk level j = 16-byte pieces [64j, 64j+64) of a segment, one segment per 64 lanes.
// Back parts (a handful of tuples each): level j = pieces [8j, 8j+8) of the c tuples that end at the segment's
// last slot, EIGHT segments per 64 lanes.  Loads are unconditional (lanes without a tuple re-read piece 0 of a
// valid segment): with predicated loads the compiler cannot count what is in flight and drains everything
// (vmcnt(0)) before the previous batch is consumed.
template <bool BACK, bool T8>
__device__ __forceinline__ void agg_fetch(const KArgs& a, const uint4* pbase, const uint32_t* pc, uint32_t w0, uint32_t lane,
                                          uint32_t j, AggBatch& b) {
    constexpr uint32_t SEGS = BACK ? 8u : 1u, PER = 64u / SEGS;  // segments per load, lanes per segment
    constexpr uint32_t TPP = T8 ? 2u : 1u;                         // tuples per 16-byte piece
    uint32_t idx[AGG_SU], seg[AGG_SU];
    b.v = 0;
#pragma unroll
    for (int s = 0; s < agg_su<T8>(); s++) {
        seg[s] = w0 + (uint32_t)s * SEGS + (BACK ? lane / PER : 0u);
        const uint32_t c = pc[min(seg[s], (uint32_t)(AGG_MAX_NWG + AGG_PAD - 1))];  // 0 past nwg
        const uint32_t first = BACK ? a.capq - c : 0u;                     // first tuple of the part (capq - c may be odd)
        const uint32_t piece = first / TPP + PER * j + (BACK ? lane % PER : lane);  // 16-byte piece of the segment
        uint32_t valid = 0;
#pragma unroll
        for (uint32_t e = 0; e < TPP; e++) {
            const uint32_t q = piece * TPP + e;  // tuple index inside the segment
            valid |= (q >= first && q < first + c) ? 1u << e : 0u;
        }
        b.v |= valid << (TPP * s);
        idx[s] = valid ? piece : 0u;
    }
#pragma unroll
    for (int s = 0; s < agg_su<T8>(); s++) b.t[s] = pbase[(size_t)min(seg[s], a.nwg - 1u) * (a.capq / TPP) + idx[s]];
}

// tuple e of a batch, as values (wide: e = piece; compact: e >> 1 = piece, e & 1 = half)
template <bool T8>
__device__ __forceinline__ void agg_vals(const AggBatch& b, int e, uint32_t part, uint32_t tb_base, TupleVals& v) {
    if (T8) {
        const uint4& q = b.t[e >> 1];
        t8_unpack((e & 1) ? make_uint2(q.z, q.w) : make_uint2(q.x, q.y), part, tb_base, v);
    } else {
        tup16_unpack(b.t[e], v);
    }
}

// slow path for one tuple (wide form): probing LDS upsert, then the device-wide table
__device__ __forceinline__ void agg_tuple(const KArgs& a, AggTable& lt, uint32_t tb_base, const uint4& t) {
    TupleVals v;
    tup16_unpack(t, v);
    uint64_t k0, k1;
    pack_key(tb_base + v.tbr, v.src_as, v.dst_as, v.etype, k0, k1);
    const uint32_t h = key_hash(k0, k1);
    const unsigned long long v2 = ((unsigned long long)v.packets << 25) | 1ull;
    if (!agg_lds_upsert(lt, k0, k1, h, v.bytes, v2)) agg_global(a, k0, k1, h, v.bytes, v.packets, 1);
}

// the queued leftovers of a wave, one per lane (LDS operations of a wave complete in order: the queue needs no fence)
__device__ __forceinline__ void agg_drain(const KArgs& a, AggTable& lt, uint32_t tb_base, uint32_t lane, const uint4* queue, uint32_t qn) {
    if (lane < qn) agg_tuple(a, lt, tb_base, queue[lane]);
}

// The common case (the key already sits in its home slot) for AGG_CH tuples at once, so that the LDS
// round trips of the segments overlap; everything else goes through the probing upsert.
template <int E0, bool T8>
__device__ __forceinline__ void agg_consume_chunk(const KArgs& a, AggTable& lt, uint32_t tb_base, uint32_t part, uint32_t lane, const AggBatch& b,
                                                  uint4* queue, uint32_t& qn) {
    uint64_t k0[AGG_CH], k1[AGG_CH];
    uint32_t h[AGG_CH];
    TupleVals tv[AGG_CH];
    unsigned long long c0[AGG_CH], c1[AGG_CH];
    if (FA_DBG(a, DBG_AGG_NO_LDS)) {  // ablation: consume the loads only
        uint32_t x = 0;
#pragma unroll
        for (int s = 0; s < agg_su<T8>(); s++) x ^= b.t[s].x ^ b.t[s].y ^ b.t[s].z ^ b.t[s].w;
        if (x == 0x12345678u) lt.s1[lane] = x;
        return;
    }
    // home slot and its successor are read together: at the table's load (<= 40 %) linear probing leaves ~30 % of
    // the keys one slot away from home and ~10 % further; only the latter (and first occurrences) take the
    // probing path below
    unsigned long long d0[AGG_CH], d1[AGG_CH];
#pragma unroll
    for (int s = 0; s < AGG_CH; s++) {
        agg_vals<T8>(b, E0 + s, part, tb_base, tv[s]);
        pack_key(tb_base + tv[s].tbr, tv[s].src_as, tv[s].dst_as, tv[s].etype, k0[s], k1[s]);
        h[s] = key_hash(k0[s], k1[s]);
        const uint32_t i = h[s] & (AGG_SLOTS - 1), j = (i + 1) & (AGG_SLOTS - 1);
        c0[s] = lt.k0[i];
        c1[s] = lt.k1[i];
        d0[s] = lt.k0[j];
        d1[s] = lt.k1[j];
    }
    uint32_t pending = 0;  // tuples that are not in their home slot (probing / claiming needed)
#pragma unroll
    for (int s = 0; s < AGG_CH; s++) {
        if (!((b.v >> (E0 + s)) & 1u)) continue;
        const unsigned long long v2 = ((unsigned long long)tv[s].packets << 25) | 1ull;
        const uint32_t i = h[s] & (AGG_SLOTS - 1);
        const bool at0 = c0[s] == k0[s] && c1[s] == k1[s], at1 = d0[s] == k0[s] && d1[s] == k1[s];
        if (at0 || at1) {
            const uint32_t t = at0 ? i : (i + 1) & (AGG_SLOTS - 1);
            if (tv[s].bytes) atomicAdd(&lt.s1[t], (unsigned long long)tv[s].bytes);
            atomicAdd(&lt.s2[t], v2);
        } else {
            pending |= 1u << s;
        }
    }
    if (FA_DBG(a, DBG_AGG_NO_SLOW)) return;
    // The leftovers (first occurrences of a group, keys two or more slots from home: ~5 % of the tuples) wait in
    // the wave's queue (as wide tuples) and take the probing path 64 at a time: handled on the spot, each round of
    // the probing loop would run with one or two active lanes.
#pragma unroll
    for (int s = 0; s < AGG_CH; s++) {
        const bool pnd = (pending >> s) & 1u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pnd);
        if (m != 0ull) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
            if (qn + cnt > 64u) {
                agg_drain(a, lt, tb_base, lane, queue, qn);
                qn = 0;
            }
            if (pnd)
                queue[qn + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] =
                    tup16_pack(tv[s].src_as, tv[s].dst_as, tv[s].bytes, tv[s].packets, tv[s].tbr, tv[s].etype);
            qn += cnt;
        }
    }
}

template <bool T8>
__device__ __forceinline__ void agg_consume(const KArgs& a, AggTable& lt, uint32_t tb_base, uint32_t part, uint32_t lane, const AggBatch& b,
                                            uint4* queue, uint32_t& qn) {
    constexpr int NT = agg_su<T8>() * (T8 ? 2 : 1);  // tuples per lane and batch
    static_assert(NT % AGG_CH == 0, "batch = whole chunks");
    agg_consume_chunk<0, T8>(a, lt, tb_base, part, lane, b, queue, qn);
    if constexpr (NT > AGG_CH) agg_consume_chunk<AGG_CH, T8>(a, lt, tb_base, part, lane, b, queue, qn);
    if constexpr (NT > 2 * AGG_CH) agg_consume_chunk<2 * AGG_CH, T8>(a, lt, tb_base, part, lane, b, queue, qn);
    if constexpr (NT > 3 * AGG_CH) agg_consume_chunk<3 * AGG_CH, T8>(a, lt, tb_base, part, lane, b, queue, qn);
    static_assert(NT <= 4 * AGG_CH, "agg_consume: add chunks");
}

template <bool T8>
__global__ __launch_bounds__(AGG_BLOCK) void agg_kernel(KArgs a) {
    constexpr uint32_t TPP = T8 ? 2u : 1u;
    __shared__ AggTable lt;
    __shared__ uint32_t pc[AGG_MAX_NWG + AGG_PAD];  // this partition's segment counts, zero padded
    const uint32_t part = blockIdx.x / AGG_SPLIT, sub = blockIdx.x % AGG_SPLIT;
    for (int i = threadIdx.x; i < AGG_SLOTS; i += AGG_BLOCK) {
        lt.k0[i] = 0;
        lt.k1[i] = 0;
        lt.s1[i] = 0;
        lt.s2[i] = 0;
    }
    __shared__ uint32_t pcb[AGG_MAX_NWG + AGG_PAD];  // ... and the counts of the segments' back parts
    __shared__ uint4 queues[(AGG_BLOCK / 64) * 64];  // per wave: tuples that need the probing path
    __shared__ uint32_t maxc_s[2];
    if (threadIdx.x < 3) maxc_s[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mymax = 0, mymaxb = 0, mysum = 0;
    for (uint32_t i = threadIdx.x; i < AGG_MAX_NWG + AGG_PAD; i += AGG_BLOCK) {
        const uint32_t c = i < a.nwg ? a.seg_counts[(size_t)part * a.nwg + i] : 0u;
        const uint32_t cb = i < a.nwg ? a.seg_counts[((size_t)NPART_MAX + part) * a.nwg + i] : 0u;
        pc[i] = c;
        pcb[i] = cb;
        mymax = max(mymax, c);
        mymaxb = max(mymaxb, cb);
        mysum += c + cb;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mymax = max(mymax, (uint32_t)__shfl_xor((int)mymax, o));
        mymaxb = max(mymaxb, (uint32_t)__shfl_xor((int)mymaxb, o));
        mysum += (uint32_t)__shfl_xor((int)mysum, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&maxc_s[0], mymax);
        atomicMax(&maxc_s[1], mymaxb);
        if (mysum) atomicAdd(&maxc_s[2], mysum);
    }
    const uint32_t tb_base = a.ctr->tb_base;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // (compact tuples: region and capq are even, so every segment starts on a 16-byte boundary)
    const uint4* pbase = a.seg + (((size_t)part * a.region) >> (T8 ? 1 : 0));
    constexpr uint32_t SU = agg_su<T8>();
    constexpr uint32_t STEP = (AGG_BLOCK / 64) * SU * AGG_SPLIT;
    __syncthreads();  // table cleared, counts staged
    const uint32_t maxc = maxc_s[0], maxcb = maxc_s[1];
    uint4* queue = queues + wave * 64;
    uint32_t qn = 0;  // wave-uniform
    // software pipeline over this wave's segment groups: the next group's loads fly during the LDS work
    // (every fetch is unconditional - clamped addresses, zero counts past the end - so that the compiler
    // can count the loads in flight and wait for the older batch only)
    // chunk levels: level j covers the 16-byte pieces [64j, 64j+64) of every segment
    const uint32_t levels = (__builtin_amdgcn_readfirstlane(maxc) + 64u * TPP - 1u) / (64u * TPP);
    for (uint32_t j = 0; j < levels; j++) {
        AggBatch b0, b1;
        uint32_t w0 = (sub * (AGG_BLOCK / 64) + wave) * SU;
        agg_fetch<false, T8>(a, pbase, pc, w0, lane, j, b0);
        while (true) {
            agg_fetch<false, T8>(a, pbase, pc, w0 + STEP, lane, j, b1);
            agg_consume<T8>(a, lt, tb_base, part, lane, b0, queue, qn);
            agg_fetch<false, T8>(a, pbase, pc, w0 + 2 * STEP, lane, j, b0);
            agg_consume<T8>(a, lt, tb_base, part, lane, b1, queue, qn);
            w0 += 2 * STEP;
            if (w0 >= a.nwg) break;
        }
    }
    // back parts: 8 pieces per segment and level; a compact back part of c tuples spans up to c / 2 + 1 pieces
    const uint32_t levels_b = (__builtin_amdgcn_readfirstlane(maxcb) / TPP + (T8 ? 1u : 0u) + 7u) >> 3;
    constexpr uint32_t STEP_B = STEP * 8u;
    for (uint32_t j = 0; j < (maxcb ? levels_b : 0u); j++) {  // the back parts (single tuples and bin leftovers of the wave-tile kernel)
        AggBatch b0, b1;
        uint32_t w0 = (sub * (AGG_BLOCK / 64) + wave) * SU * 8u;
        agg_fetch<true, T8>(a, pbase, pcb, w0, lane, j, b0);
        while (true) {
            agg_fetch<true, T8>(a, pbase, pcb, w0 + STEP_B, lane, j, b1);
            agg_consume<T8>(a, lt, tb_base, part, lane, b0, queue, qn);
            agg_fetch<true, T8>(a, pbase, pcb, w0 + 2 * STEP_B, lane, j, b0);
            agg_consume<T8>(a, lt, tb_base, part, lane, b1, queue, qn);
            w0 += 2 * STEP_B;
            if (w0 >= a.nwg) break;
        }
    }
    agg_drain(a, lt, tb_base, lane, queue, qn);
    __syncthreads();
    // every group of this partition goes to the device-wide table once; quad-grouped: one atomic line
    // transaction per group.  Uniform trip count: the whole wave takes part in the quad rounds.
    if (FA_DBG(a, DBG_AGG_NO_FLUSH)) return;
    constexpr int NF = AGG_SLOTS / AGG_BLOCK;  // slots per thread
    unsigned long long fk0[NF], fk1[NF], fs1[NF], fs2[NF];
    ulonglong2 home[NF];
    uint32_t fh[NF];
#pragma unroll
    for (int q = 0; q < NF; q++) {  // phase 1: the home-slot probes of all NF groups fly together
        const int i = q * AGG_BLOCK + threadIdx.x;
        fk0[q] = lt.k0[i];
        fk1[q] = lt.k1[i];
        fs1[q] = lt.s1[i];
        fs2[q] = lt.s2[i];
        fh[q] = key_hash(fk0[q], fk1[q]);
        home[q] = *reinterpret_cast<const ulonglong2*>(&a.tab[as_home(fk0[q], fk1[q], fh[q], a.mask, a.rlog2)]);
    }
#pragma unroll
    for (int q = 0; q < NF; q++) {
        Slot* sp = nullptr;
        const unsigned long long b = fs1[q], p = fs2[q] >> 25, c = fs2[q] & 0x1ffffffull;
        if (fk0[q] != 0 && fk1[q] != 0 && fs2[q] != 0) {
            if (home[q].x == fk0[q] && home[q].y == fk1[q]) sp = &a.tab[as_home(fk0[q], fk1[q], fh[q], a.mask, a.rlog2)];
            else sp = table_find_or_claim(a, fk0[q], fk1[q], fh[q]);
            if (!sp) spill_park(a, fk0[q], fk1[q], b, p, c);
        }
        quad_atomic_update(sp, b, p, c);
    }
}

// ---- aggregation of compact tuples -----------------------------------------------------------------
// Inside one partition a compact tuple's key is a single word: everything of the key but SrcAS[7:0] is in the
// tuple (lo = DstAS | SrcAS[19:8] << 20, kh = tbr | etcode << 4), and SrcAS[7:0] is a function of (partition, lo,
// kh) (table.cuh).  So the LDS table is keyed by ONE 64-bit word straight out of the tuple - no unpacking, no
// two-word compare, one ds_read2_b64 looks at the home slot and its successor - and the full key is only rebuilt
// once per group, when the group is added to the device-wide table.  Open addressing without wrap-around: a key
// lives in one of the AGG_PROBES slots from its home slot on (the table has that many slots of slack at the end).
constexpr int AGG8_SLOTS = 4096;
constexpr int AGG8_ALL = AGG8_SLOTS + AGG_PROBES;
struct Agg8Table {
    unsigned long long key[AGG8_ALL], s1[AGG8_ALL], s2[AGG8_ALL];  // key = 1 << 63 | kh << 32 | lo;  sums as in AggTable
};
__device__ __forceinline__ unsigned long long agg8_key(const uint2& t) { return (1ull << 63) | ((unsigned long long)(t.y >> 26) << 32) | t.x; }
__device__ __forceinline__ uint32_t agg8_mix(const uint2& t) { return (t.x ^ ((t.y >> 26) * 0x9E3779B1u)) * 0x85EBCA6Bu; }
__device__ __forceinline__ uint32_t agg8_home(const uint2& t) {
    return agg8_mix(t) >> 20;  // 12 bits; the partition came out of a different mix (t8_mix8)
}
// which pass of a multi-pass aggregation takes this tuple (bits of the mix below the home slot's)
__device__ __forceinline__ uint32_t agg8_sub(const uint2& t) { return (agg8_mix(t) >> 17) & 7u; }
__device__ __forceinline__ void agg8_global(const KArgs& a, uint32_t tb_base, uint32_t part, const uint2& t, uint32_t by, uint32_t pk) {
    TupleVals v;
    t8_unpack(t, part, tb_base, v);
    uint64_t k0, k1;
    pack_key(tb_base + v.tbr, v.src_as, v.dst_as, v.etype, k0, k1);
    agg_global(a, k0, k1, key_hash(k0, k1), by, pk, 1);
}
// probing upsert (first occurrences, keys further than one slot from home)
__device__ __forceinline__ void agg8_tuple(const KArgs& a, Agg8Table& lt, uint32_t tb_base, uint32_t part, const uint2& t) {
    const unsigned long long key = agg8_key(t);
    const uint32_t by = t.y & 0x1ffffu, pk = (t.y >> 17) & 0x1ffu;
    uint32_t i = agg8_home(t);
#pragma unroll 1
    for (int probe = 0; probe < AGG_PROBES; probe++, i++) {
        unsigned long long c = lt.key[i];
        if (c == 0) c = atomicCAS(&lt.key[i], 0ull, key);
        if (c != 0 && c != key) continue;
        if (by) atomicAdd(&lt.s1[i], (unsigned long long)by);
        atomicAdd(&lt.s2[i], ((unsigned long long)pk << 25) | 1ull);
        return;
    }
    agg8_global(a, tb_base, part, t, by, pk);  // the table is full around this key
}

// Upsert into the table region THIS workgroup owns (compact-tuple launches: region = partition, table.cuh as_region8),
// probing from slot i: plain loads and stores, one workgroup-scope CAS where a slot is claimed (two keys of the workgroup
// may want the same empty slot; nobody else writes the region while this kernel runs).  false: probe limit.
__device__ __forceinline__ bool agg8_upsert_owned(const KArgs& a, uint64_t k0, uint64_t k1, uint32_t i, uint64_t b, uint64_t p, uint64_t c,
                                                  uint32_t& created) {
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = as_next(i, a.mask, a.rlog2)) {
        Slot* s = &a.tab[i];
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&s->k0);
        if (kk.x == k0 && kk.y == k1) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&s->bytes);
            const unsigned long long cnt = s->count;
            *reinterpret_cast<ulonglong2*>(&s->bytes) = make_ulonglong2(v.x + b, v.y + p);
            s->count = cnt + c;
            return true;
        }
        if (kk.x == 0) {
            unsigned long long expect = 0ull;
            if (__hip_atomic_compare_exchange_strong(&s->k0, &expect, (unsigned long long)k0, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                s->k1 = k1;
                *reinterpret_cast<ulonglong2*>(&s->bytes) = make_ulonglong2(b, p);
                s->count = c;
                created++;
                return true;
            }
        }
        // (a slot another key of this workgroup holds - the groups of a flush are distinct - or has just claimed)
    }
    return false;
}

#ifndef FA_AGG8_SU
#define FA_AGG8_SU 2
#endif
#ifndef FA_CMS_SU
#define FA_CMS_SU 4
#endif
constexpr int AGG8_SU = FA_AGG8_SU;        // 16-byte loads per lane and batch (x 2 tuples)
constexpr int AGG8_NT = AGG8_SU * 2;
constexpr int CMS_SU = FA_CMS_SU;          // cms_agg_kernel: segments per batch (see cms_fetch)
template <int SU>
struct Agg8BatchT {
    uint4 t[SU];
    uint32_t v;  // bit e: tuple e of this lane is real (e >> 1 = load, e & 1 = half)
};
typedef Agg8BatchT<AGG8_SU> Agg8Batch;
// Loads of one work item.  Front: pieces [64j, 64j+64) of segment w0 + s, one segment per load.  Back: pieces
// [8j, 8j+8) of the c tuples that end at the segment's last slot, eight segments per load (lane / 8).  Unconditional
// (clamped addresses, zero counts past the end: see agg_fetch).
template <bool BACK, int SU>
__device__ __forceinline__ void agg8_fetch(uint32_t capq, uint32_t nwg, const uint4* pbase, const uint32_t* pc, uint32_t w0, uint32_t lane, uint32_t j,
                                           Agg8BatchT<SU>& b) {
    constexpr uint32_t SEGS = BACK ? 8u : 1u, PER = 64u / SEGS;
    uint32_t idx[SU], seg[SU];
    b.v = 0;
#pragma unroll
    for (int s = 0; s < SU; s++) {
        seg[s] = w0 + (uint32_t)s * SEGS + (BACK ? lane / PER : 0u);
        const uint32_t c = pc[min(seg[s], (uint32_t)(AGG_MAX_NWG + AGG_PAD - 1))];
        const uint32_t first = BACK ? capq - c : 0u;
        const uint32_t piece = first / 2u + PER * j + (BACK ? lane % PER : lane);
        const uint32_t q = piece * 2u;
        const uint32_t valid = ((q >= first && q < first + c) ? 1u : 0u) | ((q + 1u >= first && q + 1u < first + c) ? 2u : 0u);
        b.v |= valid << (2 * s);
        idx[s] = valid ? piece : 0u;
    }
#pragma unroll
    for (int s = 0; s < SU; s++) b.t[s] = pbase[(size_t)min(seg[s], nwg - 1u) * (capq / 2u) + idx[s]];
}
__device__ __forceinline__ void agg8_drain(const KArgs& a, Agg8Table& lt, uint32_t tb_base, uint32_t part, uint32_t lane, const uint2* queue, uint32_t qn) {
    if (lane < qn) agg8_tuple(a, lt, tb_base, part, queue[lane]);
}
// (pmask, pass: this pass takes the tuples with agg8_sub & pmask == pass; pmask = 0: all of them)
// hk: the wave's HEAVY key (< 4: none found yet, 4: none).  A stream with one dominant (SrcAS, DstAS) pair - Zipf-1.1 addresses: the
// top pair is 0.6 % of the records, half of the tuples of its partition - makes that partition's workgroup the last one
// of the kernel (1.85x the mean tuples, and its LDS adds queue up on ONE slot: 147 us against 100 us for twice the
// tuples spread evenly).  So a wave looks for a key that holds an eighth of a row (agg8_find_heavy, until it has one) and
// from then on adds that key's tuples of a row up first (two 32-bit wave sums): one pair of LDS adds per row.
// (Round 4 tried the step before this kernel: the heavy group of a partition reported to the NEXT launches' ingest workgroups,
// which pinned it in their LDS hot-key tables so that its records never became tuples.  Zipf-1.1 AS pairs: aggregation +
// second-chance kernels 122 -> 108 us, the ingest kernel 311 -> 340 us - every tile pays the mask test and a key hash for the
// mask's false positives.  profiles/r04_exp_bench_zipf_ks1_heavy{0,1}.json; dropped.)
__device__ __forceinline__ unsigned long long agg8_find_heavy(unsigned long long key, bool valid) {
    unsigned long long rest = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll 1
    for (int attempt = 0; attempt < 2 && rest != 0ull; attempt++) {
        const int l = __builtin_ctzll(rest);
        const unsigned long long k = readlane_u64(key, l);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && key == k);
        if (__builtin_popcountll(same) >= 8) return k;
        rest &= ~same;
    }
    return 0ull;
}
__device__ __forceinline__ void agg8_consume(const KArgs& a, Agg8Table& lt, uint32_t tb_base, uint32_t part, uint32_t lane, const Agg8Batch& b,
                                             uint2* queue, uint32_t& qn, uint32_t pmask, uint32_t pass, unsigned long long& hk) {
    uint2 t[AGG8_NT];
    unsigned long long key[AGG8_NT], c0[AGG8_NT], c1[AGG8_NT];
    uint32_t home[AGG8_NT];
#pragma unroll
    for (int e = 0; e < AGG8_NT; e++) {  // all the LDS reads of the batch fly together
        const uint4& q = b.t[e >> 1];
        t[e] = (e & 1) ? make_uint2(q.z, q.w) : make_uint2(q.x, q.y);
        key[e] = agg8_key(t[e]);
        home[e] = agg8_home(t[e]);
        c0[e] = lt.key[home[e]];
        c1[e] = lt.key[home[e] + 1];
    }
    if (FA_DBG(a, DBG_AGG_NO_LDS)) {
        unsigned long long x = 0;
#pragma unroll
        for (int e = 0; e < AGG8_NT; e++) x ^= c0[e] ^ c1[e] ^ key[e];
        if (x == 0x12345678ull) lt.s1[lane] = x;
        return;
    }
    uint32_t pending = 0;
    if (hk < 4ull) {  // (still looking: hk counts the batches tried; 4 = gave up - an even stream pays for four looks per wave)
        const unsigned long long f = agg8_find_heavy(key[0], (b.v & 1u) && !(pmask && (agg8_sub(t[0]) & pmask) != pass));
        hk = f ? f : hk + 1ull;
    }
    if (hk >> 63) {  // (wave-uniform; every key has bit 63 set) this wave has a heavy key: its tuples of a row are added up first
#pragma unroll
        for (int e = 0; e < AGG8_NT; e++) {
            const bool act = ((b.v >> e) & 1u) && !(pmask && (agg8_sub(t[e]) & pmask) != pass);
            const bool at0 = c0[e] == key[e], at1 = c1[e] == key[e];
            uint32_t by = t[e].y & 0x1ffffu;
            unsigned long long s2 = ((unsigned long long)((t[e].y >> 17) & 0x1ffu) << 25) | 1ull;
            bool mine = act;
            const bool hv = act && key[e] == hk && (at0 || at1);  // (the key's first tuple ever goes the normal way and opens its slot)
            const unsigned long long m = __builtin_amdgcn_ballot_w64(hv);
            if (__builtin_popcountll(m) >= 2) {
                // bytes < 2^17 and packets < 2^9 per tuple, 64 lanes: both sums and the count fit 32 bits
                const uint32_t sb = wave_sum_u32(hv ? by : 0u);
                const uint32_t sq = wave_sum_u32(hv ? (((t[e].y >> 17) & 0x1ffu) << 8) | 1u : 0u);  // packets << 8 | count
                if (hv) {
                    mine = lane == (uint32_t)__builtin_ctzll(m);
                    by = sb;
                    s2 = ((unsigned long long)(sq >> 8) << 25) | (unsigned long long)(sq & 0xffu);
                }
            }
            if (!mine) continue;
            if (at0 || at1) {
                const uint32_t s = home[e] + (at0 ? 0u : 1u);
                if (by) atomicAdd(&lt.s1[s], (unsigned long long)by);
                atomicAdd(&lt.s2[s], s2);
            } else {
                pending |= 1u << e;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < AGG8_NT; e++) {
            if (!((b.v >> e) & 1u)) continue;
            if (pmask && (agg8_sub(t[e]) & pmask) != pass) continue;
            const bool at0 = c0[e] == key[e], at1 = c1[e] == key[e];
            if (at0 || at1) {
                const uint32_t s = home[e] + (at0 ? 0u : 1u);
                const uint32_t by = t[e].y & 0x1ffffu;
                if (by) atomicAdd(&lt.s1[s], (unsigned long long)by);
                atomicAdd(&lt.s2[s], ((unsigned long long)((t[e].y >> 17) & 0x1ffu) << 25) | 1ull);
            } else {
                pending |= 1u << e;
            }
        }
    }
    if (FA_DBG(a, DBG_AGG_NO_SLOW)) return;
#pragma unroll
    for (int e = 0; e < AGG8_NT; e++) {  // leftovers wait in the wave's queue and take the probing path 64 at a time
        const bool pnd = (pending >> e) & 1u;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pnd);
        if (m != 0ull) {
            const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
            if (qn + cnt > 64u) {
                agg8_drain(a, lt, tb_base, part, lane, queue, qn);
                qn = 0;
            }
            if (pnd) queue[qn + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = t[e];
            qn += cnt;
        }
    }
}

__global__ __launch_bounds__(AGG_BLOCK) void agg8_kernel(KArgs a) {
    constexpr uint32_t WAVES = AGG_BLOCK / 64;
    constexpr uint32_t FGRP = AGG8_SU, BGRP = AGG8_SU * 8;  // segments per work item: front, back
    constexpr uint32_t NFG = (AGG_MAX_NWG + AGG_PAD) / FGRP, NBG = (AGG_MAX_NWG + AGG_PAD) / BGRP;
    __shared__ Agg8Table lt;
    __shared__ uint32_t pc[AGG_MAX_NWG + AGG_PAD], pcb[AGG_MAX_NWG + AGG_PAD];  // front / back counts of this partition's segments, zero padded
    __shared__ uint16_t flv[NFG], blv[NBG];  // levels each group of segments needs (so that a wave never walks empty levels)
    __shared__ uint2 queues[WAVES * 64];
    const uint32_t part = blockIdx.x;
    const unsigned long long tk_start = FA_DBG(a, DBG_AGG8_TIMING) ? wall_clock64() : 0ull;
    for (int i = threadIdx.x; i < AGG8_ALL; i += AGG_BLOCK) {
        lt.key[i] = 0;
        lt.s1[i] = 0;
        lt.s2[i] = 0;
    }
    for (uint32_t i = threadIdx.x; i < AGG_MAX_NWG + AGG_PAD; i += AGG_BLOCK) {
        pc[i] = i < a.nwg ? a.seg_counts[(size_t)part * a.nwg + i] : 0u;
        pcb[i] = i < a.nwg ? a.seg_counts[((size_t)NPART_MAX + part) * a.nwg + i] : 0u;
    }
    const uint32_t tb_base = a.ctr->tb_base;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* pbase = a.seg + (((size_t)part * a.region) >> 1);
    __syncthreads();
    for (uint32_t g = threadIdx.x; g < NFG; g += AGG_BLOCK) {
        uint32_t m = 0;
        for (uint32_t s = 0; s < FGRP; s++) m = max(m, pc[g * FGRP + s]);
        flv[g] = (uint16_t)((m + 127u) >> 7);  // 64 lanes x 2 tuples per level
    }
    for (uint32_t g = threadIdx.x; g < NBG; g += AGG_BLOCK) {
        uint32_t m = 0;
        for (uint32_t s = 0; s < BGRP; s++) m = max(m, pcb[g * BGRP + s]);
        blv[g] = (uint16_t)(m ? (m / 2u + 1u + 7u) >> 3 : 0u);  // 8 pieces per segment and level; c tuples span up to c / 2 + 1 pieces
    }
    __syncthreads();
    uint2* queue = queues + wave * 64;
    // Multi-pass form (a.agg_passes > 1, chosen by the host from the groups the previous launches produced): when a
    // partition holds more groups than the LDS table, every tuple that finds the table full is a chain of memory-side
    // atomics (config 5: 60-second sub-buckets, 5 k groups per partition and launch - 2.6 ms instead of 0.1).  Pass s
    // takes the tuples whose key mixes to s, folds them, adds the groups to the device table and clears the LDS table;
    // the tuples of a partition (0.5 MB) are re-read from L2.
    const uint32_t npass = max(a.agg_passes, 1u), pmask = npass - 1u;
    uint32_t my_groups = 0, my_created = 0;
#pragma unroll 1
    for (uint32_t pass = 0; pass < npass; pass++) {
    const unsigned long long tm0 = (FA_DBG(a, DBG_TIMING | DBG_AGG8_TIMING)) ? (FA_DBG(a, DBG_AGG8_TIMING) ? wall_clock64() : clock64()) : 0ull;
    uint32_t qn = 0;
    unsigned long long hk = 0ull;  // this wave's heavy key, once it has met one (agg8_consume)
    // Work items of a wave: (group g, level j), g = wave, wave + WAVES, ...; j < levels(g).  The loads of the next item
    // fly while the current one is consumed (two register buffers; every fetch is unconditional - an item past the end
    // reads clamped addresses with zero counts).
#define FA_AGG8_PASS(BACK, LV, NG, GSEGS, PCNT)                                                           \
    {                                                                                                       \
        const uint32_t ngroups = (a.nwg + (GSEGS) - 1u) / (GSEGS);                                           \
        uint32_t g = wave, j = 0;                                                                           \
        auto settle_item = [&]() {                                                                          \
            while (g < ngroups && j >= (uint32_t)__builtin_amdgcn_readfirstlane((int)LV[min(g, (uint32_t)(NG) - 1u)])) { \
                g += WAVES;                                                                                 \
                j = 0;                                                                                      \
            }                                                                                               \
        };                                                                                                  \
        settle_item();                                                                                      \
        Agg8Batch b0, b1;                                                                                   \
        agg8_fetch<BACK, AGG8_SU>(a.capq, a.nwg, pbase, PCNT, g * (GSEGS), lane, j, b0);                                          \
        while (g < ngroups) {                                                                               \
            j++;                                                                                            \
            settle_item();                                                                                  \
            agg8_fetch<BACK, AGG8_SU>(a.capq, a.nwg, pbase, PCNT, g * (GSEGS), lane, j, b1);                                      \
            agg8_consume(a, lt, tb_base, part, lane, b0, queue, qn, pmask, pass, hk);                                       \
            if (g >= ngroups) break;                                                                        \
            j++;                                                                                            \
            settle_item();                                                                                  \
            agg8_fetch<BACK, AGG8_SU>(a.capq, a.nwg, pbase, PCNT, g * (GSEGS), lane, j, b0);                                      \
            agg8_consume(a, lt, tb_base, part, lane, b1, queue, qn, pmask, pass, hk);                                       \
        }                                                                                                   \
    }
    FA_AGG8_PASS(false, flv, NFG, FGRP, pc)
    FA_AGG8_PASS(true, blv, NBG, BGRP, pcb)
#undef FA_AGG8_PASS
    agg8_drain(a, lt, tb_base, part, lane, queue, qn);
    __syncthreads();
    const unsigned long long tm1 = (FA_DBG(a, DBG_TIMING | DBG_AGG8_TIMING)) ? (FA_DBG(a, DBG_AGG8_TIMING) ? wall_clock64() : clock64()) : 0ull;
    if (FA_DBG(a, DBG_AGG_NO_FLUSH)) return;
    // every group of this partition goes to the device-wide table once (quad-grouped: one atomic line transaction per
    // group; uniform trip count: the whole wave takes part in the quad rounds)
    constexpr int NF = (AGG8_ALL + AGG_BLOCK - 1) / AGG_BLOCK, FB = 3;  // slots per thread, in blocks of FB (registers)
    auto group_of = [&](int i, uint64_t& k0, uint64_t& k1, unsigned long long& s1, unsigned long long& s2) -> bool {
        const unsigned long long key = i < AGG8_ALL ? lt.key[i] : 0ull;
        s1 = i < AGG8_ALL ? lt.s1[i] : 0ull;
        s2 = i < AGG8_ALL ? lt.s2[i] : 0ull;
        TupleVals v;
        t8_unpack(make_uint2((uint32_t)key, (uint32_t)(key >> 32) << 26), part, tb_base, v);
        pack_key(tb_base + v.tbr, v.src_as, v.dst_as, v.etype, k0, k1);
        return key != 0ull && s2 != 0ull;
    };
    if (a.rlog2 == AS_RLOG2_MAX && !(a.dbg & DBG_AGG_ATOMIC_FLUSH)) {
        // This workgroup OWNS region `part` of the device table (table.cuh: the region of a compact-eligible key is its
        // partition, and the overflow path above only reaches keys of this partition too): plain loads and stores, no
        // atomics on the sums - three memory-side atomics per group were the whole cost of this flush.
        const uint32_t rmask = a.mask >> AS_RLOG2_MAX;
#pragma unroll 1
        for (int q0 = 0; q0 < NF; q0 += FB) {
            uint64_t fk0[FB], fk1[FB];
            unsigned long long fs1[FB], fs2[FB], cnt[FB];
            ulonglong2 kk[FB], vv[FB];
            Slot* hs[FB];
            bool valid[FB];
#pragma unroll
            for (int q = 0; q < FB; q++) {  // the home slots of this thread's groups: keys and sums, all loads together
                valid[q] = group_of((q0 + q) * AGG_BLOCK + (int)threadIdx.x, fk0[q], fk1[q], fs1[q], fs2[q]);
                hs[q] = &a.tab[part * (rmask + 1u) + (key_hash(fk0[q], fk1[q]) & rmask)];
                kk[q] = *reinterpret_cast<const ulonglong2*>(&hs[q]->k0);
                vv[q] = *reinterpret_cast<const ulonglong2*>(&hs[q]->bytes);
                cnt[q] = hs[q]->count;
            }
#pragma unroll
            for (int q = 0; q < FB; q++) {
                if (!valid[q]) continue;
                my_groups++;
                const unsigned long long b = fs1[q], p = fs2[q] >> 25, c = fs2[q] & 0x1ffffffull;
                bool done = false;
                if (kk[q].x == fk0[q] && kk[q].y == fk1[q]) {
                    *reinterpret_cast<ulonglong2*>(&hs[q]->bytes) = make_ulonglong2(vv[q].x + b, vv[q].y + p);
                    hs[q]->count = cnt[q] + c;
                    done = true;
                } else if (kk[q].x == 0) {
                    unsigned long long expect = 0ull;
                    if (__hip_atomic_compare_exchange_strong(&hs[q]->k0, &expect, (unsigned long long)fk0[q], __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        hs[q]->k1 = fk1[q];
                        *reinterpret_cast<ulonglong2*>(&hs[q]->bytes) = make_ulonglong2(b, p);
                        hs[q]->count = c;
                        my_created++;
                        done = true;
                    }
                }
                if (!done && !agg8_upsert_owned(a, fk0[q], fk1[q], as_next((uint32_t)(hs[q] - a.tab), a.mask, a.rlog2), b, p, c, my_created))
                    spill_park(a, fk0[q], fk1[q], b, p, c);
            }
        }
    } else {
    // (small tables - fewer than 256 regions: a region is shared by several partitions - keep the atomic protocol)
#pragma unroll 1
    for (int q0 = 0; q0 < NF; q0 += FB) {
        unsigned long long fk0[FB], fk1[FB], fs1[FB], fs2[FB];
        ulonglong2 home[FB];
        uint32_t fh[FB];
        bool valid[FB], claim[FB];
#pragma unroll
        for (int q = 0; q < FB; q++) {  // phase 1: the home-slot probes of this thread's groups fly together
            uint64_t k0, k1;
            valid[q] = group_of((q0 + q) * AGG_BLOCK + (int)threadIdx.x, k0, k1, fs1[q], fs2[q]);
            fk0[q] = k0;
            fk1[q] = k1;
            fh[q] = as_home(k0, k1, key_hash(k0, k1), a.mask, a.rlog2);
            home[q] = *reinterpret_cast<const ulonglong2*>(&a.tab[fh[q]]);
        }
        // phase 2: groups whose home slot is empty claim it - the two key words with one CAS each, issued for all FB
        // groups of the thread before the first answer is awaited
        unsigned long long r0[FB], r1[FB];
#pragma unroll
        for (int q = 0; q < FB; q++) {
            claim[q] = valid[q] && home[q].x == 0;
            r0[q] = home[q].x;
            r1[q] = home[q].y;
            if (claim[q]) r0[q] = atomicCAS(&a.tab[fh[q]].k0, 0ull, fk0[q]);
        }
#pragma unroll
        for (int q = 0; q < FB; q++) {
            if (claim[q]) {
                if (r0[q] == 0) r0[q] = fk0[q];
                claim[q] = r0[q] == fk0[q];  // (lost the slot to another key: the probing path below)
                if (claim[q]) r1[q] = atomicCAS(&a.tab[fh[q]].k1, 0ull, fk1[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < FB; q++) {
            Slot* sp = nullptr;
            const unsigned long long b = fs1[q], p = fs2[q] >> 25, c = fs2[q] & 0x1ffffffull;
            if (valid[q]) {
                my_groups++;
                if (claim[q] && r1[q] == 0) {  // this lane created the group
                    count_created(&a.ctr->used);
                    r1[q] = fk1[q];
                }
                if (r0[q] == fk0[q] && r1[q] == fk1[q]) sp = &a.tab[fh[q]];
                else sp = table_find_or_claim(a, fk0[q], fk1[q], key_hash(fk0[q], fk1[q]));
                if (!sp) spill_park(a, fk0[q], fk1[q], b, p, c);
            }
            quad_atomic_update(sp, b, p, c);
        }
    }
    }
    if (FA_DBG(a, DBG_AGG8_TIMING) && threadIdx.x == 0) {  // (FA_DEBUG_FLAGS=8388608: 100 MHz ticks per workgroup - set-up / segment walk / flush)
        __builtin_amdgcn_s_waitcnt(0);
        atomicAdd(&a.ctr->t_total, tm0 - tk_start);
        atomicAdd(&a.ctr->t_wait, tm1 - tm0);
        atomicAdd(&a.ctr->t_work, wall_clock64() - tm1);
        atomicAdd(&a.ctr->t_tiles, 1ull);
    }
    if ((FA_DBG(a, DBG_TIMING)) && threadIdx.x == 0) {  // (FA_DEBUG_FLAGS=1024: core clocks per workgroup and pass - fold / add to the device table)
        __builtin_amdgcn_s_waitcnt(0);
        atomicAdd(&a.ctr->t_wait, tm1 - tm0);
        atomicAdd(&a.ctr->t_work, (unsigned long long)clock64() - tm1);
        atomicAdd(&a.ctr->t_tiles, 1ull);
    }
    if (pass + 1 < npass) {  // next pass: empty table
        __syncthreads();
        for (int i = threadIdx.x; i < AGG8_ALL; i += AGG_BLOCK) {
            lt.key[i] = 0;
            lt.s1[i] = 0;
            lt.s2[i] = 0;
        }
        __syncthreads();
    }
    }
    // pass-count feedback: groups this launch added to the device table
    const uint32_t gw = (uint32_t)wave_sum_u64(my_groups), cw = (uint32_t)wave_sum_u64(my_created);
    __syncthreads();
    if (threadIdx.x == 0) pc[0] = pc[1] = 0;
    __syncthreads();
    if (lane == 0 && gw) atomicAdd(&pc[0], gw);
    if (lane == 0 && cw) atomicAdd(&pc[1], cw);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (pc[1]) atomicAdd(&a.ctr->used, (unsigned long long)pc[1]);
        if (pc[0]) atomicAdd(&a.ctr->agg_groups, (unsigned long long)pc[0]);
        if (blockIdx.x == 0) atomicAdd(&a.ctr->agg_launches, 1ull);
    }
}

// ---- Count-Min scatter sink: fold the sketch tuples -------------------------------------------------
// One work unit per (sketch, partition) - or per slice of a heavy one: the partition's counters (depth rows x 2^sub
// columns) live in a dense LDS array for the unit, every tuple {l1, l2, weight} of the partition's segments is `depth`
// LDS adds, and the array is folded into copy 0 of the sketch with plain, coalesced 64-bit read-modify-writes (the
// partition's `depth` blocks of 2^sub counters belong to this unit alone; slices of one partition use atomics; the
// atomic paths of other kernels use the other copies).  One persistent workgroup per CU takes units heaviest first.
// Segment geometry and walk: those of the wide (16-byte) flows_5m tuples, agg_fetch<BACK, false>.
// Batches of CMS_SU segments (round 3 measured 8 against 4 - twice the loads in flight per wave: the segment walk of a
// workgroup went from 41.6 to 44.7 us; it is not the latency of the loads that bounds it).
constexpr int CMS_PAD = CMS_SU * 8;  // zero counts behind the last segment (the back pass reads 8 segments per load)
struct CmsBatch {
    uint4 t[CMS_SU];
    uint32_t v;  // bit s = t[s] is a tuple of this lane (not a dummy load)
};
// agg_fetch<BACK, false> for batches of CMS_SU segments (same geometry, same unconditional loads)
template <bool BACK>
__device__ __forceinline__ void cms_fetch(const KArgs& a, const uint4* pbase, const uint32_t* pc, uint32_t w0, uint32_t lane, uint32_t j, CmsBatch& b) {
    constexpr uint32_t SEGS = BACK ? 8u : 1u, PER = 64u / SEGS;
    uint32_t idx[CMS_SU], seg[CMS_SU];
    b.v = 0;
#pragma unroll
    for (int s = 0; s < CMS_SU; s++) {
        seg[s] = w0 + (uint32_t)s * SEGS + (BACK ? lane / PER : 0u);
        const uint32_t c = pc[min(seg[s], (uint32_t)(AGG_MAX_NWG + CMS_PAD - 1))];  // 0 past nwg
        const uint32_t first = BACK ? a.capq - c : 0u;
        const uint32_t piece = first + PER * j + (BACK ? lane % PER : lane);
        const uint32_t valid = piece < first + c ? 1u : 0u;
        b.v |= valid << s;
        idx[s] = valid ? piece : 0u;
    }
#pragma unroll
    for (int s = 0; s < CMS_SU; s++) b.t[s] = pbase[(size_t)min(seg[s], a.nwg - 1u) * a.capq + idx[s]];
}

__global__ __launch_bounds__(AGG_BLOCK) void cms_agg_kernel(KArgs a, uint32_t set_mask, uint32_t cpar) {
    __shared__ unsigned long long arr[1u << CMS_PART_LOG2_MAX];  // 128 KiB
    __shared__ uint32_t pc[AGG_MAX_NWG + CMS_PAD], pcb[AGG_MAX_NWG + CMS_PAD];
    __shared__ uint32_t maxc_s[3];
    constexpr uint32_t CNFG = (AGG_MAX_NWG + CMS_PAD) / CMS_SU, CNBG = (AGG_MAX_NWG + CMS_PAD) / (CMS_SU * 8);
    __shared__ uint16_t flv[CNFG], blv[CNBG + 1];  // levels each group of segments needs: front parts, back parts
    __shared__ uint32_t fold_scr[(AGG_BLOCK / 64) * 192];  // 768 bytes per wave: wave_fold_lds
    // Which (sketch, partition) this workgroup takes: the HEAVIEST first, and the heavy ones in SLICES.  A partition that
    // holds a heavy hitter's key gets 3-4x the mean number of tuples (every wave tile of the chip that is not served by
    // its workgroup's hot-address cache sends one), one workgroup fits a CU, the grid is two rounds of them - and the
    // kernel lasted as long as the one workgroup of the heaviest partition.  So the schedule has cms_extra_units(nlog) spare
    // units: a partition above 1.25x the mean is cut into k <= 8 slices (every k-th group of segments; each slice
    // folds into its own LDS array and adds it to the sketch with atomics), heaviest first while spares last.  The sizes
    // are the PREVIOUS launch's (streams are stationary; copy `cpar` of cms_psize - the parity of THIS kernel's launches -
    // constant during this launch; this launch's sizes go to copy `cpar ^ 1`): every workgroup derives the same schedule from the same array - ranks with
    // ties by index are a permutation whatever the sizes, unit u of the schedule belongs to workgroup u.
    __shared__ uint32_t psz[CMS_SETS * CMS_NPART];
    __shared__ uint16_t byrank[CMS_SETS * CMS_NPART], want[CMS_SETS * CMS_NPART];  // partition of rank r; slices it wants
    __shared__ uint16_t ustart[CMS_SETS * CMS_NPART + 1];                          // first unit of rank r
    __shared__ uint32_t mine_s[4];
    const unsigned long long tk00 = FA_DBG(a, DBG_CMS_TIMING) ? wall_clock64() : 0ull;
    const uint32_t nlog = CMS_NPART * (set_mask == 3u ? 2u : 1u);  // logical ids: enabled sketches x partitions
    const uint32_t spare = cms_extra_units(nlog);
    for (uint32_t i = threadIdx.x; i < nlog; i += AGG_BLOCK) psz[i] = a.cms_psize[cpar * (CMS_SETS * CMS_NPART) + i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nlog; i += AGG_BLOCK) {
        const uint32_t v = psz[i];
        uint32_t rank = 0, tot = 0;
        for (uint32_t q = 0; q < nlog; q++) {
            rank += (psz[q] > v || (psz[q] == v && q < i)) ? 1u : 0u;
            tot += psz[q];
        }
        // slices: ceil(size / (1.25 x mean)); "heavy" (fold the rows by key first): above 1.5x the mean
        const unsigned long long vn = (unsigned long long)v * nlog;
        const uint32_t k = tot ? (uint32_t)min((vn * 4ull + 5ull * tot - 1ull) / (5ull * tot), 8ull) : 1u;
        byrank[rank] = (uint16_t)i;
        want[rank] = (uint16_t)(max(k, 1u) | (vn * 2ull > (unsigned long long)tot * 3ull ? 0x100u : 0u));
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive prefix of the extra slices over the ranks: 8 ranks per lane (nlog <= 512)
        uint32_t e[8], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t r = threadIdx.x * 8u + q;
            e[q] = r < nlog ? (want[r] & 0xffu) - 1u : 0u;
            sum += e[q];
        }
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
            if ((int)threadIdx.x >= o) inc += t;
        }
        uint32_t ex = inc - sum;
        if (threadIdx.x == 0) ustart[0] = 0;
#pragma unroll
        for (uint32_t q = 0; q < 8; q++) {
            const uint32_t r = threadIdx.x * 8u + q;
            ex += e[q];
            if (r < nlog) ustart[r + 1] = (uint16_t)(r + 1u + min(ex, spare));  // (extras granted in rank order until the spares are gone)
        }
    }
    __syncthreads();
    // The workgroups are PERSISTENT (one per CU: the 128 KiB array fills its LDS): unit blockIdx.x first, then whatever
    // unit the launch's counter hands out next - no second round of workgroups that all start together and wait for the
    // slowest, and the schedule above is worked out once per CU.  (Counter: word `cpar` behind the sizes; workgroup 0
    // clears word `cpar ^ 1` for the next launch of this kernel.)
    uint32_t* const next_unit = a.cms_psize + 2u * (CMS_SETS * CMS_NPART);
    if (blockIdx.x == 0 && threadIdx.x == 0) next_unit[cpar ^ 1u] = 0u;
    const uint32_t nunits = ustart[nlog];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned long long tk0 = tk00;
    for (uint32_t unit = blockIdx.x; unit < nunits;) {  // (workgroup-uniform)
    for (uint32_t r = threadIdx.x; r < nlog; r += AGG_BLOCK) {
        const uint32_t u0 = ustart[r], u1 = ustart[r + 1];
        if (unit >= u0 && unit < u1) {
            mine_s[0] = byrank[r];
            mine_s[1] = unit - u0;  // slice
            mine_s[2] = u1 - u0;    // of k
            mine_s[3] = want[r] >> 8;
        }
    }
    __syncthreads();
    const uint32_t logical = mine_s[0], slice = mine_s[1], nslice = mine_s[2];
    const bool heavy = mine_s[3] != 0u;  // workgroup-uniform
    // logical id -> (sketch, partition): only the enabled sketches have workgroups
    const uint32_t set = (set_mask == 3u) ? logical / CMS_NPART : (set_mask >> 1);
    const uint32_t prefix = logical % CMS_NPART, part = set * CMS_NPART + prefix;
    const uint32_t sub = a.cms_sub, ncnt = a.cms_depth << sub;
    for (uint32_t i = threadIdx.x; i < ncnt; i += AGG_BLOCK) arr[i] = 0;
    if (threadIdx.x < 3) maxc_s[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mymax = 0, mymaxb = 0, mysum = 0;
    for (uint32_t i = threadIdx.x; i < AGG_MAX_NWG + CMS_PAD; i += AGG_BLOCK) {
        const uint32_t c = i < a.nwg ? a.cseg_counts[(size_t)part * a.nwg + i] : 0u;
        const uint32_t cb = i < a.nwg ? a.cseg_counts[((size_t)CMS_SETS * CMS_NPART + part) * a.nwg + i] : 0u;
        pc[i] = c;
        pcb[i] = cb;
        mymax = max(mymax, c);
        mymaxb = max(mymaxb, cb);
        mysum += c + cb;
    }
    for (int o = 32; o > 0; o >>= 1) {
        mymax = max(mymax, (uint32_t)__shfl_xor((int)mymax, o));
        mymaxb = max(mymaxb, (uint32_t)__shfl_xor((int)mymaxb, o));
        mysum += (uint32_t)__shfl_xor((int)mysum, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&maxc_s[0], mymax);
        atomicMax(&maxc_s[1], mymaxb);
        if (mysum) atomicAdd(&maxc_s[2], mysum);
    }
    const uint4* pbase = a.cseg + (size_t)part * a.cregion;
    KArgs g = a;  // (cms_fetch reads the segment geometry from capq / nwg)
    g.capq = a.ccapq;
    __syncthreads();
    const uint32_t maxcb = maxc_s[1];
    if (threadIdx.x == 0 && slice == 0u) a.cms_psize[(cpar ^ 1u) * (CMS_SETS * CMS_NPART) + logical] = maxc_s[2];  // (the next launch's schedule)
    for (uint32_t gq = threadIdx.x; gq < CNFG; gq += AGG_BLOCK) {
        uint32_t m = 0;
        for (uint32_t q = 0; q < (uint32_t)CMS_SU; q++) m = max(m, pc[gq * CMS_SU + q]);
        flv[gq] = (uint16_t)((m + 63u) >> 6);  // level j: tuples [64j, 64j+64) of a segment
    }
    for (uint32_t gq = threadIdx.x; gq < CNBG; gq += AGG_BLOCK) {
        uint32_t m = 0;
        for (uint32_t q = 0; q < (uint32_t)CMS_SU * 8u; q++) m = max(m, pcb[gq * CMS_SU * 8u + q]);
        blv[gq] = (uint16_t)((m + 7u) >> 3);   // back parts: 8 tuples per segment and level
    }
    __syncthreads();
    // A heavy hitter sends one tuple per wave tile: a good part of every 64 consecutive tuples of its partition carries the
    // SAME key, and LDS atomics on one address serialize - across the lanes of a wave and across the 16 waves of the
    // workgroup: the partition of the stream's heaviest key set the duration of the whole kernel (2.5x the mean workgroup).
    // So the 64 tuples of a load are folded by key first, whatever their mix (wave_fold_lds: one hash-claim round in a
    // wave-private LDS table - the first lane of a key keeps it and receives the others' weights): one lane per distinct
    // key adds the row's sum.  (Round 2 probed only the first and the last lane's key: a heavy key that holds a third of
    // the lanes was caught half of the time.)
    uint32_t* const fscr = fold_scr + wave * 192u;
    auto consume = [&](const CmsBatch& b) {
#pragma unroll
        for (int e = 0; e < CMS_SU; e++) {
            const uint4& q = b.t[e];
            uint64_t w = (uint64_t)q.w << 32 | q.z;
            bool v = ((b.v >> e) & 1u) && w;
            if (__builtin_amdgcn_ballot_w64(v) == 0ull) continue;
            if (FA_DBG(a, DBG_AGG_NO_LDS)) {  // (ablation: the loads only)
                if (v && (q.x ^ q.y ^ q.z) == 0x12345u && q.w == 0x777u) arr[0] = 1;
                continue;
            }
            if (heavy) wave_fold_lds(fscr, v, (uint64_t)q.y << 32 | q.x, (uint64_t)0, w);
            if (v)
                for (uint32_t r = 0; r < a.cms_depth; r++) atomicAdd(&arr[(r << sub) + cms_low(q.x, q.y, r, sub)], (unsigned long long)w);
        }
    };
    // Work items of a wave: (group g of CMS_SU segments, level j), g = the wave's share of this slice's groups; j <
    // levels(g) - a wave never walks the empty levels of short segments up to the partition's longest one (a heavy
    // hitter's workgroup fills its segment of the key's partition 4-5x beyond the mean: round 2 walked every segment to
    // that length).  The loads of the next item fly while the current one is consumed (two register buffers; every fetch
    // is unconditional - an item past the end reads clamped addresses with zero counts).
    // (Round 3 measured: items drawn one at a time from an LDS counter even the waves out - the slowest wave of a
    // workgroup walks 1.2x the mean here - and the walk takes as long as before; so does a walk with the LDS adds
    // removed (47.7 vs 52.7 us per unit) and one with twice the loads in flight.  What bounds it is the CU's share of the
    // memory system: ~0.7 MB of 1-KiB pieces per unit.)
    constexpr uint32_t WAVES = AGG_BLOCK / 64;
#define FA_CMS_PASS(BACK, LV, GSEGS, PCNT)                                                       \
    {                                                                                              \
        const uint32_t ngroups = (a.nwg + (GSEGS) - 1u) / (GSEGS);                                  \
        uint32_t gi = wave * nslice + slice, j = 0;  /* this slice: groups = slice (mod nslice) */   \
        auto settle_item = [&]() {                                                                 \
            while (gi < ngroups && j >= (uint32_t)__builtin_amdgcn_readfirstlane((int)LV[gi])) {   \
                gi += WAVES * nslice;                                                              \
                j = 0;                                                                             \
            }                                                                                      \
        };                                                                                         \
        settle_item();                                                                             \
        CmsBatch b0, b1;                                                                           \
        cms_fetch<BACK>(g, pbase, PCNT, gi * (GSEGS), lane, j, b0);                                \
        while (gi < ngroups) {                                                                     \
            j++;                                                                                   \
            settle_item();                                                                         \
            cms_fetch<BACK>(g, pbase, PCNT, gi * (GSEGS), lane, j, b1);                            \
            consume(b0);                                                                           \
            if (gi >= ngroups) break;                                                              \
            j++;                                                                                   \
            settle_item();                                                                         \
            cms_fetch<BACK>(g, pbase, PCNT, gi * (GSEGS), lane, j, b0);                            \
            consume(b1);                                                                           \
        }                                                                                          \
    }
    const unsigned long long tk1 = FA_DBG(a, DBG_CMS_TIMING) ? wall_clock64() : 0ull;
    FA_CMS_PASS(false, flv, (uint32_t)CMS_SU, pc)
    if (maxcb) FA_CMS_PASS(true, blv, (uint32_t)CMS_SU * 8u, pcb)
#undef FA_CMS_PASS
    if (FA_DBG(a, DBG_CMS_TIMING) && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        atomicAdd(&a.ctr->t_total, wall_clock64() - tk1);  // (sum over the waves: mean wave vs the workgroup's walk = imbalance)
    }
    __syncthreads();
    const unsigned long long tk2 = FA_DBG(a, DBG_CMS_TIMING) ? wall_clock64() : 0ull;
    // (eight counters per thread and round: the loads of a round are issued together - a load behind the previous
    // counter's conditional store cannot move above it, and sixteen dependent round trips to HBM were a tenth of a
    // workgroup's time)
    unsigned long long* sk = set ? a.cms_dst : a.cms_src;
    constexpr uint32_t FL = 8;
    for (uint32_t i0 = threadIdx.x; i0 < ncnt; i0 += AGG_BLOCK * FL) {
        unsigned long long v[FL], old[FL];
        size_t at[FL];
#pragma unroll
        for (uint32_t q = 0; q < FL; q++) {
            const uint32_t i = min(i0 + q * AGG_BLOCK, ncnt - 1u);
            v[q] = i0 + q * AGG_BLOCK < ncnt ? arr[i] : 0ull;
            at[q] = ((size_t)(i >> sub) << a.cms_wl2) + ((size_t)prefix << sub) + (i & ((1u << sub) - 1u));
        }
        if (nslice > 1u) {  // (workgroup-uniform) the partition's other slices add to the same counters
#pragma unroll
            for (uint32_t q = 0; q < FL; q++)
                if (v[q]) atomicAdd(&sk[at[q]], v[q]);
            continue;
        }
#pragma unroll
        for (uint32_t q = 0; q < FL; q++) old[q] = sk[at[q]];
#pragma unroll
        for (uint32_t q = 0; q < FL; q++)
            if (v[q]) sk[at[q]] = old[q] + v[q];
    }
    if (FA_DBG(a, DBG_CMS_TIMING) && threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tk3 = wall_clock64();
        atomicAdd(&a.ctr->t_wait, (tk1 - tk0) + (tk3 - tk2));
        atomicAdd(&a.ctr->t_work, tk2 - tk1);
        atomicAdd(&a.ctr->t_tiles, 1ull);
        tk0 = tk3;
    }
    __syncthreads();  // (the array, the counts and mine_s are the next unit's from here)
    if (threadIdx.x == 0) mine_s[0] = gridDim.x + atomicAdd(&next_unit[cpar], 1u);
    __syncthreads();
    unit = mine_s[0];
    __syncthreads();
    }
}

}  // namespace fa
