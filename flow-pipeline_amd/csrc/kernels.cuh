// kernels.cuh - the gfx950 kernels of libflowagg.
//
// Hot path (per batch):  probe_kernel -> wtile_kernel<KEYSETS> -> deferred_kernel -> agg_kernel
//   wire bytes in HBM --global_load_lds_dwordx4 nt (async DMA)--> the wave's LDS tile (<= 64 records)
//   -> one record per lane parsed out of LDS (wire.cuh, parse_canon)
//   -> key = (TimeReceived/granule, SrcAS, DstAS, EType)   [create.sh:92-110]
//   -> per-workgroup LDS hash table absorbs hot keys (mocker.go:61-62 has 9 groups)
//   -> everything else leaves the workgroup as a 16-byte tuple in the workgroup's PRIVATE segment of the
//      key's hash partition, 8 tuples = one aligned 128-byte line at a time (LDS bins; positions from LDS
//      counters - no global atomics: MI355X retires only ~23.7 G global-atomic line requests/s,
//      tools/sink_bench.hip);
//   (tile_kernel = the 256-thread workgroup-tile form: decode path, direct sink, FA_TILE=wg)
//   agg_kernel: one 1024-thread workgroup per key partition streams the partition's segments back,
//      aggregates them in an LDS hash table (two packed 64-bit LDS atomics per tuple) and adds each
//      group to the device-wide table once.
//   Records parse_canon is not sure about go to deferred_kernel (parse_fast, any field order, then
//   parse_generic, complete semantics); values that do not fit a tuple take
//   the direct device-wide-table path (64-bit atomics).
//
// Roofline: HBM-bound integer/byte work; algorithmic bytes = wire bytes, read
// once (DESIGN.md "Roofline").  No MFMA anywhere - nothing here is a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gen.cuh"
#include "table.cuh"
#include "wide.cuh"
#include "wire.cuh"

namespace fa {

constexpr int BLOCK = 256;
constexpr int TILE_BYTES = 21760;  // one LDS tile buffer: 256 records x 85 B (framed mocker records are <= 85)
constexpr int TILE_PAD = 112;      // readable slack behind the staged bytes (window / address reads)
constexpr int TILE_STRIDE = TILE_BYTES + TILE_PAD;
constexpr int LDS_SLOTS = 64;      // per-workgroup hot-key slots (2.5 KiB)
constexpr int LDS_PROBES = 2;
constexpr int PART_LOG2_MAX = 8;   // key partitions of the scatter sink (<= 256: see tools/scatter_bench.hip)
constexpr int NPART_MAX = 1 << PART_LOG2_MAX;
constexpr int WBLOCK = 512;         // wave-tile kernel: 8 waves, each with a private LDS tile of <= 64 records
constexpr int WT_RECS = 64;
constexpr int WT_STRIDE = 5472;     // 64 records x 85 B + alignment slack (16-byte multiple); overreads land in the next tile / the bins
constexpr uint32_t BIN_CAP = 8;     // a bin = one 128-byte line of tuples per key partition (256 x 8 x 16 B = 32 KiB per workgroup)
constexpr int AGG_BLOCK = 1024;    // agg_kernel: 16 waves share one LDS table
#ifndef FA_AGG_SLOTS
#define FA_AGG_SLOTS 4096
#endif
#ifndef FA_AGG_SPLIT
#define FA_AGG_SPLIT 1
#endif
constexpr int AGG_SLOTS = FA_AGG_SLOTS;  // 32 B of LDS per slot (4096: 128 KiB)
constexpr int AGG_SPLIT = FA_AGG_SPLIT;  // workgroups per key partition (each with its own LDS table)
constexpr int AGG_PROBES = 16;
constexpr uint32_t TUPLE_TB_SPAN = 16;        // time buckets a batch may span on the tuple path
constexpr uint32_t TUPLE_MAX_BYTES = 1u << 28, TUPLE_MAX_PACKETS = 1u << 15, TUPLE_MAX_ETYPE = 1u << 16;
constexpr uint32_t AGG_MAX_BATCH = 1u << 24;  // count <= 2^24 per slot keeps the packed LDS sums exact
static_assert(TILE_STRIDE % 16 == 0, "LDS tile buffers must stay 16-byte aligned");

enum { MODE_INGEST = 0, MODE_DECODE = 1 };
// Kernel variants are compiled for the key-set masks 1..7 (rollup and/or sketches); every other
// combination runs the KS_ALL variant, which parses the union of the columns and tests the runtime mask.
constexpr uint32_t KS_ALL = 0xFFu;
constexpr uint32_t FA_KEYS_WIDE = FA_KEYS_ADDR_PORT_PROTO | FA_KEYS_PORT_HIST | FA_KEYS_MINUTE_SERIES;
constexpr uint32_t PORT_DENSE = 65536;  // ports below this live in the dense histograms
// ablation switches (env FA_DEBUG_FLAGS; measurement only - results are wrong when set)
enum { DBG_NO_SINK = 1, DBG_LOOP_PARSER = 2, DBG_NO_LDS_TABLE = 4, DBG_NO_GLOBAL = 8, DBG_NO_PARSE = 16, DBG_NO_TUPLE_STORE = 32,
       DBG_AGG_NO_LDS = 64, DBG_AGG_NO_FLUSH = 128, DBG_AGG_NO_SLOW = 256, DBG_DMA_NO_NT = 512, DBG_TIMING = 1024, DBG_TUPLE_NT = 2048, DBG_TUPLE_SC = 4096, DBG_NO_LANE_OFF = 8192, DBG_SYNTH_TILES = 16384, DBG_NOT_MINE = 32768, DBG_NO_FRAME = 131072 };

struct SpillEntry {
    unsigned long long k0, k1, bytes, packets, count;
};

struct Counters {
    unsigned long long ok, bad, slow, spill_lost, used, direct, retried;
    unsigned int exotic_count, spill_count, rows_count, retry_count, tb_base, ks_overflow, ks_rows, pad;
    unsigned long long t_wait, t_work, t_tiles, t_total;  // DBG_TIMING: core-clock cycles of wave 0 of every workgroup
    unsigned long long wused, wspill_lost;  // wide table (wide.cuh)
    unsigned int wspill_count, wrows_count;
};

// Distinct-address set behind fa_topk (the dashboards rank EVERY address,
// viz-ch.json:233,479).  32-byte slots; tag = 0 (empty) | bit 63 (claimed) | bit 62 (key written) |
// 62 hash bits of the key.
struct __attribute__((aligned(32))) KeySlot {
    unsigned long long tag, lo, hi, pad;
};
constexpr unsigned long long KS_CLAIMED = 1ull << 63, KS_READY = 1ull << 62;
struct TopkRow {
    unsigned long long lo, hi, weight;
};

struct ColumnPtrs {
    uint64_t *time_received, *time_flow_start, *sampling_rate, *bytes, *packets;
    uint32_t *sequence_num, *src_as, *dst_as, *etype, *proto, *src_port, *dst_port;
    uint4 *sampler_address, *src_addr, *dst_addr;
    uint8_t* status;
};

struct KArgs {
    const uint8_t* buf;   // 16-byte aligned device pointer
    const uint32_t* off;  // n+1 offsets
    uint32_t n;
    uint32_t framed;
    uint32_t gran;
    Slot* tab;
    uint32_t mask;
    SpillEntry* spill;
    uint32_t spill_cap;
    Counters* ctr;
    uint32_t* exotic_idx;
    unsigned long long* cms_src;
    unsigned long long* cms_dst;
    uint32_t cms_depth, cms_wl2;
    uint64_t cms_seed;
    KeySlot* ks_src;  // distinct SrcAddr / DstAddr values seen (nullptr when the key set is off)
    KeySlot* ks_dst;
    uint32_t ks_mask;
    ColumnPtrs cols;
    uint32_t tile_recs;  // records per tile (<= BLOCK), chosen by the host from the mean record size
    uint32_t dbg;  // FA_DEBUG_FLAGS ablation switches (0 in production)
    uint32_t* retry_idx;   // records parse_canon deferred
    double gran_recip;     // (1/gran)(1+2^-40): floor(t * gran_recip) == t / gran for every u32 t
    // scatter sink (seg == nullptr: every record takes the direct device-wide-table path)
    uint4* seg;            // [NPART][region] tuples; partition p, workgroup w: seg[p*region + w*capq + q]
    uint32_t* seg_counts;  // [2][NPART][nwg]: tuples at the front of a segment, tuples at its back (wave-tile kernel only)
    uint32_t capq;         // tuples per (partition, workgroup) segment (multiple of 8 = 128-byte lines)
    uint32_t capf, capb;   // wave-tile kernel: front part (full lines, grows up from 0) and back part (single tuples, grows down from capq-1)
    uint32_t nwg;          // workgroups of the tile kernel that filled the segments
    unsigned long long region;  // tuples per partition (nwg*capq plus a skew against power-of-two strides)
    uint32_t plog2;        // log2(key partitions)
    // wide key sets (wide.cuh)
    uint32_t key_sets;     // runtime mask (the KS_ALL kernel variant tests it)
    WSlot* wtab;
    uint32_t wmask;
    WSpillEntry* wspill;
    uint32_t wspill_cap;
    ulonglong2* port_hist;  // [2][PORT_DENSE] {sum(Bytes*SamplingRate), count()}: SrcPort, then DstPort
};

__device__ __forceinline__ WArgs wargs(const KArgs& a) {
    return WArgs{a.wtab, a.wmask, a.wspill, a.wspill_cap, &a.ctr->wspill_count, &a.ctr->wspill_lost, &a.ctr->wused};
}
// does this kernel variant serve key set X for this launch?
template <uint32_t KEYSETS>
__device__ __forceinline__ bool ks_on(const KArgs& a, uint32_t x) {
    return (KEYSETS & x) != 0 && (KEYSETS != KS_ALL || (a.key_sets & x) != 0);
}

// ---- sinks ------------------------------------------------------------------
__device__ __forceinline__ void agg_global(const KArgs& a, uint64_t k0, uint64_t k1, uint32_t h,
                                           uint64_t b, uint64_t p, uint64_t c) {
    uint32_t i = h & a.mask;
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = (i + 1) & a.mask) {
        Slot* s = &a.tab[i];
        unsigned long long c0 = s->k0;
        if (c0 == 0) c0 = atomicCAS(&s->k0, 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        unsigned long long c1 = s->k1;
        if (c1 == 0) {
            c1 = atomicCAS(&s->k1, 0ull, (unsigned long long)k1);
            if (c1 == 0) atomicAdd(&a.ctr->used, 1ull);  // this lane created the group
        }
        if (c1 != 0 && c1 != k1) continue;
        if (b) atomicAdd(&s->bytes, (unsigned long long)b);
        if (p) atomicAdd(&s->packets, (unsigned long long)p);
        atomicAdd(&s->count, (unsigned long long)c);
        return;
    }
    // probe limit: park the partial aggregate; the host grows the table and replays it
    unsigned int j = atomicAdd(&a.ctr->spill_count, 1u);
    if (j < a.spill_cap) {
        a.spill[j] = SpillEntry{k0, k1, b, p, c};
    } else {
        atomicAdd(&a.ctr->spill_lost, 1ull);
    }
}

__device__ __forceinline__ uint64_t cms_hash(uint64_t lo, uint64_t hi, uint64_t seed, uint32_t row) {
    uint64_t h = mix64(lo ^ mix64(seed + 0x9E3779B97F4A7C15ull * (row + 1)));
    return mix64(h ^ hi);
}
// The sketch is kept in CMS_REPLICAS copies; a workgroup adds to copy blockIdx % CMS_REPLICAS and the copies
// are summed into copy 0 before anything reads the sketch (cms_fold_kernel).  Counters of heavy hitters are
// hit by every wave of the chip, and same-address atomics serialize at the memory side (~10 ns each:
// 1.9 M updates of the top Zipf-1.1 key per launch cost ~19 ms on one copy); u64 sums commute, so the folded
// sketch is bit-identical to a single-copy one.
constexpr uint32_t CMS_REPLICAS = 8;
__device__ __forceinline__ void cms_add(unsigned long long* cms, uint32_t depth, uint32_t wl2,
                                        uint64_t seed, const uint32_t key[4], uint64_t w) {
    if (w == 0) return;
    uint64_t lo = (uint64_t)key[1] << 32 | key[0], hi = (uint64_t)key[3] << 32 | key[2];
    unsigned long long* copy = cms + (size_t)(blockIdx.x % CMS_REPLICAS) * ((size_t)depth << wl2);
    for (uint32_t r = 0; r < depth; r++) {
        uint64_t h = cms_hash(lo, hi, seed, r);
        atomicAdd(&copy[((size_t)r << wl2) + (size_t)(h >> (64 - wl2))], (unsigned long long)w);
    }
}
__global__ void cms_fold_kernel(unsigned long long* cms, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long sum = 0;
        for (uint32_t r = 1; r < CMS_REPLICAS; r++) {
            const unsigned long long v = cms[r * words + i];
            if (v) {
                sum += v;
                cms[r * words + i] = 0;
            }
        }
        if (sum) cms[i] += sum;
    }
}

// Folds the lanes of a wave that carry the same 16-byte key: the first lane to claim the key's slot in a
// 64-entry LDS table keeps the key and receives the weights of the others (valid = false for those).  One
// round whatever the key distribution (wave_combine gives up on skewed mixes of hot and cold keys); lanes that
// lose the slot to a DIFFERENT key just stay on their own.  scratch: 768 bytes of wave-private LDS.
__device__ __forceinline__ void wave_fold_lds(uint32_t* scratch, bool& valid, uint64_t lo, uint64_t hi, uint64_t& w) {
    uint32_t* owner = scratch;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(scratch + 64);
    const uint32_t ln = __lane_id();
    owner[ln] = 0xffffffffu;
    acc[ln] = 0;
    uint32_t h = (uint32_t)lo * 0x9E3779B1u ^ (uint32_t)(lo >> 32) * 0x85EBCA6Bu ^ (uint32_t)hi * 0xC2B2AE35u ^ (uint32_t)(hi >> 32) * 0x27D4EB2Fu;
    h ^= h >> 15;
    const uint32_t slot = (h * 0x2545F491u) >> 26;
    uint32_t win = ln;
    if (valid) {
        const uint32_t prev = atomicCAS(&owner[slot], 0xffffffffu, ln);
        win = prev == 0xffffffffu ? ln : prev;
    }
    const uint64_t wlo = (uint64_t)(uint32_t)__shfl((int)(uint32_t)(lo >> 32), (int)win) << 32 | (uint32_t)__shfl((int)(uint32_t)lo, (int)win);
    const uint64_t whi = (uint64_t)(uint32_t)__shfl((int)(uint32_t)(hi >> 32), (int)win) << 32 | (uint32_t)__shfl((int)(uint32_t)hi, (int)win);
    const bool same = valid && win != ln && wlo == lo && whi == hi;
    if (same) {
        if (w) atomicAdd(&acc[slot], (unsigned long long)w);
        valid = false;
    }
    if (valid && win == ln) w += acc[slot];  // (behind the adds: LDS operations of a wave complete in order)
}

// Inserts a FixedString(16) key into the distinct-key set.  The per-XCD L2s are not coherent, so a plain
// load may show an OLD version of a slot - harmless for the fast path (a slot never changes once its
// key is written, so a complete match is always true), but everything else must come from the memory
// side: the slot is claimed by CAS on its tag (hash of the key), the key words are written with
// returning atomics, then the READY bit is set; a lane that needs to compare against a slot owned by
// an equal tag reads the key words with atomics as well.  A lane that meets an equal tag whose key is
// not written yet cannot compare and moves on, so a key may (rarely) be stored twice - fa_topk removes
// duplicates.  The set is exact in content: a key is dropped only when the table is full, and that is
// reported (ks_overflow -> FA_ERR_TABLE_FULL).
__device__ __forceinline__ void keyset_insert(const KArgs& a, KeySlot* tab, const uint32_t key[4]) {
    const unsigned long long lo = (unsigned long long)key[1] << 32 | key[0], hi = (unsigned long long)key[3] << 32 | key[2];
    uint32_t h = key[0] * 0x9E3779B1u + key[1];
    h ^= h >> 15;
    h = (h ^ key[2]) * 0x85EBCA6Bu + key[3];
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    uint32_t g = (key[3] ^ 0x27D4EB2Fu) * 0x165667B1u + key[2];
    g ^= g >> 15;
    g = (g ^ key[1]) * 0xD3A2646Du + key[0];
    g ^= g >> 14;
    const unsigned long long mytag = KS_CLAIMED | (((unsigned long long)g << 32 | h) & (KS_READY - 1));
    uint32_t i = h & a.ks_mask;
    for (int probe = 0; probe < 256; probe++, i = (i + 1) & a.ks_mask) {
        KeySlot* s = &tab[i];
        // fastest path: the key is already there and this XCD's L2 knows it.  Plain (cached) loads may be stale,
        // but a slot never changes once READY, so a complete match is always true; anything else is looked at
        // again through the memory side below.
        {
            const ulonglong2 c01 = *reinterpret_cast<const ulonglong2*>(&s->tag);  // tag, lo
            if (c01.x == (mytag | KS_READY) && c01.y == lo && s->hi == hi) return;
        }
        // the key may be there: system-scope loads are served by the memory side, past the (incoherent) per-XCD
        // L2s
        unsigned long long t = __hip_atomic_load(&s->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (t == (mytag | KS_READY)) {
            const unsigned long long l = __hip_atomic_load(&s->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long q = __hip_atomic_load(&s->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (l == lo && q == hi) return;
            continue;  // equal tag, different key
        }
        if (t != 0 && t != mytag) continue;  // somebody else's slot
        bool done = false;
        if (t == 0) {
            t = atomicCAS(&s->tag, 0ull, mytag);
            if (t == 0) {  // claimed: publish the key, then mark it readable
                const unsigned long long o1 = atomicExch(&s->lo, lo), o2 = atomicExch(&s->hi, hi);
                if ((o1 & o2) != ~0ull) atomicOr(&s->tag, KS_READY);  // (always true: orders the OR behind both writes)
                done = true;
            }
        }
        // (claimers of this wave have published by now; owners in other waves are a few instructions away)
        if (!done && (t | KS_READY) == (mytag | KS_READY)) {
            for (int spin = 0; spin < 4096 && !(t & KS_READY); spin++) t = atomicOr(&s->tag, 0ull);
            if (t & KS_READY) {
                const unsigned long long l = atomicAdd(&s->lo, 0ull), q = atomicAdd(&s->hi, 0ull);  // memory-side reads
                done = l == lo && q == hi;
            }
        }
        if (done) return;
    }
    atomicAdd(&a.ctr->ks_overflow, 1u);
}

__device__ __forceinline__ void store_columns(const ColumnPtrs& c, uint32_t idx, const Rec& r,
                                              uint8_t status) {
    c.time_received[idx] = r.time_received;
    c.time_flow_start[idx] = r.time_flow_start;
    c.sampling_rate[idx] = r.sampling_rate;
    c.bytes[idx] = r.bytes;
    c.packets[idx] = r.packets;
    c.sequence_num[idx] = r.sequence_num;
    c.src_as[idx] = r.src_as;
    c.dst_as[idx] = r.dst_as;
    c.etype[idx] = r.etype;
    c.proto[idx] = r.proto;
    c.src_port[idx] = r.src_port;
    c.dst_port[idx] = r.dst_port;
    c.sampler_address[idx] = make_uint4(r.sampler[0], r.sampler[1], r.sampler[2], r.sampler[3]);
    c.src_addr[idx] = make_uint4(r.src[0], r.src[1], r.src[2], r.src[3]);
    c.dst_addr[idx] = make_uint4(r.dst[0], r.dst[1], r.dst[2], r.dst[3]);
    c.status[idx] = status;
}

template <uint32_t KEYSETS>
constexpr uint32_t cols_for_keysets() {
    uint32_t c = 0;
    if (KEYSETS & FA_KEYS_AS_PAIR) c |= COLS_AS_ROLLUP;
    if (KEYSETS & FA_KEYS_SRCADDR_CMS) c |= COL_SRC_ADDR | COL_BYTES | COL_SAMPLING_RATE;
    if (KEYSETS & FA_KEYS_DSTADDR_CMS) c |= COL_DST_ADDR | COL_BYTES | COL_SAMPLING_RATE;
    if (KEYSETS & FA_KEYS_ADDR_PORT_PROTO) c |= COL_TIME_RECEIVED | COL_SRC_ADDR | COL_DST_PORT | COL_PROTO | COL_BYTES | COL_PACKETS;
    if (KEYSETS & FA_KEYS_PORT_HIST) c |= COL_SRC_PORT | COL_DST_PORT | COL_BYTES | COL_SAMPLING_RATE;
    if (KEYSETS & FA_KEYS_MINUTE_SERIES) c |= COL_TIME_FLOW_START | COL_BYTES | COL_SAMPLING_RATE;
    return c;
}

// ---- LDS DMA staging ------------------------------------------------------------
// Copies nbytes (rounded up to 16) from 16-byte-aligned global memory into an LDS
// buffer with `global_load_lds_dwordx4`: 1 KiB per wave-instruction, no VGPR round
// trip, asynchronous (tracked by vmcnt).  Lanes past the end are masked off.
template <int AUX = 0>
__device__ __forceinline__ void dma_to_lds(const uint8_t* g, uint32_t nbytes, uint32_t* lds) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t npieces = (nbytes + 1023u) >> 10;
    for (uint32_t p = wave; p < npieces; p += BLOCK / 64) {
        const uint32_t o = p * 1024u + lane * 16u;
        if (o < nbytes) {
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(g + o),
                (__attribute__((address_space(3))) void*)(lds + p * 256u), 16, 0, AUX);
        }
    }
}
// vmcnt(0) through the builtin (not inline asm) so that the compiler's own waitcnt
// scoreboard learns the DMA has landed and does not re-drain before LDS reads.
__device__ __forceinline__ void dma_wait_all() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// ---- device-wide table probe ------------------------------------------------------
// Finds or claims the slot of (k0,k1); returns nullptr when the probe limit is hit.
__device__ __forceinline__ Slot* table_find_or_claim(const KArgs& a, uint64_t k0, uint64_t k1, uint32_t h) {
    uint32_t i = h & a.mask;
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = (i + 1) & a.mask) {
        Slot* s = &a.tab[i];
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(s);  // one 16-byte load: k0,k1
        unsigned long long c0 = kk.x, c1 = kk.y;
        if (c0 == k0 && c1 == k1) return s;  // common case: no atomics on the key words
        if (c0 == 0) c0 = atomicCAS(&s->k0, 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        if (c1 == 0) {
            c1 = atomicCAS(&s->k1, 0ull, (unsigned long long)k1);
            if (c1 == 0) atomicAdd(&a.ctr->used, 1ull);  // this lane created the group
        }
        if (c1 != 0 && c1 != k1) continue;
        return s;
    }
    return nullptr;
}

// Broadcast lane Q of every quad to the 4 lanes of that quad (DPP quad_perm, VALU only).
template <int Q>
__device__ __forceinline__ uint64_t quad_bcast_u64(uint64_t v) {
    constexpr int CTRL = Q | (Q << 2) | (Q << 4) | (Q << 6);
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xf, 0xf, false);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xf, 0xf, false);
    return (uint64_t)hi << 32 | lo;
}

// Quad-grouped atomics.  The memory side retires ~23.7 G atomic cache-line
// transactions/s no matter how many lanes of one instruction hit the line
// (tools/atomics_bench.hip), so the three sums of a slot are issued by three
// adjacent lanes of ONE instruction: in round Q every quad works on the record of
// its lane Q; lane w of the quad adds word w (bytes, packets, count).  One
// transaction per record instead of three.  Must be called by the full wave.
template <int Q, int VOFF>
__device__ __forceinline__ void quad_round(uint64_t ptr, uint64_t b, uint64_t p, uint64_t c, uint32_t w) {
    const uint64_t qp = quad_bcast_u64<Q>(ptr);
    const uint64_t qb = quad_bcast_u64<Q>(b), qq = quad_bcast_u64<Q>(p), qc = quad_bcast_u64<Q>(c);
    const uint64_t v = w == 0 ? qb : w == 1 ? qq : qc;
    if (qp != 0 && w < 3 && v != 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(qp) + VOFF + w, (unsigned long long)v);
}
// slot = base pointer of a 64-byte slot whose three sums start at word VOFF (2: Slot, 4: WSlot); 0 = nothing to do
template <int VOFF>
__device__ __forceinline__ void quad_atomic_update_at(uint64_t ptr, uint64_t b, uint64_t p, uint64_t c) {
    const uint32_t w = threadIdx.x & 3;
    quad_round<0, VOFF>(ptr, b, p, c, w);
    quad_round<1, VOFF>(ptr, b, p, c, w);
    quad_round<2, VOFF>(ptr, b, p, c, w);
    quad_round<3, VOFF>(ptr, b, p, c, w);
}
__device__ __forceinline__ void quad_atomic_update(Slot* sp, uint64_t b, uint64_t p, uint64_t c) {
    quad_atomic_update_at<2>((uint64_t)sp, b, p, c);
}

// Dense port histograms, quad-grouped: in round Q the quad works on its lane Q's record; lanes 0,1 add
// {weight, 1} to the SrcPort entry and lanes 2,3 to the DstPort entry (two 16-byte entries = two atomic
// line transactions per record instead of four).  0 = no entry for that direction.  Full wave.
template <int Q>
__device__ __forceinline__ void port_round(uint64_t ps, uint64_t pd, uint64_t wgt, uint32_t w) {
    const uint64_t qs = quad_bcast_u64<Q>(ps), qd = quad_bcast_u64<Q>(pd), qw = quad_bcast_u64<Q>(wgt);
    const uint64_t base = w < 2 ? qs : qd;
    const uint64_t v = (w & 1) ? 1ull : qw;
    if (base != 0 && v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(base) + (w & 1), (unsigned long long)v);
}
__device__ __forceinline__ void port_hist_update(uint64_t ps, uint64_t pd, uint64_t wgt) {
    const uint32_t w = threadIdx.x & 3;
    port_round<0>(ps, pd, wgt, w);
    port_round<1>(ps, pd, wgt, w);
    port_round<2>(ps, pd, wgt, w);
    port_round<3>(ps, pd, wgt, w);
}

// ---- wide key sets ----------------------------------------------------------------------------------
__device__ __forceinline__ void app_key(const KArgs& a, const Rec& r, uint32_t tb, WKey& k) {
    wkey_pack(WK_APP, tb, (uint64_t)r.src[1] << 32 | r.src[0], (uint64_t)r.src[3] << 32 | r.src[2], r.dst_port, r.proto, k);
}
__device__ __forceinline__ uint32_t minute_of(const Rec& r) {
    return (uint32_t)r.time_flow_start / 60u;  // UInt64 -> DateTime (create.sh:40), toStartOfMinute (viz-ch.json:74)
}

// Per-lane form (deferred records, no wave cooperation): every wide key set through plain atomics.
template <uint32_t KEYSETS>
__device__ __forceinline__ void wide_sink_slow(const KArgs& a, const Rec& r, uint32_t tb) {
    const WArgs t = wargs(a);
    const uint64_t wgt = r.bytes * r.sampling_rate;  // viz-ch.json:74,358,604 sum(Bytes*SamplingRate), UInt64 wrap
    if (ks_on<KEYSETS>(a, FA_KEYS_ADDR_PORT_PROTO)) {
        WKey k;
        app_key(a, r, tb, k);
        wagg_global(t, k, r.bytes, r.packets, 1);
    }
    if (ks_on<KEYSETS>(a, FA_KEYS_PORT_HIST)) {
        for (int d = 0; d < 2; d++) {
            const uint32_t port = d ? r.dst_port : r.src_port;
            if (port < PORT_DENSE) {
                unsigned long long* e = reinterpret_cast<unsigned long long*>(&a.port_hist[(size_t)d * PORT_DENSE + port]);
                if (wgt) atomicAdd(e, (unsigned long long)wgt);
                atomicAdd(e + 1, 1ull);
            } else {
                WKey k;
                wkey_pack(d ? WK_DSTPORT : WK_SRCPORT, 0, 0, 0, port, 0, k);
                wagg_global(t, k, wgt, 0, 1);
            }
        }
    }
    if (ks_on<KEYSETS>(a, FA_KEYS_MINUTE_SERIES)) {
        WKey k;
        wkey_pack(WK_MINUTE, 0, 0, 0, minute_of(r), 0, k);
        wagg_global(t, k, wgt, 0, 1);
    }
}

// Full-wave form (tile kernel): one atomic line transaction per record and key set.
template <uint32_t KEYSETS>
__device__ __forceinline__ void wide_sink_wave(const KArgs& a, LdsMinutes& lm, const Rec& r, bool sure, uint32_t tb) {
    const WArgs t = wargs(a);
    const uint64_t wgt = r.bytes * r.sampling_rate;
    if (ks_on<KEYSETS>(a, FA_KEYS_ADDR_PORT_PROTO)) {
        WSlot* sp = nullptr;
        if (sure) {
            WKey k;
            app_key(a, r, tb, k);
            sp = wtable_find_or_claim(t, k, wkey_hash(k));
            if (!sp) wspill_park(t, k, r.bytes, r.packets, 1);
        }
        quad_atomic_update_at<4>((uint64_t)sp, r.bytes, r.packets, 1);
    }
    if (ks_on<KEYSETS>(a, FA_KEYS_PORT_HIST)) {
        uint64_t ps = 0, pd = 0;
        if (sure) {
            if (r.src_port < PORT_DENSE) {
                ps = (uint64_t)&a.port_hist[r.src_port];
            } else {
                WKey k;
                wkey_pack(WK_SRCPORT, 0, 0, 0, r.src_port, 0, k);
                wagg_global(t, k, wgt, 0, 1);
            }
            if (r.dst_port < PORT_DENSE) {
                pd = (uint64_t)&a.port_hist[(size_t)PORT_DENSE + r.dst_port];
            } else {
                WKey k;
                wkey_pack(WK_DSTPORT, 0, 0, 0, r.dst_port, 0, k);
                wagg_global(t, k, wgt, 0, 1);
            }
        }
        port_hist_update(ps, pd, wgt);
    }
    if (ks_on<KEYSETS>(a, FA_KEYS_MINUTE_SERIES)) {
        // lanes of a wave almost always share one or two minutes: fold them, then one LDS update per group
        const uint32_t minute = minute_of(r);
        uint64_t w0 = wgt, z = 0, c = 1;
        bool valid = sure;
        wave_combine<4, 2>(valid, (uint64_t)minute, 1ull, w0, z, c);
        if (valid && !lds_minutes_add(lm, minute, w0, c)) {
            WKey k;
            wkey_pack(WK_MINUTE, 0, 0, 0, minute, 0, k);
            wagg_global(t, k, w0, 0, c);
        }
    }
}

__device__ __forceinline__ void spill_park(const KArgs& a, uint64_t k0, uint64_t k1, uint64_t b, uint64_t p, uint64_t c) {
    unsigned int j = atomicAdd(&a.ctr->spill_count, 1u);
    if (j < a.spill_cap)
        a.spill[j] = SpillEntry{k0, k1, b, p, c};
    else
        atomicAdd(&a.ctr->spill_lost, 1ull);
}

// t / gran for a runtime granule without an integer division (see KArgs::gran_recip)
__device__ __forceinline__ uint32_t time_bucket(const KArgs& a, uint32_t t32) {
    return (uint32_t)((double)t32 * a.gran_recip);
}

// varint(len) frame prefix of 1 or 2 bytes (records < 16 KiB) straight from the first window
__device__ __forceinline__ bool frame_short(uint32_t x, uint32_t rec_len, uint32_t& prefix_len) {
    const uint32_t b0 = x & 0xffu, b1 = (x >> 8) & 0xffu;
    const bool one = b0 < 0x80u;
    const uint32_t val = one ? b0 : ((b0 & 0x7fu) | (b1 << 7));
    prefix_len = one ? 1u : 2u;
    return (one || b1 < 0x80u) && rec_len >= prefix_len && val == rec_len - prefix_len;
}

// Full bins leave as whole, aligned 128-byte lines, up to 8 bins per pass: lane group g (8 lanes) takes the
// g-th filled bin, each lane copies one tuple - one store instruction writes 8 complete lines, no partial
// lines and no workgroup barrier.  The producers of a bin's other slots may sit in other waves: the high
// half of the bin word counts the slots WRITTEN, and nobody can take a slot of a full bin, so the spin
// below only ever waits for straight-line code of waves that never wait for us (producers of this wave
// finished in lockstep inside lane_work).  fill_part: the bin this lane filled (or ~0); scratch: 32 bytes of
// wave-private LDS.  Must be called by the full wave.
__device__ __forceinline__ void bins_flush(const KArgs& a, uint4* bins, uint32_t* bin_cnt, uint32_t* part_cnt, uint32_t* scratch,
                                           uint32_t fill_part, uint32_t tb_base, uint32_t& n_direct) {
    const unsigned long long fm = __builtin_amdgcn_ballot_w64(fill_part != 0xffffffffu);
    if (fm == 0ull) return;
    const uint32_t ln = __lane_id(), g = ln >> 3, sub = ln & 7u;
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << ln) - 1ull));
    const uint32_t todo = (uint32_t)__builtin_popcountll(fm);
    for (uint32_t base = 0; base < todo; base += 8u) {
        if (fill_part != 0xffffffffu && rank - base < 8u) scratch[rank - base] = fill_part;
        const bool act = g < min(8u, todo - base);
        const uint32_t fp = act ? scratch[g] : 0u;
        // the written-slot check, the tuple read and the line allocation are issued back to back (LDS operations
        // of a wave complete in order, so the read sees what the check saw); only a bin that is still being
        // written costs further round trips
        // (acquire / release pair with the producers' written-slot count: without it the COMPILER may move the
        // tuple read above the check - it did, and rows differed from the oracle at 16 M records)
        const uint32_t c0 = __hip_atomic_load(&bin_cnt[fp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint4 tq = bins[fp * BIN_CAP + sub];
        uint32_t line = 0;
        if (act && sub == 0) line = atomicAdd(&part_cnt[fp], 1u) & 0xffffu;  // low half: lines at the front
        bool late = false;
        if (__builtin_amdgcn_ballot_w64(act && (c0 >> 16) < BIN_CAP) != 0ull) {
            late = true;
            while (__builtin_amdgcn_ballot_w64(act && (__hip_atomic_load(&bin_cnt[fp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> 16) < BIN_CAP) != 0ull) {}
        }
        line = (uint32_t)__shfl((int)line, (int)(ln & ~7u));
        if (act) {
            const uint4 tv = late ? bins[fp * BIN_CAP + sub] : tq;
            if ((line + 1u) * BIN_CAP <= a.capf) {
                if (!(a.dbg & DBG_NO_TUPLE_STORE))
                    a.seg[(size_t)fp * a.region + (size_t)blockIdx.x * a.capq + line * BIN_CAP + sub] = tv;
            } else {  // front part full (skewed batch): straight to the device-wide table
                const uint32_t qby = tv.z & 0x0fffffffu, qtbr = tv.z >> 28, qpk = tv.w & 0x7fffu, qet = tv.w >> 15;
                uint64_t q0, q1;
                pack_key(tb_base + qtbr, tv.x, tv.y, qet, q0, q1);
                agg_global(a, q0, q1, key_hash(q0, q1), qby, qpk, 1);
                n_direct++;
            }
            // (release: behind the tuple reads above)
            if (sub == 0) __hip_atomic_store(&bin_cnt[fp], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// ---- per-lane work on a staged record (called by every lane of the workgroup) ------
template <int MODE, uint32_t KEYSETS, uint32_t COLS>
__device__ __forceinline__ void lane_work(const KArgs& a, LdsTable<LDS_SLOTS>& lt, LdsMinutes& lm, uint32_t* part_cnt, const uint32_t* tile,
                                          bool mine, uint32_t pos, uint32_t end, uint32_t rec_idx, uint32_t tb_base,
                                          uint32_t& n_ok, uint32_t& n_direct, uint32_t& lt_seen, uint32_t& lt_hits,
                                          uint4* bins, uint32_t* bin_cnt, uint32_t& fill_out) {
    // ---- parse (divergent: only lanes that own a staged record) ----
    bool sure = false;
    Rec r;
    rec_clear(r);
    if (mine) {
        LdsSrc src{tile};
        sure = true;
        if (a.framed && !(a.dbg & DBG_NO_FRAME)) {
            uint32_t pl = 0;
            const uint32_t i = pos >> 2;
            sure = frame_short(fa_alignbyte(src.dw(i + 1), src.dw(i), pos), end - pos, pl);
            pos += pl;
        }
        if (sure && !(a.dbg & DBG_NO_PARSE)) {
            if (a.dbg & DBG_LOOP_PARSER) sure = parse_fast<COLS>(src, pos, end, r);
            else sure = parse_canon<COLS>(src, pos, end, r);
        }
        if (!sure) {
            unsigned int j = atomicAdd(&a.ctr->retry_count, 1u);
            a.retry_idx[j] = rec_idx;
        }
    }
    // ---- sink ----
    if (MODE == MODE_DECODE) {
        if (sure) store_columns(a.cols, rec_idx, r, 0);
        return;
    }
    n_ok += sure ? 1 : 0;
    if (a.dbg & DBG_NO_SINK) {
        n_ok += (uint32_t)(r.time_received ^ r.bytes ^ r.packets ^ r.src_as ^ r.dst_as ^ r.etype) & 1;
        return;
    }
    const uint32_t t32 = (uint32_t)r.time_received;  // UInt64 -> DateTime (create.sh:39)
    const uint32_t tb = time_bucket(a, t32);
    if (ks_on<KEYSETS>(a, FA_KEYS_AS_PAIR)) {
        uint64_t k0, k1;
        pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
        const uint32_t h = key_hash(k0, k1);
        const uint64_t b = r.bytes, p = r.packets, c = 1;
        bool pending = sure;
        // hot-key table: worth its LDS atomics only while it absorbs records.  Every wave keeps score (ballots:
        // wave-uniform, no LDS traffic) and stops offering records once fewer than 1 in 8 of its first 256 stuck
        // (64 k uniform AS pairs never do; the mocker's 9 groups always do).  lt_seen == ~0u: switched off.
        if (lt_seen != 0xffffffffu && !(a.dbg & DBG_NO_LDS_TABLE)) {
            if (pending) pending = !lds_table_add<LDS_SLOTS, LDS_PROBES>(lt, k0, k1, h, b, p, c);
            lt_seen += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sure));
            lt_hits += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sure && !pending));
            if (lt_seen >= 256u) {
                if (lt_hits * 8u < lt_seen) lt_seen = 0xffffffffu;
                else lt_seen = lt_hits = 0;
            }
        }
        // tuple path: 16 bytes to this workgroup's private segment of the key's partition
        uint32_t fill_part = 0xffffffffu;  // wave-tile kernel: the bin this lane has just filled
        if (pending && a.seg) {
            const uint32_t tbr = tb - tb_base;
            const bool fits = tbr < TUPLE_TB_SPAN && b < TUPLE_MAX_BYTES && p < TUPLE_MAX_PACKETS && r.etype < TUPLE_MAX_ETYPE;
            if (fits) {
                const uint32_t part = h >> (32 - a.plog2);
                const uint4 tv = make_uint4(r.src_as, r.dst_as, (uint32_t)b | (tbr << 28), (uint32_t)p | (r.etype << 15));
                if (bins) {
                    // wave-tile kernel: the tuple waits in the workgroup's LDS bin of its partition; the lane that takes
                    // the last slot of a bin sends the 8 tuples off as one full, aligned 128-byte line (below)
                    // (acquire: the tuple write below must not move above the claim - the previous occupants of the bin
                    // are read by the flusher until it resets the word; release: the tuple is written before it counts)
                    const uint32_t slot = __hip_atomic_fetch_add(&bin_cnt[part], 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffffu;  // low half: slots taken, high half: slots written
                    if (slot < BIN_CAP) {
                        bins[part * BIN_CAP + slot] = tv;
                        __hip_atomic_fetch_add(&bin_cnt[part], 0x10000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        fill_part = slot == BIN_CAP - 1 ? part : fill_part;
                        pending = false;
                    } else {  // the bin is on its way out: single 16-byte store to the back part of the segment
                        const uint32_t ob = atomicAdd(&part_cnt[part], 0x10000u) >> 16;
                        if (ob < a.capb) {
                            if (!(a.dbg & DBG_NO_TUPLE_STORE))
                                a.seg[(size_t)part * a.region + (size_t)blockIdx.x * a.capq + (a.capq - 1u - ob)] = tv;
                            pending = false;
                        }
                    }
                } else {
                    const uint32_t q = atomicAdd(&part_cnt[part], 1u);
                    if (q < a.capq) {
                        if (!(a.dbg & DBG_NO_TUPLE_STORE)) {
                            uint4* dstp = &a.seg[(size_t)part * a.region + (size_t)blockIdx.x * a.capq + q];
                            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                            const v4u tvv = {tv.x, tv.y, tv.z, tv.w};
                            if (a.dbg & DBG_TUPLE_NT) __builtin_nontemporal_store(tvv, reinterpret_cast<v4u*>(dstp));
                            else if (a.dbg & DBG_TUPLE_SC) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dstp), "v"(tvv) : "memory");
                            else *dstp = tv;
                        }
                        pending = false;
                    }
                }
            }
        }
        fill_out = fill_part;  // full bins leave in bins_flush(), which the wave-tile kernel runs in the shadow of its next DMA
        // direct path (what is left): device-wide table, one atomic line transaction per record
        if (__builtin_amdgcn_ballot_w64(pending) != 0ull && !(a.dbg & DBG_NO_GLOBAL)) {  // wave-uniform
            Slot* sp = nullptr;
            if (pending) {
                n_direct++;
                sp = table_find_or_claim(a, k0, k1, h);
                if (!sp) spill_park(a, k0, k1, b, p, c);
            }
            quad_atomic_update(sp, b, p, c);
        }
    }
    if (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) {
        // lanes of a wave that carry the same address (heavy hitters) are folded first: one sketch update and
        // one distinct-set probe per address and wave (wave-tile kernel: its parsed tile buffer is the scratch)
        const uint64_t w = r.bytes * r.sampling_rate;  // viz-ch.json:233 sum(Bytes*SamplingRate), UInt64 wrap
        if (ks_on<KEYSETS>(a, FA_KEYS_SRCADDR_CMS)) {
            uint64_t ws = w;
            bool valid = sure;
            if (bins) wave_fold_lds(const_cast<uint32_t*>(tile), valid, (uint64_t)r.src[1] << 32 | r.src[0], (uint64_t)r.src[3] << 32 | r.src[2], ws);
            if (valid) {
                cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, ws);
                keyset_insert(a, a.ks_src, r.src);
            }
        }
        if (ks_on<KEYSETS>(a, FA_KEYS_DSTADDR_CMS)) {
            uint64_t ws = w;
            bool valid = sure;
            if (bins) wave_fold_lds(const_cast<uint32_t*>(tile), valid, (uint64_t)r.dst[1] << 32 | r.dst[0], (uint64_t)r.dst[3] << 32 | r.dst[2], ws);
            if (valid) {
                cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, ws);
                keyset_insert(a, a.ks_dst, r.dst);
            }
        }
    }
    if (KEYSETS & FA_KEYS_WIDE) wide_sink_wave<KEYSETS>(a, lm, r, sure, tb);
}

// End-of-kernel counters: one global atomic per WORKGROUP.  All waves of the grid finish at about the same
// time, and same-address atomics serialize at the memory side: one atomic per wave (8192 of them) was a
// ~30 us tail on a 0.4 ms launch.
__device__ __forceinline__ void block_counters_add(uint32_t* lds2, Counters* ctr, uint32_t n_ok, uint32_t n_direct) {
    if (threadIdx.x < 2) lds2[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t ok = (uint32_t)wave_sum_u64(n_ok), direct = (uint32_t)wave_sum_u64(n_direct);
    if (__lane_id() == 0) {
        if (ok) atomicAdd(&lds2[0], ok);
        if (direct) atomicAdd(&lds2[1], direct);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lds2[0]) atomicAdd(&ctr->ok, (unsigned long long)lds2[0]);
        if (lds2[1]) atomicAdd(&ctr->direct, (unsigned long long)lds2[1]);
    }
}

// ---- the tile kernel ----------------------------------------------------------
// Persistent workgroups; tile = 256 consecutive records (one per lane).  The wire
// bytes of tile t+1 stream into the second LDS buffer (async DMA) while tile t is
// parsed and aggregated, so the HBM latency hides behind the integer work.
// The descriptor is two loads (lo, hi) whose values must NOT be looked at before the tile's turn comes: any
// arithmetic on them right after the loads makes the compiler wait for them - and, vmcnt being in-order, for
// the DMA issued just before - and only then issue the per-lane offset loads: two serialized memory round
// trips per tile (this cost 25 % of the staging bandwidth, tools/read_bench2.hip).  fits() is evaluated
// when the tile is current.
struct TileDesc {
    uint32_t r0, nrec, lo, hi;  // records [r0,r0+nrec), wire bytes [lo,hi)
};
__device__ __forceinline__ TileDesc tile_desc(const KArgs& a, uint32_t t, uint32_t ntiles) {
    TileDesc d{0, 0, 0, 0};
    if (t < ntiles) {
        d.r0 = t * a.tile_recs;
        d.nrec = min(a.tile_recs, a.n - d.r0);
        // the indices are laundered through VGPRs: for provably uniform addresses the compiler moves the loaded
        // values to SGPRs at once (v_readfirstlane right behind the loads = the same premature wait)
        uint32_t i0 = d.r0, i1 = d.r0 + d.nrec;
        asm volatile("" : "+v"(i0), "+v"(i1));
        d.lo = a.off[i0];
        d.hi = a.off[i1];
    }
    return d;
}
// the bounds of the tile whose turn has come, as wave-uniform scalars
__device__ __forceinline__ TileDesc tile_current(const TileDesc& d) {
    return TileDesc{d.r0, d.nrec, (uint32_t)__builtin_amdgcn_readfirstlane((int)d.lo), (uint32_t)__builtin_amdgcn_readfirstlane((int)d.hi)};
}
template <int BYTES>
__device__ __forceinline__ bool tile_fits(const TileDesc& d) {  // whole tile fits one LDS buffer (the normal case)
    return d.nrec != 0 && d.hi >= d.lo && (d.hi - (d.lo & ~15u)) <= (uint32_t)BYTES;
}

// Persistent workgroups; tile = up to 256 consecutive records (one per lane) staged
// in ONE LDS buffer.  LDS bounds the number of records a CU can hold, and the parse
// is a long dependent chain per record, so the LDS goes to as many co-resident
// workgroups as possible (6 per CU = 6 waves/SIMD): while one workgroup waits for
// its DMA, the others parse.  (A double-buffered variant with 3 workgroups/CU
// staged at 4.0 TB/s but left the parse latency-bound at 3 waves/SIMD.)
template <int MODE, uint32_t KEYSETS>
__global__ __launch_bounds__(BLOCK) void tile_kernel(KArgs a) {
    constexpr uint32_t COLS = MODE == MODE_DECODE ? (uint32_t)COL_ALL : cols_for_keysets<KEYSETS>();
    __shared__ __attribute__((aligned(16))) uint32_t tile[TILE_STRIDE / 4];
    __shared__ LdsTable<LDS_SLOTS> lt;
    __shared__ uint32_t part_cnt[NPART_MAX];  // tuples this workgroup appended per key partition
    __shared__ LdsMinutes lm;                 // per-minute series pre-aggregation (KS_ALL variant only)

    const uint32_t tid = threadIdx.x;
    if (MODE == MODE_INGEST && (KEYSETS & FA_KEYS_AS_PAIR)) {
        lds_table_clear(lt);
        for (int i = tid; i < NPART_MAX; i += BLOCK) part_cnt[i] = 0;
    }
    if (MODE == MODE_INGEST && (KEYSETS & FA_KEYS_MINUTE_SERIES)) lds_minutes_clear(lm);
    const uint32_t tb_base = MODE == MODE_INGEST ? a.ctr->tb_base : 0u;

    uint32_t n_ok = 0, n_direct = 0, lt_seen = 0, lt_hits = 0, no_fill = 0;
    const uint32_t ntiles = (a.n + a.tile_recs - 1) / a.tile_recs;
    const uint32_t stride = gridDim.x;
    uint32_t t = blockIdx.x;
    TileDesc cur = tile_current(tile_desc(a, t, ntiles));
    uint32_t o0 = 0, o1 = 0;  // this lane's record of the current tile
    if (t < ntiles && tid < cur.nrec) {
        o0 = a.off[cur.r0 + tid];
        o1 = a.off[cur.r0 + tid + 1];
    }
    __syncthreads();  // LDS table cleared

    const bool timing = (a.dbg & DBG_TIMING) != 0 && tid == 0;
    uint32_t tm_wait = 0, tm_work = 0, tm_tiles = 0;
    const uint32_t tm_start = timing ? (uint32_t)clock64() : 0u;
    for (; t < ntiles; t += stride) {
        const bool cur_fits = tile_fits<TILE_BYTES>(cur);
        const uint32_t tm0 = timing ? (uint32_t)clock64() : 0u;
        // (1) stream this tile's wire bytes into LDS (async DMA) ...
        if (cur_fits) {
            // nt: the wire bytes are read exactly once; keeping them out of the way of the L2's open tuple lines
            // is worth 11 % of the launch (MI355X, tools/knobs.sh FA_DEBUG_FLAGS=512)
            if (a.dbg & DBG_DMA_NO_NT) dma_to_lds<0>(a.buf + (cur.lo & ~15u), cur.hi - (cur.lo & ~15u), tile);
            else dma_to_lds<2>(a.buf + (cur.lo & ~15u), cur.hi - (cur.lo & ~15u), tile);
        }
        // ... and meanwhile fetch the next tile's descriptor and offsets
        const TileDesc nxt = tile_desc(a, t + stride, ntiles);
        uint32_t n0 = 0, n1 = 0;
        if (tid < nxt.nrec) {
            n0 = a.off[nxt.r0 + tid];
            n1 = a.off[nxt.r0 + tid + 1];
        }
        dma_wait_all();
        __syncthreads();
        const uint32_t tm1 = timing ? (uint32_t)clock64() : 0u;

        // (2) parse + aggregate out of LDS
        if (cur_fits) {
            const uint32_t cbase = cur.lo & ~15u;
            const bool mine = tid < cur.nrec && o1 >= o0 && o0 >= cur.lo && o1 <= cur.hi;
            if (tid < cur.nrec && !mine) {  // broken offsets: let the generic path judge it
                unsigned int j = atomicAdd(&a.ctr->exotic_count, 1u);
                a.exotic_idx[j] = cur.r0 + tid;
            }
            lane_work<MODE, KEYSETS, COLS>(a, lt, lm, part_cnt, tile, mine, o0 - cbase, o1 - cbase, cur.r0 + tid, tb_base, n_ok, n_direct, lt_seen, lt_hits, nullptr, nullptr, no_fill);
        } else {
            // rare: the tile's bytes exceed the LDS buffer (big records): stage it in passes
            uint32_t done = 0;
            while (done < cur.nrec) {
                const uint32_t first = a.off[cur.r0 + done];
                const uint32_t cbase = first & ~15u;
                const uint32_t climit = cbase + TILE_BYTES;
                const uint32_t stage_end = min(cur.hi, climit);
                if (stage_end > cbase) dma_to_lds(a.buf + cbase, stage_end - cbase, tile);
                const uint32_t k = done + tid;
                uint32_t p0 = 0, p1 = 0;
                bool mine = false;
                if (k < cur.nrec) {
                    p0 = a.off[cur.r0 + k];
                    p1 = a.off[cur.r0 + k + 1];
                    mine = p1 <= climit && p1 >= p0 && p0 >= cbase && p1 <= cur.hi;
                }
                dma_wait_all();
                const int nfit = __syncthreads_count(mine);  // offsets are monotone: a prefix fits
                if (nfit == 0) {
                    // one record larger than the LDS buffer (or broken offsets): generic path
                    if (tid == 0) {
                        unsigned int j = atomicAdd(&a.ctr->exotic_count, 1u);
                        a.exotic_idx[j] = cur.r0 + done;
                    }
                    done += 1;
                } else {
                    lane_work<MODE, KEYSETS, COLS>(a, lt, lm, part_cnt, tile, mine, p0 - cbase, p1 - cbase, cur.r0 + k, tb_base, n_ok, n_direct, lt_seen, lt_hits, nullptr, nullptr, no_fill);
                    done += nfit;
                }
                __syncthreads();  // the buffer is restaged by the next pass
            }
        }
        __syncthreads();  // everyone is done reading the tile
        if (timing) {
            const uint32_t tm2 = (uint32_t)clock64();
            tm_wait += tm1 - tm0;
            tm_work += tm2 - tm1;
            tm_tiles++;
        }
        cur = tile_current(nxt);
        o0 = n0;
        o1 = n1;
    }
    if (timing) {
        atomicAdd(&a.ctr->t_wait, (unsigned long long)tm_wait);
        atomicAdd(&a.ctr->t_work, (unsigned long long)tm_work);
        atomicAdd(&a.ctr->t_tiles, (unsigned long long)tm_tiles);
        atomicAdd(&a.ctr->t_total, (unsigned long long)((uint32_t)clock64() - tm_start));
    }
    if (MODE == MODE_INGEST) {
        if (KEYSETS & FA_KEYS_MINUTE_SERIES) {
            __syncthreads();
            if (tid < LDS_MINUTES && lm.key[tid] != 0 && lm.c[tid] != 0) {
                WKey k;
                wkey_pack(WK_MINUTE, 0, 0, 0, lm.key[tid] - 1u, 0, k);
                wagg_global(wargs(a), k, lm.w[tid], 0, lm.c[tid]);
            }
        }
        if (KEYSETS & FA_KEYS_AS_PAIR) {
            __syncthreads();
            // hot-key table -> device-wide table, one atomic line transaction per group (uniform trip count:
            // the quad rounds need the whole wave)
            for (int i0 = 0; i0 < LDS_SLOTS; i0 += BLOCK) {
                const int i = i0 + tid;
                Slot* sp = nullptr;
                unsigned long long b = 0, p = 0, c = 0;
                if (i < LDS_SLOTS) {
                    const unsigned long long k0 = lt.k0[i], k1 = lt.k1[i];
                    b = lt.bytes[i];
                    p = lt.packets[i];
                    c = lt.count[i];
                    if (k0 != 0 && k1 != 0 && c != 0) {
                        sp = table_find_or_claim(a, k0, k1, key_hash(k0, k1));
                        if (!sp) spill_park(a, k0, k1, b, p, c);
                    }
                }
                quad_atomic_update(sp, b, p, c);
            }
        }
        if ((KEYSETS & FA_KEYS_AS_PAIR) && a.seg) {
            for (int i = tid; i < (1 << a.plog2); i += BLOCK)
{
                a.seg_counts[(size_t)i * a.nwg + blockIdx.x] = min(part_cnt[i], a.capq);
                a.seg_counts[((size_t)NPART_MAX + i) * a.nwg + blockIdx.x] = 0;
            }
        }
        block_counters_add(part_cnt, a.ctr, n_ok, n_direct);  // (part_cnt has been written out: reused as scratch)
    }
}

// ---- the wave-tile kernel ---------------------------------------------------------------------------
// The production ingest kernel of the scatter sink.  Same per-record work as tile_kernel, different
// residency: 2 workgroups of 8 waves per CU; every WAVE stages its own tile of <= 64 records into a private
// LDS buffer (its next DMA is issued the moment the tile is consumed) and parses it - there is no workgroup
// barrier anywhere in the steady state.  The LDS this frees (the 256-thread kernel spends all of it on
// co-resident tiles) holds the tuple bins: a tuple waits in the 8-slot bin of its key partition, and a full bin
// leaves as ONE aligned 128-byte line (lane_work), instead of as eight 16-byte stores whose cache line is
// evicted from the L2 long before its neighbours arrive (DESIGN.md "Measurements").  A segment therefore has a
// front part of whole lines and a back part for the odd tuples (bin leftovers at the end of the launch, tuples
// that met a bin on its way out).  The workgroup's segments are 3x longer than tile_kernel's, which also
// suits agg_kernel's 64-lane loads.
typedef TileDesc WTileDesc;  // (same rule: the loaded bounds are not looked at before the tile's turn)
__device__ __forceinline__ WTileDesc wtile_desc(const KArgs& a, uint32_t t, uint32_t ntiles) {
    WTileDesc d{0, 0, 0, 0};
    if (t < ntiles) {
        d.r0 = t * a.tile_recs;
        d.nrec = min(a.tile_recs, a.n - d.r0);
        if (a.dbg & DBG_SYNTH_TILES) {  // measurement only: fixed-size aligned tiles, no descriptor loads
            d.lo = t * 4608u;
            d.hi = d.lo + 4608u;
            return d;
        }
        uint32_t i0 = d.r0, i1 = d.r0 + d.nrec;
        asm volatile("" : "+v"(i0), "+v"(i1));  // (see tile_desc)
        d.lo = a.off[i0];
        d.hi = a.off[i1];
    }
    return d;
}

template <uint32_t KEYSETS>
__global__ __launch_bounds__(WBLOCK) void wtile_kernel(KArgs a) {
    constexpr uint32_t COLS = cols_for_keysets<KEYSETS>();
    constexpr int WAVES = WBLOCK / 64;
    __shared__ __attribute__((aligned(16))) uint32_t tiles[WAVES * WT_STRIDE / 4];
    __shared__ __attribute__((aligned(16))) uint4 bins[NPART_MAX * BIN_CAP];
    __shared__ uint32_t bin_cnt[NPART_MAX];
    __shared__ uint32_t part_cnt[NPART_MAX];
    __shared__ uint32_t flush_scratch[WAVES * 8];
    __shared__ LdsTable<LDS_SLOTS> lt;
    __shared__ LdsMinutes lm;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (KEYSETS & FA_KEYS_AS_PAIR) {
        lds_table_clear(lt);
        for (int i = tid; i < NPART_MAX; i += WBLOCK) {
            part_cnt[i] = 0;
            bin_cnt[i] = 0;
        }
    }
    if (KEYSETS & FA_KEYS_MINUTE_SERIES) lds_minutes_clear(lm);
    const uint32_t tb_base = a.ctr->tb_base;
    uint32_t* tile = tiles + wave * (WT_STRIDE / 4);

    uint32_t n_ok = 0, n_direct = 0, lt_seen = 0, lt_hits = 0;
    const uint32_t ntiles = (a.n + a.tile_recs - 1) / a.tile_recs;
    const uint32_t stride = gridDim.x * WAVES;
    const uint32_t rounds = (ntiles + stride - 1) / stride;  // the same for every wave of the grid (flush barriers)
    uint32_t t = blockIdx.x * WAVES + wave;
    WTileDesc cur = tile_current(wtile_desc(a, t, ntiles));
    uint32_t o0 = 0, o1 = 0;
    const bool lane_off = !(a.dbg & DBG_NO_LANE_OFF);
    // one offset load per lane: a record's end is its neighbour's start (lane nrec-1: the tile's end, already known)
    if (lane_off && lane < cur.nrec) o0 = a.off[cur.r0 + lane];
    o1 = (uint32_t)__shfl_down((int)o0, 1);
    if (lane + 1 >= cur.nrec) o1 = cur.hi;
    __syncthreads();  // LDS state cleared

    auto issue_dma = [&](const WTileDesc& d) {
        if (tile_fits<WT_STRIDE - 16>(d)) {
            const uint32_t cbase = d.lo & ~15u, nbytes = d.hi - cbase;
            for (uint32_t o = lane * 16u; o < nbytes; o += 1024u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.buf + cbase + o),
                                                 (__attribute__((address_space(3))) void*)(tile + (o - lane * 16u) / 4u), 16, 0, 2);
        }
    };
    // pipeline: at the top of a round the wave's tile is already on its way (issued right after the previous
    // tile was consumed, BEFORE the flush barriers, so that the memory system stays busy during a flush) and
    // the descriptor + offsets of the tile after it are in flight
    issue_dma(cur);
    WTileDesc nxt = wtile_desc(a, t + stride, ntiles);
    uint32_t n0 = 0;
    if (lane_off && lane < nxt.nrec) n0 = a.off[nxt.r0 + lane];
    for (uint32_t round = 0; round < rounds; round++, t += stride) {
        dma_wait_all();  // the wave's own DMA has landed: no workgroup barrier on this path
        uint32_t fill = 0xffffffffu;  // the bin this lane fills in this round
        // parse + sink
        if (cur.nrec != 0) {
            const uint32_t cbase = cur.lo & ~15u;
            bool mine = tile_fits<WT_STRIDE - 16>(cur) && lane < cur.nrec && o1 >= o0 && o0 >= cur.lo && o1 <= cur.hi;
            if (a.dbg & DBG_NOT_MINE) mine = mine && o0 == 0x7fffffffu;
            if (lane < cur.nrec && !mine && !(a.dbg & (DBG_NO_LANE_OFF | DBG_SYNTH_TILES | DBG_NOT_MINE))) {  // tile larger than the buffer / broken offsets
                unsigned int j = atomicAdd(&a.ctr->exotic_count, 1u);
                a.exotic_idx[j] = cur.r0 + lane;
            }
            lane_work<MODE_INGEST, KEYSETS, COLS>(a, lt, lm, part_cnt, tile, mine, o0 - cbase, o1 - cbase, cur.r0 + lane, tb_base, n_ok,
                                                  n_direct, lt_seen, lt_hits, bins, bin_cnt, fill);
        }
        // full bins leave BEFORE the next DMA is issued: behind it their stores would sit in the in-order vmcnt
        // queue and the wave would wait for the write acknowledgements on top of its tile (measured: +15 %)
        if ((KEYSETS & FA_KEYS_AS_PAIR) && a.seg) bins_flush(a, bins, bin_cnt, part_cnt, flush_scratch + wave * 8, fill, tb_base, n_direct);
        cur = tile_current(nxt);
        o0 = n0;
        o1 = (uint32_t)__shfl_down((int)o0, 1);
        if (lane + 1 >= cur.nrec) o1 = cur.hi;
        issue_dma(cur);  // next tile (the buffer is free: every read of the old tile has returned)
        nxt = wtile_desc(a, t + 2 * stride, ntiles);
        n0 = 0;
        if (lane_off && lane < nxt.nrec) n0 = a.off[nxt.r0 + lane];
    }
    // what is left in the bins (fewer than BIN_CAP tuples each) goes to the back part of the segments
    if ((KEYSETS & FA_KEYS_AS_PAIR) && a.seg) {
        __syncthreads();
        for (uint32_t idx = tid; idx < (uint32_t)NPART_MAX * BIN_CAP; idx += WBLOCK) {
            const uint32_t p = idx / BIN_CAP, sl = idx % BIN_CAP;
            const uint32_t cnt = min(bin_cnt[p] & 0xffffu, BIN_CAP);
            if (sl < cnt) {
                const uint32_t ob = (part_cnt[p] >> 16) + sl;
                const uint4 tv = bins[idx];
                if (ob < a.capb) {
                    if (!(a.dbg & DBG_NO_TUPLE_STORE)) a.seg[(size_t)p * a.region + (size_t)blockIdx.x * a.capq + (a.capq - 1u - ob)] = tv;
                } else {  // back part full (skewed batch): straight to the device-wide table
                    const uint32_t by = tv.z & 0x0fffffffu, tbr = tv.z >> 28, pk = tv.w & 0x7fffu, et = tv.w >> 15;
                    uint64_t k0, k1;
                    pack_key(tb_base + tbr, tv.x, tv.y, et, k0, k1);
                    agg_global(a, k0, k1, key_hash(k0, k1), by, pk, 1);
                    n_direct++;
                }
            }
        }
        __syncthreads();
        if (tid < NPART_MAX) {
            part_cnt[tid] += min(bin_cnt[tid] & 0xffffu, BIN_CAP) << 16;
            bin_cnt[tid] = 0;
        }
        __syncthreads();
    }
    if (KEYSETS & FA_KEYS_MINUTE_SERIES) {
        __syncthreads();
        if (tid < LDS_MINUTES && lm.key[tid] != 0 && lm.c[tid] != 0) {
            WKey k;
            wkey_pack(WK_MINUTE, 0, 0, 0, lm.key[tid] - 1u, 0, k);
            wagg_global(wargs(a), k, lm.w[tid], 0, lm.c[tid]);
        }
    }
    if (KEYSETS & FA_KEYS_AS_PAIR) {
        __syncthreads();
        for (int i0 = 0; i0 < LDS_SLOTS; i0 += WBLOCK) {  // hot-key table -> device-wide table (full waves: quad rounds)
            const int i = i0 + tid;
            Slot* sp = nullptr;
            unsigned long long b = 0, p = 0, c = 0;
            if (i < LDS_SLOTS) {
                const unsigned long long k0 = lt.k0[i], k1 = lt.k1[i];
                b = lt.bytes[i];
                p = lt.packets[i];
                c = lt.count[i];
                if (k0 != 0 && k1 != 0 && c != 0) {
                    sp = table_find_or_claim(a, k0, k1, key_hash(k0, k1));
                    if (!sp) spill_park(a, k0, k1, b, p, c);
                }
            }
            quad_atomic_update(sp, b, p, c);
        }
        if (a.seg)
            for (int i = tid; i < (1 << a.plog2); i += WBLOCK) {
                const uint32_t w = part_cnt[i];
                a.seg_counts[(size_t)i * a.nwg + blockIdx.x] = min((w & 0xffffu) * BIN_CAP, a.capf);
                a.seg_counts[((size_t)NPART_MAX + i) * a.nwg + blockIdx.x] = min(w >> 16, a.capb);
            }
    }
    block_counters_add(bin_cnt, a.ctr, n_ok, n_direct);  // (the bins are empty by now: reused as scratch)
}

// ---- probe: where in time does this batch sit? ---------------------------------------------
// 64 evenly spaced records are decoded; tb_base = (smallest time bucket
// seen) - 2, so that the 4-bit relative bucket of the tuple path covers the batch (Kafka partitions
// are close to time-ordered; records outside [tb_base, tb_base+16) take the direct path).
__global__ __launch_bounds__(64) void probe_kernel(KArgs a) {
    __shared__ uint32_t lo;
    if (threadIdx.x == 0) lo = 0xffffffffu;
    __syncthreads();
    const uint32_t idx = a.n <= 64 ? threadIdx.x : (uint32_t)(((uint64_t)threadIdx.x * (a.n - 1)) / 63u);
    if (idx < a.n) {
        uint32_t pos = a.off[idx], end = a.off[idx + 1];
        // tb_base is only a hint (it decides which records may use the tuple path, never a result), so
        // the order-free fast parser is enough: samples it is not sure about are skipped
        GlobalSrc src{reinterpret_cast<const uint32_t*>(a.buf)};
        bool ok = end >= pos;
        if (ok && a.framed) {
            uint32_t pl = 0;
            ok = frame_fast(window64(src, pos), end - pos, pl);
            pos += pl;
        }
        if (ok) {
            // what proto.Marshal emits (mocker.go:97): [Type 08 xx] then TimeReceived 10 <varint> - one or two
            // cache-resident windows instead of a walk over the whole record; anything else: the general parser
            uint64_t w = window64(src, pos);
            if ((w & 0x80ffu) == 0x0008u) {
                pos += 2;
                w = window64(src, pos);
            }
            uint32_t vl;
            uint64_t val;
            if ((w & 0xffu) == 0x10u && varint6(w >> 8, vl, val) && pos + 1 + vl <= end) {
                atomicMin(&lo, time_bucket(a, (uint32_t)val));
            } else {
                Rec r;
                rec_clear(r);
                if (parse_fast<COL_TIME_RECEIVED>(src, pos, end, r)) atomicMin(&lo, time_bucket(a, (uint32_t)r.time_received));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.ctr->tb_base = lo == 0xffffffffu ? 0u : (lo > 2u ? lo - 2u : 0u);
        a.ctr->exotic_count = 0;  // the batch's deferral lists start empty (saves two memset dispatches per batch)
        a.ctr->retry_count = 0;
    }
}

// Records the tile kernel could not stage (broken offsets, tiles larger than the LDS buffer): complete
// semantics, one record per lane straight from HBM.
template <int MODE, uint32_t KEYSETS>
__device__ __forceinline__ void exotic_pass(const KArgs& a) {
    const uint32_t cnt = a.ctr->exotic_count;
    for (uint32_t j = blockIdx.x * BLOCK + threadIdx.x; j < cnt; j += gridDim.x * BLOCK) {
        uint32_t idx = a.exotic_idx[j];
        const uint8_t* p = a.buf + a.off[idx];
        const uint8_t* end = a.buf + a.off[idx + 1];
        bool ok = end >= p;
        if (ok && a.framed) ok = frame_generic(p, end);
        Rec r;
        if (ok)
            ok = parse_generic(p, end, r);
        if (!ok) rec_clear(r);
        atomicAdd(&a.ctr->slow, 1ull);
        if (MODE == MODE_DECODE) {
            store_columns(a.cols, idx, r, ok ? 0 : 1);
            continue;
        }
        if (!ok) {
            atomicAdd(&a.ctr->bad, 1ull);
            continue;
        }
        atomicAdd(&a.ctr->ok, 1ull);
        const uint32_t tb = (uint32_t)r.time_received / a.gran;
        if (ks_on<KEYSETS>(a, FA_KEYS_AS_PAIR)) {
            uint64_t k0, k1;
            pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
            agg_global(a, k0, k1, key_hash(k0, k1), r.bytes, r.packets, 1);
        }
        uint64_t w = r.bytes * r.sampling_rate;
        if (ks_on<KEYSETS>(a, FA_KEYS_SRCADDR_CMS)) {
            cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, w);
            keyset_insert(a, a.ks_src, r.src);
        }
        if (ks_on<KEYSETS>(a, FA_KEYS_DSTADDR_CMS)) {
            cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, w);
            keyset_insert(a, a.ks_dst, r.dst);
        }
        if (KEYSETS & FA_KEYS_WIDE) wide_sink_slow<KEYSETS>(a, r, tb);
    }
}

// ---- second chance: records parse_canon deferred ---------------------------------------------
// One record per lane straight from HBM/L2 with the order-free fast parser; what it is not sure
// about is decided in place by the complete parser.  Updates go to the device-wide table after a wave-level combine
// (this tier is about staying exact and tolerable on producers that do not emit canonical order).
template <int MODE, uint32_t KEYSETS>
__global__ __launch_bounds__(BLOCK) void deferred_kernel(KArgs a) {
    constexpr uint32_t COLS = MODE == MODE_DECODE ? (uint32_t)COL_ALL : cols_for_keysets<KEYSETS>();
    exotic_pass<MODE, KEYSETS>(a);
    const uint32_t cnt = a.ctr->retry_count;
    const uint32_t rounds = (cnt + gridDim.x * BLOCK - 1) / (gridDim.x * BLOCK);
    uint32_t n_ok = 0;
    for (uint32_t it = 0; it < rounds; it++) {  // whole waves stay together (wave_combine below)
        const uint32_t j = (it * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x;
        bool sure = false;
        Rec r;
        rec_clear(r);
        uint32_t idx = 0;
        if (j < cnt) {
            idx = a.retry_idx[j];
            uint32_t pos = a.off[idx], end = a.off[idx + 1];
            GlobalSrc src{reinterpret_cast<const uint32_t*>(a.buf)};
            sure = end >= pos;
            if (sure && a.framed) {
                uint32_t pl = 0;
                sure = frame_fast(window64(src, pos), end - pos, pl);
                pos += pl;
            }
            if (sure) sure = parse_fast<COLS>(src, pos, end, r);
            if (!sure) {  // third tier, in place: the complete parser decides
                const uint8_t* p = a.buf + a.off[idx];
                const uint8_t* pe = a.buf + a.off[idx + 1];
                bool ok = pe >= p;
                if (ok && a.framed) ok = frame_generic(p, pe);
                if (ok) ok = parse_generic(p, pe, r);
                if (!ok) rec_clear(r);
                atomicAdd(&a.ctr->slow, 1ull);
                if (MODE == MODE_DECODE) store_columns(a.cols, idx, r, ok ? 0 : 1);
                else if (!ok) atomicAdd(&a.ctr->bad, 1ull);
                sure = ok && MODE != MODE_DECODE;
            }
        }
        if (MODE == MODE_DECODE) {
            if (sure) store_columns(a.cols, idx, r, 0);
            continue;
        }
        n_ok += sure ? 1 : 0;
        const uint32_t tb = time_bucket(a, (uint32_t)r.time_received);
        if (ks_on<KEYSETS>(a, FA_KEYS_AS_PAIR)) {
            uint64_t k0, k1;
            pack_key(tb, r.src_as, r.dst_as, r.etype, k0, k1);
            uint64_t b = r.bytes, p = r.packets, c = 1;
            bool valid = sure;
            wave_combine<16, 2>(valid, k0, k1, b, p, c);
            if (valid) agg_global(a, k0, k1, key_hash(k0, k1), b, p, c);
        }
        if (sure && (KEYSETS & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS))) {
            const uint64_t w = r.bytes * r.sampling_rate;
            if (ks_on<KEYSETS>(a, FA_KEYS_SRCADDR_CMS)) {
                cms_add(a.cms_src, a.cms_depth, a.cms_wl2, a.cms_seed, r.src, w);
                keyset_insert(a, a.ks_src, r.src);
            }
            if (ks_on<KEYSETS>(a, FA_KEYS_DSTADDR_CMS)) {
                cms_add(a.cms_dst, a.cms_depth, a.cms_wl2, a.cms_seed, r.dst, w);
                keyset_insert(a, a.ks_dst, r.dst);
            }
        }
        if (sure && (KEYSETS & FA_KEYS_WIDE)) wide_sink_slow<KEYSETS>(a, r, tb);
    }
    if (MODE == MODE_INGEST) {
        uint64_t tot = wave_sum_u64(n_ok);
        if (__lane_id() == 0 && tot) {
            atomicAdd(&a.ctr->ok, (unsigned long long)tot);
            atomicAdd(&a.ctr->retried, (unsigned long long)tot);
        }
    }
}

// ---- aggregation of the scattered tuples -----------------------------------------------------
// One 1024-thread workgroup per key partition.  LDS table slot = key (2 words, same claim protocol
// as the other tables) + two packed sums:  s1 = sum(bytes) (< 2^28 * 2^24),
// s2 = sum(packets) << 25 | count  (packets < 2^15, count <= 2^24: no carry between the fields).
struct AggTable {
    unsigned long long k0[AGG_SLOTS], k1[AGG_SLOTS], s1[AGG_SLOTS], s2[AGG_SLOTS];
};
static_assert(sizeof(AggTable) == 32 * AGG_SLOTS, "agg_kernel LDS table");

// slow path of the LDS upsert: claim / probe; false = the table is full around this hash
// (skip = leading slots of the probe sequence already known to hold other keys: a key never changes)
__device__ __forceinline__ bool agg_lds_upsert(AggTable& lt, uint64_t k0, uint64_t k1, uint32_t h, uint32_t by,
                                               unsigned long long v2, uint32_t skip = 0) {
    uint32_t i = (h + skip) & (AGG_SLOTS - 1);
#pragma unroll 1
    for (int probe = (int)skip; probe < AGG_PROBES; probe++, i = (i + 1) & (AGG_SLOTS - 1)) {
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

struct AggBatch {
    uint4 t[AGG_SU];
    uint32_t v;  // bit s: t[s] is a tuple of this lane (not a dummy load)
};

// issue the loads of lanes [0,64) of AGG_SU consecutive segments starting at w0.  The segment counts
// come from LDS (pc, zero padded): a count read from global memory would put a full vmcnt drain between
// consecutive tuple loads.
// Front parts: chunk level j = tuples [64j, 64j+64) of a segment, one segment per 64 lanes.
// Back parts (a handful of tuples each): level j = tuples [8j, 8j+8) of the c tuples that end at the segment's
// last slot, EIGHT segments per 64 lanes.  Loads are unconditional (lanes without a tuple re-read slot 0 of a
// valid segment): with predicated loads the compiler cannot count what is in flight and drains everything
// (vmcnt(0)) before the previous batch is consumed.
template <bool BACK>
__device__ __forceinline__ void agg_fetch(const KArgs& a, const uint4* pbase, const uint32_t* pc, uint32_t w0, uint32_t lane,
                                          uint32_t j, AggBatch& b) {
    constexpr uint32_t SEGS = BACK ? 8u : 1u, PER = 64u / SEGS;  // segments per load, lanes per segment
    uint32_t idx[AGG_SU], seg[AGG_SU];
    b.v = 0;
#pragma unroll
    for (int s = 0; s < AGG_SU; s++) {
        seg[s] = w0 + (uint32_t)s * SEGS + (BACK ? lane / PER : 0u);
        const uint32_t c = pc[min(seg[s], (uint32_t)(AGG_MAX_NWG + AGG_PAD - 1))];  // 0 past nwg
        const uint32_t q = PER * j + (BACK ? lane % PER : lane);
        const bool valid = q < c;
        b.v |= valid ? 1u << s : 0u;
        idx[s] = valid ? (BACK ? a.capq - c : 0u) + q : 0u;
    }
#pragma unroll
    for (int s = 0; s < AGG_SU; s++) b.t[s] = pbase[(size_t)min(seg[s], a.nwg - 1u) * a.capq + idx[s]];
}

__device__ __forceinline__ void agg_tuple(const KArgs& a, AggTable& lt, uint32_t tb_base, const uint4& t) {
    const uint32_t by = t.z & 0x0fffffffu, tbr = t.z >> 28, pk = t.w & 0x7fffu, et = t.w >> 15;
    uint64_t k0, k1;
    pack_key(tb_base + tbr, t.x, t.y, et, k0, k1);
    const uint32_t h = key_hash(k0, k1);
    const unsigned long long v2 = ((unsigned long long)pk << 25) | 1ull;
    if (!agg_lds_upsert(lt, k0, k1, h, by, v2)) agg_global(a, k0, k1, h, by, pk, 1);
}

// The common case (the key already sits in its home slot) for all AGG_SU tuples at once, so that the LDS
// round trips of the segments overlap; everything else goes through the probing upsert.
// the queued leftovers of a wave, one per lane (LDS operations of a wave complete in order: the queue needs no fence)
__device__ __forceinline__ void agg_drain(const KArgs& a, AggTable& lt, uint32_t tb_base, uint32_t lane, const uint4* queue, uint32_t qn) {
    if (lane < qn) agg_tuple(a, lt, tb_base, queue[lane]);
}

template <int S0>
__device__ __forceinline__ void agg_consume_chunk(const KArgs& a, AggTable& lt, uint32_t tb_base, uint32_t lane, const AggBatch& b,
                                                  uint4* queue, uint32_t& qn) {
    uint64_t k0[AGG_CH], k1[AGG_CH];
    uint32_t h[AGG_CH];
    unsigned long long c0[AGG_CH], c1[AGG_CH];
    if (a.dbg & DBG_AGG_NO_LDS) {  // ablation: consume the loads only
        uint32_t x = 0;
#pragma unroll
        for (int s = 0; s < AGG_CH; s++) x ^= b.t[S0 + s].x ^ b.t[S0 + s].y ^ b.t[S0 + s].z ^ b.t[S0 + s].w;
        if (x == 0x12345678u) lt.s1[lane] = x;
        return;
    }
    // home slot and its successor are read together: at the table's load (<= 40 %) linear probing leaves ~30 % of
    // the keys one slot away from home and ~10 % further; only the latter (and first occurrences) take the
    // probing path below
    unsigned long long d0[AGG_CH], d1[AGG_CH];
#pragma unroll
    for (int s = 0; s < AGG_CH; s++) {
        const uint32_t tbr = b.t[S0 + s].z >> 28, et = b.t[S0 + s].w >> 15;
        pack_key(tb_base + tbr, b.t[S0 + s].x, b.t[S0 + s].y, et, k0[s], k1[s]);
        h[s] = key_hash(k0[s], k1[s]);
        const uint32_t i = h[s] & (AGG_SLOTS - 1), j = (i + 1) & (AGG_SLOTS - 1);
        c0[s] = lt.k0[i];
        c1[s] = lt.k1[i];
        d0[s] = lt.k0[j];
        d1[s] = lt.k1[j];
    }
    uint32_t pending = 0;  // segments whose tuple is not in its home slot (probing / claiming needed)
    uint32_t skipw = 0;    // 2 bits per segment: leading probe slots known to hold other keys
#pragma unroll
    for (int s = 0; s < AGG_CH; s++) {
        if (!((b.v >> (S0 + s)) & 1u)) continue;
        const uint32_t by = b.t[S0 + s].z & 0x0fffffffu, pk = b.t[S0 + s].w & 0x7fffu;
        const unsigned long long v2 = ((unsigned long long)pk << 25) | 1ull;
        const uint32_t i = h[s] & (AGG_SLOTS - 1);
        const bool at0 = c0[s] == k0[s] && c1[s] == k1[s], at1 = d0[s] == k0[s] && d1[s] == k1[s];
        if (at0 || at1) {
            const uint32_t t = at0 ? i : (i + 1) & (AGG_SLOTS - 1);
            if (by) atomicAdd(&lt.s1[t], (unsigned long long)by);
            atomicAdd(&lt.s2[t], v2);
        } else {
            // slots that definitely belong to other keys need no second look on the probing path
            const bool o0 = (c0[s] != 0 && c0[s] != k0[s]) || (c0[s] == k0[s] && c1[s] != 0 && c1[s] != k1[s]);
            const bool o1 = (d0[s] != 0 && d0[s] != k0[s]) || (d0[s] == k0[s] && d1[s] != 0 && d1[s] != k1[s]);
            skipw |= (o0 ? (o1 ? 2u : 1u) : 0u) << (2 * s);
            pending |= 1u << s;
        }
    }
    if (a.dbg & DBG_AGG_NO_SLOW) return;
    // The leftovers (first occurrences of a group, keys two or more slots from home: ~5 % of the tuples) wait in
    // the wave's queue and take the probing path 64 at a time: handled on the spot, each round of the probing
    // loop would run with one or two active lanes.
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
            if (pnd) queue[qn + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = b.t[S0 + s];
            qn += cnt;
        }
    }
}

__device__ __forceinline__ void agg_consume(const KArgs& a, AggTable& lt, uint32_t tb_base, uint32_t lane, const AggBatch& b,
                                            uint4* queue, uint32_t& qn) {
    agg_consume_chunk<0>(a, lt, tb_base, lane, b, queue, qn);
    if (AGG_SU > AGG_CH) agg_consume_chunk<AGG_SU - AGG_CH>(a, lt, tb_base, lane, b, queue, qn);
}

__global__ __launch_bounds__(AGG_BLOCK) void agg_kernel(KArgs a) {
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
    if (threadIdx.x < 2) maxc_s[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mymax = 0, mymaxb = 0;
    for (uint32_t i = threadIdx.x; i < AGG_MAX_NWG + AGG_PAD; i += AGG_BLOCK) {
        const uint32_t c = i < a.nwg ? a.seg_counts[(size_t)part * a.nwg + i] : 0u;
        const uint32_t cb = i < a.nwg ? a.seg_counts[((size_t)NPART_MAX + part) * a.nwg + i] : 0u;
        pc[i] = c;
        pcb[i] = cb;
        mymax = max(mymax, c);
        mymaxb = max(mymaxb, cb);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mymax = max(mymax, (uint32_t)__shfl_xor((int)mymax, o));
        mymaxb = max(mymaxb, (uint32_t)__shfl_xor((int)mymaxb, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&maxc_s[0], mymax);
        atomicMax(&maxc_s[1], mymaxb);
    }
    const uint32_t tb_base = a.ctr->tb_base;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint4* pbase = a.seg + (size_t)part * a.region;
    constexpr uint32_t STEP = (AGG_BLOCK / 64) * AGG_SU * AGG_SPLIT;
    __syncthreads();  // table cleared, counts staged
    const uint32_t maxc = maxc_s[0], maxcb = maxc_s[1];
    uint4* queue = queues + wave * 64;
    uint32_t qn = 0;  // wave-uniform
    // software pipeline over this wave's segment groups: the next group's loads fly during the LDS work
    // (every fetch is unconditional - clamped addresses, zero counts past the end - so that the compiler
    // can count the loads in flight and wait for the older batch only)
    // chunk levels: level j covers tuples [64j, 64j+64) of every segment (segments of the 512-thread tile kernel
    // hold ~127 tuples, those of the 256-thread one ~42)
    const uint32_t levels = (__builtin_amdgcn_readfirstlane(maxc) + 63u) >> 6;
    for (uint32_t j = 0; j < levels; j++) {
        AggBatch b0, b1;
        uint32_t w0 = (sub * (AGG_BLOCK / 64) + wave) * AGG_SU;
        agg_fetch<false>(a, pbase, pc, w0, lane, j, b0);
        while (true) {
            agg_fetch<false>(a, pbase, pc, w0 + STEP, lane, j, b1);
            agg_consume(a, lt, tb_base, lane, b0, queue, qn);
            agg_fetch<false>(a, pbase, pc, w0 + 2 * STEP, lane, j, b0);
            agg_consume(a, lt, tb_base, lane, b1, queue, qn);
            w0 += 2 * STEP;
            if (w0 >= a.nwg) break;
        }
    }
    const uint32_t levels_b = (__builtin_amdgcn_readfirstlane(maxcb) + 7u) >> 3;
    constexpr uint32_t STEP_B = STEP * 8u;
    for (uint32_t j = 0; j < levels_b; j++) {  // the back parts (single tuples and bin leftovers of the wave-tile kernel)
        AggBatch b0, b1;
        uint32_t w0 = (sub * (AGG_BLOCK / 64) + wave) * AGG_SU * 8u;
        agg_fetch<true>(a, pbase, pcb, w0, lane, j, b0);
        while (true) {
            agg_fetch<true>(a, pbase, pcb, w0 + STEP_B, lane, j, b1);
            agg_consume(a, lt, tb_base, lane, b0, queue, qn);
            agg_fetch<true>(a, pbase, pcb, w0 + 2 * STEP_B, lane, j, b0);
            agg_consume(a, lt, tb_base, lane, b1, queue, qn);
            w0 += 2 * STEP_B;
            if (w0 >= a.nwg) break;
        }
    }
    agg_drain(a, lt, tb_base, lane, queue, qn);
    __syncthreads();
    // every group of this partition goes to the device-wide table once; quad-grouped: one atomic line
    // transaction per group.  Uniform trip count: the whole wave takes part in the quad rounds.
    if (a.dbg & DBG_AGG_NO_FLUSH) return;
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
        home[q] = *reinterpret_cast<const ulonglong2*>(&a.tab[fh[q] & a.mask]);
    }
#pragma unroll
    for (int q = 0; q < NF; q++) {
        Slot* sp = nullptr;
        const unsigned long long b = fs1[q], p = fs2[q] >> 25, c = fs2[q] & 0x1ffffffull;
        if (fk0[q] != 0 && fk1[q] != 0 && fs2[q] != 0) {
            if (home[q].x == fk0[q] && home[q].y == fk1[q]) sp = &a.tab[fh[q] & a.mask];
            else sp = table_find_or_claim(a, fk0[q], fk1[q], fh[q]);
            if (!sp) spill_park(a, fk0[q], fk1[q], b, p, c);
        }
        quad_atomic_update(sp, b, p, c);
    }
}

// ---- window close ---------------------------------------------------------------
struct Row5m {
    uint32_t date, timeslot, src_as, dst_as, etype, pad;
    unsigned long long bytes, packets, count;
};

// Appends rows whose time bucket lies in [tb_lo, tb_hi) to `rows`.
__global__ void extract_kernel(const Slot* tab, uint32_t nslots, uint32_t gran, uint32_t tb_lo,
                               uint32_t tb_hi, Row5m* rows, uint32_t rows_cap, Counters* ctr) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        const Slot& s = tab[i];
        if (s.k0 == 0 || s.k1 == 0 || s.count == 0) continue;
        uint32_t tb, sa, da, et;
        unpack_key(s.k0, s.k1, tb, sa, da, et);
        if (tb < tb_lo || tb >= tb_hi) continue;
        unsigned int j = atomicAdd(&ctr->rows_count, 1u);
        if (j < rows_cap) {
            uint32_t ts = tb * gran;
            rows[j] = Row5m{ts / 86400u, ts, sa, da, et, 0, s.bytes, s.packets, s.count};
        }
    }
}

// Re-inserts every row outside [tb_lo, tb_hi) into a fresh table (window removal / growth).
__global__ void rebuild_kernel(const Slot* old_tab, uint32_t old_slots, uint32_t tb_lo, uint32_t tb_hi,
                               KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < old_slots; i += gridDim.x * blockDim.x) {
        const Slot& s = old_tab[i];
        if (s.k0 == 0 || s.k1 == 0 || s.count == 0) continue;
        uint32_t tb, sa, da, et;
        unpack_key(s.k0, s.k1, tb, sa, da, et);
        if (tb >= tb_lo && tb < tb_hi) continue;
        agg_global(a, s.k0, s.k1, key_hash(s.k0, s.k1), s.bytes, s.packets, s.count);
    }
}

__global__ void replay_spill_kernel(const SpillEntry* sp, uint32_t n, KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        agg_global(a, sp[i].k0, sp[i].k1, key_hash(sp[i].k0, sp[i].k1), sp[i].bytes, sp[i].packets,
                   sp[i].count);
}

// rows produced elsewhere (another GPU / Kafka partition) folded into this table
__global__ void merge_rows_kernel(const Row5m* rows, uint32_t n, KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t k0, k1;
        pack_key(rows[i].timeslot / a.gran, rows[i].src_as, rows[i].dst_as, rows[i].etype, k0, k1);
        agg_global(a, k0, k1, key_hash(k0, k1), rows[i].bytes, rows[i].packets, rows[i].count);
    }
}

// ---- wide table maintenance ---------------------------------------------------------------------
struct WRow {
    unsigned long long w[4], v0, v1, v2;
};
// which rows: kind_mask bit k selects kind k; WK_APP rows additionally need tb in [tb_lo, tb_hi)
__device__ __forceinline__ bool wrow_selected(const unsigned long long w[4], uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi) {
    uint32_t kind, tb, port, proto;
    uint64_t lo, hi;
    wkey_unpack(w, kind, tb, lo, hi, port, proto);
    if (!((kind_mask >> kind) & 1u)) return false;
    return kind != WK_APP || (tb >= tb_lo && tb < tb_hi);
}
__global__ void wextract_kernel(const WSlot* tab, uint32_t nslots, uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi, WRow* rows,
                                uint32_t rows_cap, Counters* ctr) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        const WSlot& s = tab[i];
        if (s.w[0] == 0 || s.w[1] == 0 || s.w[2] == 0 || s.w[3] == 0 || s.v2 == 0) continue;
        if (!wrow_selected(s.w, kind_mask, tb_lo, tb_hi)) continue;
        const unsigned int j = atomicAdd(&ctr->wrows_count, 1u);
        if (j < rows_cap) rows[j] = WRow{{s.w[0], s.w[1], s.w[2], s.w[3]}, s.v0, s.v1, s.v2};
    }
}
// Re-inserts every row that is NOT selected into a fresh table (window removal / reset / growth).
__global__ void wrebuild_kernel(const WSlot* old_tab, uint32_t old_slots, uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi, KArgs a) {
    const WArgs t = wargs(a);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < old_slots; i += gridDim.x * blockDim.x) {
        const WSlot& s = old_tab[i];
        if (s.w[0] == 0 || s.w[1] == 0 || s.w[2] == 0 || s.w[3] == 0 || s.v2 == 0) continue;
        if (wrow_selected(s.w, kind_mask, tb_lo, tb_hi)) continue;
        WKey k{{s.w[0], s.w[1], s.w[2], s.w[3]}};
        wagg_global(t, k, s.v0, s.v1, s.v2);
    }
}
// parked updates / rows produced elsewhere (another GPU / Kafka partition) folded into this table
__global__ void wmerge_kernel(const WRow* rows, uint32_t n, KArgs a) {
    const WArgs t = wargs(a);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        WKey k{{rows[i].w[0], rows[i].w[1], rows[i].w[2], rows[i].w[3]}};
        wagg_global(t, k, rows[i].v0, rows[i].v1, rows[i].v2);
    }
}

// ---- heavy hitters ---------------------------------------------------------------------------
// One row per stored key: its Count-Min estimate = min over the sketch rows (>= the exact
// sum(Bytes*SamplingRate), viz-ch.json:233).  The host sorts, removes duplicate keys and cuts at k.
__global__ void topk_rows_kernel(const KeySlot* ks, uint32_t nslots, const unsigned long long* cms, uint32_t depth,
                                 uint32_t wl2, uint64_t seed, TopkRow* rows, uint32_t rows_cap, Counters* ctr) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += gridDim.x * blockDim.x) {
        if ((ks[i].tag & KS_READY) == 0) continue;
        const unsigned long long lo = ks[i].lo, hi = ks[i].hi;
        unsigned long long best = ~0ull;
        for (uint32_t r = 0; r < depth; r++) {
            const unsigned long long v = cms[((size_t)r << wl2) + (size_t)(cms_hash(lo, hi, seed, r) >> (64 - wl2))];
            best = v < best ? v : best;
        }
        const unsigned int j = atomicAdd(&ctr->ks_rows, 1u);
        if (j < rows_cap) rows[j] = TopkRow{lo, hi, best};
    }
}

// keys found by other GPUs / Kafka partitions join this context's candidate set (window close)
__global__ void keyset_merge_kernel(const uint4* keys, uint32_t n, KeySlot* tab, KArgs a) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t k[4] = {keys[i].x, keys[i].y, keys[i].z, keys[i].w};
        keyset_insert(a, tab, k);
    }
}

// ---- synthetic producer ------------------------------------------------------------
__global__ void gen_len_kernel(fa_mock_params g, uint64_t i0, uint32_t n, uint32_t* len) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t tmp[208];
    len[i] = gen_encode(g, i0 + i, tmp);
}
__global__ void gen_write_kernel(fa_mock_params g, uint64_t i0, uint32_t n, const uint32_t* off,
                                 uint8_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t tmp[208];
    uint32_t l = gen_encode(g, i0 + i, tmp);
    uint8_t* p = out + off[i];
    for (uint32_t k = 0; k < l; k++) p[k] = tmp[k];
}

}  // namespace fa
