// sinks.cuh - shared state of the gfx950 kernels (constants, counters, kernel arguments) and the sinks a decoded
// record can go to: device-wide group-by table (quad-grouped atomics), Count-Min sketch copies, distinct-address
// set, SoA columns, wide-key table, port histograms.  See kernels.cuh for the map of the hot path.
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
// FA_WBLOCK / FA_WT_STRIDE / FA_BIN_BYTES: geometry experiments (tools/gpu_ab.sh builds variants of the library).
// Geometry of the wave-tile kernel, AS-rollup variant: 2 workgroups x 12 waves per CU = 6 waves per SIMD, tile buffers
// of 4864 bytes (64 records of 76 bytes), half-line bins (64 bytes = 8 compact tuples).  Round 1 ran 2 x 8 waves with
// 5472-byte tiles and full-line bins (all the LDS there was next to 16-byte tuples); with compact tuples a half-line bin
// holds as many tuples as a full one did, and the LDS this frees goes to 8 more waves per CU: +3.5 % on config 2
// (same-box A/B: half-line bins alone -2.8 %, 24 waves with them +3.3..3.6 %, 20 waves -23 % - uneven over the 4 SIMDs).
#ifndef FA_WBLOCK
#define FA_WBLOCK 768
#endif
#ifndef FA_WT_STRIDE
#define FA_WT_STRIDE 4864
#endif
#ifndef FA_BIN_BYTES
#define FA_BIN_BYTES 64
#endif
constexpr int WBLOCK = FA_WBLOCK;   // 12 waves, each with a private LDS tile of <= 64 records
constexpr int WT_RECS = 64;
constexpr int WT_STRIDE = FA_WT_STRIDE;  // 4864 = 64 records x 76 B (16-byte multiple); longer records: fewer per tile; overreads land in the next tile / the bins
constexpr int WT_STRIDE_CMS = 5056;  // tile buffers of the kernel variants that serve a sketch (ingest.cuh, wtile_stride): 16 waves, what the 160 KiB leave
constexpr int WT_WG_PER_CU = 2;
constexpr int WBLOCK_CMS = 2 * FA_WBLOCK > 1024 ? 1024 : 2 * FA_WBLOCK;  // sketch variants: one big workgroup per CU (ingest.cuh, wtile_block)
static_assert(WBLOCK % 64 == 0 && WBLOCK >= 256 && WBLOCK <= 1024 && WT_STRIDE % 16 == 0, "wave-tile geometry");
// A bin = one store unit of tuples per key partition, in uint4 (16-byte) words: a whole 128-byte line (8) in every kernel
// variant that serves a sketch or a wide key set (16 waves per workgroup; half-line bins measured -6 % there), half a
// line (4 = 8 compact tuples) in the flows_5m-only variant (24 waves per CU, see FA_WBLOCK above).  One function for
// the kernel templates and the host (segment geometry): what launch_tiles dispatches on is the key-set mask.
static_assert(FA_BIN_BYTES == 128 || FA_BIN_BYTES == 64, "bin = one or half a cache line");
// (wt_lean: the variants without a sketch - 12-wave workgroups, two per CU: flows_5m alone, and config 5's pair
// flows_5m + (SrcAddr,DstPort,Proto); every other mask runs the 16-wave geometry, WBLOCK_CMS / WT_STRIDE_CMS)
__host__ __device__ constexpr bool wt_lean(uint32_t key_sets) { return key_sets == FA_KEYS_AS_PAIR || key_sets == (FA_KEYS_AS_PAIR | FA_KEYS_ADDR_PORT_PROTO); }
// tile buffer of a variant; the ones that scatter (SrcAddr,DstPort,Proto) tuples give 128-160 bytes per wave to the 512 region
// counters (two wagg_kernel workgroups per CU) - and still fit two workgroups per CU / the 160 KiB
__host__ __device__ constexpr int wt_stride(uint32_t key_sets) {
    return (wt_lean(key_sets) ? WT_STRIDE : WT_STRIDE_CMS) - ((key_sets & FA_KEYS_ADDR_PORT_PROTO) ? (wt_lean(key_sets) ? 128 : 160) : 0);
}
__host__ __device__ constexpr uint32_t bin_line(uint32_t key_sets) { return wt_lean(key_sets) ? FA_BIN_BYTES / 16u : 8u; }
// tuples per bin: wide (16-byte) or compact (8-byte) tuples
template <bool T8, uint32_t BL>
constexpr uint32_t bin_cap() { return (T8 ? 2u : 1u) * BL; }
// (Round 4 measured a second geometry for long records - GoFlow's 156-byte sFlow samples fill 30 of a wave's 64 lanes -: half
// the waves per workgroup, tile buffers twice as long, ~60 records per wave and round in the same LDS.  1.08 ms per launch
// against 0.75 ms: the walk is a chain of dependent LDS reads per record, what hides it is the number of waves, not the
// number of busy lanes.  profiles/r04_goflow_long_geometry.json; the variant is gone.)
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
constexpr uint32_t AGG_MAX_BATCH = 1u << 24;  // wide tuples: count <= 2^24 per slot keeps the packed LDS sums exact (Packets < 2^15: 15 + 24 + 25 bits)
constexpr uint32_t AGG8_MAX_BATCH = (1u << 25) - 1u;  // compact tuples (Packets < 2^9): 25 bits of count
static_assert(TILE_STRIDE % 16 == 0, "LDS tile buffers must stay 16-byte aligned");

enum { MODE_INGEST = 0, MODE_DECODE = 1 };
// Kernel variants are compiled for the key-set masks 1..7 (rollup and/or sketches); every other
// combination runs the KS_ALL variant, which parses the union of the columns and tests the runtime mask.
constexpr uint32_t KS_ALL = 0xFFu;
constexpr uint32_t FA_KEYS_WIDE = FA_KEYS_ADDR_PORT_PROTO | FA_KEYS_PORT_HIST | FA_KEYS_MINUTE_SERIES;
constexpr uint32_t PORT_DENSE = 65536;  // ports below this live in the dense histograms
// ablation switches (env FA_DEBUG_FLAGS; measurement only - results are wrong when set)
// FA_DEBUG_FLAGS ablation switches (results are wrong by design).  Production builds compile them OUT (FA_ABLATE 0:
// every test below folds to false - no scalar tests, no dead branches in the hot kernels); the measurement scripts under
// tools/ build a variant with EXTRA=-DFA_ABLATE=1.  DBG_AGG_ATOMIC_FLUSH stays a runtime switch (exact: an A/B of two
// correct flush protocols).
#ifndef FA_ABLATE
#define FA_ABLATE 0
#endif
#ifndef FA_TL_WINDOW
#define FA_TL_WINDOW 256u
#endif
#define FA_DBG(a, flags) (FA_ABLATE != 0 && ((a).dbg & (flags)) != 0)
enum { DBG_NO_SINK = 1, DBG_LOOP_PARSER = 2, DBG_NO_LDS_TABLE = 4, DBG_NO_GLOBAL = 8, DBG_NO_PARSE = 16, DBG_NO_TUPLE_STORE = 32,
       DBG_AGG_NO_LDS = 64, DBG_AGG_NO_FLUSH = 128, DBG_AGG_NO_SLOW = 256, DBG_DMA_NO_NT = 512, DBG_TIMING = 1024, DBG_TUPLE_NT = 2048, DBG_TUPLE_SC = 4096, DBG_NO_LANE_OFF = 8192, DBG_SYNTH_TILES = 16384, DBG_NOT_MINE = 32768, DBG_NO_SECOND = 65536, DBG_NO_FRAME = 131072, DBG_NO_KEYSET = 262144, DBG_NO_CMS = 524288, DBG_NO_HOT = 1048576, DBG_CMS_TUPLE_LOCAL = 268435456 /* sketch tuples of full bins go to a 4 KiB window per workgroup: their HBM writes go, everything else stays (what narrower sketch tuples could buy on the write side) */, DBG_NO_SINGLES = 134217728 /* tuples that meet a closing bin are dropped instead of leaving as single stores: what the singles cost in WRITE_SIZE and time */, DBG_TUPLE_LOCAL = 67108864 /* tuple stores go to a 4 KiB window per workgroup (L2-resident): the store INSTRUCTIONS and their acknowledgements stay, the HBM write traffic goes */, DBG_CAND_NO_SET = 33554432 /* candidates mode: addresses above the threshold are neither looked up in nor added to the set */, DBG_AGG_ATOMIC_FLUSH = 2097152 /* exact: agg8_kernel adds its groups with atomics although it owns the region (A/B) */,
       DBG_AGG8_TIMING = 8388608 /* agg8_kernel: 100 MHz ticks per workgroup - set-up / segment walk / flush */,
       DBG_CMS_TIMING = 4194304 /* cms_agg_kernel: 100 MHz ticks per workgroup - schedule + counts + flush / segment walk; the waves' own walk times (imbalance) */ };

struct SpillEntry {
    unsigned long long k0, k1, bytes, packets, count;
};

struct Counters {
    unsigned long long ok, bad, slow, spill_lost, used, direct, retried;
    unsigned long long misfit8;  // records of compact-tuple launches that would have fitted a wide tuple only (format feedback)
    // the deferral lists' fill levels come in two copies: batch B appends under copy B & 1 and its deferred kernel
    // clears the OTHER copy for batch B + 1 (no memset dispatches, no reset race inside one kernel)
    unsigned int exotic_count[2], retry_count[2];
    unsigned int spill_count, rows_count, ks_overflow, ks_rows;
    // the launch's time base and, behind it, the bucket range of the (SrcAddr,DstPort,Proto) tuples the launch left in its segments
    // (smallest bucket, ~largest bucket; both start at ~0 before every launch): copied into a wide-log chunk as one piece
    unsigned int tb_base, wtb_min, wtb_nmax, pad;
    unsigned long long t_wait, t_work, t_tiles, t_total;  // DBG_TIMING: core-clock cycles of wave 0 of every workgroup
    unsigned long long wfold_n;  // tuples of wide-log chunks folded into the table so far (beside wused: do folds still open rows?)
    unsigned long long wused, wspill_lost;  // wide table (wide.cuh)
    unsigned int wspill_count, wrows_count;
    unsigned long long agg_groups, agg_launches;  // agg8_kernel: groups it added to the device table, launches (pass-count feedback)
    unsigned long long late;  // records below KArgs::late_below (flows_5m windows that were closed before they arrived)
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
// fa_topk's pre-selection (maintenance.cuh, topk_hist_kernel): a monotone map of an estimate onto 2048 bins - values below 64
// exactly, above that 32 steps per octave (e = 6: bins 64..95 ... e = 63: bins 1888..1919).  a <= b => topk_bin(a) <= topk_bin(b).
constexpr uint32_t TK_BINS = 2048;
__host__ __device__ __forceinline__ uint32_t topk_bin(unsigned long long est) {
    if (est < 64ull) return (uint32_t)est;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t e = 63u - (uint32_t)__clzll((long long)est);
#else
    const uint32_t e = 63u - (uint32_t)__builtin_clzll(est);
#endif
    return ((e - 4u) << 5) | ((uint32_t)(est >> (e - 5u)) & 31u);
}
__host__ __device__ __forceinline__ unsigned long long topk_bin_floor(uint32_t b) {  // the smallest estimate that maps to bin b
    if (b < 64u) return b;
    const uint32_t e = (b >> 5) + 4u;
    return (1ull << e) | ((unsigned long long)(b & 31u) << (e - 5u));
}

// candidates mode (maintenance.cuh "candidates mode"): the state of a sketch's launch boundary, in device memory
struct CandState {
    unsigned int hist[TK_BINS];
    unsigned int sel[4];       // [0] bin of rank K, [1] candidates in the bins >= it, [2] candidates held
    unsigned long long total;  // N_t
    unsigned long long theta;  // theta_t (written by cand_bits_kernel: fa_stats / tests)
};

struct ColumnPtrs {
    uint64_t *time_received, *time_flow_start, *sampling_rate, *bytes, *packets;
    uint32_t *sequence_num, *src_as, *dst_as, *etype, *proto, *src_port, *dst_port;
    uint4 *sampler_address, *src_addr, *dst_addr;
    uint8_t* status;
};

struct HotSeed;
struct KArgs {
    const uint8_t* buf;   // 16-byte aligned device pointer
    const uint32_t* off;  // n+1 offsets
    uint32_t n;
    uint32_t len;     // wire bytes in buf: record bounds beyond it are broken offsets (never dereferenced)
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
    // fa_config.topk_mode = FA_TOPK_CANDIDATES: the sets hold CANDIDATES only - addresses whose Count-Min estimate at the previous
    // launch boundary was >= the threshold of that boundary (maintenance.cuh, cand_*_kernel).  One bit per sketch counter
    // ("counter >= threshold"), depth rows of 2^cms_wl2 bits: an address passes when the bits of all its counters are set -
    // exactly estimate >= threshold.  512 KiB per sketch at 4 x 2^20: L2-resident, against one random 64-byte HBM line per
    // address instance for the exact-universe sets.  nullptr: every address ever seen is kept (the exact mode).
    const uint32_t* cand_src;
    const uint32_t* cand_dst;
    uint32_t cms_nrep;  // sketch copies the atomic paths spread over (CMS_REPLICAS; 1 in candidates mode: the boundary reads copy 0)
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
    uint32_t par;          // batch parity: which copy of the deferral counters this batch uses
    uint32_t rlog2;        // log2(regions of the device-wide table) = as_rlog2(log2 slots)  (table.cuh)
    uint32_t agg_passes;   // agg8_kernel: passes over a partition's tuples (1, 2, 4, 8), each with 1 / passes of the groups in the LDS table
    // Count-Min scatter sink (cseg == nullptr: every sketch update is a memory-side atomic, cms_add)
    uint4* cseg;            // [CMS_SETS * CMS_NPART][cregion] sketch tuples {l1, l2, weight}; partition p, workgroup w: cseg[p*cregion + w*ccapq + q]
    uint32_t* cseg_counts;  // [2][CMS_SETS * CMS_NPART][nwg]: tuples at the front (whole 64-byte chunks) / at the back of a segment
    uint32_t ccapq, ccapf, ccapb;
    unsigned long long cregion;
    uint32_t cms_sub;       // log2(counters per row of a sketch partition) = cms_wl2 - 8 on the scatter path
    HotSeed* hot_seed;       // [nwg][CMS_SETS][HOT_SLOTS] the hot-address caches' entries of the previous launch (tags: hot_seed_tag)
    uint32_t* hot_seed_tag;  // [nwg][CMS_SETS][HOT_SLOTS] 0 = empty
    uint32_t hot_epoch;      // launches of the sketch variants so far (which entries have to re-earn their admission)
    uint32_t* cms_psize;    // [2][CMS_SETS * CMS_NPART] (+ 2 words: the unit counters of cms_agg_kernel's persistent workgroups): sketch tuples per partition in the previous launch (copy `par`) / this one (copy `par ^ 1`): cms_agg_kernel's schedule
    // wide key sets (wide.cuh)
    uint32_t key_sets;     // runtime mask (the KS_ALL kernel variant tests it)
    WSlot* wtab;
    uint32_t wmask;
    WSpillEntry* wspill;
    uint32_t wspill_cap;
    // scatter sink of the (SrcAddr,DstPort,Proto) key set (wseg == nullptr: every update is a chain of memory-side atomics)
    uint4* wseg;             // [2^wplog2][wregion] 32-byte tuples (wide.cuh, wtup_pack); region p, workgroup w: tuple p*wregion + w*wcapq + q
    uint32_t* wseg_counts;   // [2^wplog2][nwg]
    uint32_t wcapq;
    uint32_t wplog2;         // log2(regions of the wide table) = wide_plog2(log2 slots)
    unsigned long long wregion;
    ulonglong2* port_hist;  // [2][PORT_DENSE] {sum(Bytes*SamplingRate), count()}: SrcPort, then DstPort
    uint32_t late_below;    // time buckets below it were closed (flows_5m): records that still arrive for them are counted
};

// The kernel's arguments, read again from its kernarg segment - for the RARE blocks of the ingest kernels (direct path, deferral
// lists, segment-overflow fallbacks).  Kernel arguments are loop-invariant and always loadable, so the compiler loads every one of
// them at the kernel's entry and keeps it in an SGPR for the kernel's life: wtile_kernel<1, true> needs 106 SGPRs that way and pays
// for it with 58 v_writelane / 139 v_readlane (SGPRs spilled to VGPR lanes), a dozen and a half of them per tile.  A block that
// runs once in a thousand tiles reads its table pointer, mask, spill buffer and counters HERE instead - the empty asm makes the
// segment pointer opaque, so the loads stay inside the block - and the by-value copies of those fields die early.
// (every kernel that calls this takes its KArgs as the FIRST parameter: offset 0 of the segment)
__device__ __forceinline__ KArgs cold_args() {
#if defined(__HIP_DEVICE_COMPILE__)
    const KArgs* p = (const KArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // (address space 4 -> generic: the loads below become flat loads - fine for blocks this rare)
    asm volatile("" : "+s"(p));
    return *p;
#else
    return KArgs{};  // (the host pass only parses device code)
#endif
}

__device__ __forceinline__ WArgs wargs(const KArgs& a) {
    return WArgs{a.wtab, a.wmask, a.wmask >> a.wplog2, a.wspill, a.wspill_cap, &a.ctr->wspill_count, &a.ctr->wspill_lost, &a.ctr->wused};
}
// does this kernel variant serve key set X for this launch?
template <uint32_t KEYSETS>
__device__ __forceinline__ bool ks_on(const KArgs& a, uint32_t x) {
    return (KEYSETS & x) != 0 && (KEYSETS != KS_ALL || (a.key_sets & x) != 0);
}

// ---- sinks ------------------------------------------------------------------
__device__ __forceinline__ void agg_global(const KArgs& a, uint64_t k0, uint64_t k1, uint32_t h,
                                           uint64_t b, uint64_t p, uint64_t c) {
    uint32_t i = as_home(k0, k1, h, a.mask, a.rlog2);
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = as_next(i, a.mask, a.rlog2)) {
        Slot* s = &a.tab[i];
        unsigned long long c0 = s->k0;
        if (c0 == 0) c0 = atomicCAS(&s->k0, 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        unsigned long long c1 = s->k1;
        if (c1 == 0) {
            c1 = atomicCAS(&s->k1, 0ull, (unsigned long long)k1);
            if (c1 == 0) count_created(&a.ctr->used);  // this lane created the group
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

// The sketch (DESIGN.md "Sketch"; the test infrastructure restates it in C and in numpy): a PREFIX-PARTITIONED
// Count-Min sketch.  Two 64-bit hashes per key (two mix64 per key, not per key and row):
//   a = mix64(lo ^ mix64(seed + phi));  h1 = mix64(a ^ hi);  h2 = a | 1
//   pbits = min(8, width_log2 - 4), sub = width_log2 - pbits
//   prefix = h1 & (2^pbits - 1);  l1 = h1 >> 32;  l2 = (h2 >> 32) | 1      (32-bit double hashing, Kirsch & Mitzenmacher)
//   column(r) = prefix << sub | (uint32)(l1 + r * l2) >> (32 - sub)
// = 2^pbits independent sketches of width 2^sub; the key picks one with hash bits that take no part in the row hashes.
// Same expected error e / 2^width_log2 * sum(W) as the flat layout, rows independent given the prefix - and all `depth`
// counters of a key sit in ONE partition: an update is one 16-byte tuple {l1, l2, weight} for the scatter sink below
// instead of `depth` tuples (round 2 measured the sink at 4 tuples per address: 40 % of the ingest kernel's time).
__host__ __device__ __forceinline__ void cms_hash2(uint64_t lo, uint64_t hi, uint64_t seed, uint64_t& h1, uint64_t& h2) {
    const uint64_t a = mix64(lo ^ mix64(seed + 0x9E3779B97F4A7C15ull));
    h1 = mix64(a ^ hi);
    h2 = a | 1ull;
}
__host__ __device__ __forceinline__ uint32_t cms_pbits(uint32_t wl2) { return wl2 - 4u < 8u ? wl2 - 4u : 8u; }
struct CmsKey {
    uint32_t prefix, l1, l2;
};
__host__ __device__ __forceinline__ CmsKey cms_key(uint64_t h1, uint64_t h2, uint32_t wl2) {
    return CmsKey{(uint32_t)h1 & ((1u << cms_pbits(wl2)) - 1u), (uint32_t)(h1 >> 32), (uint32_t)(h2 >> 32) | 1u};
}
// column of row r inside the key's partition (sub = width_log2 - pbits bits)
__host__ __device__ __forceinline__ uint32_t cms_low(uint32_t l1, uint32_t l2, uint32_t r, uint32_t sub) { return (l1 + r * l2) >> (32u - sub); }
__host__ __device__ __forceinline__ uint32_t cms_column(const CmsKey& k, uint32_t r, uint32_t wl2) {
    const uint32_t sub = wl2 - cms_pbits(wl2);
    return (k.prefix << sub) | cms_low(k.l1, k.l2, r, sub);
}
// The sketch is kept in CMS_REPLICAS copies; a workgroup adds to copy blockIdx % CMS_REPLICAS and the copies
// are summed into copy 0 before anything reads the sketch (cms_fold_kernel).  Counters of heavy hitters are
// hit by every wave of the chip, and same-address atomics serialize at the memory side (~10 ns each:
// 1.9 M updates of the top Zipf-1.1 key per launch cost ~19 ms on one copy); u64 sums commute, so the folded
// sketch is bit-identical to a single-copy one.
constexpr uint32_t CMS_REPLICAS = 8;
__device__ __forceinline__ void cms_add_key(unsigned long long* cms, uint32_t depth, uint32_t wl2, const CmsKey& k, uint64_t w, uint32_t nrep) {
    if (w == 0) return;
    unsigned long long* copy = cms + (size_t)(blockIdx.x % nrep) * ((size_t)depth << wl2);
    for (uint32_t r = 0; r < depth; r++) atomicAdd(&copy[((size_t)r << wl2) + cms_column(k, r, wl2)], (unsigned long long)w);
}
__device__ __forceinline__ void cms_add(unsigned long long* cms, uint32_t depth, uint32_t wl2,
                                        uint64_t seed, const uint32_t key[4], uint64_t w, uint32_t nrep) {
    if (w == 0) return;
    uint64_t lo = (uint64_t)key[1] << 32 | key[0], hi = (uint64_t)key[3] << 32 | key[2];
    uint64_t h, h2;
    cms_hash2(lo, hi, seed, h, h2);
    cms_add_key(cms, depth, wl2, cms_key(h, h2, wl2), w, nrep);
}
// ---- Count-Min scatter sink --------------------------------------------------------------------------
// Memory-side atomics retire ~24 G/s whatever their scope (tools/atomics_bench.hip): at depth 4 and two sketches that is
// 8 per record - 2.6 G records/s, 2.5 % of the HBM roofline.  The wave-tile kernel therefore treats sketch updates like
// flows_5m tuples: a key's counters all sit in the partition its prefix names (256 partitions per sketch), an update
// leaves the workgroup as ONE 16-byte tuple {l1, l2, weight} through an LDS bin of its partition (4 tuples = one 64-byte
// chunk per store) into the workgroup's private segment of the partition, and cms_agg_kernel (agg.cuh) adds each
// partition up in a dense LDS array (depth x 2^sub counters) and folds it into the sketch with plain coalesced
// read-modify-writes - no atomics at all.  u64 sums commute: bit-identical to the CPU sketch.
// Needs 256 partitions (width_log2 >= 12) of <= 2^14 counters (depth 4: width_log2 <= 20, the default 32 MiB sketch);
// anything else (bigger sketches, deferred records, the workgroup-tile kernel) keeps the atomic path.
constexpr uint32_t CMS_NPART = 256, CMS_SETS = 2, CMS_BIN = 4, CMS_PART_LOG2_MAX = 14;
__host__ __device__ constexpr uint32_t cms_extra_units(uint32_t nlog) { return nlog / 8u; }  // cms_agg_kernel: spare work units for the slices of heavy partitions
struct CmsLds {
    uint4 bins[CMS_SETS * CMS_NPART * CMS_BIN];   // 32 KiB: one 64-byte chunk per partition
    uint32_t bin_cnt[CMS_SETS * CMS_NPART];       // low half: slots taken, high half: slots written (like the tuple bins)
    uint32_t part_cnt[CMS_SETS * CMS_NPART];      // low half: chunks at the front of the segment, high half: tuples at its back
};
// does this sketch geometry go through the scatter sink?
__host__ __device__ __forceinline__ bool cms_scatterable(uint32_t depth, uint32_t wl2) {
    return wl2 >= 12u && ((uint64_t)depth << (wl2 - 8u)) <= (1ull << CMS_PART_LOG2_MAX);
}

// a tuple that found no room in its segment (a heavy hitter's partition): atomics
__device__ __forceinline__ void cms_atomic_tuple(const KArgs& a, uint32_t p, const uint4& t) {
    const unsigned long long w = (unsigned long long)t.w << 32 | t.z;
    cms_add_key((p >> 8) ? a.cms_dst : a.cms_src, a.cms_depth, a.cms_wl2, CmsKey{p & (CMS_NPART - 1u), t.x, t.y}, w, a.cms_nrep);
}
// Full CMS bins leave as whole 64-byte chunks: lane group g (4 lanes) takes the g-th filled bin, each lane copies
// 16 bytes (one tuple) - up to 16 chunks per store instruction.  Same hand-over protocol as bins_flush.  A chunk that
// finds the front part of its segment full is added to the sketch with atomics (skewed batches).
__device__ __forceinline__ void cms_bins_flush(const KArgs& a, CmsLds& cl, uint32_t* scratch, uint32_t fill_part) {
    const unsigned long long fm = __builtin_amdgcn_ballot_w64(fill_part != 0xffffffffu);
    if (fm == 0ull) return;
    const uint32_t ln = __lane_id(), g = ln >> 2, sub = ln & 3u;
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << ln) - 1ull));
    const uint32_t todo = (uint32_t)__builtin_popcountll(fm);
    const uint4* bins4 = cl.bins;
    for (uint32_t base = 0; base < todo; base += 16u) {
        if (fill_part != 0xffffffffu && rank - base < 16u) scratch[rank - base] = fill_part;
        const bool act = g < min(16u, todo - base);
        const uint32_t fp = act ? scratch[g] : 0u;
        const uint32_t c0 = __hip_atomic_load(&cl.bin_cnt[fp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint4 tq = bins4[fp * 4u + sub];
        uint32_t chunk = 0;
        if (act && sub == 0) chunk = atomicAdd(&cl.part_cnt[fp], 1u) & 0xffffu;
        bool late = false;
        if (__builtin_amdgcn_ballot_w64(act && (c0 >> 16) < CMS_BIN) != 0ull) {
            late = true;
            while (__builtin_amdgcn_ballot_w64(act && (__hip_atomic_load(&cl.bin_cnt[fp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> 16) < CMS_BIN) != 0ull) {}
        }
        chunk = (uint32_t)__shfl((int)chunk, (int)(ln & ~3u));
        if (act) {
            const uint4 tv = late ? bins4[fp * 4u + sub] : tq;
            if ((chunk + 1u) * CMS_BIN <= a.ccapf) {
                // (cregion and ccapq are multiples of 4 tuples: every segment starts on a 64-byte boundary)
                if (FA_DBG(a, DBG_CMS_TUPLE_LOCAL)) a.cseg[(size_t)blockIdx.x * 256u + ((chunk * CMS_BIN + sub) & 255u)] = tv;
                else a.cseg[(size_t)fp * a.cregion + (size_t)blockIdx.x * a.ccapq + chunk * CMS_BIN + sub] = tv;
            } else {
                cms_atomic_tuple(a, fp, tv);
                if (sub == 0) atomicSub(&cl.part_cnt[fp], 1u);  // (the chunk was not stored: the 16-bit counter stays <= its cap, never carries)
            }
            if (sub == 0) __hip_atomic_store(&cl.bin_cnt[fp], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}
// ---- per-workgroup hot-address cache ----------------------------------------------------------------
// Heavy hitters (Zipf 1.1: the top address carries 8 % of the records, the top 64 carry 40 %) would send one tuple per
// wave-tile to each of their counters - twice the load of an average sketch slice on the few slices they hit - and one
// distinct-set probe each.  An address that shows up at least twice inside one wave-tile (the wave-level fold tells) may
// claim an entry here; from then on every occurrence in this workgroup is ONE LDS add.  At the end of the launch each
// entry that was hit is worth one sketch update per row and one distinct-set insert.
// 32 sets x 4 ways per sketch (set = 5 bits of the key hash); tag: 0 empty, 1 being written, else a fingerprint (>= 2).
// The entries SURVIVE the launch (KArgs::hot_seed, one copy per workgroup): an address of rank 60-130 shows up twice in
// a wave-tile once or twice per launch and workgroup - admitted anew every launch it was absorbed for half of it or
// never (round 3 measured: 70 % of the address instances still left as tuples, against 54 % for a perfect 64-entry
// cache).  An entry is dropped at the end of a launch in which it was never hit; of a set whose four ways are all taken
// the LIGHTEST entry (weight added during the launch) has to earn its admission again every other launch - a heavy
// address is back within a few tiles, a colder one makes room for whatever shows up twice in a tile first.
constexpr int HOT_SETS = 32, HOT_WAYS = 4, HOT_SLOTS = HOT_SETS * HOT_WAYS;
struct HotAddrs {
    unsigned long long tag2[CMS_SETS][HOT_SLOTS / 2];  // 32-bit tags, the four ways of a set side by side: two 8-byte reads
    __device__ __forceinline__ unsigned int* tag(uint32_t set, uint32_t slot) { return reinterpret_cast<unsigned int*>(&tag2[set][0]) + slot; }
    unsigned long long lo[CMS_SETS][HOT_SLOTS], hi[CMS_SETS][HOT_SLOTS], w[CMS_SETS][HOT_SLOTS];
    unsigned char touched[CMS_SETS][HOT_SLOTS];  // hit during this launch (a seeded entry nobody hits adds nothing and is dropped)
};
struct HotSeed {  // what survives (+ the tag): 20 bytes per entry
    unsigned long long lo, hi;
};
// true = absorbed (the caller neither updates the sketch nor probes the distinct set for this record)
__device__ __forceinline__ bool hot_add(HotAddrs& ht, uint32_t set, uint64_t lo, uint64_t hi, uint64_t h1, uint64_t w, bool admit) {
    const uint32_t base = ((uint32_t)(h1 >> 8) & (HOT_SETS - 1)) * HOT_WAYS;
    const unsigned int fp = (unsigned int)(h1 >> 32) | 2u;
    const unsigned long long ta = __hip_atomic_load(&ht.tag2[set][base / 2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long tb = __hip_atomic_load(&ht.tag2[set][base / 2 + 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned int t[4] = {(unsigned int)ta, (unsigned int)(ta >> 32), (unsigned int)tb, (unsigned int)(tb >> 32)};
    uint32_t way = 4u, free_way = 4u;
    bool busy = false;
#pragma unroll
    for (int q = 3; q >= 0; q--) {
        way = t[q] == fp ? (uint32_t)q : way;
        free_way = t[q] == 0u ? (uint32_t)q : free_way;
        busy = busy || t[q] == 1u;
    }
    if (way < 4u) {
        const uint32_t slot = base + way;
        const unsigned long long klo = ht.lo[set][slot], khi = ht.hi[set][slot];  // (both halves in flight together: one wait)
        if ((klo == lo) & (khi == hi)) {
            if (w) atomicAdd(&ht.w[set][slot], (unsigned long long)w);
            ht.touched[set][slot] = 1;
            return true;
        }
        return false;  // (a different address with this fingerprint: 2^-31)
    }
    // admission: a free way, and none of the set being written right now (it might be this very address, from another wave)
    if (admit && !busy && free_way < 4u) {
        const uint32_t slot = base + free_way;
        if (atomicCAS(ht.tag(set, slot), 0u, 1u) == 0u) {
            ht.lo[set][slot] = lo;
            ht.hi[set][slot] = hi;
            ht.w[set][slot] = w;
            ht.touched[set][slot] = 1;
            __hip_atomic_store(ht.tag(set, slot), fp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (behind the key and the weight)
            return true;
        }
    }
    return false;
}

// One record's update of one sketch (set = 0 SrcAddr, 1 DstAddr): ONE tuple through the bin of the key's partition - slot
// claim, tuple write, and a flush of every bin the wave has filled.  valid = false lanes only take part in the flush
// (which needs the whole wave).  list: >= 16 words of wave-private LDS (the wave's dead tile buffer).
__device__ __forceinline__ void cms_scatter(const KArgs& a, CmsLds& cl, uint32_t* list, uint32_t set, bool valid, uint64_t w, uint64_t h1, uint64_t h2) {
    valid = valid && w != 0;
    const CmsKey k = cms_key(h1, h2, a.cms_wl2);
    const uint32_t p = set * CMS_NPART + k.prefix;
    const uint4 t = make_uint4(k.l1, k.l2, (uint32_t)w, (uint32_t)(w >> 32));
    uint32_t fill = 0xffffffffu;
    if (valid) {
        const uint32_t slot = __hip_atomic_fetch_add(&cl.bin_cnt[p], 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffffu;
        if (slot < CMS_BIN) {
            cl.bins[p * CMS_BIN + slot] = t;
            __hip_atomic_fetch_add(&cl.bin_cnt[p], 0x10000u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (slot == CMS_BIN - 1) fill = p;
        } else {  // the bin is on its way out: single store to the back part of the segment
            const uint32_t ob = atomicAdd(&cl.part_cnt[p], 0x10000u) >> 16;
            if (ob < a.ccapb) {
                a.cseg[(size_t)p * a.cregion + (size_t)blockIdx.x * a.ccapq + (a.ccapq - 1u - ob)] = t;
            } else {
                cms_atomic_tuple(a, p, t);
                atomicSub(&cl.part_cnt[p], 0x10000u);  // (not stored: the back count never wraps into stored tuples)
            }
        }
    }
    cms_bins_flush(a, cl, list, fill);
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
    // (behind the adds: LDS operations of a wave complete in order.  The adds and this read sit in different
    // branches of ONE thread's program, so nothing in the language orders them: the barrier pins the emitted order -
    // the same class of compiler reordering bins_flush once hit)
    asm volatile("" ::: "memory");
    if (valid && win == ln) w += acc[slot];
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
// Hash of the set: the sketch's first hash of the key (cms_hash2) - the ingest kernel has it already.  Slot index =
// its high half, tag = its low 62 bits.
struct KsProbe {                // the home slot as a plain (cached, possibly stale) load saw it
    ulonglong2 c01;             // tag and low key word
    unsigned long long chi;     // high key word
};
__device__ __forceinline__ unsigned long long keyset_tag(uint64_t h1) { return KS_CLAIMED | (h1 & (KS_READY - 1)); }
__device__ __forceinline__ uint32_t keyset_home(const KArgs& a, uint64_t h1) { return (uint32_t)(h1 >> 32) & a.ks_mask; }
// issue the home-slot loads of a key (nothing waits here: the caller does other work before keyset_finish; slot and tag
// are functions of h1 - not kept in registers meanwhile)
__device__ __forceinline__ KsProbe keyset_probe(const KArgs& a, KeySlot* tab, uint64_t h1) {
    KsProbe p;
    const KeySlot* s = &tab[keyset_home(a, h1)];
    p.c01 = *reinterpret_cast<const ulonglong2*>(&s->tag);
    p.chi = s->hi;
    return p;
}
// One step of the probing path: looks at slots i and i + 1 (one 64-byte line when i is even; both plain loads are in
// flight together) and returns true when the key is in the set - found or inserted; otherwise i has moved on.
// Fastest paths, on plain (cached) loads: they may be stale - but stale only ever means OLDER, and a slot never changes
// once READY: a READY slot seen here is final.  So (a) a complete match is always true, and (b) a READY slot that holds
// another tag (a displaced key walks over those: ~12 % of the probes at a quarter load) or the same tag with another
// key is skipped without consulting the memory side.  Only EMPTY or claimed-but-not-ready views go to the memory side
// (system-scope loads and atomics are served past the incoherent per-XCD L2s).
// (pre: the caller already holds a plain view of slot i - the home slot's, from keyset_probe)
__device__ __forceinline__ bool keyset_step(const KArgs& a, KeySlot* tab, unsigned long long lo, unsigned long long hi, unsigned long long mytag, uint32_t& i,
                                            bool pre = false, ulonglong2 pre01 = make_ulonglong2(0, 0), unsigned long long prehi = 0) {
    const uint32_t i1 = (i + 1) & a.ks_mask;
    ulonglong2 c0 = pre01;
    unsigned long long h0 = prehi;
    if (!pre) {
        c0 = *reinterpret_cast<const ulonglong2*>(&tab[i].tag);  // tag, lo
        h0 = tab[i].hi;
    }
    const ulonglong2 c1 = *reinterpret_cast<const ulonglong2*>(&tab[i1].tag);
    const unsigned long long h1 = tab[i1].hi;
    KeySlot* s = &tab[i];
    if (c0.x & KS_READY) {
        if (c0.x == (mytag | KS_READY) && c0.y == lo && h0 == hi) return true;
        if (c1.x & KS_READY) {
            if (c1.x == (mytag | KS_READY) && c1.y == lo && h1 == hi) return true;
            i = (i + 2) & a.ks_mask;
            return false;
        }
        s = &tab[i1];
        i = i1;
    }
    i = (i + 1) & a.ks_mask;  // (where the walk goes on if slot s is not the one)
    // a slot the plain view shows EMPTY: straight to the claim (the CAS answers with the slot's real tag either way - a
    // memory-side load in front of it was one more round trip in the chain every NEW key pays, and a streaming batch has
    // a new key in nearly every tile); a claimed-not-ready view: ask the memory side first
    const unsigned long long seen = s == &tab[i1] && (c0.x & KS_READY) ? c1.x : c0.x;
    unsigned long long t = seen == 0ull ? atomicCAS(&s->tag, 0ull, mytag) : __hip_atomic_load(&s->tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bool won = seen == 0ull && t == 0ull;
    if (!won && t == 0ull) {  // (the plain view showed a claim the memory side does not have)
        t = atomicCAS(&s->tag, 0ull, mytag);
        won = t == 0ull;
    }
    if (won) {  // claimed: publish the key, then mark it readable
        const unsigned long long o1 = atomicExch(&s->lo, lo), o2 = atomicExch(&s->hi, hi);
        if ((o1 & o2) != ~0ull) atomicOr(&s->tag, KS_READY);  // (always true: orders the OR behind both writes)
    }
    // Several lanes of ONE wave may carry the same new key (a heavy hitter's first tile; every instance of a fresh candidate in
    // candidates mode): one of them wins the claim above, the others wait below for its READY bit - and the winner only sets it
    // if its block has run by then.  In the C++ model the two are unrelated threads and round 4's form (early returns out of
    // both branches) left their order to the block layout: in the workgroup-tile kernel the losers came first, spun their 4096
    // rounds against a lane that could not run, gave up and stored the key AGAIN one slot further - 16 copies per heavy key
    // (found by the candidates mode's threshold, which counts the set's rows), and milliseconds of spinning in a kernel's
    // first launch.  The ballot is a convergent operation: the publish block cannot sink below it, the wait cannot rise above.
    const unsigned long long claimers = __builtin_amdgcn_ballot_w64(won);
    asm volatile("" ::"s"(claimers) : "memory");
    if (won) return true;
    if (t == (mytag | KS_READY)) {
        const unsigned long long l = __hip_atomic_load(&s->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long q = __hip_atomic_load(&s->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return l == lo && q == hi;  // (false: equal tag, different key)
    }
    if ((t | KS_READY) != (mytag | KS_READY)) return false;  // somebody else's slot
    // an equal tag whose key is still being written: by a lane of this wave (it has published: above) or of another wave (a
    // few instructions away)
    for (int spin = 0; spin < 4096 && !(t & KS_READY); spin++) t = atomicOr(&s->tag, 0ull);
    if (t & KS_READY) {
        const unsigned long long l = atomicAdd(&s->lo, 0ull), q = atomicAdd(&s->hi, 0ull);  // memory-side reads
        return l == lo && q == hi;
    }
    return false;
}
__device__ __forceinline__ void keyset_insert_slow(const KArgs& a, KeySlot* tab, unsigned long long lo, unsigned long long hi, unsigned long long mytag, uint32_t i) {
    for (int step = 0; step < 256; step++)
        if (keyset_step(a, tab, lo, hi, mytag, i)) return;
    atomicAdd(&a.ctr->ks_overflow, 1u);
}
// the common case - the key sits in its home slot - costs the two loads keyset_probe issued; everything else
// (other slot, first occurrence) takes the probing path
__device__ __forceinline__ bool keyset_at_home(const KsProbe& p, uint64_t h1, unsigned long long lo, unsigned long long hi) {
    return p.c01.x == (keyset_tag(h1) | KS_READY) && p.c01.y == lo && p.chi == hi;
}
__device__ __forceinline__ void keyset_finish(const KArgs& a, KeySlot* tab, const KsProbe& p, uint64_t h1, unsigned long long lo, unsigned long long hi) {
    if (keyset_at_home(p, h1, lo, hi)) return;
    keyset_insert_slow(a, tab, lo, hi, keyset_tag(h1), keyset_home(a, h1));
}
// (Round 4 measured the probing path OUT of the ingest kernel: keys not found in their home slot appended to per-workgroup
// lists and inserted by a kernel behind it, a key per lane.  While the sets fill up the ingest kernel gained a third (994 vs
// 1455 us) and the list kernel cost 863 us; in the steady state of the 1 B-record stream both forms take 1.16 ms per launch
// and the lists add launches of 1.63 ms.  profiles/r04_exp_config3_400M_defer{0,1}.json; the lists are gone.)
// Both addresses of a record (the ingest kernel's form): ONE probing loop for what is left of the two sets.  A wave almost
// always has a lane whose key is not in its home slot (12 % of the keys at a quarter load), every step of the probing
// path is a round trip to memory with the whole wave waiting, and two loops in a row - source set, then destination set
// - were two such chains per tile; a lane with both takes them one after the other, different lanes side by side.
__device__ __forceinline__ void keyset_finish2(const KArgs& a, bool vs, const KsProbe& ps, uint64_t sh1, unsigned long long slo, unsigned long long shi, bool vd,
                                               const KsProbe& pd, uint64_t dh1, unsigned long long dlo, unsigned long long dhi) {
    const bool ns = vs && !keyset_at_home(ps, sh1, slo, shi);
    bool nd = vd && !keyset_at_home(pd, dh1, dlo, dhi);
    if (__builtin_amdgcn_ballot_w64(ns || nd) == 0ull) return;
    KeySlot* tab = ns ? a.ks_src : a.ks_dst;
    unsigned long long lo = ns ? slo : dlo, hi = ns ? shi : dhi, mytag = keyset_tag(ns ? sh1 : dh1);
    uint32_t i = keyset_home(a, ns ? sh1 : dh1), steps = 0;
    ulonglong2 pre01 = ns ? ps.c01 : pd.c01;  // the home slot as keyset_probe saw it: the first step does not load it again
    unsigned long long prehi = ns ? ps.chi : pd.chi;
    bool active = ns || nd, pre = true;
    nd = nd && ns;  // (from here on: the destination address still waits behind the source address)
    while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
        if (active) {
            bool done = keyset_step(a, tab, lo, hi, mytag, i, pre, pre01, prehi);
            pre = false;
            if (!done && ++steps >= 256u) {
                atomicAdd(&a.ctr->ks_overflow, 1u);
                done = true;
            }
            if (done) {
                active = nd;
                if (nd) {
                    tab = a.ks_dst;
                    lo = dlo;
                    hi = dhi;
                    mytag = keyset_tag(dh1);
                    i = keyset_home(a, dh1);
                    pre01 = pd.c01;
                    prehi = pd.chi;
                    pre = true;
                    steps = 0;
                    nd = false;
                }
            }
        }
    }
}
// ---- candidates mode: is this address a candidate? ---------------------------------------------------------------------------
// bits: depth rows of 2^wl2 bits, bit (r, column) = "counter (r, column) was >= the threshold at the last launch boundary".
// (flat counter index = row << wl2 | column; bit i lives in word i >> 5)
__device__ __forceinline__ uint32_t cand_word(const uint32_t* bits, uint32_t wl2, uint32_t row, uint32_t col) { return bits[(((size_t)row << wl2) + col) >> 5]; }
// (the row-0 word is loaded by the caller ahead of time: cand_word(bits, wl2, 0, cms_column(k, 0, wl2)))
__device__ __forceinline__ bool cand_pass(const uint32_t* bits, uint32_t depth, uint32_t wl2, const CmsKey& k, uint32_t word0) {
    if (!((word0 >> (cms_column(k, 0, wl2) & 31u)) & 1u)) return false;  // (all but ~0.1 % of the addresses stop here)
    for (uint32_t r = 1; r < depth; r++) {
        const uint32_t col = cms_column(k, r, wl2);
        if (!((cand_word(bits, wl2, r, col) >> ((((size_t)r << wl2) + col) & 31u)) & 1u)) return false;
    }
    return true;
}
__device__ __forceinline__ void keyset_insert_h(const KArgs& a, KeySlot* tab, unsigned long long lo, unsigned long long hi, uint64_t h1) {
    const KsProbe p = keyset_probe(a, tab, h1);
    keyset_finish(a, tab, p, h1, lo, hi);
}
// an address the ingest paths saw (set = 0 SrcAddr, 1 DstAddr): kept - always in the exact mode, when it passes the
// candidate test in candidates mode.  (fa_topk_merge_keys inserts unconditionally: keyset_insert.)
__device__ __forceinline__ void keyset_offer(const KArgs& a, uint32_t set, const uint32_t key[4]) {
    const uint64_t lo = (uint64_t)key[1] << 32 | key[0], hi = (uint64_t)key[3] << 32 | key[2];
    uint64_t h1, h2;
    cms_hash2(lo, hi, a.cms_seed, h1, h2);
    const uint32_t* bits = set ? a.cand_dst : a.cand_src;
    if (bits) {
        const CmsKey k = cms_key(h1, h2, a.cms_wl2);
        if (!cand_pass(bits, a.cms_depth, a.cms_wl2, k, cand_word(bits, a.cms_wl2, 0, cms_column(k, 0, a.cms_wl2)))) return;
    }
    keyset_insert_h(a, set ? a.ks_dst : a.ks_src, lo, hi, h1);
}
__device__ __forceinline__ void keyset_insert(const KArgs& a, KeySlot* tab, const uint32_t key[4]) {
    const unsigned long long lo = (unsigned long long)key[1] << 32 | key[0], hi = (unsigned long long)key[3] << 32 | key[2];
    uint64_t h1, h2;
    cms_hash2(lo, hi, a.cms_seed, h1, h2);
    const KsProbe p = keyset_probe(a, tab, h1);
    keyset_finish(a, tab, p, h1, lo, hi);
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
    uint32_t i = as_home(k0, k1, h, a.mask, a.rlog2);
    for (int probe = 0; probe < FA_MAX_PROBES; probe++, i = as_next(i, a.mask, a.rlog2)) {
        Slot* s = &a.tab[i];
        const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(s);  // one 16-byte load: k0,k1
        unsigned long long c0 = kk.x, c1 = kk.y;
        if (c0 == k0 && c1 == k1) return s;  // common case: no atomics on the key words
        if (c0 == 0) c0 = atomicCAS(&s->k0, 0ull, (unsigned long long)k0);
        if (c0 != 0 && c0 != k0) continue;
        if (c1 == 0) {
            c1 = atomicCAS(&s->k1, 0ull, (unsigned long long)k1);
            if (c1 == 0) count_created(&a.ctr->used);  // this lane created the group
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
__device__ __forceinline__ void wide_sink_wave(const KArgs& a, LdsMinutes& lm, const Rec& r, bool sure, uint32_t tb, uint32_t tb_base = 0,
                                               uint32_t* wpart_cnt = nullptr) {
    const WArgs t = wargs(a);
    const uint64_t wgt = r.bytes * r.sampling_rate;
    if (ks_on<KEYSETS>(a, FA_KEYS_ADDR_PORT_PROTO)) {
        WSlot* sp = nullptr;
        bool logged = false;  // this lane's update left as a tuple in a segment
        if (sure) {
            WKey k;
            app_key(a, r, tb, k);
            const uint32_t h = wkey_hash(k);
            bool done = false;
            // wave-tile kernel: the update leaves as one 32-byte tuple into this workgroup's segment of the key's table
            // region (one sector store; wagg_kernel folds the region's tuples without atomics on the sums)
            if (wpart_cnt && a.wseg && wtup_fits(tb - tb_base, r.packets, r.dst_port, r.proto)) {
                const uint32_t part = (h & t.mask) >> __builtin_popcount(t.rmask);  // region of the home slot
                const uint32_t pos = atomicAdd(&wpart_cnt[part], 1u);
                if (pos < a.wcapq) {
                    uint4 q0, q1;
                    wtup_pack(r.src, tb - tb_base, r.bytes, r.packets, r.dst_port, r.proto, q0, q1);
                    uint4* dst = a.wseg + 2u * ((size_t)part * a.wregion + (size_t)blockIdx.x * a.wcapq + pos);
                    dst[0] = q0;
                    dst[1] = q1;
                    done = true;
                }
            }
            logged = done;
            if (!done) {
                sp = wtable_find_or_claim(t, k, h);
                if (!sp) wspill_park(t, k, r.bytes, r.packets, 1);
            }
        }
#ifndef FA_NO_WRANGE  // (A/B of the bookkeeping's cost only: reads of log chunks need it)
        if (wpart_cnt) {
            // the bucket range of the tuples this workgroup leaves in the segments (two LDS words behind the region counts): a close
            // compares its range with the chunk's and knows when nothing of a chunk is left - without a scan of the chunk.  A tile's
            // records almost always share one bucket: one lane's LDS atomics then.
            const unsigned long long dm = __builtin_amdgcn_ballot_w64(logged);
            if (dm != 0ull) {
                const uint32_t first = (uint32_t)__builtin_ctzll(dm);
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)tb, (int)first);
                uint32_t lo = t0, hi = t0;
                if (__builtin_amdgcn_ballot_w64(logged && tb != t0) != 0ull) {
                    lo = logged ? tb : 0xffffffffu;
                    hi = logged ? tb : 0u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
                        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
                    }
                }
                if (__lane_id() == first) {
                    uint32_t* wrange = wpart_cnt + (1u << WIDE_PLOG2_MAX);
                    atomicMin(&wrange[0], lo);
                    atomicMin(&wrange[1], ~hi);
                }
            }
        }
#endif
        if (__builtin_amdgcn_ballot_w64(sp != nullptr) != 0ull) quad_atomic_update_at<4>((uint64_t)sp, r.bytes, r.packets, 1);
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
// g-th filled bin, each lane copies 16 bytes (one wide tuple or two compact ones) - one store instruction writes 8
// complete lines, no partial lines and no workgroup barrier.  The producers of a bin's other slots may sit in other
// waves: the high half of the bin word counts the slots WRITTEN, and nobody can take a slot of a full bin, so the
// spin below only ever waits for straight-line code of waves that never wait for us (producers of this wave
// finished in lockstep inside lane_work).  fill_part: the bin this lane filled (or ~0); scratch: 32 bytes of
// wave-private LDS.  Must be called by the full wave.
template <bool T8, uint32_t BL>
__device__ __forceinline__ void bins_flush(const KArgs& a, uint4* bins, uint32_t* bin_cnt, uint32_t* part_cnt, uint32_t* scratch,
                                           uint32_t fill_part, uint32_t tb_base, uint32_t& n_direct) {
    constexpr uint32_t TB = bin_cap<T8, BL>();
    const unsigned long long fm = __builtin_amdgcn_ballot_w64(fill_part != 0xffffffffu);
    if (fm == 0ull) return;
    constexpr uint32_t LPB = BL, GROUPS = 64u / LPB;  // lanes per bin (16 bytes each), bins per pass
    const uint32_t ln = __lane_id(), g = ln / LPB, sub = ln % LPB;
    const uint32_t rank = (uint32_t)__builtin_popcountll(fm & ((1ull << ln) - 1ull));
    const uint32_t todo = (uint32_t)__builtin_popcountll(fm);
    for (uint32_t base = 0; base < todo; base += GROUPS) {
        if (fill_part != 0xffffffffu && rank - base < GROUPS) scratch[rank - base] = fill_part;
        const bool act = g < min(GROUPS, todo - base);
        const uint32_t fp = act ? scratch[g] : 0u;
        // the written-slot check, the tuple read and the line allocation are issued back to back (LDS operations
        // of a wave complete in order, so the read sees what the check saw); only a bin that is still being
        // written costs further round trips
        // (acquire / release pair with the producers' written-slot count: without it the COMPILER may move the
        // tuple read above the check - it did, and rows differed from the oracle at 16 M records)
        const uint32_t c0 = __hip_atomic_load(&bin_cnt[fp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint4 tq = bins[fp * BL + sub];
        uint32_t line = 0;
        if (act && sub == 0) line = atomicAdd(&part_cnt[fp], 1u) & 0xffffu;  // low half: lines at the front
        bool late = false;
        if (__builtin_amdgcn_ballot_w64(act && (c0 >> 16) < TB) != 0ull) {
            late = true;
            while (__builtin_amdgcn_ballot_w64(act && (__hip_atomic_load(&bin_cnt[fp], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >> 16) < TB) != 0ull) {}
        }
        line = (uint32_t)__shfl((int)line, (int)(ln - sub));
        if (act) {
            const uint4 tv = late ? bins[fp * BL + sub] : tq;
            if ((line + 1u) * TB <= a.capf) {
                // (compact tuples: region and capq are even, so the segment starts on a uint4 boundary)
                const size_t seg0 = ((size_t)fp * a.region + (size_t)blockIdx.x * a.capq) >> (T8 ? 1 : 0);
                // (plain stores: the L2 merges the half lines of a bin's two flushes and the back parts' single tuples before
                // they go to HBM - streaming (nt) stores measured 6-11 % slower on BASELINE config 2, round 3)
                // (FA_TL_WINDOW: uint4s of the window, measurement builds only.  256 = 4 KiB: every partition's lines land on the same 2 KiB.
                // Larger windows spread the partitions (128 uint4 apart = a launch's lines of one partition): 1024 / 4096 / 16384 / 32768 per
                // workgroup = 8 / 32 / 128 / 256 MB over 512 workgroups - inside the L2s, their sum, the Infinity Cache, the real footprint)
                if (FA_DBG(a, DBG_TUPLE_LOCAL)) a.seg[(size_t)blockIdx.x * FA_TL_WINDOW + ((fp * (FA_TL_WINDOW > 256u ? 128u : 0u) + line * BL + sub) & (FA_TL_WINDOW - 1u))] = tv;
                else if (!(FA_DBG(a, DBG_NO_TUPLE_STORE))) a.seg[seg0 + line * BL + sub] = tv;
            } else {  // front part full (skewed batch): straight to the device-wide table
                TupleVals v[2];
                if (T8) {
                    t8_unpack(make_uint2(tv.x, tv.y), fp, tb_base, v[0]);
                    t8_unpack(make_uint2(tv.z, tv.w), fp, tb_base, v[1]);
                } else {
                    tup16_unpack(tv, v[0]);
                }
#pragma unroll
                for (int e = 0; e < (T8 ? 2 : 1); e++) {
                    uint64_t q0, q1;
                    pack_key(tb_base + v[e].tbr, v[e].src_as, v[e].dst_as, v[e].etype, q0, q1);
                    agg_global(cold_args(), q0, q1, key_hash(q0, q1), v[e].bytes, v[e].packets, 1);
                    n_direct++;
                }
                if (sub == 0) atomicSub(&part_cnt[fp], 1u);  // (the line was not stored: the 16-bit counter stays <= its cap)
            }
            // (release: behind the tuple reads above)
            if (sub == 0) __hip_atomic_store(&bin_cnt[fp], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

}  // namespace fa
