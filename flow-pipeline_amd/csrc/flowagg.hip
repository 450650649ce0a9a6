// flowagg.hip - C-ABI implementation of libflowagg (include/flowagg.h).
//
// Host side of the MI355X flow-aggregation stage: owns the HIP stream, the
// device group-by table, sketches, pinned staging and the SoA projection
// buffers; launches the gfx950 kernels of kernels.cuh.  Mirrors the shape of
// the reference sink (inserter/inserter.go:90-165: buffer -> flush) with the
// ClickHouse semantics of compose/clickhouse/create.sh:5-110.
// There is deliberately no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/flowagg.h"
#include "kernels.cuh"

using namespace fa;

static thread_local std::string g_create_error;

struct fa_ctx {
    fa_config cfg{};
    uint32_t gran = 300;
    hipStream_t stream = nullptr;
    // three events per ingest launch: before / after the tile kernel, after the aggregation kernel
    struct LaunchEvents { hipEvent_t e0, e1, e2; };
    std::vector<LaunchEvents> ev_pool;
    size_t ev_used = 0;
    std::vector<LaunchEvents> dev_pool;  // ... of fa_decode_device launches
    size_t dev_used = 0;

    Slot* tab = nullptr;
    uint32_t cap_log2 = 20;
    SpillEntry* spill = nullptr;
    // Parked updates that met a full table.  A record parks at most one update, so 2 x max_batch_records (+ slack for
    // the per-workgroup flushes) covers everything the host may have in flight before it looks at the counters again
    // (pre_launch_guard): aggregates are never dropped, the table grows and the parked updates are replayed.
    uint32_t spill_cap = 0;
    Counters* d_ctr = nullptr;
    Counters* h_ctr = nullptr;  // pinned
    // counter snapshots, one per ingest launch (ring): what the device had counted when that launch finished
    static constexpr int NSNAP = 8;
    Counters* h_snap = nullptr;  // pinned, NSNAP entries
    hipEvent_t snap_ev[NSNAP] = {};
    uint64_t snap_records[NSNAP] = {};  // records launched up to and including the snapshot's batch
    uint64_t snap_seq[NSNAP] = {};      // launch sequence number (0 = unused)
    uint64_t launch_seq = 0, known_seq = 0;
    uint64_t launched_records = 0, known_records = 0;
    Counters known{};                   // newest snapshot (or settle) the host has seen
    uint32_t* d_exotic = nullptr;  // deferral lists: [0,cap) exotic, [cap,2cap) retry
    size_t exotic_cap = 0;
    // scatter sink
    uint4* seg = nullptr;
    size_t seg_bytes = 0;
    uint32_t* seg_counts = nullptr;
    size_t seg_counts_cap = 0;
    // Count-Min scatter sink (sinks.cuh): sketch tuples' segments
    uint4* cseg = nullptr;
    size_t cseg_bytes = 0;
    uint32_t* cseg_counts = nullptr;
    size_t cseg_counts_cap = 0;
    HotSeed* hot_seed = nullptr;   // [hot_seed_wgs][CMS_SETS][HOT_SLOTS] entries of the hot-address caches that survive a launch
    uint32_t* hot_seed_tag = nullptr;
    uint32_t hot_seed_wgs = 0, hot_epoch = 0;
    uint32_t cms_par = 0;          // parity of the next cms_agg_kernel launch (its size copies and unit counters)
    uint32_t* cms_psize = nullptr;  // [2][CMS_SETS * CMS_NPART] tuples per sketch partition, last launch / this launch (cms_agg_kernel: heaviest first)
    // scatter sink of the (SrcAddr,DstPort,Proto) key set (wagg.cuh)
    uint4* wseg = nullptr;
    size_t wseg_bytes = 0;
    uint32_t* wseg_counts = nullptr;
    size_t wseg_counts_cap = 0;
    int wide_mode = 0;         // env FA_WIDE: 0 adaptive, 1 "atomic" (every update through memory-side atomics), 2 "scatter" (always the scatter sink),
                               // 3 "log": scatter, and the launch's tuples stay in their segments (wlog below) instead of being folded at once
    // ---- wide log (FA_WIDE=log) ----
    // A (SrcAddr,DstPort,Proto) stream opens a row for nearly every record: folding a launch's tuples into the hash table is
    // two random HBM accesses per record that aggregate nothing - and the window close sorts the rows anyway.  In log mode a
    // launch's segment buffers are simply KEPT (a chunk); reads take the table's rows AND the chunks' tuples through the
    // same sort + segmented sums; a close moves the chunks' watermark; only when more than wlog_max chunks are pending (or
    // the table is rebuilt) the oldest is folded into the table after all (wagg_kernel, or the atomic replay when the
    // table's geometry has changed since).
    struct WChunk {
        uint4* seg = nullptr;
        size_t seg_bytes = 0;
        uint32_t* counts = nullptr;   // [nparts][nwg] (+ 4 words: [counts_cap] = the launch's time base)
        size_t counts_cap = 0;
        uint32_t nwg = 0, wcapq = 0, wplog2 = 0, wmask = 0, wm = 0;  // wm: buckets below it were dropped after this launch
        size_t wregion = 0;
        uint64_t n = 0;               // records of the launch (upper bound of its tuples)
        uint32_t minb = 0, maxb = 0;  // smallest / largest bucket among the chunk's live tuples when they were looked for (wlog_drop; valid: minb_known)
        bool minb_known = false;
    };
    std::vector<WChunk> wlog, wlog_free;
    bool wide_probed = false;      // the ctx has seen its first launch (ingest_device_records: a first big launch is probed with its first 2^20 records)
    bool wide_defer = false;       // adaptive (FA_WIDE unset): log mode from the moment more than half of a million records opened new rows - for
                                   // the rest of the ctx's life (deferred launches tell nothing about new rows; a stream that stops opening
                                   // rows folds its chunks through wagg_kernel once more than wlog_max are pending: the scatter sink's cost)
    bool wlog_now = false;         // the launch being prepared runs in log mode
    uint64_t seen_wfold = 0;       // Counters::wfold_n at the last feedback look
    size_t wlog_max = 8;
    // Upper bound of the wide table's rows at any moment (wide_rows_bound): what the newest counter snapshot the host has
    // seen counted, plus every row that whatever was queued BEHIND that snapshot could still open - ingest launches
    // (a record opens at most wide_per_record rows, through the scatter sink's fold or the atomic paths) and folds of log
    // chunks (one row per tuple).  wpot_total only grows; every snapshot remembers its value (snap_wpot).
    uint64_t wpot_total = 0, known_wpot = 0;
    uint64_t snap_wpot[NSNAP] = {};
    uint32_t wseg_budget = 0;      // env FA_WSEG_BUDGET (tests only): segment buffers the ctx may hold at a time - the next allocation "fails"
    uint32_t late_below = 0;       // time buckets below it belong to flows_5m windows that were closed: records that still arrive for them are counted (stats.records_late)
    uint64_t wlog_recorded = 0, wlog_folded = 0, wlog_replayed = 0, wlog_dropped = 0, wlog_wm_moves = 0, wlog_nomem_folds = 0;  // chunks (fa_stats, FA_VERBOSE)
    bool wide_scatter = true;  // adaptive: the scatter sink while a good share of the records open new rows (7 atomics each
                               // on the atomic path); a stream that mostly hits existing rows (one atomic line transaction
                               // each) is cheaper without the detour through the segments
    uint64_t seen_wused = 0, seen_ok_w = 0;
    bool cms_atomic = false;  // env FA_CMS=atomic (A/B, tests): every sketch update through memory-side atomics
    bool cms_scatter_ok = false;  // the sketch geometry fits the scatter sink (256 partitions of <= 2^14 counters)
    int sink_mode = 0;  // 0 auto, 1 direct, 2 scatter (env FA_SINK)
    bool use_wave_tiles = false;  // decision for the batch being launched
    bool use_t8 = false;          // ... compact 8-byte tuples (table.cuh) for it
    // tuple format feedback: compact tuples while (almost) every record fits them.  A launch whose misfits (records
    // that only a wide tuple holds - they took the direct path) exceed 1/16 of its records switches the ctx to wide
    // tuples for the next 64 launches, then compact is tried again.  Only speed depends on this, never results.
    int t8_mode = 0;              // env FA_TUPLE: 0 adaptive, 1 always compact ("8"), 2 always wide ("16")
    uint64_t t8_wide_until = 0;   // batches counter value up to which wide tuples are used
    uint64_t seen_misfit8 = 0, seen_ok = 0;  // counter values at the last look
    uint64_t seen_agg_groups = 0, seen_agg_launches = 0;
    uint32_t agg_passes_forced = 0;  // env FA_AGG_PASSES (tests, A/B)
    uint32_t agg_passes = 1;      // agg8_kernel passes for the next launch (1, 2, 4, 8): groups per launch / (partitions x passes) <= half the LDS table
    unsigned stage_threads = 8;   // host threads of the staging copy (fa_ingest)
    bool agg_generic = false;     // env FA_AGG=generic (A/B): compact tuples through the two-word-key aggregation kernel
    uint32_t par = 0;             // parity of the next launch (Counters::exotic_count / retry_count copies)
    uint32_t seg_cap_limit = 0;   // env FA_SEG_CAP (tests only): upper bound on tuples per segment
    uint32_t last_nwg = 0;        // workgroups of the last scatter-sink launch (FA_VERBOSE: reads its segment counts back)
    // ingest kernel (env FA_TILE=wave|wg, measurement / tests): wave-private tiles + LDS tuple bins is the
    // production kernel of the scatter sink (never slower than the 256-thread workgroup-tile kernel on the
    // workloads measured, 15-20 % faster when most records leave as tuples); the workgroup kernel serves the
    // decode path, the direct sink (small batches) and key sets without the flows_5m rollup.
    int tile_mode = 0;  // 0 default (wave), 1 wave, 2 workgroup
    uint32_t plog2 = PART_LOG2_MAX, wgpc_cap = 0;  // experiment knobs (env FA_PLOG2, FA_WGPC)

    // host-fed path: pinned staging (double buffered) + device input
    uint8_t* h_stage[2] = {nullptr, nullptr};
    size_t h_stage_cap[2] = {0, 0};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_cur = 0;
    uint8_t* d_in[2] = {nullptr, nullptr};
    size_t d_in_cap[2] = {0, 0};

    // SoA projection
    void* col_block = nullptr;
    size_t col_cap = 0;
    ColumnPtrs cols{};

    // window close (rows_host.inc): rows collected out of the device state, merge scratch, merged / ordered rows
    void* rc_buf = nullptr;          // collected rows (public format, unsorted)
    size_t rc_cap = 0;
    void* rw_buf = nullptr;          // port / minute rows (wide-table rows + the dense histogram's entries)
    size_t rw_cap = 0;
    void* m_scratch = nullptr;       // sort scratch: 2 key arrays, 4 index arrays, hipcub temporary storage
    size_t m_scratch_cap = 0;
    void* m_out[2] = {nullptr, nullptr};  // merged rows; rows in emit order
    size_t m_out_cap[2] = {0, 0};
    void* fs_scratch = nullptr;      // device-side framing (framing.cuh): block starts (two copies), counts, bases, error flags, counters
    size_t fs_scratch_cap = 0;
    void* fs_off = nullptr;          // ... the offsets it produces
    size_t fs_off_cap = 0;
    void* wl_scratch = nullptr;      // window reads of log chunks: per-segment counts, their scan, hipcub storage
    size_t wl_scratch_cap = 0;
    void* part_buf = nullptr;        // fa_rows_partition_device: the rows grouped by destination rank
    size_t part_cap = 0;
    unsigned int* part_cnt = nullptr;  // [3][RPART_MAX_WORLD]: counts, starts, cursors
    void* h_rows = nullptr;          // pinned: rows on their way to the caller
    size_t h_rows_cap = 0;
    hipEvent_t copy_ev[2] = {nullptr, nullptr};  // the two halves of h_rows while a large result leaves in pieces (rows_host.inc)

    unsigned long long* cms_src = nullptr;
    unsigned long long* cms_dst = nullptr;
    size_t cms_words = 0;   // words of ONE copy; the buffers hold CMS_REPLICAS copies
    bool cms_dirty = false;  // copies > 0 may hold counts (cms_fold)
    // merged view (window close across GPUs): all-rank sums of the sketches, filled by fa_merge_allreduce or by the
    // caller's collective (fa_device_state_get + fa_merged_view_set); stale as soon as this ctx ingests again
    unsigned long long* cms_src_m = nullptr;
    unsigned long long* cms_dst_m = nullptr;
    bool merged_valid = false;
    KeySlot* ks_src = nullptr;  // distinct-address sets (fa_topk)
    KeySlot* ks_dst = nullptr;
    uint32_t ks_log2 = 20;

    // wide key sets (wide.cuh): one table + the dense port histograms
    WSlot* wtab = nullptr;
    uint32_t wcap_log2 = 20;
    WSpillEntry* wspill = nullptr;
    uint32_t wspill_cap = 0;      // (updates a record can park in the wide table) x 2 x max_batch_records + slack
    uint32_t wide_per_record = 0;  // wide-table updates one record can cause (enabled wide key sets)
    uint64_t wused_base = 0;
    uint64_t wide_dead = 0;       // slots of the wide table whose rows a window close zeroed (wdrop_kernel): occupied, but no rows - purged by the next rebuild
    ulonglong2* port_hist = nullptr;  // [2][PORT_DENSE]

    fa_stats_t stats{};
    uint64_t used_base = 0;  // groups created before the current counter epoch
    std::string err;
    int sticky = 0;  // sticky error from async work
    int num_cus = 256;
    uint32_t dbg = 0;
};

#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t e__ = (expr);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                    \
            return FA_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

// Every entry point runs on the ctx's device whatever device the calling thread had current (several ctxs on
// different GPUs in one process; Go moves goroutines between OS threads): allocations and launches must not land on
// the caller's device.
#define FA_ON_DEVICE(c)                                \
    do {                                               \
        if (c) (void)hipSetDevice((c)->cfg.device);    \
    } while (0)

static uint32_t log2_ceil(uint64_t v);
static int fail(fa_ctx* c, int code, const char* msg) {
    if (c) c->err = msg;
    return code;
}

extern "C" uint32_t fa_abi_version(void) { return FA_ABI_VERSION; }

extern "C" const char* fa_last_error(const fa_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static KArgs make_args(fa_ctx* c) {
    KArgs a{};
    a.framed = c->cfg.framed ? 1u : 0u;
    a.gran = c->gran;
    a.tab = c->tab;
    a.mask = (1u << c->cap_log2) - 1;
    a.rlog2 = as_rlog2(c->cap_log2);
    a.spill = c->spill;
    a.spill_cap = c->spill_cap;
    a.ctr = c->d_ctr;
    a.exotic_idx = c->d_exotic;
    a.cms_src = c->cms_src;
    a.cms_dst = c->cms_dst;
    a.cms_depth = c->cfg.cms_depth;
    a.cms_wl2 = c->cfg.cms_width_log2;
    a.cms_seed = c->cfg.cms_seed;
    a.ks_src = c->ks_src;
    a.ks_dst = c->ks_dst;
    a.ks_mask = (1u << c->ks_log2) - 1;
    a.cols = c->cols;
    a.dbg = c->dbg;
    a.tile_recs = BLOCK;
    a.retry_idx = c->d_exotic ? c->d_exotic + c->exotic_cap : nullptr;
    a.key_sets = c->cfg.key_sets;
    a.wtab = c->wtab;
    a.wmask = (1u << c->wcap_log2) - 1;
    a.wplog2 = wide_plog2(c->wcap_log2);
    a.wspill = c->wspill;
    a.wspill_cap = c->wspill_cap;
    a.port_hist = c->port_hist;
    a.gran_recip = (1.0 / (double)c->gran) * (1.0 + 1.0 / 1099511627776.0);
    a.par = c->par;
    a.agg_passes = c->agg_passes_forced ? c->agg_passes_forced : c->agg_passes;
    a.late_below = c->late_below;
    return a;
}

// Persistent grid: exactly the number of workgroups that are co-resident
// (CUs x LDS/VGPR-limited workgroups per CU); tiles are grid-strided.
template <class K>
static int grid_for(fa_ctx* c, K kernel, uint32_t n, uint32_t tile_recs) {
    uint32_t tiles = (n + tile_recs - 1) / tile_recs;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0) != hipSuccess || per_cu < 1)
        per_cu = 2;
    if (c->wgpc_cap && (int)c->wgpc_cap < per_cu) per_cu = (int)c->wgpc_cap;
    uint32_t g = std::min<uint32_t>((uint32_t)c->num_cus * (uint32_t)per_cu, AGG_MAX_NWG);
    return (int)std::max(1u, std::min(tiles, g));
}

// Records per LDS tile: as many as fit one tile buffer at the batch's mean record
// size (one record per lane, at most BLOCK).  Tiles that still overflow (outliers)
// take the multi-pass path inside the kernel.
static uint32_t tile_recs_for(size_t len, size_t n) {
    if (n == 0) return BLOCK;
    double avg = (double)len / (double)n + 0.5;
    double r = ((double)TILE_BYTES - 15.0) / avg;
    if (r >= (double)BLOCK) return BLOCK;
    if (r < 1.0) return 1;
    return (uint32_t)r;
}

extern "C" int fa_create(const fa_config* cfg_in, fa_ctx** out) {
    if (!cfg_in || !out) {
        g_create_error = "fa_create: null argument";
        return FA_ERR_ARG;
    }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "fa_create: no HIP device (libflowagg has no CPU fallback)";
        return FA_ERR_NO_DEVICE;
    }
    fa_config cfg = *cfg_in;
    if (cfg.window_secs == 0) cfg.window_secs = 300;
    if (cfg.table_capacity_log2 == 0) cfg.table_capacity_log2 = 20;
    if (cfg.cms_depth == 0) cfg.cms_depth = 4;
    if (cfg.cms_width_log2 == 0) cfg.cms_width_log2 = 20;
    if (cfg.key_sets == 0) cfg.key_sets = FA_KEYS_AS_PAIR;
    if (cfg.topk_capacity_log2 == 0) cfg.topk_capacity_log2 = 20;
    if (cfg.wide_capacity_log2 == 0) cfg.wide_capacity_log2 = 20;
    if (cfg.max_batch_records == 0) cfg.max_batch_records = AGG_MAX_BATCH;
    if (cfg.max_batch_records > AGG8_MAX_BATCH) cfg.max_batch_records = AGG8_MAX_BATCH;  // (launches of wide tuples are split at 2^24)
    uint32_t gran = cfg.subwindow_secs ? cfg.subwindow_secs : cfg.window_secs;
    if (cfg.device < 0 || cfg.device >= ndev || gran < 60 || 86400 % gran != 0 ||
        cfg.window_secs % gran != 0 || 86400 % cfg.window_secs != 0 || cfg.table_capacity_log2 < 10 ||
        cfg.table_capacity_log2 > 30 || cfg.cms_depth > 16 || cfg.cms_width_log2 < 4 ||
        cfg.cms_width_log2 > 28 || (cfg.key_sets & ~63u) || cfg.topk_capacity_log2 < 8 || cfg.topk_capacity_log2 > 30 ||
        cfg.wide_capacity_log2 < 8 || cfg.wide_capacity_log2 > 30) {
        g_create_error = "fa_create: invalid configuration";
        return FA_ERR_ARG;
    }
    fa_ctx* c = new fa_ctx();
    c->cfg = cfg;
    c->gran = gran;
    c->cap_log2 = cfg.table_capacity_log2;
    if (const char* d = getenv("FA_DEBUG_FLAGS")) c->dbg = (uint32_t)strtoul(d, nullptr, 0);
    if (const char* d = getenv("FA_PLOG2")) c->plog2 = std::min<uint32_t>(PART_LOG2_MAX, std::max<uint32_t>(4, (uint32_t)atoi(d)));
    if (const char* d = getenv("FA_WGPC")) c->wgpc_cap = (uint32_t)atoi(d);
    if (const char* d = getenv("FA_SEG_CAP")) c->seg_cap_limit = std::max<uint32_t>(40u, (uint32_t)atoi(d) & ~7u);
    if (const char* d = getenv("FA_TILE")) c->tile_mode = !strcmp(d, "wave") ? 1 : !strcmp(d, "wg") ? 2 : 0;
    if (const char* d = getenv("FA_SINK")) c->sink_mode = !strcmp(d, "direct") ? 1 : !strcmp(d, "scatter") ? 2 : 0;
    if (const char* d = getenv("FA_AGG")) c->agg_generic = !strcmp(d, "generic");
    if (const char* d = getenv("FA_CMS")) c->cms_atomic = !strcmp(d, "atomic");
    if (const char* d = getenv("FA_WIDE")) c->wide_mode = !strcmp(d, "atomic") ? 1 : !strcmp(d, "scatter") ? 2 : !strcmp(d, "log") ? 3 : 0;
    if (const char* d = getenv("FA_WIDE_LOG_CHUNKS")) c->wlog_max = (size_t)std::max(0, atoi(d));
    if (const char* d = getenv("FA_WSEG_BUDGET")) c->wseg_budget = (uint32_t)std::max(1, atoi(d));
    if (const char* d = getenv("FA_AGG_PASSES")) {
        const int v = atoi(d);
        c->agg_passes_forced = (v == 1 || v == 2 || v == 4 || v == 8) ? (uint32_t)v : 0u;
    }
    c->cms_scatter_ok = cms_scatterable(cfg.cms_depth, cfg.cms_width_log2);
    if (const char* d = getenv("FA_STAGE_THREADS")) c->stage_threads = (unsigned)std::min(64, std::max(1, atoi(d)));
    c->stage_threads = std::min(c->stage_threads, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* d = getenv("FA_TUPLE")) c->t8_mode = !strcmp(d, "8") ? 1 : !strcmp(d, "16") ? 2 : 0;
    auto bail = [&](const char* what, hipError_t e) {
        g_create_error = std::string("fa_create: ") + what + ": " + hipGetErrorString(e);
        fa_destroy(c);
        return e == hipErrorOutOfMemory ? FA_ERR_NOMEM : FA_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(cfg.device)) != hipSuccess) return bail("hipSetDevice", e);
    {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg.device) == hipSuccess && v > 0)
            c->num_cus = v;
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    for (int i = 0; i < 2; i++)
        if ((e = hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&c->copy_ev[i], hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    for (int i = 0; i < fa_ctx::NSNAP; i++)
        if ((e = hipEventCreateWithFlags(&c->snap_ev[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipHostMalloc(&c->h_snap, sizeof(Counters) * fa_ctx::NSNAP)) != hipSuccess) return bail("hipHostMalloc", e);
    c->spill_cap = 2u * cfg.max_batch_records + (1u << 21);
    c->wide_per_record = ((cfg.key_sets & FA_KEYS_ADDR_PORT_PROTO) ? 1u : 0u) + ((cfg.key_sets & FA_KEYS_PORT_HIST) ? 2u : 0u) +
                         ((cfg.key_sets & FA_KEYS_MINUTE_SERIES) ? 1u : 0u);
    // (capped at 2^25 entries = 1.9 GB: launches that could park more are split, fa_ingest_device)
    c->wspill_cap = (uint32_t)std::min<uint64_t>((uint64_t)c->wide_per_record * 2u * cfg.max_batch_records + (1u << 21), 1u << 25);
    size_t tab_bytes = sizeof(Slot) << c->cap_log2;
    if ((e = hipMalloc(&c->tab, tab_bytes)) != hipSuccess) return bail("hipMalloc(table)", e);
    if ((e = hipMemsetAsync(c->tab, 0, tab_bytes, c->stream)) != hipSuccess) return bail("memset", e);
    if ((e = hipMalloc(&c->spill, sizeof(SpillEntry) * c->spill_cap)) != hipSuccess)
        return bail("hipMalloc(spill)", e);
    if ((e = hipMalloc(&c->d_ctr, sizeof(Counters))) != hipSuccess) return bail("hipMalloc(ctr)", e);
    if ((e = hipMemsetAsync(c->d_ctr, 0, sizeof(Counters), c->stream)) != hipSuccess)
        return bail("memset", e);
    if ((e = hipHostMalloc(&c->h_ctr, sizeof(Counters))) != hipSuccess) return bail("hipHostMalloc", e);
    if (cfg.key_sets & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) {
        c->cms_words = (size_t)cfg.cms_depth << cfg.cms_width_log2;
        if (cfg.key_sets & FA_KEYS_SRCADDR_CMS) {
            if ((e = hipMalloc(&c->cms_src, c->cms_words * 8 * CMS_REPLICAS)) != hipSuccess) return bail("hipMalloc(cms)", e);
            if ((e = hipMemsetAsync(c->cms_src, 0, c->cms_words * 8 * CMS_REPLICAS, c->stream)) != hipSuccess)
                return bail("memset", e);
        }
        if (cfg.key_sets & FA_KEYS_DSTADDR_CMS) {
            if ((e = hipMalloc(&c->cms_dst, c->cms_words * 8 * CMS_REPLICAS)) != hipSuccess) return bail("hipMalloc(cms)", e);
            if ((e = hipMemsetAsync(c->cms_dst, 0, c->cms_words * 8 * CMS_REPLICAS, c->stream)) != hipSuccess)
                return bail("memset", e);
        }
        c->ks_log2 = cfg.topk_capacity_log2;
        const size_t ks_bytes = sizeof(KeySlot) << c->ks_log2;
        if (cfg.key_sets & FA_KEYS_SRCADDR_CMS) {
            if ((e = hipMalloc(&c->ks_src, ks_bytes)) != hipSuccess) return bail("hipMalloc(key set)", e);
            if ((e = hipMemsetAsync(c->ks_src, 0, ks_bytes, c->stream)) != hipSuccess) return bail("memset", e);
        }
        if (cfg.key_sets & FA_KEYS_DSTADDR_CMS) {
            if ((e = hipMalloc(&c->ks_dst, ks_bytes)) != hipSuccess) return bail("hipMalloc(key set)", e);
            if ((e = hipMemsetAsync(c->ks_dst, 0, ks_bytes, c->stream)) != hipSuccess) return bail("memset", e);
        }
    }
    if (cfg.key_sets & FA_KEYS_WIDE) {
        c->wcap_log2 = cfg.wide_capacity_log2;
        const size_t wbytes = sizeof(WSlot) << c->wcap_log2;
        if ((e = hipMalloc(&c->wtab, wbytes)) != hipSuccess) return bail("hipMalloc(wide table)", e);
        if ((e = hipMemsetAsync(c->wtab, 0, wbytes, c->stream)) != hipSuccess) return bail("memset", e);
        if ((e = hipMalloc(&c->wspill, sizeof(WSpillEntry) * c->wspill_cap)) != hipSuccess) return bail("hipMalloc(wide spill)", e);
        c->stats.wide_capacity = 1ull << c->wcap_log2;
    }
    if (cfg.key_sets & FA_KEYS_PORT_HIST) {
        const size_t hbytes = sizeof(ulonglong2) * 2 * PORT_DENSE;
        if ((e = hipMalloc(&c->port_hist, hbytes)) != hipSuccess) return bail("hipMalloc(port histograms)", e);
        if ((e = hipMemsetAsync(c->port_hist, 0, hbytes, c->stream)) != hipSuccess) return bail("memset", e);
    }
    {  // pinned staging of window-close rows, sized for the initial table (grows with it): pinning at the first close
       // would cost more than the close itself
        const size_t bytes = std::min<size_t>((size_t)sizeof(Row5m) << c->cap_log2, (size_t)64 << 20);
        if (hipHostMalloc(&c->h_rows, bytes) == hipSuccess) c->h_rows_cap = bytes;
        else c->h_rows = nullptr;
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return bail("sync", e);
    c->stats.table_capacity = 1ull << c->cap_log2;
    *out = c;
    return FA_OK;
}

extern "C" void fa_destroy(fa_ctx* c) {
    FA_ON_DEVICE(c);
    if (!c) return;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (getenv("FA_VERBOSE") && c->d_ctr && c->h_ctr && hipMemcpy(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost) == hipSuccess)
        fprintf(stderr, "[flowagg] agg8: %llu groups added in %llu launches, passes now %u; direct %llu, ok %llu, wide rows %llu\n",
                (unsigned long long)c->h_ctr->agg_groups, (unsigned long long)c->h_ctr->agg_launches, c->agg_passes,
                (unsigned long long)c->h_ctr->direct, (unsigned long long)c->h_ctr->ok, (unsigned long long)(c->wused_base + c->h_ctr->wused));
    if (getenv("FA_VERBOSE") && c->wlog_recorded)
        fprintf(stderr, "[flowagg] wide log: %llu chunks recorded, %llu folded by wagg_kernel, %llu by the atomic replay, %llu dropped whole, %zu pending\n",
                (unsigned long long)c->wlog_recorded, (unsigned long long)c->wlog_folded, (unsigned long long)c->wlog_replayed, (unsigned long long)c->wlog_dropped, c->wlog.size());
    if ((c->dbg & DBG_TIMING) && c->d_ctr && c->h_ctr && hipMemcpy(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost) == hipSuccess &&
        c->h_ctr->t_tiles)
        fprintf(stderr, "[flowagg timing] per tile (wave 0 of every workgroup, core clocks): wait %.0f  work %.0f  | tiles %llu  total/wg-launch %.0f\n",
                (double)c->h_ctr->t_wait / (double)c->h_ctr->t_tiles, (double)c->h_ctr->t_work / (double)c->h_ctr->t_tiles,
                (unsigned long long)c->h_ctr->t_tiles, (double)c->h_ctr->t_total);
    if ((c->dbg & DBG_AGG8_TIMING) && c->d_ctr && c->h_ctr && hipMemcpy(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost) == hipSuccess &&
        c->h_ctr->t_tiles)
        fprintf(stderr, "[flowagg timing] agg8_kernel per workgroup and pass (us): set-up %.2f  segment walk %.2f  flush %.2f  | %llu\n",
                (double)c->h_ctr->t_total / (double)c->h_ctr->t_tiles * 0.01, (double)c->h_ctr->t_wait / (double)c->h_ctr->t_tiles * 0.01,
                (double)c->h_ctr->t_work / (double)c->h_ctr->t_tiles * 0.01, (unsigned long long)c->h_ctr->t_tiles);
    if ((c->dbg & DBG_CMS_TIMING) && c->d_ctr && c->h_ctr && hipMemcpy(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost) == hipSuccess &&
        c->h_ctr->t_tiles)
        fprintf(stderr, "[flowagg timing] cms_agg_kernel per workgroup (us): schedule + counts + flush %.2f  segment walk %.2f  | workgroups %llu  mean wave's walk %.2f\n",
                (double)c->h_ctr->t_wait / (double)c->h_ctr->t_tiles * 0.01, (double)c->h_ctr->t_work / (double)c->h_ctr->t_tiles * 0.01,
                (unsigned long long)c->h_ctr->t_tiles, (double)c->h_ctr->t_total / (double)c->h_ctr->t_tiles / (double)(AGG_BLOCK / 64) * 0.01);
    if (getenv("FA_VERBOSE") && c->seg_counts && c->last_nwg) {  // how the last launch's tuples left: whole store units (front) or single tuples (back)
        std::vector<uint32_t> cnt((size_t)c->last_nwg * NPART_MAX * 2);
        if (hipMemcpy(cnt.data(), c->seg_counts, cnt.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long front = 0, back = 0;
            for (size_t i = 0; i < cnt.size() / 2; i++) front += cnt[i], back += cnt[cnt.size() / 2 + i];
            fprintf(stderr, "[flowagg] last launch's tuples: %llu in whole store units, %llu single (%.2f %%)\n", front, back, 100.0 * (double)back / (double)std::max(1ull, front + back));
            // (layout [2][NPART_MAX][nwg]: balance over the key partitions - agg8_kernel runs one workgroup per partition)
            unsigned long long mx = 0;
            for (size_t p = 0; p < (size_t)NPART_MAX; p++) {
                unsigned long long sp = 0;
                for (uint32_t w = 0; w < c->last_nwg; w++) sp += (unsigned long long)cnt[p * c->last_nwg + w] + cnt[((size_t)NPART_MAX + p) * c->last_nwg + w];
                mx = std::max(mx, sp);
            }
            fprintf(stderr, "[flowagg] ... per key partition: mean %.0f max %llu (%.2fx)\n", (double)(front + back) / NPART_MAX, mx,
                    (double)mx * NPART_MAX / (double)std::max(1ull, front + back));
        }
    }
    if (getenv("FA_VERBOSE") && c->cseg_counts && c->last_nwg) {  // balance of the last launch's sketch tuples over the 2 x 256 partitions
        const size_t np = (size_t)CMS_SETS * CMS_NPART;
        std::vector<uint32_t> cnt(np * c->last_nwg * 2);
        if (hipMemcpy(cnt.data(), c->cseg_counts, cnt.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            unsigned long long tot = 0, mx = 0, mxseg = 0;
            for (size_t p = 0; p < np; p++) {
                unsigned long long s = 0;
                for (uint32_t w = 0; w < c->last_nwg; w++) {
                    const unsigned long long v = (unsigned long long)cnt[p * c->last_nwg + w] + cnt[(np + p) * c->last_nwg + w];
                    s += v;
                    mxseg = std::max(mxseg, v);
                }
                tot += s;
                mx = std::max(mx, s);
            }
            fprintf(stderr, "[flowagg] last launch's sketch tuples: %llu, per partition mean %.0f max %llu (%.2fx), longest segment %llu (mean %.1f)\n", tot,
                    (double)tot / (double)np, mx, (double)mx * (double)np / (double)std::max(1ull, tot), mxseg, (double)tot / (double)(np * c->last_nwg));
        }
    }
    (void)hipFree(c->tab);
    (void)hipFree(c->spill);
    (void)hipFree(c->d_ctr);
    if (c->h_ctr) (void)hipHostFree(c->h_ctr);
    if (c->h_snap) (void)hipHostFree(c->h_snap);
    for (int i = 0; i < fa_ctx::NSNAP; i++)
        if (c->snap_ev[i]) (void)hipEventDestroy(c->snap_ev[i]);
    (void)hipFree(c->d_exotic);
    (void)hipFree(c->seg);
    (void)hipFree(c->seg_counts);
    (void)hipFree(c->cseg);
    (void)hipFree(c->cseg_counts);
    (void)hipFree(c->cms_psize);
    (void)hipFree(c->hot_seed);
    (void)hipFree(c->hot_seed_tag);
    (void)hipFree(c->wseg);
    (void)hipFree(c->wseg_counts);
    for (auto* v : {&c->wlog, &c->wlog_free})
        for (auto& k : *v) {
            (void)hipFree(k.seg);
            (void)hipFree(k.counts);
        }
    for (int i = 0; i < 2; i++) {
        if (c->h_stage[i]) (void)hipHostFree(c->h_stage[i]);
        (void)hipFree(c->d_in[i]);
        if (c->stage_ev[i]) (void)hipEventDestroy(c->stage_ev[i]);
        if (c->copy_ev[i]) (void)hipEventDestroy(c->copy_ev[i]);
    }
    (void)hipFree(c->col_block);
    (void)hipFree(c->rc_buf);
    (void)hipFree(c->rw_buf);
    (void)hipFree(c->m_scratch);
    (void)hipFree(c->m_out[0]);
    (void)hipFree(c->m_out[1]);
    (void)hipFree(c->fs_scratch);
    (void)hipFree(c->fs_off);
    (void)hipFree(c->wl_scratch);
    (void)hipFree(c->part_buf);
    (void)hipFree(c->part_cnt);
    if (c->h_rows) (void)hipHostFree(c->h_rows);
    (void)hipFree(c->cms_src);
    (void)hipFree(c->cms_dst);
    (void)hipFree(c->cms_src_m);
    (void)hipFree(c->cms_dst_m);
    (void)hipFree(c->ks_src);
    (void)hipFree(c->ks_dst);
    (void)hipFree(c->wtab);
    (void)hipFree(c->wspill);
    (void)hipFree(c->port_hist);
    for (auto* pool : {&c->ev_pool, &c->dev_pool})
        for (auto& p : *pool) {
            (void)hipEventDestroy(p.e0);
            (void)hipEventDestroy(p.e1);
            (void)hipEventDestroy(p.e2);
        }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- table maintenance -------------------------------------------------------------
// Builds a fresh table of 2^new_log2 slots holding every row outside [tb_lo,tb_hi).
static int rebuild_table(fa_ctx* c, uint32_t new_log2, uint32_t tb_lo, uint32_t tb_hi) {
    Slot* nt = nullptr;
    size_t bytes = sizeof(Slot) << new_log2;
    hipError_t e = hipMalloc(&nt, bytes);
    if (e != hipSuccess) return fail(c, FA_ERR_NOMEM, "rebuild_table: hipMalloc failed");
    if ((e = hipMemsetAsync(nt, 0, bytes, c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(&c->d_ctr->used, 0, sizeof(unsigned long long), c->stream)) != hipSuccess) {
        (void)hipFree(nt);  // the old table stays in place
        c->err = std::string("rebuild_table: ") + hipGetErrorString(e);
        return FA_ERR_HIP;
    }
    Slot* old = c->tab;
    uint32_t old_slots = 1u << c->cap_log2;
    c->tab = nt;
    c->cap_log2 = new_log2;
    KArgs a = make_args(c);
    hipLaunchKernelGGL(rebuild_kernel, dim3(1024), dim3(256), 0, c->stream, old, old_slots, tb_lo, tb_hi, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(old));
    c->used_base = 0;
    c->stats.table_capacity = 1ull << c->cap_log2;
    return FA_OK;
}

// Wide table: fresh table of 2^new_log2 slots holding every row that is not selected by
// (kind_mask, [tb_lo,tb_hi)) - see wrow_selected().
static int rebuild_wide(fa_ctx* c, uint32_t new_log2, uint32_t kind_mask, uint32_t tb_lo, uint32_t tb_hi) {
    WSlot* nt = nullptr;
    size_t bytes = sizeof(WSlot) << new_log2;
    if (hipMalloc(&nt, bytes) != hipSuccess) return fail(c, FA_ERR_NOMEM, "rebuild_wide: hipMalloc failed");
    hipError_t e;
    if ((e = hipMemsetAsync(nt, 0, bytes, c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(&c->d_ctr->wused, 0, sizeof(unsigned long long), c->stream)) != hipSuccess) {
        (void)hipFree(nt);  // the old table stays in place
        c->err = std::string("rebuild_wide: ") + hipGetErrorString(e);
        return FA_ERR_HIP;
    }
    WSlot* old = c->wtab;
    uint32_t old_slots = 1u << c->wcap_log2;
    c->wtab = nt;
    c->wcap_log2 = new_log2;
    KArgs a = make_args(c);
    hipLaunchKernelGGL(wrebuild_kernel, dim3(1024), dim3(256), 0, c->stream, old, old_slots, kind_mask, tb_lo, tb_hi, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(old));
    c->wused_base = 0;
    c->wide_dead = 0;  // (slots without rows were not carried over)
    c->stats.wide_capacity = 1ull << c->wcap_log2;
    return FA_OK;
}

static uint32_t log2_ceil(uint64_t v) {
    uint32_t l = 0;
    while ((1ull << l) < v) l++;
    return l;
}

// grows the wide table / replays parked updates until nothing is pending (h = fresh copy of the counters)
static int settle_wide(fa_ctx* c, Counters& h) {
    if (!c->wtab) return FA_OK;
    c->stats.wide_used = c->wused_base + h.wused;
    if (h.wspill_lost) {
        c->sticky = FA_ERR_TABLE_FULL;
        return fail(c, FA_ERR_TABLE_FULL, "wide-key table and its spill buffer overflowed; aggregates were lost");
    }
    int guard = 0;
    while (h.wspill_count || c->stats.wide_used * 2 > (1ull << c->wcap_log2)) {
        // room for every group that exists plus every parked update, at <= 50 % load, in ONE step - at the SAME size when
        // the slots that closed windows left dead (wdrop_kernel) are what fills the table
        const uint64_t live = c->stats.wide_used - std::min(c->wide_dead, c->stats.wide_used);
        const uint32_t want = c->wide_dead ? std::max(c->wcap_log2, log2_ceil(2 * (live + h.wspill_count))) :
                                             std::max(c->wcap_log2 + 1, log2_ceil(2 * (c->stats.wide_used + h.wspill_count)));
        if (want > 30 || ++guard > 8) return fail(c, FA_ERR_TABLE_FULL, "wide-key table cannot grow further");
        const uint32_t nspill = h.wspill_count;
        // the parked updates move to a private copy and the buffer is emptied BEFORE the rebuild, so that anything the
        // rebuild or the replay parks again is kept for the next round of this loop
        WSpillEntry* tmp = nullptr;
        if (nspill) {
            if (hipMalloc(&tmp, sizeof(WSpillEntry) * nspill) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(wide spill copy) failed");
            hipError_t e = hipMemcpyAsync(tmp, c->wspill, sizeof(WSpillEntry) * nspill, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) { (void)hipFree(tmp); c->err = "settle_wide: copy failed"; return FA_ERR_HIP; }
        }
        hipError_t e0 = hipMemsetAsync(&c->d_ctr->wspill_count, 0, sizeof(unsigned int), c->stream);
        int rc = e0 == hipSuccess ? rebuild_wide(c, want, 0, 0, 0 /* nothing selected: keep everything */) : FA_ERR_HIP;
        if (rc == FA_OK && nspill) {
            KArgs a = make_args(c);
            static_assert(sizeof(WSpillEntry) == sizeof(WRow), "spill entries replay as rows");
            hipLaunchKernelGGL(wmerge_kernel, dim3(256), dim3(256), 0, c->stream, (const WRow*)tmp, nspill, a);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = FA_ERR_HIP;
        }
        if (tmp) (void)hipFree(tmp);
        if (rc) return rc == FA_ERR_HIP ? fail(c, FA_ERR_HIP, "settle_wide: replay failed") : rc;
        HIPCHK(c, hipMemcpy(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost));
        h = *c->h_ctr;
        c->stats.wide_used = c->wused_base + h.wused;
        if (h.wspill_lost) return fail(c, FA_ERR_TABLE_FULL, "wide spill buffer overflowed during replay");
    }
    return FA_OK;
}

static int cms_fold(fa_ctx* c);

// tuple-format feedback (fa_ctx::use_t8): h = counters at least as new as the last look
static void format_feedback(fa_ctx* c, const Counters& h) {
    const uint64_t d_mis = h.misfit8 - c->seen_misfit8, d_ok = h.ok - c->seen_ok;
    if (d_ok && d_mis * 16 > d_ok) c->t8_wide_until = c->stats.batches + 64;
    c->seen_misfit8 = h.misfit8;
    c->seen_ok = h.ok;
    if (c->wtab && !c->wide_defer) {  // (SrcAddr,DstPort,Proto): scatter sink or atomics, by the share of records that opened a row lately
        if (h.wused < c->seen_wused) c->seen_wused = h.wused;  // (table rebuilt: the count starts over)
        if (h.ok < c->seen_ok_w) c->seen_ok_w = h.ok;
        const uint64_t dw = h.wused - c->seen_wused, dn = h.ok - c->seen_ok_w;
        if (dn >= (1u << 20)) {
            if (dw * 5 > dn) c->wide_scatter = true;
            else if (dw * 10 < dn) c->wide_scatter = false;
            // ... and no table at all for a stream that opens a row for most of its records: the log (fa_ctx::wlog)
            if (c->wide_mode == 0 && c->wide_scatter && dw * 2 > dn && c->wlog_max > 0) {
                c->wide_defer = true;
                c->seen_wfold = h.wfold_n;
            }
            c->seen_wused = h.wused;
            c->seen_ok_w = h.ok;
        }
    }
    if (c->wtab && c->wide_defer && c->wide_mode == 0) {  // log mode chosen by the library: do the chunks that ARE folded still open rows?
        const uint64_t dn = h.wfold_n - c->seen_wfold;
        if (h.wused < c->seen_wused || h.wfold_n < c->seen_wfold) {  // (table rebuilt / counters restarted: no verdict from this look)
            c->seen_wused = h.wused;
            c->seen_wfold = h.wfold_n;
        } else if (dn >= (1u << 20)) {
            // fewer than half of the folded tuples opened a row: this stream aggregates - back to the table (scatter sink or
            // atomics, by the feedback above); the chunks still pending are folded one per launch (fa_ingest_device)
            if ((h.wused - c->seen_wused) * 2 < dn) {
                c->wide_defer = false;
                c->seen_ok_w = h.ok;
            }
            c->seen_wused = h.wused;
            c->seen_wfold = h.wfold_n;
        }
    }
    // passes of agg8_kernel: the groups a launch adds to the device table, per partition and pass, against the LDS table
    if (h.agg_launches < c->seen_agg_launches || h.agg_groups < c->seen_agg_groups) c->seen_agg_launches = c->seen_agg_groups = 0;
    const uint64_t d_l = h.agg_launches - c->seen_agg_launches, d_g = h.agg_groups - c->seen_agg_groups;
    if (d_l) {
        const uint64_t per_part = d_g / d_l >> c->plog2;
        uint32_t want = 1;
        while (want < 8 && per_part > (uint64_t)(AGG8_SLOTS / 2) * want) want *= 2;
        // (a launch that overflowed its table under-counts: move up one step at a time, down only when clearly below)
        if (want > c->agg_passes) c->agg_passes = std::min(want, c->agg_passes * 2);
        else if (want < c->agg_passes && per_part * 3 < (uint64_t)(AGG8_SLOTS / 2) * c->agg_passes) c->agg_passes = std::max(want, c->agg_passes / 2);
        c->seen_agg_launches = h.agg_launches;
        c->seen_agg_groups = h.agg_groups;
    }
}

// Waits for the stream, folds device counters into stats, replays spills after growing.
static int settle(fa_ctx* c) {
    if (c->cms_dirty) {
        int frc = cms_fold(c);
        if (frc) return frc;
    }
    HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].e0, c->ev_pool[i].e1) == hipSuccess) {
            c->stats.kernel_ns = (uint64_t)((double)ms * 1e6);
            c->stats.kernel_ns_total += c->stats.kernel_ns;
            c->stats.kernel_launches += 1;
        }
        if (hipEventElapsedTime(&ms, c->ev_pool[i].e0, c->ev_pool[i].e2) == hipSuccess)
            c->stats.batch_ns_total += (uint64_t)((double)ms * 1e6);
    }
    c->ev_used = 0;
    for (size_t i = 0; i < c->dev_used; i++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->dev_pool[i].e0, c->dev_pool[i].e2) == hipSuccess) {
            c->stats.decode_ns_total += (uint64_t)((double)ms * 1e6);
            c->stats.decode_launches += 1;
        }
    }
    c->dev_used = 0;
    Counters h = *c->h_ctr;
    c->stats.records_ok = h.ok;
    c->stats.records_bad = h.bad;
    c->stats.records_slow = h.slow;
    c->stats.records_direct = h.direct;
    c->stats.records_retried = h.retried;
    c->stats.records_misfit_compact = h.misfit8;
    c->stats.records_late = h.late;
    c->stats.table_used = c->used_base + h.used;
    format_feedback(c, h);
    if (h.spill_lost) {
        c->sticky = FA_ERR_TABLE_FULL;
        return fail(c, FA_ERR_TABLE_FULL, "group-by table and spill buffer overflowed; aggregates were lost");
    }
    int guard = 0;
    while (h.spill_count || c->stats.table_used * 2 > (1ull << c->cap_log2)) {
        // room for every group that exists plus every parked update, at <= 50 % load, in ONE step
        const uint32_t want = std::max(c->cap_log2 + 1, log2_ceil(2 * (c->stats.table_used + h.spill_count)));
        if (want > 30 || ++guard > 8) return fail(c, FA_ERR_TABLE_FULL, "group-by table cannot grow further");
        const uint32_t nspill = h.spill_count;
        // the parked aggregates move to a private copy and the buffer is emptied BEFORE the rebuild, so that anything
        // the rebuild or the replay parks again is kept for the next round of this loop
        SpillEntry* tmp = nullptr;
        if (nspill) {
            if (hipMalloc(&tmp, sizeof(SpillEntry) * nspill) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(spill copy) failed");
            hipError_t e = hipMemcpyAsync(tmp, c->spill, sizeof(SpillEntry) * nspill, hipMemcpyDeviceToDevice, c->stream);
            if (e != hipSuccess) { (void)hipFree(tmp); c->err = "settle: copy failed"; return FA_ERR_HIP; }
        }
        hipError_t e0 = hipMemsetAsync(&c->d_ctr->spill_count, 0, sizeof(unsigned int), c->stream);
        int rc = e0 == hipSuccess ? rebuild_table(c, want, 1, 0 /* empty range: keep everything */) : FA_ERR_HIP;
        if (rc == FA_OK && nspill) {
            KArgs a = make_args(c);
            hipLaunchKernelGGL(replay_spill_kernel, dim3(1024), dim3(256), 0, c->stream, tmp, nspill, a);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = FA_ERR_HIP;
        }
        if (tmp) (void)hipFree(tmp);
        if (rc) return rc == FA_ERR_HIP ? fail(c, FA_ERR_HIP, "settle: replay failed") : rc;
        HIPCHK(c, hipMemcpy(c->h_ctr, c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost));
        h = *c->h_ctr;
        c->stats.table_used = c->used_base + h.used;
        if (h.spill_lost) return fail(c, FA_ERR_TABLE_FULL, "spill buffer overflowed during replay");
    }
    int rc = settle_wide(c, h);
    // everything launched so far is accounted for
    c->known = h;
    c->known_records = c->launched_records;
    c->known_seq = c->launch_seq;
    c->known_wpot = c->wpot_total;
    return rc;
}

// ---- wide log (FA_WIDE=log; fa_ctx::wlog) -----------------------------------------------------------------------------
static WChunkArgs wchunk_args(const fa_ctx::WChunk& k) {
    return WChunkArgs{k.seg, k.counts, k.counts + k.counts_cap, 1u << k.wplog2, k.nwg, k.wcapq, k.wm, k.wregion};
}
// room in the table for `more` new rows at <= 50 % load (host-side bound first; a settle - and a one-step growth - only when
// the bound says so)
static uint64_t wide_rows_bound(const fa_ctx* c) { return c->wused_base + c->known.wused + (c->wpot_total - c->known_wpot); }
static int wlog_make_room(fa_ctx* c, uint64_t more) {
    if ((wide_rows_bound(c) + more) * 2 <= (1ull << c->wcap_log2)) return FA_OK;
    int rc = settle(c);  // (exact count; grows the table by itself when it is above 50 % already)
    if (rc) return rc;
    if ((c->stats.wide_used + more) * 2 <= (1ull << c->wcap_log2)) return FA_OK;
    const uint32_t want = log2_ceil(2 * (c->stats.wide_used + more));
    if (want > 30) return fail(c, FA_ERR_TABLE_FULL, "wide-key table cannot grow further");
    return rebuild_wide(c, want, 0, 0, 0);  // (pending chunks stay pending: they are folded by the atomic replay from now on)
}
// one pending chunk into the table: wagg_kernel while the table still has the geometry the tuples were scattered for (its
// workgroups own "their" regions' tuples), the atomic replay otherwise; the buffers go to the free list
static int wlog_fold(fa_ctx* c, const fa_ctx::WChunk& k) {
    KArgs a = make_args(c);
    if (k.wplog2 == a.wplog2 && k.wmask == a.wmask) {
        a.wseg = k.seg;
        a.wseg_counts = k.counts;
        a.nwg = k.nwg;
        a.wcapq = k.wcapq;
        a.wregion = k.wregion;
        hipLaunchKernelGGL(wagg_kernel, dim3(1u << a.wplog2), dim3(WAGG_BLOCK), 0, c->stream, a, (const uint32_t*)(k.counts + k.counts_cap), k.wm);
        c->wlog_folded++;
    } else {
        hipLaunchKernelGGL(wlog_replay_kernel, dim3(2048), dim3(256), 0, c->stream, wchunk_args(k), a);
        c->wlog_replayed++;
    }
    HIPCHK(c, hipGetLastError());
    c->wpot_total += k.n;
    c->wlog_free.push_back(k);
    return FA_OK;
}
static int wlog_flush_oldest(fa_ctx* c) {
    const fa_ctx::WChunk k = c->wlog.front();
    int rc = wlog_make_room(c, k.n);
    if (rc) return rc;
    c->wlog.erase(c->wlog.begin());
    return wlog_fold(c, k);
}
static int wlog_flush_all(fa_ctx* c) {
    while (!c->wlog.empty()) {
        int rc = wlog_flush_oldest(c);
        if (rc) return rc;
    }
    return FA_OK;
}
// behind a log-mode launch: its segment buffers become the newest chunk (the ctx allocates or recycles others for the next
// launch); more than wlog_max pending: the oldest is folded into the table after all
static int wlog_record(fa_ctx* c, const KArgs& a, size_t n) {
    fa_ctx::WChunk k;
    k.seg = c->wseg;
    k.seg_bytes = c->wseg_bytes;
    k.counts = c->wseg_counts;
    k.counts_cap = c->wseg_counts_cap;
    k.nwg = a.nwg;
    k.wcapq = a.wcapq;
    k.wplog2 = a.wplog2;
    k.wmask = a.wmask;
    k.wregion = a.wregion;
    k.n = n;
    c->wlog.push_back(k);
    c->wlog_recorded++;
    c->wseg = nullptr;
    c->wseg_bytes = 0;
    c->wseg_counts = nullptr;
    c->wseg_counts_cap = 0;
    while (c->wlog.size() > c->wlog_max) {
        int rc = wlog_flush_oldest(c);
        if (rc) return rc;
    }
    return FA_OK;
}
// a drop of buckets [lo, hi) as far as the pending chunks go: when nothing older is alive in them it is their watermark
// (tuples below it are skipped by every later read and fold); anything else folds them into the table first, where the
// caller's rebuild removes the range.  "Older" is judged by the smallest bucket among a chunk's live tuples - looked for
// once per chunk (a scan of its segments, the word behind the time base) and kept: with the launch's time base instead
// (the smallest SAMPLED bucket - 2) every close of a real timeslot found "something older" and folded all chunks.
// every pending chunk's bucket range [minb, maxb] (live tuples; one scan per chunk, kept)
static int wlog_bucket_ranges(fa_ctx* c) {
    bool any = false;
    for (auto& k : c->wlog)
        if (!k.minb_known) {
            uint32_t* word = k.counts + k.counts_cap + 1;  // (the words behind the time base: min, max)
            HIPCHK(c, hipMemsetAsync(word, 0xff, sizeof(uint32_t), c->stream));
            HIPCHK(c, hipMemsetAsync(word + 1, 0, sizeof(uint32_t), c->stream));
            hipLaunchKernelGGL(wlog_minbucket_kernel, dim3(2048), dim3(256), 0, c->stream, wchunk_args(k), word);
            HIPCHK(c, hipGetLastError());
            any = true;
        }
    if (!any) return FA_OK;
    std::vector<uint32_t> mm(2 * c->wlog.size());
    for (size_t i = 0; i < c->wlog.size(); i++)
        if (!c->wlog[i].minb_known) HIPCHK(c, hipMemcpyAsync(&mm[2 * i], c->wlog[i].counts + c->wlog[i].counts_cap + 1, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < c->wlog.size(); i++) {
        fa_ctx::WChunk& k = c->wlog[i];
        if (!k.minb_known) {
            k.minb = mm[2 * i];  // (~0: no live tuple at all)
            k.maxb = mm[2 * i + 1];
            k.minb_known = true;
        }
    }
    return FA_OK;
}
static int wlog_drop(fa_ctx* c, uint32_t lo, uint32_t hi) {
    if (c->wlog.empty()) return FA_OK;
    int rc0 = wlog_bucket_ranges(c);
    if (rc0) return rc0;
    uint32_t oldest = 0xFFFFFFFFu;  // smallest bucket any pending chunk still holds
    for (size_t i = 0; i < c->wlog.size(); i++) {
        fa_ctx::WChunk& k = c->wlog[i];
        if (k.minb != 0xFFFFFFFFu && k.maxb >= k.wm) oldest = std::min(oldest, std::max(k.minb, k.wm));
    }
    if (lo > oldest) return wlog_flush_all(c);  // (a range that is not the oldest: the table's rebuild has to remove it)
    for (size_t i = 0; i < c->wlog.size();) {
        fa_ctx::WChunk& k = c->wlog[i];
        if (hi > k.wm) {
            k.wm = hi;
            c->wlog_wm_moves++;
        }
        if (k.minb == 0xFFFFFFFFu || k.maxb < k.wm) {  // nothing of this chunk is alive
            c->wlog_free.push_back(k);
            c->wlog_dropped++;
            c->wlog.erase(c->wlog.begin() + (long)i);
        } else {
            i++;
        }
    }
    return FA_OK;
}

// ---- counter snapshots: never lose aggregates ------------------------------------------------------------
// Every ingest launch is followed by an asynchronous copy of the device counters into a pinned ring slot.  Before
// the next launch the host looks at the newest snapshot that has landed (no waiting): a table above 50 % load or
// parked updates are settled (grow + replay) right away, and a launch is only queued while the updates that could
// be parked by everything in flight still fit the spill buffers - otherwise the host first waits for the newest
// snapshot.  Steady state costs one 300-byte copy per launch and no synchronisation.
static void poll_snapshots(fa_ctx* c, bool wait_newest) {
    if (c->launch_seq == c->known_seq) return;
    if (wait_newest) (void)hipEventSynchronize(c->snap_ev[(c->launch_seq - 1) % fa_ctx::NSNAP]);
    for (uint64_t q = c->launch_seq; q > c->known_seq; q--) {  // newest first
        const int k = (int)((q - 1) % fa_ctx::NSNAP);
        if (c->snap_seq[k] != q) break;  // overwritten: older ones are gone too
        if (hipEventQuery(c->snap_ev[k]) == hipSuccess) {
            c->known = c->h_snap[k];
            c->known_records = c->snap_records[k];
            c->known_seq = q;
            c->known_wpot = c->snap_wpot[k];
            format_feedback(c, c->known);
            return;
        }
    }
}
static int pre_launch_guard(fa_ctx* c, size_t n) {
    auto verdict = [&]() -> int {  // 0 go, 1 look again after waiting, 2 settle
        const Counters& h = c->known;
        if (h.spill_count || h.wspill_count || (c->used_base + h.used) * 2 > (1ull << c->cap_log2) ||
            (c->wtab && (c->wused_base + h.wused) * 2 > (1ull << c->wcap_log2)))
            return 2;
        const uint64_t pot = (c->launched_records - c->known_records) + n;  // records whose updates the host has not seen settle
        if (pot + (1u << 20) > c->spill_cap || (c->wtab && pot * c->wide_per_record + (1u << 20) > c->wspill_cap)) return 1;
        return 0;
    };
    poll_snapshots(c, false);
    int v = verdict();
    if (v == 1) {
        poll_snapshots(c, true);
        v = verdict() ? 2 : 0;
    }
    return v == 2 ? settle(c) : FA_OK;
}
static int post_launch_snapshot(fa_ctx* c, size_t n) {
    c->launched_records += n;
    c->launch_seq += 1;
    const int k = (int)((c->launch_seq - 1) % fa_ctx::NSNAP);
    HIPCHK(c, hipEventSynchronize(c->snap_ev[k]));  // (the slot's previous copy, NSNAP launches ago)
    HIPCHK(c, hipMemcpyAsync(&c->h_snap[k], c->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->snap_ev[k], c->stream));
    c->snap_records[k] = c->launched_records;
    c->snap_seq[k] = c->launch_seq;
    c->snap_wpot[k] = c->wpot_total;
    return FA_OK;
}

extern "C" int fa_sync(fa_ctx* c) {
    FA_ON_DEVICE(c);
    if (!c) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    return settle(c);
}

// ---- ingest ---------------------------------------------------------------------------
// Launch order on the ctx stream: tile -> deferred -> [agg]   (wave-tile kernel: it finds the batch's time base
// itself; the workgroup-tile kernel with the scatter sink still takes it from probe_kernel).
template <int MODE>
static int launch_tiles(fa_ctx* c, KArgs& a, int grid, fa_ctx::LaunchEvents* ev = nullptr) {
    dim3 b(BLOCK);
    dim3 g(grid);
    dim3 ge(std::min(256u, (a.n + BLOCK - 1) / BLOCK));
    a.par = c->par;
    c->par ^= 1u;
    const bool wave_tiles = MODE == MODE_INGEST && a.seg != nullptr && a.tile_recs <= (uint32_t)WT_RECS && c->use_wave_tiles;
    const bool t8 = wave_tiles && c->use_t8;
    // (the second-chance kernel runs in line: on a side stream beside the aggregation it cost MORE - 66 vs 58 us for
    // deferred + aggregation per launch, the cross-stream hand-over being slower than the 4.5 us kernel - and its atomic
    // upserts would race with the region-owned plain stores of agg8_kernel / cms_agg_kernel; the knob is gone)
    hipStream_t dstream = c->stream;
    if (ev) (void)hipEventRecord(ev->e0, c->stream);
    if (MODE == MODE_INGEST && a.seg && !wave_tiles) hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, c->stream, a);
    if (wave_tiles) c->stats.wave_tile_launches += 1;
    if (t8) c->stats.compact_tuple_launches += 1;
#define FA_LAUNCH_W(KS)                                                                         \
    do {                                                                                        \
        if (t8) hipLaunchKernelGGL((wtile_kernel<KS, true>), g, dim3(wtile_block<KS>()), 0, c->stream, a); \
        else hipLaunchKernelGGL((wtile_kernel<KS, false>), g, dim3(wtile_block<KS>()), 0, c->stream, a);   \
    } while (0)
#define FA_LAUNCH(KS)                                                                           \
    case KS: {                                                                                  \
        if constexpr (MODE == MODE_INGEST) {                                                    \
            if (wave_tiles) FA_LAUNCH_W(KS);                                                    \
            else hipLaunchKernelGGL((tile_kernel<MODE, KS>), g, b, 0, c->stream, a);            \
        } else {                                                                                \
            hipLaunchKernelGGL((tile_kernel<MODE, KS>), g, b, 0, c->stream, a);                 \
        }                                                                                       \
        if (ev) (void)hipEventRecord(ev->e1, c->stream);                                        \
        hipLaunchKernelGGL((deferred_kernel<MODE, KS>), ge, b, 0, dstream, a);                  \
        break;                                                                                  \
    }
    if constexpr (MODE == MODE_DECODE) {
        switch (1u) { FA_LAUNCH(1u) }
    } else {
        switch (c->cfg.key_sets) {
            FA_LAUNCH(1u) FA_LAUNCH(2u) FA_LAUNCH(3u) FA_LAUNCH(4u) FA_LAUNCH(5u) FA_LAUNCH(6u) FA_LAUNCH(7u) FA_LAUNCH(9u)
        default:  // any wide key set: the generic variant (runtime mask)
            if (wave_tiles) FA_LAUNCH_W(KS_ALL);
            else hipLaunchKernelGGL((tile_kernel<MODE, KS_ALL>), g, b, 0, c->stream, a);
            if (ev) (void)hipEventRecord(ev->e1, c->stream);
            hipLaunchKernelGGL((deferred_kernel<MODE, KS_ALL>), ge, b, 0, dstream, a);
            break;
        }
    }
#undef FA_LAUNCH
#undef FA_LAUNCH_W
    if (MODE == MODE_INGEST && a.seg) {
        const dim3 ga((1u << a.plog2) * AGG_SPLIT);
        if (t8 && AGG_SPLIT == 1 && !c->agg_generic) hipLaunchKernelGGL(agg8_kernel, ga, dim3(AGG_BLOCK), 0, c->stream, a);
        else if (t8) hipLaunchKernelGGL(agg_kernel<true>, ga, dim3(AGG_BLOCK), 0, c->stream, a);
        else hipLaunchKernelGGL(agg_kernel<false>, ga, dim3(AGG_BLOCK), 0, c->stream, a);
    }
    if (MODE == MODE_INGEST && wave_tiles && a.cseg) {  // fold the sketch tuples (Count-Min scatter sink)
        const uint32_t set_mask = (c->cfg.key_sets >> 1) & 3u;
        const uint32_t nlog = CMS_NPART * (set_mask == 3u ? 2u : 1u);
        hipLaunchKernelGGL(cms_agg_kernel, dim3(std::min<uint32_t>((uint32_t)c->num_cus, nlog)), dim3(AGG_BLOCK), 0, c->stream, a, set_mask, c->cms_par);  // persistent: one per CU
        c->cms_par ^= 1u;
    }
    // fold the (SrcAddr,DstPort,Proto) tuples: one workgroup per table region, plain loads and stores - behind every
    // dispatch of this launch that updates the wide table with atomics (wagg.cuh)
    if (MODE == MODE_INGEST && wave_tiles && a.wseg) {
        if (c->wlog_now) {  // log mode: the tuples stay where they are (the chunk is taken over behind the launch: wlog_record)
            HIPCHK(c, hipMemcpyAsync(c->wseg_counts + c->wseg_counts_cap, &c->d_ctr->tb_base, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        } else {
            hipLaunchKernelGGL(wagg_kernel, dim3(1u << a.wplog2), dim3(WAGG_BLOCK), 0, c->stream, a, (const uint32_t*)nullptr, 0u);
        }
    }
    if (ev) (void)hipEventRecord(ev->e2, c->stream);
    HIPCHK(c, hipGetLastError());
    return FA_OK;
}

template <int MODE>
static int tile_grid(fa_ctx* c, uint32_t n, uint32_t tile_recs) {
    if constexpr (MODE == MODE_DECODE) return grid_for(c, tile_kernel<MODE_DECODE, 1u>, n, tile_recs);
    switch (c->cfg.key_sets) {
    case 1u: return grid_for(c, tile_kernel<MODE_INGEST, 1u>, n, tile_recs);
    case 2u: return grid_for(c, tile_kernel<MODE_INGEST, 2u>, n, tile_recs);
    case 3u: return grid_for(c, tile_kernel<MODE_INGEST, 3u>, n, tile_recs);
    case 4u: return grid_for(c, tile_kernel<MODE_INGEST, 4u>, n, tile_recs);
    case 5u: return grid_for(c, tile_kernel<MODE_INGEST, 5u>, n, tile_recs);
    case 6u: return grid_for(c, tile_kernel<MODE_INGEST, 6u>, n, tile_recs);
    case 7u: return grid_for(c, tile_kernel<MODE_INGEST, 7u>, n, tile_recs);
    case 9u: return grid_for(c, tile_kernel<MODE_INGEST, 9u>, n, tile_recs);
    default: return grid_for(c, tile_kernel<MODE_INGEST, KS_ALL>, n, tile_recs);
    }
}

// deferral lists for n records: exotic [0,cap) and retry [cap,2cap)
static int ensure_exotic(fa_ctx* c, size_t n) {
    if (c->exotic_cap >= n) return FA_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_exotic);
    c->d_exotic = nullptr;
    size_t cap = std::max<size_t>(n, 1 << 16);
    if (hipMalloc(&c->d_exotic, 2 * cap * sizeof(uint32_t)) != hipSuccess)
        return fail(c, FA_ERR_NOMEM, "hipMalloc(deferral lists) failed");
    c->exotic_cap = cap;
    return FA_OK;
}

// Tuple segments for a batch of n records processed by nwg workgroups: capacity per (partition,
// workgroup) = 2x the mean + 32 (a Poisson mean of m never reaches 2m+32; skewed batches overflow into
// the direct path), in whole 128-byte lines (8 wide / 16 compact tuples).  The region stride gets a skew of
// three lines so that consecutive partitions do not alias in L2.
static int ensure_segments(fa_ctx* c, size_t n, uint32_t nwg, bool t8, KArgs& a) {
    const size_t NPART = (size_t)1 << c->plog2;
    const size_t avg = n / ((size_t)nwg * NPART);
    const uint32_t tpl = (t8 ? 2u : 1u) * bin_line(c->cfg.key_sets);  // tuples per store unit (a 128-byte line; flows_5m alone: half a line)
    uint32_t capq = (uint32_t)((2 * avg + 32 + tpl - 1) & ~(size_t)(tpl - 1));
    if (c->seg_cap_limit) capq = std::max<uint32_t>(std::min(capq, c->seg_cap_limit) & ~(tpl - 1), 2 * tpl);  // (tests: force the segment-overflow fallbacks)
    const size_t region = (size_t)nwg * capq + 3 * tpl;
    const size_t bytes = region * NPART * (t8 ? 8 : 16);
    if (c->seg_bytes < bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->seg);
        c->seg = nullptr;
        c->seg_bytes = 0;
        if (hipMalloc(&c->seg, bytes) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(tuple segments) failed");
        c->seg_bytes = bytes;
    }
    const size_t ncnt = (size_t)nwg * NPART_MAX * 2;  // front and back counts
    if (c->seg_counts_cap < ncnt) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->seg_counts);
        c->seg_counts = nullptr;
        c->seg_counts_cap = 0;
        if (hipMalloc(&c->seg_counts, ncnt * sizeof(uint32_t)) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(segment counts) failed");
        c->seg_counts_cap = ncnt;
    }
    a.seg = c->seg;
    a.seg_counts = c->seg_counts;
    a.capq = capq;
    // (the LDS position counters are 16-bit halves of one word: caps leave room for the few increments that are in
    // flight before a full segment's overflow is taken back - sinks.cuh)
    a.capb = std::min<uint32_t>(std::max<uint32_t>(2 * tpl, (capq / 4) & ~(tpl - 1)), 0xff00u - tpl);  // back part: single tuples, bin leftovers
    a.capb = std::min(a.capb, capq - tpl);
    a.capf = std::min<uint32_t>(capq - a.capb, 0xff00u * tpl);                                           // front part: full lines
    a.nwg = nwg;
    a.region = region;
    a.plog2 = c->plog2;
    c->last_nwg = nwg;
    return FA_OK;
}

// Segments of the Count-Min scatter sink for a batch of n records processed by nwg workgroups: per (sketch partition,
// workgroup) 3x the mean + 32 tuples of 16 bytes (a heavy hitter adds up to one tuple per wave-tile to its partition after
// the wave-level fold: about as much again as the partition's mean; what still overflows is added with atomics).
static int ensure_csegments(fa_ctx* c, size_t n, uint32_t nwg, KArgs& a) {
    const size_t nparts = (size_t)CMS_SETS * CMS_NPART;
    const size_t mean = n / ((size_t)CMS_NPART * nwg);
    uint32_t capq = (uint32_t)((3 * mean + 32 + 3) & ~(size_t)3);
    if (c->seg_cap_limit) capq = std::max<uint32_t>(std::min(capq, c->seg_cap_limit) & ~3u, 32u);  // (tests: force the overflow fallbacks)
    const size_t region = (size_t)nwg * capq + 12;
    const size_t bytes = region * nparts * sizeof(uint4);
    if (c->cseg_bytes < bytes) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->cseg);
        c->cseg = nullptr;
        c->cseg_bytes = 0;
        if (hipMalloc(&c->cseg, bytes) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(sketch tuple segments) failed");
        c->cseg_bytes = bytes;
    }
    const size_t ncnt = (size_t)nwg * nparts * 2;
    if (c->cseg_counts_cap < ncnt) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->cseg_counts);
        c->cseg_counts = nullptr;
        c->cseg_counts_cap = 0;
        if (hipMalloc(&c->cseg_counts, ncnt * sizeof(uint32_t)) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(sketch segment counts) failed");
        c->cseg_counts_cap = ncnt;
    }
    if (c->hot_seed_wgs < nwg) {  // the hot-address caches' surviving entries, one set per workgroup (sinks.cuh, HotAddrs)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->hot_seed);
        (void)hipFree(c->hot_seed_tag);
        c->hot_seed = nullptr;
        c->hot_seed_tag = nullptr;
        c->hot_seed_wgs = 0;
        const size_t ne = (size_t)nwg * CMS_SETS * HOT_SLOTS;
        if (hipMalloc(&c->hot_seed, ne * sizeof(HotSeed)) != hipSuccess || hipMalloc(&c->hot_seed_tag, ne * sizeof(uint32_t)) != hipSuccess)
            return fail(c, FA_ERR_NOMEM, "hipMalloc(hot-address seeds) failed");
        HIPCHK(c, hipMemsetAsync(c->hot_seed_tag, 0, ne * sizeof(uint32_t), c->stream));
        c->hot_seed_wgs = nwg;
    }
    a.hot_seed = c->hot_seed;
    a.hot_seed_tag = c->hot_seed_tag;
    a.hot_epoch = c->hot_epoch++;
    if (!c->cms_psize) {
        const size_t bytes = (2 * (size_t)CMS_SETS * CMS_NPART + 2) * sizeof(uint32_t);  // (+ the two unit counters)
        if (hipMalloc(&c->cms_psize, bytes) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(sketch partition sizes) failed");
        HIPCHK(c, hipMemsetAsync(c->cms_psize, 0, bytes, c->stream));
    }
    a.cms_psize = c->cms_psize;
    a.cseg = c->cseg;
    a.cseg_counts = c->cseg_counts;
    a.ccapq = capq;
    a.ccapb = std::min<uint32_t>(std::max<uint32_t>(8u, (capq / 8) & ~3u), 0xff00u);
    a.ccapf = std::min<uint32_t>(capq - a.ccapb, 0xff00u * CMS_BIN);
    a.cregion = region;
    a.cms_sub = c->cfg.cms_width_log2 - 8u;
    return FA_OK;
}

// Segments of the wide scatter sink for a batch of n records processed by nwg workgroups: per (table region, workgroup)
// 2x the mean + 32 tuples of 32 bytes (what overflows - a heavy key's region - takes the atomic path).
static int ensure_wsegments(fa_ctx* c, size_t n, uint32_t nwg, KArgs& a) {
    const size_t nparts = (size_t)1 << a.wplog2;
    const size_t mean = n / (nparts * nwg);
    uint32_t capq = (uint32_t)(2 * mean + 32);
    if (c->seg_cap_limit) capq = std::max<uint32_t>(std::min(capq, c->seg_cap_limit), 4u);  // (tests: force the overflow fallback)
    const size_t region = (size_t)nwg * capq + 6;  // (skew against power-of-two strides)
    const size_t bytes = region * nparts * 2 * sizeof(uint4);
    const size_t ncnt = (size_t)nwg * nparts;
    // Log mode keeps up to wlog_max + 1 pairs of buffers alive (the pending chunks and the one being filled; recycled through
    // wlog_free).  When another pair cannot be had - hipMalloc fails (FA_WSEG_BUDGET: the tests' stand-in for that) - the
    // ingest does not fail: buffers on the free list that are too small are released, then the OLDEST pending chunk is folded
    // into the table after all and its buffers are taken over; only a ctx that holds no chunk at all reports FA_ERR_NOMEM.
    auto held = [&]() { return (uint32_t)((c->wseg || c->wseg_counts ? 1 : 0) + c->wlog.size() + c->wlog_free.size()); };
    for (;;) {
        if (!c->wseg && !c->wseg_counts && !c->wlog_free.empty()) {  // the buffers of a chunk that has been folded or dropped
            const fa_ctx::WChunk k = c->wlog_free.back();
            c->wlog_free.pop_back();
            c->wseg = k.seg;
            c->wseg_bytes = k.seg_bytes;
            c->wseg_counts = k.counts;
            c->wseg_counts_cap = k.counts_cap;
        }
        bool ok = true;
        if (c->wseg_bytes < bytes || c->wseg_counts_cap < ncnt) {
            HIPCHK(c, hipStreamSynchronize(c->stream));  // (a recycled pair may still be read by its chunk's fold)
            (void)hipFree(c->wseg);
            (void)hipFree(c->wseg_counts);
            c->wseg = nullptr;
            c->wseg_counts = nullptr;
            c->wseg_bytes = c->wseg_counts_cap = 0;
            ok = !(c->wseg_budget && held() + 1u > c->wseg_budget) && hipMalloc(&c->wseg, bytes) == hipSuccess;
            if (ok) {
                c->wseg_bytes = bytes;
                ok = hipMalloc(&c->wseg_counts, (ncnt + 4) * sizeof(uint32_t)) == hipSuccess;  // (+ the time base and minimum bucket words of a log chunk)
                if (ok) c->wseg_counts_cap = ncnt;
            }
            if (!ok) {
                (void)hipGetLastError();
                (void)hipFree(c->wseg);
                c->wseg = nullptr;
                c->wseg_bytes = 0;
            }
        }
        if (ok) break;
        if (!c->wlog_free.empty()) continue;  // (the next pair of the free list - released if it is too small as well)
        if (c->wlog.empty()) return fail(c, FA_ERR_NOMEM, "hipMalloc(wide tuple segments) failed");
        int rc = wlog_flush_oldest(c);  // its buffers land on the free list
        if (rc) return rc;
        c->wlog_nomem_folds++;
    }
    a.wseg = c->wseg;
    a.wseg_counts = c->wseg_counts;
    a.wcapq = capq;
    a.wregion = region;
    a.nwg = nwg;
    return FA_OK;
}

static int ensure_dev(fa_ctx* c, void** p, size_t* cap, size_t bytes, const char* what);
static int frame_split_host(const uint8_t* buf, size_t len, std::vector<uint64_t>& off);
// Device-side framing (framing.cuh): d_buf[0, len) is a chain of varint(len)-framed records -> *d_off = n + 1 offsets in HBM
// (owned by the ctx, valid until the next split), *n_out = records.  FA_ERR_FRAMING when it is not such a chain.
static int frame_split_device(fa_ctx* c, const uint8_t* d_buf, size_t len, const uint32_t** d_off, size_t* n_out) {
    *d_off = nullptr;
    *n_out = 0;
    if (len == 0) return FA_OK;
    const uint32_t nb = (uint32_t)((len + FS_BLOCK - 1) / FS_BLOCK);
    size_t tmp_scan = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_scan, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)nb, c->stream);
    const size_t arr = ((size_t)nb * 4 + 255) & ~(size_t)255;
    int rc = ensure_dev(c, &c->fs_scratch, &c->fs_scratch_cap, 7 * arr + tmp_scan + 512, "framing scratch");
    if (rc) return rc;
    uint8_t* base = (uint8_t*)c->fs_scratch;
    uint32_t* start = (uint32_t*)base;
    uint32_t* exits = (uint32_t*)(base + arr);
    uint32_t* cnt = (uint32_t*)(base + 2 * arr);
    uint32_t* bases = (uint32_t*)(base + 3 * arr);
    uint8_t* err = base + 4 * arr;             // (nb bytes)
    uint8_t* trust[2] = {base + 5 * arr, base + 6 * arr};  // (nb bytes each)
    unsigned int* flag = (unsigned int*)(base + 7 * arr);
    void* tmp = base + 7 * arr + 256;
    const dim3 gl((nb + 255) / 256), bl(256);
    HIPCHK(c, hipMemsetAsync(start, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(fs_guess_kernel, dim3((nb + 3) / 4), bl, 0, c->stream, d_buf, (uint32_t)len, nb, start);
    HIPCHK(c, hipGetLastError());
    bool settled = false;
    int rounds = 0;
    for (int round = 0; round < FS_MAX_ROUNDS && !settled; round++, rounds++) {
        const uint8_t* tin = round ? trust[(round - 1) & 1] : nullptr;
        if (round) hipLaunchKernelGGL(fs_apply_kernel, gl, bl, 0, c->stream, nb, start, (const uint32_t*)exits, tin);
        HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL(fs_walk_kernel, gl, bl, 0, c->stream, d_buf, (uint32_t)len, nb, start, cnt, err, tin, trust[round & 1], exits, flag);
        HIPCHK(c, hipGetLastError());
        unsigned int differ = 0;
        HIPCHK(c, hipMemcpyAsync(&differ, flag, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        settled = differ == 0;
    }
    if (getenv("FA_VERBOSE")) fprintf(stderr, "[flowagg framing] %zu bytes, %u blocks: %s after %d round(s)\n", len, nb, settled ? "settled" : "NOT settled (host walk)", rounds);
    std::vector<uint32_t> h_off;
    if (settled) {
        HIPCHK(c, hipMemsetAsync(flag, 0, sizeof(unsigned int), c->stream));
        hipLaunchKernelGGL(fs_err_kernel, dim3(64), dim3(256), 0, c->stream, (const uint8_t*)err, nb, flag);
        size_t tb = tmp_scan;
        if (hipcub::DeviceScan::ExclusiveSum(tmp, tb, cnt, bases, (int)nb, c->stream) != hipSuccess) return fail(c, FA_ERR_HIP, "scan failed");
        unsigned int bad = 0;
        uint32_t last[2] = {0, 0};
        HIPCHK(c, hipMemcpyAsync(&bad, flag, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&last[0], bases + (nb - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&last[1], cnt + (nb - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (bad) return fail(c, FA_ERR_FRAMING, "stream is not a chain of varint-framed records");
        const size_t n = (size_t)last[0] + last[1];
        rc = ensure_dev(c, &c->fs_off, &c->fs_off_cap, (n + 1) * sizeof(uint32_t), "frame offsets");
        if (rc) return rc;
        hipLaunchKernelGGL(fs_emit_kernel, dim3((nb + 255) / 256), dim3(256), 0, c->stream, d_buf, (uint32_t)len, nb, (const uint32_t*)start, (const uint32_t*)bases,
                           (uint32_t*)c->fs_off, (uint32_t)n);
        HIPCHK(c, hipGetLastError());
        *d_off = (const uint32_t*)c->fs_off;
        *n_out = n;
        return FA_OK;
    }
    // the guesses did not settle (records longer than several blocks, adversarial bytes): the host walks the stream
    std::vector<uint8_t> h(len);
    HIPCHK(c, hipMemcpy(h.data(), d_buf, len, hipMemcpyDeviceToHost));
    std::vector<uint64_t> off64;
    rc = frame_split_host(h.data(), len, off64);
    if (rc) return fail(c, rc, "stream is not a chain of varint-framed records");
    h_off.assign(off64.begin(), off64.end());
    rc = ensure_dev(c, &c->fs_off, &c->fs_off_cap, h_off.size() * sizeof(uint32_t), "frame offsets");
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(c->fs_off, h_off.data(), h_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    *d_off = (const uint32_t*)c->fs_off;
    *n_out = h_off.size() - 1;
    return FA_OK;
}

// One call's records -> launches.  len = bytes of the whole buffer the offsets point into (the bound the kernels clamp
// to); bytes = wire bytes of THESE n records (exact or estimated: tile sizing and the bytes_in statistic).
static int ingest_device_records(fa_ctx* c, const void* d_buf, size_t len, size_t bytes, const void* d_off, size_t n);

extern "C" int fa_ingest_device(fa_ctx* c, const void* d_buf, size_t len, const void* d_off, size_t n) {
    FA_ON_DEVICE(c);
    if (!c) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!d_off) {  // offsets == NULL: the stream is split on the device (framing.cuh); n is ignored
        if (!c->cfg.framed) return fail(c, FA_ERR_ARG, "fa_ingest_device: offsets are required for bare (unframed) records");
        if (len == 0) return FA_OK;
        if (!d_buf || len >= (1ull << 32) || ((uintptr_t)d_buf & 15)) return fail(c, FA_ERR_ARG, "fa_ingest_device: bad buffer (16-byte aligned, < 4 GiB)");
        const uint32_t* off = nullptr;
        size_t total = 0;
        int rc = frame_split_device(c, (const uint8_t*)d_buf, len, &off, &total);
        if (rc) return rc;
        size_t used = 0;
        for (size_t i = 0; i < total;) {
            const size_t m = std::min<size_t>(c->cfg.max_batch_records, total - i);
            const size_t bytes = i + m == total ? len - used : (size_t)((double)len * (double)m / (double)total);  // (an estimate per launch; exact in sum)
            rc = ingest_device_records(c, d_buf, len, bytes, off + i, m);
            if (rc) return rc;
            used += bytes;
            i += m;
        }
        return FA_OK;
    }
    if (n == 0) return FA_OK;
    if (!d_buf || len >= (1ull << 32) || n > c->cfg.max_batch_records || ((uintptr_t)d_buf & 15) || ((uintptr_t)d_off & 3))
        return fail(c, FA_ERR_ARG, "fa_ingest_device: bad buffer (16-byte aligned, < 4 GiB, n <= max_batch_records)");
    return ingest_device_records(c, d_buf, len, len, d_off, n);
}

static int ingest_device_records(fa_ctx* c, const void* d_buf, size_t len, size_t bytes, const void* d_off, size_t n) {
    if (n == 0) return FA_OK;
    // The first big launch of a ctx with the (SrcAddr,DstPort,Proto) key set teaches it what the stream looks like: its first 2^20 (and
    // a margin) records go ahead as a launch of their own and the host looks at the counters before the rest follows - a stream that
    // opens a row per record (BASELINE config 5) keeps the rest of that launch in the log instead of folding 16 M rows into the
    // hash table first (one extra launch and one synchronisation in a ctx's life).
    if (c->wtab && (c->cfg.key_sets & FA_KEYS_ADDR_PORT_PROTO) && c->wide_mode == 0 && !c->wide_defer && !c->wide_probed && n >= (1u << 22)) {
        c->wide_probed = true;
        const size_t m = ((size_t)1 << 20) + ((size_t)1 << 17);  // (format_feedback wants 2^20 decoded records; some may be refused)
        const size_t b0 = (size_t)((double)bytes * (double)m / (double)n);
        int rc0 = ingest_device_records(c, d_buf, len, b0, d_off, m);
        if (rc0) return rc0;
        rc0 = settle(c);  // (the counters of those records: format_feedback decides between table, scatter sink and log)
        if (rc0) return rc0;
        return ingest_device_records(c, d_buf, len, bytes - b0, (const uint32_t*)d_off + m, n - m);
    }
    c->wide_probed = true;
    if (c->wide_per_record) {
        // a launch may not be able to park more wide-table updates than the spill buffer holds: split it
        const size_t lim = (c->wspill_cap - (1u << 20)) / c->wide_per_record;
        if (n > lim) {
            for (size_t i = 0; i < n; i += lim) {
                const size_t m = std::min(lim, n - i);
                int rc1 = ingest_device_records(c, d_buf, len, (size_t)((double)bytes * (double)m / (double)n), (const uint32_t*)d_off + i, m);
                if (rc1) return rc1;
            }
            return FA_OK;
        }
    }
    // small batches are not worth a second pass: they go straight to the device-wide table
    const bool scatter = (c->cfg.key_sets & FA_KEYS_AS_PAIR) && (c->sink_mode == 2 || (c->sink_mode == 0 && n >= (1u << 15)));
    c->use_wave_tiles = scatter && c->tile_mode != 2;
    // tuple format of this launch (table.cuh): compact 8-byte tuples on the wave-tile kernel with 256 partitions,
    // unless recent launches showed that this stream's records do not fit them
    c->use_t8 = c->use_wave_tiles && c->plog2 == 8 && c->t8_mode != 2 && (c->t8_mode == 1 || c->stats.batches >= c->t8_wide_until);
    if (n > AGG_MAX_BATCH && !c->use_t8) {  // the packed LDS sums of the wide-tuple aggregation hold 2^24 records per launch
        const size_t h = n / 2;
        int rc1 = ingest_device_records(c, d_buf, len, bytes / 2, d_off, h);
        if (rc1) return rc1;
        return ingest_device_records(c, d_buf, len, bytes - bytes / 2, (const uint32_t*)d_off + h, n - h);
    }
    int rc = pre_launch_guard(c, n);
    if (rc) return rc;
    rc = ensure_exotic(c, n);
    if (rc) return rc;
    KArgs a = make_args(c);
    a.buf = (const uint8_t*)d_buf;
    a.off = (const uint32_t*)d_off;
    a.n = (uint32_t)n;
    a.len = (uint32_t)len;
    a.tile_recs = tile_recs_for(bytes, n);
    int grid = tile_grid<MODE_INGEST>(c, a.n, a.tile_recs);
    if (c->use_wave_tiles) {  // wave-private tiles: <= 64 records per wave, WBLOCK / 64 waves per workgroup, WT_WG_PER_CU workgroups per CU
        const double avg = (double)bytes / (double)n;
        // Tiles are sized by RECORDS: 64 (one per lane) whenever the mean record allows, otherwise as many as fit the
        // buffer with about two sigma of byte headroom (sigma of a tile ~ 12 B x sqrt(records): a mix of 60- and 84-byte
        // records).  A tile whose bytes still exceed the buffer is not lost to the slow path any more: the wave takes
        // its rest as one more part (ingest.cuh) - about 1 tile in 80 on BASELINE config 2, where this fills all 64
        // lanes instead of 61.
        // (the kernel variants that serve a sketch run one 16-wave workgroup per CU instead of two 12-wave ones, with
        // slightly shorter tile buffers: wtile_block, wtile_stride)
        const bool big_wg = !wt_lean(c->cfg.key_sets);
        const uint32_t ks = c->cfg.key_sets;
        auto recs_for = [&](double cap) {
            double r = cap / avg;
            r = (cap - 2.0 * 12.0 * std::sqrt(std::min(r, (double)WT_RECS))) / avg;
            return r >= (double)WT_RECS ? (uint32_t)WT_RECS : r < 1.0 ? 1u : (uint32_t)r;
        };
        const double cap = (double)wt_stride((ks >= 1u && ks <= 7u) || ks == 9u ? ks : KS_ALL) - 16.0 - 15.0;  // (the instantiation launch_tiles picks)
        a.tile_recs = recs_for(cap);
        const uint32_t wtiles = (a.n + a.tile_recs - 1) / a.tile_recs;
        const uint32_t waves = (uint32_t)(big_wg ? WBLOCK_CMS : WBLOCK) / 64u;
        const uint32_t wgs = (wtiles + waves - 1) / waves;
        grid = (int)std::max(1u, std::min<uint32_t>(wgs, (uint32_t)c->num_cus * (uint32_t)(big_wg ? 1 : WT_WG_PER_CU)));
        if (a.tile_recs > (uint32_t)WT_RECS) c->use_t8 = false;
    }
    if (scatter) {
        rc = ensure_segments(c, n, (uint32_t)grid, c->use_t8, a);
        if (rc) return rc;
    }
    if (c->use_wave_tiles && (c->cfg.key_sets & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) && !c->cms_atomic && c->cms_scatter_ok) {
        rc = ensure_csegments(c, n, (uint32_t)grid, a);
        if (rc) return rc;
    }
    if (c->use_wave_tiles && c->wtab && (c->cfg.key_sets & FA_KEYS_ADDR_PORT_PROTO) && grid <= WAGG_MAX_NWG &&
        (c->wide_mode == 2 || c->wide_mode == 3 || (c->wide_mode == 0 && c->wide_scatter))) {
        rc = ensure_wsegments(c, n, (uint32_t)grid, a);
        if (rc) return rc;
        c->wlog_now = c->wide_mode == 3 || (c->wide_mode == 0 && c->wide_defer);
    } else {
        c->wlog_now = false;
    }
    if (c->ev_used == c->ev_pool.size()) {
        if (c->ev_pool.size() >= 4096) {  // bound the pool: fold what is pending
            rc = settle(c);
            if (rc) return rc;
        } else {
            fa_ctx::LaunchEvents e{};
            HIPCHK(c, hipEventCreate(&e.e0));
            HIPCHK(c, hipEventCreate(&e.e1));
            HIPCHK(c, hipEventCreate(&e.e2));
            c->ev_pool.push_back(e);
        }
    }
    fa_ctx::LaunchEvents* evp = &c->ev_pool[c->ev_used++];
    if (c->wtab) c->wpot_total += (uint64_t)n * c->wide_per_record;  // (rows this launch may open in the wide table: wide_rows_bound)
    rc = launch_tiles<MODE_INGEST>(c, a, grid, evp);
    if (rc) return rc;
    if (c->wlog_now && a.wseg) {
        rc = wlog_record(c, a, n);
        if (rc) return rc;
    } else if (c->wide_mode == 0 && !c->wide_defer && !c->wlog.empty()) {
        // the library left the log mode: what is pending drains - a chunk per launch while wagg_kernel can take it (1.2 ms), one
        // in eight launches when the table has grown since and it is the atomic replay (13 ms for 16.67 M tuples)
        const fa_ctx::WChunk& k = c->wlog.front();
        if ((k.wplog2 == a.wplog2 && k.wmask == a.wmask) || c->stats.batches % 8 == 0) {
            rc = wlog_flush_oldest(c);
            if (rc) return rc;
        }
    }
    rc = post_launch_snapshot(c, n);
    if (rc) return rc;
    c->stats.bytes_in += bytes;
    c->stats.batches += 1;
    if (c->cfg.key_sets & (FA_KEYS_SRCADDR_CMS | FA_KEYS_DSTADDR_CMS)) {
        c->cms_dirty = true;
        c->merged_valid = false;
    }
    return FA_OK;
}

// Splits a chain of framed records on the host (offsets == NULL).
static int frame_split_host(const uint8_t* buf, size_t len, std::vector<uint64_t>& off) {
    size_t p = 0;
    off.clear();
    while (p < len) {
        off.push_back(p);
        uint64_t v = 0;
        int i = 0;
        for (;; i++) {
            if (i >= 10 || p >= len) return FA_ERR_FRAMING;
            uint8_t b = buf[p++];
            if (i < 9)
                v |= (uint64_t)(b & 0x7f) << (7 * i);
            else
                v |= (uint64_t)(b & 1) << 63;
            if (!(b & 0x80)) break;
        }
        if (v > len - p) return FA_ERR_FRAMING;
        p += v;
    }
    off.push_back(len);
    return FA_OK;
}

static int ensure_stage(fa_ctx* c, int s, size_t bytes) {
    if (c->h_stage_cap[s] < bytes) {
        if (c->h_stage[s]) (void)hipHostFree(c->h_stage[s]);
        c->h_stage[s] = nullptr;
        size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 20);
        if (hipHostMalloc(&c->h_stage[s], cap) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipHostMalloc(staging) failed");
        c->h_stage_cap[s] = cap;
    }
    if (c->d_in_cap[s] < bytes) {
        (void)hipFree(c->d_in[s]);
        c->d_in[s] = nullptr;
        size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 20);
        if (hipMalloc(&c->d_in[s], cap) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(input) failed");
        c->d_in_cap[s] = cap;
    }
    return FA_OK;
}

// Copies [buf,len) + offsets into pinned staging slot and uploads.  On return the
// caller's memory is no longer referenced.  Layout in the slot: bytes, pad to 16,
// 32 B slack, then n+1 uint32 offsets.
static int stage_and_upload(fa_ctx* c, const uint8_t* buf, size_t len, const uint64_t* off, size_t n,
                            const uint8_t** d_buf, const uint32_t** d_off, int* slot) {
    if (len >= (1ull << 32) - 64) return fail(c, FA_ERR_ARG, "batch larger than 4 GiB; split it");
    int s = c->stage_cur;
    c->stage_cur ^= 1;
    size_t off_pos = ((len + 15) & ~(size_t)15) + 32;
    size_t total = off_pos + (n + 1) * sizeof(uint32_t);
    HIPCHK(c, hipEventSynchronize(c->stage_ev[s]));  // slot free again?
    int rc = ensure_stage(c, s, total);
    if (rc) return rc;
    uint8_t* hs = c->h_stage[s];
    uint32_t* ho = reinterpret_cast<uint32_t*>(hs + off_pos);
    // the copy into pinned memory and the u64 -> u32 narrowing of the offsets run on a few host threads: one thread
    // moves ~10 GB/s, the PCIe Gen5 link behind it 63 GB/s (env FA_STAGE_THREADS, default 8; small batches: inline)
    unsigned nthr = c->stage_threads;
    if (len + n * 12 < (8u << 20)) nthr = 1;
    std::atomic<int> bad{0};
    auto work = [&](unsigned t) {
        const size_t b0 = len * t / nthr, b1 = len * (t + 1) / nthr;
        if (b1 > b0) memcpy(hs + b0, buf + b0, b1 - b0);
        const size_t i0 = (n + 1) * t / nthr, i1 = (n + 1) * (t + 1) / nthr;
        uint64_t prev = i0 ? off[i0 - 1] : 0;
        for (size_t i = i0; i < i1; i++) {
            const uint64_t o = off[i];
            if (o < prev || o > len) {
                bad.store(1);
                return;
            }
            prev = o;
            ho[i] = (uint32_t)o;
        }
    };
    if (nthr <= 1) {
        nthr = 1;
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; t++) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
    }
    if (bad.load()) return fail(c, FA_ERR_ARG, "offsets must be non-decreasing and <= len");
    memset(hs + len, 0, off_pos - len);
    HIPCHK(c, hipMemcpyAsync(c->d_in[s], hs, total, hipMemcpyHostToDevice, c->stream));
    *slot = s;  // caller records stage_ev[s] once the kernels reading d_in[s] are enqueued
    *d_buf = c->d_in[s];
    *d_off = reinterpret_cast<const uint32_t*>(c->d_in[s] + off_pos);
    return FA_OK;
}

extern "C" int fa_ingest(fa_ctx* c, const uint8_t* buf, size_t len, const uint64_t* offsets, size_t n) {
    FA_ON_DEVICE(c);
    if (!c) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!buf && len) return fail(c, FA_ERR_ARG, "fa_ingest: null buffer");
    std::vector<uint64_t> split;
    if (!offsets && c->cfg.framed && len && len <= (1ull << 30)) {
        // a framed stream without offsets: the bytes are uploaded as they are and cut into records on the device (framing.cuh);
        // the host walk below - 1 GB/s - only serves streams beyond one staging buffer
        const uint64_t whole[2] = {0, len};
        const uint8_t* d_buf;
        const uint32_t* d_off;
        int slot = 0;
        int rc = stage_and_upload(c, buf, len, whole, 1, &d_buf, &d_off, &slot);
        if (rc) return rc;
        rc = fa_ingest_device(c, d_buf, len, nullptr, 0);
        HIPCHK(c, hipEventRecord(c->stage_ev[slot], c->stream));
        return rc;
    }
    if (!offsets) {
        if (!c->cfg.framed) return fail(c, FA_ERR_ARG, "fa_ingest: offsets are required for bare (unframed) records");
        int rc = frame_split_host(buf, len, split);
        if (rc) return fail(c, rc, "fa_ingest: stream is not a chain of varint-framed records");
        offsets = split.data();
        n = split.size() - 1;
    }
    if (n == 0) return FA_OK;
    // chunk so that each launch stays under max_batch_records and 1 GiB of wire bytes
    size_t i = 0;
    while (i < n) {
        size_t j = std::min(n, i + (size_t)c->cfg.max_batch_records);
        while (j > i + 1 && offsets[j] - offsets[i] > (1ull << 30)) j = i + (j - i) / 2;
        std::vector<uint64_t> rel;
        const uint64_t* o = offsets + i;
        uint64_t base = offsets[i];
        if (base) {
            rel.resize(j - i + 1);
            for (size_t k = 0; k <= j - i; k++) {
                if (offsets[i + k] < base) return fail(c, FA_ERR_ARG, "offsets must be non-decreasing");
                rel[k] = offsets[i + k] - base;
            }
            o = rel.data();
        }
        if (offsets[j] > len || offsets[j] < base) return fail(c, FA_ERR_ARG, "offsets exceed len");
        const uint8_t* d_buf;
        const uint32_t* d_off;
        int slot = 0;
        int rc = stage_and_upload(c, buf + base, (size_t)(offsets[j] - base), o, j - i, &d_buf, &d_off, &slot);
        if (rc) return rc;
        rc = fa_ingest_device(c, d_buf, (size_t)(offsets[j] - base), d_off, j - i);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(c->stage_ev[slot], c->stream));
        i = j;
    }
    return FA_OK;
}

// ---- decode + project ---------------------------------------------------------------------
static int ensure_columns(fa_ctx* c, size_t n) {
    if (c->col_cap >= n) return FA_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(c->col_block);
    c->col_block = nullptr;
    size_t cap = std::max<size_t>(n, 1 << 16);
    cap = (cap + 63) & ~(size_t)63;
    size_t bytes = cap * (5 * 8 + 7 * 4 + 3 * 16 + 1);
    if (hipMalloc(&c->col_block, bytes) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(columns) failed");
    uint8_t* p = (uint8_t*)c->col_block;
    auto take = [&](size_t elem) {
        uint8_t* r = p;
        p += cap * elem;
        return r;
    };
    c->cols.sampler_address = (uint4*)take(16);
    c->cols.src_addr = (uint4*)take(16);
    c->cols.dst_addr = (uint4*)take(16);
    c->cols.time_received = (uint64_t*)take(8);
    c->cols.time_flow_start = (uint64_t*)take(8);
    c->cols.sampling_rate = (uint64_t*)take(8);
    c->cols.bytes = (uint64_t*)take(8);
    c->cols.packets = (uint64_t*)take(8);
    c->cols.sequence_num = (uint32_t*)take(4);
    c->cols.src_as = (uint32_t*)take(4);
    c->cols.dst_as = (uint32_t*)take(4);
    c->cols.etype = (uint32_t*)take(4);
    c->cols.proto = (uint32_t*)take(4);
    c->cols.src_port = (uint32_t*)take(4);
    c->cols.dst_port = (uint32_t*)take(4);
    c->cols.status = (uint8_t*)take(1);
    c->col_cap = cap;
    return FA_OK;
}

extern "C" int fa_decode_device(fa_ctx* c, const void* d_buf, size_t len, const void* d_off, size_t n,
                                fa_columns* out) {
    FA_ON_DEVICE(c);
    FA_ON_DEVICE(c);
    if (!c || !out) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!d_buf || !d_off || len >= (1ull << 32) || n > c->cfg.max_batch_records || ((uintptr_t)d_buf & 15))
        return fail(c, FA_ERR_ARG, "fa_decode_device: bad buffer");
    int rc = ensure_columns(c, std::max<size_t>(n, 1));
    if (rc) return rc;
    rc = ensure_exotic(c, std::max<size_t>(n, 1));
    if (rc) return rc;
    if (n) {
        KArgs a = make_args(c);
        a.buf = (const uint8_t*)d_buf;
        a.off = (const uint32_t*)d_off;
        a.n = (uint32_t)n;
        a.len = (uint32_t)len;
        a.tile_recs = tile_recs_for(len, n);
        if (c->dev_used == c->dev_pool.size()) {
            if (c->dev_pool.size() >= 1024) {  // bound the pool: fold what is pending
                rc = settle(c);
                if (rc) return rc;
            } else {
                fa_ctx::LaunchEvents e{};
                HIPCHK(c, hipEventCreate(&e.e0));
                HIPCHK(c, hipEventCreate(&e.e1));
                HIPCHK(c, hipEventCreate(&e.e2));
                c->dev_pool.push_back(e);
            }
        }
        rc = launch_tiles<MODE_DECODE>(c, a, tile_grid<MODE_DECODE>(c, a.n, a.tile_recs), &c->dev_pool[c->dev_used++]);
        if (rc) return rc;
    }
    out->time_received = c->cols.time_received;
    out->time_flow_start = c->cols.time_flow_start;
    out->sampling_rate = c->cols.sampling_rate;
    out->bytes = c->cols.bytes;
    out->packets = c->cols.packets;
    out->sequence_num = c->cols.sequence_num;
    out->src_as = c->cols.src_as;
    out->dst_as = c->cols.dst_as;
    out->etype = c->cols.etype;
    out->proto = c->cols.proto;
    out->src_port = c->cols.src_port;
    out->dst_port = c->cols.dst_port;
    out->sampler_address = (const uint8_t*)c->cols.sampler_address;
    out->src_addr = (const uint8_t*)c->cols.src_addr;
    out->dst_addr = (const uint8_t*)c->cols.dst_addr;
    out->status = c->cols.status;
    return FA_OK;
}

extern "C" int fa_decode(fa_ctx* c, const uint8_t* buf, size_t len, const uint64_t* offsets, size_t n,
                         fa_flow_row* out) {
    FA_ON_DEVICE(c);
    FA_ON_DEVICE(c);
    if (!c || (!out && n)) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!offsets) return fail(c, FA_ERR_ARG, "fa_decode: offsets required");
    size_t done = 0;
    while (done < n) {
        size_t j = std::min(n, done + (size_t)std::min<uint32_t>(c->cfg.max_batch_records, 1u << 22));
        while (j > done + 1 && offsets[j] - offsets[done] > (1ull << 30)) j = done + (j - done) / 2;
        size_t m = j - done;
        uint64_t base = offsets[done];
        std::vector<uint64_t> rel(m + 1);
        for (size_t k = 0; k <= m; k++) {
            if (offsets[done + k] < base || offsets[done + k] > len) return fail(c, FA_ERR_ARG, "bad offsets");
            rel[k] = offsets[done + k] - base;
        }
        const uint8_t* d_buf;
        const uint32_t* d_off;
        int slot = 0;
        int rc = stage_and_upload(c, buf + base, (size_t)rel[m], rel.data(), m, &d_buf, &d_off, &slot);
        if (rc) return rc;
        fa_columns cols;
        rc = fa_decode_device(c, d_buf, (size_t)rel[m], d_off, m, &cols);
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        // gather SoA -> AoS on the host
        std::vector<uint64_t> u64(m);
        std::vector<uint32_t> u32(m);
        std::vector<uint8_t> b16(m * 16), st(m);
        fa_flow_row* o = out + done;
        memset(o, 0, m * sizeof(fa_flow_row));
#define GET64(member, src)                                                                  \
    HIPCHK(c, hipMemcpy(u64.data(), src, m * 8, hipMemcpyDeviceToHost));                    \
    for (size_t k = 0; k < m; k++) o[k].member = u64[k];
#define GET32(member, src)                                                                  \
    HIPCHK(c, hipMemcpy(u32.data(), src, m * 4, hipMemcpyDeviceToHost));                    \
    for (size_t k = 0; k < m; k++) o[k].member = u32[k];
#define GET16(member, src)                                                                  \
    HIPCHK(c, hipMemcpy(b16.data(), src, m * 16, hipMemcpyDeviceToHost));                   \
    for (size_t k = 0; k < m; k++) memcpy(o[k].member, &b16[k * 16], 16);
        GET64(time_received, cols.time_received)
        GET64(time_flow_start, cols.time_flow_start)
        GET64(sampling_rate, cols.sampling_rate)
        GET64(bytes, cols.bytes)
        GET64(packets, cols.packets)
        GET32(sequence_num, cols.sequence_num)
        GET32(src_as, cols.src_as)
        GET32(dst_as, cols.dst_as)
        GET32(etype, cols.etype)
        GET32(proto, cols.proto)
        GET32(src_port, cols.src_port)
        GET32(dst_port, cols.dst_port)
        GET16(sampler_address, cols.sampler_address)
        GET16(src_addr, cols.src_addr)
        GET16(dst_addr, cols.dst_addr)
#undef GET64
#undef GET32
#undef GET16
        HIPCHK(c, hipMemcpy(st.data(), cols.status, m, hipMemcpyDeviceToHost));
        for (size_t k = 0; k < m; k++) o[k].status = st[k];
        done = j;
    }
    return FA_OK;
}

// ---- window close: see rows_host.inc (included below) ------------------------------------------------
// timeslot -> bucket range.  With sub-windows a window [timeslot, timeslot+window_secs)
// is the sum of window_secs/gran consecutive sub-buckets (sliding windows share them).
static bool bucket_range(const fa_ctx* c, uint32_t timeslot, uint32_t& lo, uint32_t& hi) {
    if (timeslot == 0xFFFFFFFFu) {
        lo = 0;
        hi = 0xFFFFFFFFu;
        return true;
    }
    if (timeslot % c->gran) return false;
    lo = timeslot / c->gran;
    hi = lo + c->cfg.window_secs / c->gran;
    return true;
}


// ---- RowBinary sink -------------------------------------------------------------------------------
// ClickHouse RowBinary: fixed-width little-endian integers, arrays as LEB128 length + elements; Nested
// columns travel as one array per sub-column (format restated from the ClickHouse documentation; no
// server in this image to load it into).
extern "C" int fa_rows_to_rowbinary(const fa_row5m* rows, size_t n, uint8_t* out, size_t cap, size_t* bytes_out) {
    if ((!rows && n) || !bytes_out) return FA_ERR_ARG;
    const size_t need = n * (size_t)FA_ROWBINARY_ROW5M_BYTES;
    *bytes_out = need;
    if (need > cap || (!out && need)) return FA_ERR_CAPACITY;
    uint8_t* p = out;
    auto put = [&](uint64_t v, int bytes) {
        for (int i = 0; i < bytes; i++) *p++ = (uint8_t)(v >> (8 * i));
    };
    for (size_t i = 0; i < n; i++) {
        const fa_row5m& r = rows[i];
        if (r.date > 0xFFFFu) return FA_ERR_ARG;
        put(r.date, 2);
        put(r.timeslot, 4);
        put(r.src_as, 4);
        put(r.dst_as, 4);
        put(1, 1); put(r.etype, 4);    // ETypeMap.EType   [EType]
        put(1, 1); put(r.bytes, 8);    // ETypeMap.Bytes   [Bytes]
        put(1, 1); put(r.packets, 8);  // ETypeMap.Packets [Packets]
        put(1, 1); put(r.count, 8);    // ETypeMap.Count   [Count]
        put(r.bytes, 8);
        put(r.packets, 8);
        put(r.count, 8);
    }
    return FA_OK;
}

// ---- dashboard address rendering (viz-ch.json:233,479) ---------------------------------------------
// IPv6NumToString is restated from BIND's inet_ntop6 (the algorithm ClickHouse's formatIPv6 follows and glibc
// ships; the CPU test-suite checks this function against glibc on random and structured addresses).
extern "C" int fa_format_addr(const uint8_t addr[16], uint32_t etype, char* out, size_t cap) {
    if (!addr || !out) return FA_ERR_ARG;
    char tmp[FA_ADDR_STRLEN + 2];
    char* p = tmp;
    auto dec = [&](unsigned v) {
        if (v >= 100) *p++ = (char)('0' + v / 100);
        if (v >= 10) *p++ = (char)('0' + v / 10 % 10);
        *p++ = (char)('0' + v % 10);
    };
    auto dotted = [&](const uint8_t* b) {
        for (int i = 0; i < 4; i++) {
            if (i) *p++ = '.';
            dec(b[i]);
        }
    };
    if (etype == 0x800) {
        dotted(addr);
    } else {
        unsigned w[8];
        for (int i = 0; i < 8; i++) w[i] = ((unsigned)addr[2 * i] << 8) | addr[2 * i + 1];
        int best = -1, best_len = 0, cur = -1, cur_len = 0;
        for (int i = 0; i < 8; i++) {
            if (w[i] == 0) {
                if (cur < 0) cur = i, cur_len = 1;
                else cur_len++;
            } else if (cur >= 0) {
                if (best < 0 || cur_len > best_len) best = cur, best_len = cur_len;
                cur = -1;
            }
        }
        if (cur >= 0 && (best < 0 || cur_len > best_len)) best = cur, best_len = cur_len;
        if (best >= 0 && best_len < 2) best = -1;
        bool done = false;
        for (int i = 0; i < 8 && !done; i++) {
            if (best >= 0 && i >= best && i < best + best_len) {
                if (i == best) *p++ = ':';
                continue;
            }
            if (i) *p++ = ':';
            if (i == 6 && best == 0 && (best_len == 6 || (best_len == 5 && w[5] == 0xffffu))) {
                dotted(addr + 12);  // encapsulated IPv4
                done = true;
                break;
            }
            static const char hex[] = "0123456789abcdef";
            bool lead = true;
            for (int s = 12; s >= 0; s -= 4) {
                unsigned d = (w[i] >> s) & 15u;
                if (d || !lead || s == 0) *p++ = hex[d], lead = false;
            }
        }
        if (!done && best >= 0 && best + best_len == 8) *p++ = ':';
    }
    *p++ = 0;
    const size_t need = (size_t)(p - tmp);
    if (need > cap) return FA_ERR_CAPACITY;
    memcpy(out, tmp, need);
    return FA_OK;
}

// ---- wide key sets: window close and dashboard reads -----------------------------------------------
// rows (already packed keys) -> device -> wmerge_kernel
static int merge_wide(fa_ctx* c, const std::vector<WRow>& rows) {
    if (!c->wtab) return fail(c, FA_ERR_ARG, "key set not enabled");
    if (rows.empty()) return FA_OK;
    WRow* d = nullptr;
    if (hipMalloc(&d, rows.size() * sizeof(WRow)) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpyAsync(d, rows.data(), rows.size() * sizeof(WRow), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        KArgs a = make_args(c);
        hipLaunchKernelGGL(wmerge_kernel, dim3(256), dim3(256), 0, c->stream, d, (uint32_t)rows.size(), a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) {
        c->err = std::string("merge_wide: ") + hipGetErrorString(e);
        return FA_ERR_HIP;
    }
    return settle(c);
}

extern "C" int fa_merge_rows_app(fa_ctx* c, const fa_row_app* rows, size_t n) {
    FA_ON_DEVICE(c);
    if (!c || (!rows && n)) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!(c->cfg.key_sets & FA_KEYS_ADDR_PORT_PROTO)) return fail(c, FA_ERR_ARG, "FA_KEYS_ADDR_PORT_PROTO not enabled");
    std::vector<WRow> w(n);
    for (size_t i = 0; i < n; i++) {
        if (rows[i].timeslot % c->gran) return fail(c, FA_ERR_ARG, "fa_merge_rows_app: timeslot not on this ctx's bucket grid");
        uint64_t lo, hi;
        memcpy(&lo, rows[i].src_addr, 8);
        memcpy(&hi, rows[i].src_addr + 8, 8);
        WKey k;
        wkey_pack(WK_APP, rows[i].timeslot / c->gran, lo, hi, rows[i].dst_port, rows[i].proto, k);
        w[i] = WRow{{k.w[0], k.w[1], k.w[2], k.w[3]}, rows[i].bytes, rows[i].packets, rows[i].count};
    }
    return merge_wide(c, w);
}

__global__ void port_merge_kernel(const fa_port_row* rows, uint32_t n, ulonglong2* hist) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned long long* e = reinterpret_cast<unsigned long long*>(&hist[rows[i].port]);
        if (rows[i].weight) atomicAdd(e, (unsigned long long)rows[i].weight);
        atomicAdd(e + 1, (unsigned long long)rows[i].count);
    }
}

extern "C" int fa_merge_ports(fa_ctx* c, int dst, const fa_port_row* rows, size_t n) {
    FA_ON_DEVICE(c);
    if (!c || (!rows && n) || (dst != 0 && dst != 1)) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!c->port_hist) return fail(c, FA_ERR_ARG, "FA_KEYS_PORT_HIST not enabled");
    std::vector<fa_port_row> small;
    std::vector<WRow> big;
    for (size_t i = 0; i < n; i++) {
        if (rows[i].port < PORT_DENSE) {
            small.push_back(rows[i]);
        } else {
            WKey k;
            wkey_pack(dst ? WK_DSTPORT : WK_SRCPORT, 0, 0, 0, rows[i].port, 0, k);
            big.push_back(WRow{{k.w[0], k.w[1], k.w[2], k.w[3]}, rows[i].weight, 0, rows[i].count});
        }
    }
    if (!small.empty()) {
        fa_port_row* d = nullptr;
        if (hipMalloc(&d, small.size() * sizeof(fa_port_row)) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc failed");
        hipError_t e = hipMemcpyAsync(d, small.data(), small.size() * sizeof(fa_port_row), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(port_merge_kernel, dim3(64), dim3(256), 0, c->stream, d, (uint32_t)small.size(),
                               c->port_hist + (size_t)dst * PORT_DENSE);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d);
        if (e != hipSuccess) {
            c->err = std::string("fa_merge_ports: ") + hipGetErrorString(e);
            return FA_ERR_HIP;
        }
    }
    return merge_wide(c, big);
}

extern "C" int fa_merge_minutes(fa_ctx* c, const fa_minute_row* rows, size_t n) {
    FA_ON_DEVICE(c);
    if (!c || (!rows && n)) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    if (!(c->cfg.key_sets & FA_KEYS_MINUTE_SERIES)) return fail(c, FA_ERR_ARG, "FA_KEYS_MINUTE_SERIES not enabled");
    std::vector<WRow> w(n);
    for (size_t i = 0; i < n; i++) {
        if (rows[i].minute % 60u) return fail(c, FA_ERR_ARG, "fa_merge_minutes: not a minute boundary");
        WKey k;
        wkey_pack(WK_MINUTE, 0, 0, 0, rows[i].minute / 60u, 0, k);
        w[i] = WRow{{k.w[0], k.w[1], k.w[2], k.w[3]}, rows[i].weight, 0, rows[i].count};
    }
    return merge_wide(c, w);
}

extern "C" int fa_dashboard_reset(fa_ctx* c) {
    FA_ON_DEVICE(c);
    if (!c) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    int rc = settle(c);
    if (rc) return rc;
    if (c->port_hist) HIPCHK(c, hipMemsetAsync(c->port_hist, 0, sizeof(ulonglong2) * 2 * PORT_DENSE, c->stream));
    if (c->wtab && (c->cfg.key_sets & (FA_KEYS_PORT_HIST | FA_KEYS_MINUTE_SERIES)))
        return rebuild_wide(c, c->wcap_log2, (1u << WK_SRCPORT) | (1u << WK_DSTPORT) | (1u << WK_MINUTE), 0, 0);
    return FA_OK;
}

// ---- sketches -----------------------------------------------------------------------------------
// Sums the sketch copies into copy 0 (sinks.cuh, cms_add).  Every reader of a sketch calls this first.
static int cms_fold(fa_ctx* c) {
    if (!c->cms_dirty) return FA_OK;
    for (unsigned long long* p : {c->cms_src, c->cms_dst})
        if (p) hipLaunchKernelGGL(cms_fold_kernel, dim3(2048), dim3(256), 0, c->stream, p, c->cms_words);
    HIPCHK(c, hipGetLastError());
    c->cms_dirty = false;
    return FA_OK;
}

// the sketch readers answer from: the merged (all-rank) view while it is valid, the ctx's own sketch otherwise
static unsigned long long* cms_of(fa_ctx* c, uint32_t key_set) {
    if (key_set == FA_KEYS_SRCADDR_CMS) return c->merged_valid && c->cms_src_m ? c->cms_src_m : c->cms_src;
    if (key_set == FA_KEYS_DSTADDR_CMS) return c->merged_valid && c->cms_dst_m ? c->cms_dst_m : c->cms_dst;
    return nullptr;
}
static int ensure_merged_view(fa_ctx* c) {
    for (int d = 0; d < 2; d++) {
        unsigned long long* own = d ? c->cms_dst : c->cms_src;
        unsigned long long** m = d ? &c->cms_dst_m : &c->cms_src_m;
        if (own && !*m && hipMalloc(m, c->cms_words * 8) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc(merged sketch view) failed");
    }
    return FA_OK;
}

extern "C" int fa_cms_read(fa_ctx* c, uint32_t key_set, uint64_t* out, size_t cap_words) {
    FA_ON_DEVICE(c);
    if (!c || !out) return FA_ERR_ARG;
    unsigned long long* p = cms_of(c, key_set);
    if (!p) return fail(c, FA_ERR_ARG, "key set not enabled");
    if (cap_words < c->cms_words) return fail(c, FA_ERR_CAPACITY, "sketch buffer too small");
    int rc = settle(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(out, p, c->cms_words * 8, hipMemcpyDeviceToHost));
    return FA_OK;
}

extern "C" int fa_cms_reset(fa_ctx* c, uint32_t key_set) {
    FA_ON_DEVICE(c);
    if (!c) return FA_ERR_ARG;
    unsigned long long* p = key_set == FA_KEYS_SRCADDR_CMS ? c->cms_src : key_set == FA_KEYS_DSTADDR_CMS ? c->cms_dst : nullptr;
    if (!p) return fail(c, FA_ERR_ARG, "key set not enabled");
    c->merged_valid = false;
    HIPCHK(c, hipMemsetAsync(p, 0, c->cms_words * 8 * CMS_REPLICAS, c->stream));
    KeySlot* ks = key_set == FA_KEYS_SRCADDR_CMS ? c->ks_src : c->ks_dst;
    if (ks) HIPCHK(c, hipMemsetAsync(ks, 0, sizeof(KeySlot) << c->ks_log2, c->stream));
    return FA_OK;
}

static uint32_t cms_column_host(const uint8_t key[16], uint64_t seed, uint32_t wl2, uint32_t row) {
    uint64_t lo, hi, h1, h2;
    memcpy(&lo, key, 8);
    memcpy(&hi, key + 8, 8);
    cms_hash2(lo, hi, seed, h1, h2);  // (sinks.cuh: the one definition, host and device)
    return cms_column(cms_key(h1, h2, wl2), row, wl2);
}

extern "C" int fa_cms_query(fa_ctx* c, uint32_t key_set, const uint8_t key[16], uint64_t* weight) {
    FA_ON_DEVICE(c);
    if (!c || !key || !weight) return FA_ERR_ARG;
    unsigned long long* p = cms_of(c, key_set);
    if (!p) return fail(c, FA_ERR_ARG, "key set not enabled");
    int rc = settle(c);
    if (rc) return rc;
    uint64_t best = ~0ull;
    for (uint32_t r = 0; r < c->cfg.cms_depth; r++) {
        size_t idx = ((size_t)r << c->cfg.cms_width_log2) + cms_column_host(key, c->cfg.cms_seed, c->cfg.cms_width_log2, r);
        unsigned long long v;
        HIPCHK(c, hipMemcpy(&v, p + idx, 8, hipMemcpyDeviceToHost));
        best = std::min<uint64_t>(best, v);
    }
    *weight = best;
    return FA_OK;
}

extern "C" int fa_topk_merge_keys(fa_ctx* c, uint32_t key_set, const uint8_t* keys, size_t n) {
    FA_ON_DEVICE(c);
    if (!c || (!keys && n)) return FA_ERR_ARG;
    if (c->sticky) return c->sticky;
    KeySlot* ks = key_set == FA_KEYS_SRCADDR_CMS ? c->ks_src : key_set == FA_KEYS_DSTADDR_CMS ? c->ks_dst : nullptr;
    if (!ks) return fail(c, FA_ERR_ARG, "fa_topk_merge_keys: key set not enabled");
    if (!n) return FA_OK;
    uint4* d = nullptr;
    if (hipMalloc(&d, n * 16) != hipSuccess) return fail(c, FA_ERR_NOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpyAsync(d, keys, n * 16, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        KArgs a = make_args(c);
        hipLaunchKernelGGL(keyset_merge_kernel, dim3(256), dim3(256), 0, c->stream, d, (uint32_t)n, ks, a);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) {
        c->err = std::string("fa_topk_merge_keys: ") + hipGetErrorString(e);
        return FA_ERR_HIP;
    }
    return FA_OK;
}

#include "rows_host.inc"

extern "C" int fa_device_state_get(fa_ctx* c, fa_device_state* out) {
    FA_ON_DEVICE(c);
    if (!c || !out) return FA_ERR_ARG;
    int rc = settle(c);
    if (rc) return rc;
    out->cms_src = c->cms_src;
    out->cms_dst = c->cms_dst;
    out->cms_words = c->cms_words;
    out->port_hist = c->port_hist;
    out->port_hist_words = c->port_hist ? (size_t)4 * PORT_DENSE : 0;
    rc = ensure_merged_view(c);
    if (rc) return rc;
    out->cms_src_merged = c->cms_src_m;
    out->cms_dst_merged = c->cms_dst_m;
    return FA_OK;
}

extern "C" int fa_merged_view_set(fa_ctx* c, int valid) {
    FA_ON_DEVICE(c);
    if (!c) return FA_ERR_ARG;
    if (valid && ((c->cms_src && !c->cms_src_m) || (c->cms_dst && !c->cms_dst_m))) return fail(c, FA_ERR_ARG, "fa_merged_view_set: no merged view (call fa_device_state_get first)");
    c->merged_valid = valid != 0;
    return FA_OK;
}

// RCCL is bound lazily so that libflowagg.so loads on hosts without librccl.
#include <dlfcn.h>
extern "C" int fa_merge_allreduce(fa_ctx* c, void* comm) {
    FA_ON_DEVICE(c);
    if (!c || !comm) return FA_ERR_ARG;
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_fn fn = nullptr;
    if (!fn) {
        void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (h) fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!fn) return fail(c, FA_ERR_UNSUPPORTED, "librccl.so / ncclAllReduce not found");
    }
    int rc = settle(c);  // (folds the sketch copies into copy 0)
    if (rc) return rc;
    rc = ensure_merged_view(c);
    if (rc) return rc;
    const int ncclUint64 = 5, ncclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t
    c->merged_valid = false;
    if (c->cms_src && fn(c->cms_src, c->cms_src_m, c->cms_words, ncclUint64, ncclSum, comm, c->stream) != 0)
        return fail(c, FA_ERR_HIP, "ncclAllReduce(cms_src) failed");
    if (c->cms_dst && fn(c->cms_dst, c->cms_dst_m, c->cms_words, ncclUint64, ncclSum, comm, c->stream) != 0)
        return fail(c, FA_ERR_HIP, "ncclAllReduce(cms_dst) failed");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->merged_valid = true;
    return FA_OK;
}

extern "C" int fa_stats(fa_ctx* c, fa_stats_t* out) {
    FA_ON_DEVICE(c);
    if (!c || !out) return FA_ERR_ARG;
    int rc = settle(c);
    c->stats.wide_log_chunks = c->wlog.size();
    uint64_t lb = 0, lr = 0;
    for (const auto& k : c->wlog) {
        lb += k.seg_bytes + (k.counts_cap + 4) * sizeof(uint32_t);
        lr += k.n;
    }
    for (const auto& k : c->wlog_free) lb += k.seg_bytes + (k.counts_cap + 4) * sizeof(uint32_t);
    c->stats.wide_log_bytes = lb;
    c->stats.wide_log_records = lr;
    c->stats.wide_log_recorded = c->wlog_recorded;
    c->stats.wide_log_folded = c->wlog_folded;
    c->stats.wide_log_replayed = c->wlog_replayed;
    c->stats.wide_log_dropped = c->wlog_dropped;
    c->stats.wide_log_watermark_moves = c->wlog_wm_moves;
    c->stats.wide_log_nomem_folds = c->wlog_nomem_folds;
    c->stats.wide_log_mode = (c->wide_mode == 3 || (c->wide_mode == 0 && c->wide_defer)) ? 1 : 0;
    *out = c->stats;
    return rc;
}

// ---- synthetic producer ------------------------------------------------------------------------
extern "C" int fa_mock_generate_device(fa_ctx* c, const fa_mock_params* g, uint64_t i0, uint64_t n, void* d_buf,
                                       size_t cap, void* d_off, uint64_t* bytes_out) {
    FA_ON_DEVICE(c);
    FA_ON_DEVICE(c);
    if (!c || !g || !d_buf || !d_off || n == 0 || n >= (1ull << 31)) return FA_ERR_ARG;
    uint32_t* off = (uint32_t*)d_off;
    uint32_t* len = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    HIPCHK(c, hipMalloc(&len, (n + 1) * sizeof(uint32_t)));
    int rc = FA_OK;
    do {
        hipLaunchKernelGGL(gen_len_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, *g, i0, (uint32_t)n, len);
        if (hipMemsetAsync(len + n, 0, 4, c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, len, off, (int)(n + 1), c->stream);
        if (hipMalloc(&tmp, tmp_bytes) != hipSuccess) { rc = FA_ERR_NOMEM; break; }
        if (hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, len, off, (int)(n + 1), c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        uint32_t total = 0;
        if (hipMemcpyAsync(&total, off + n, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        if ((size_t)total + 32 > cap) { rc = FA_ERR_CAPACITY; if (bytes_out) *bytes_out = total; break; }
        hipLaunchKernelGGL(gen_write_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, *g, i0, (uint32_t)n, off, (uint8_t*)d_buf);
        if (hipMemsetAsync((uint8_t*)d_buf + total, 0, 32, c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        if (bytes_out) *bytes_out = total;
    } while (0);
    (void)hipFree(len);
    (void)hipFree(tmp);
    if (rc == FA_ERR_HIP) c->err = std::string("fa_mock_generate_device: ") + hipGetErrorString(hipGetLastError());
    if (rc == FA_ERR_CAPACITY) c->err = "fa_mock_generate_device: buffer too small (needs bytes + 32 slack)";
    return rc;
}

extern "C" int fa_mock_generate_host(const fa_mock_params* g, uint64_t i0, uint64_t n, uint8_t* buf, size_t cap,
                                     uint64_t* offsets, uint64_t* bytes_out) {
    if (!g || !buf) return FA_ERR_ARG;
    size_t pos = 0;
    uint8_t tmp[FA_MOCK_MAX_RECORD];
    for (uint64_t k = 0; k < n; k++) {
        uint32_t l = gen_encode(*g, i0 + k, tmp);
        if (pos + l > cap) return FA_ERR_CAPACITY;
        memcpy(buf + pos, tmp, l);
        if (offsets) offsets[k] = pos;
        pos += l;
    }
    if (offsets) offsets[n] = pos;
    if (bytes_out) *bytes_out = pos;
    return FA_OK;
}
