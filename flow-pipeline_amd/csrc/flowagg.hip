// flowagg.hip - C-ABI implementation of libflowagg (include/flowagg.h).
//
// Host side of the MI355X flow-aggregation stage: owns the HIP stream, the
// device group-by table, sketches, pinned staging and the SoA projection
// buffers; launches the gfx950 kernels of kernels.cuh.  Mirrors the shape of
// the reference sink (inserter/inserter.go:90-165: buffer -> flush) with the
// ClickHouse semantics of compose/clickhouse/create.sh:5-110.
// There is deliberately no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>
#include <sched.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
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
    uint64_t seen_retried = 0, seq_until = 0;  // launches before batch seq_until run the learnt-field-order variant (format_feedback)
    int seq_mode = 0;                          // FA_SEQ: 1 always, 2 never
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
    void* cut_buf = nullptr;         // fa_read_window_app48 into page-locked memory: the window's rows cut in two by key (rows_host.inc)
    size_t cut_cap = 0;
    hipStream_t copy_stream = nullptr;  // ... the first half's rows leave on it while the second half is sorted
    // contexts with a sketch: the flows_5m tuple aggregation of a launch runs on a stream of its own BESIDE the sketch fold (and the
    // candidates mode's boundary kernels) - it touches neither sketch nor set, they touch neither tuple segments nor table (launch_tiles)
    hipStream_t cand_stream = nullptr;
    hipEvent_t cand_ev[2] = {nullptr, nullptr};  // ingest + second-chance kernels done (main -> side), aggregation done (side -> main)
    void* fs_scratch = nullptr;      // device-side framing (framing.cuh): block starts, exits, counts, bases, error / trust flags, sub-block entries, counters
    size_t fs_scratch_cap = 0;
    void* fs_off = nullptr;          // ... the offsets it produces
    size_t fs_off_cap = 0;
    void* wl_scratch = nullptr;      // window reads of log chunks: per-segment counts, their scan, hipcub storage
    size_t wl_scratch_cap = 0;
    void* read_clk = nullptr;        // FA_VERBOSE: the ReadClock of the read in progress (collect's own phases report into it)
    void* part_buf = nullptr;        // fa_rows_partition_device: the rows grouped by destination rank
    size_t part_cap = 0;
    unsigned int* part_cnt = nullptr;  // [3][RPART_MAX_WORLD]: counts, starts, cursors
    // fa_group_*: the buffer the group's exchange writes into (peer copies from the other members), owned by the ctx so that it
    // can be reserved where the rows it will hold come into being (reserve_window_read) instead of inside the first close
    void* xch_buf = nullptr;
    size_t xch_cap = 0;
    uint32_t group_members = 0;        // > 0 while the ctx is a member of a group (of that many contexts)
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
    // candidates mode (cfg.topk_mode = FA_TOPK_CANDIDATES; maintenance.cuh "candidates mode"): one bit per sketch counter, rebuilt
    // behind every ingest launch; the sets then hold candidates only
    // fa_topk: a bin the k-th estimate of a sketch is known to reach (rows_host.inc: the k-th row of the last read; valid for
    // k' <= tk_lb_k on the same view - own sketch or merged - until fa_cms_reset)
    uint32_t tk_lb_bin[2] = {0, 0};
    size_t tk_lb_k[2] = {0, 0};
    bool tk_lb_merged[2] = {false, false};
    unsigned short* cand_chunkmax = nullptr;  // candidates mode: scratch of the boundary's scan (2 sets)
    uint32_t* cand_bits[2] = {nullptr, nullptr};
    size_t cand_bits_bytes = 0;
    CandState* cand_state = nullptr;  // [2]

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
    a.cand_src = c->cand_bits[0];
    a.cand_dst = c->cand_bits[1];
    a.cms_nrep = c->cand_state ? 1u : CMS_REPLICAS;
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
    if (cfg.topk_track == 0) cfg.topk_track = 256;
    if (cfg.max_batch_records == 0) cfg.max_batch_records = AGG_MAX_BATCH;
    if (cfg.max_batch_records > AGG8_MAX_BATCH) cfg.max_batch_records = AGG8_MAX_BATCH;  // (launches of wide tuples are split at 2^24)
    uint32_t gran = cfg.subwindow_secs ? cfg.subwindow_secs : cfg.window_secs;
    if (cfg.device < 0 || cfg.device >= ndev || gran < 60 || 86400 % gran != 0 ||
        cfg.window_secs % gran != 0 || 86400 % cfg.window_secs != 0 || cfg.table_capacity_log2 < 10 ||
        cfg.table_capacity_log2 > 30 || cfg.cms_depth > 16 || cfg.cms_width_log2 < 4 ||
        cfg.cms_width_log2 > 28 || (cfg.key_sets & ~63u) || cfg.topk_mode > FA_TOPK_CANDIDATES || cfg.topk_capacity_log2 < 8 || cfg.topk_capacity_log2 > 30 ||
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
    if (const char* d = getenv("FA_SEQ")) c->seq_mode = !strcmp(d, "1") ? 1 : !strcmp(d, "0") ? 2 : 0;  // learnt-field-order kernel: always / never (default: by the counters)
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
        // (the side stream of the ingest path - launch_tiles: the flows_5m tuple aggregation runs on it beside the sketch fold - is
        // created here: a stream costs milliseconds, not a thing for the first ingest launch)
        const bool want_side = cfg.topk_mode == FA_TOPK_CANDIDATES && (getenv("FA_AGG_SIDE") == nullptr || strcmp(getenv("FA_AGG_SIDE"), "0") != 0);
        if (want_side) {  // (only the candidates mode uses it: every stream of a process competes for its few hardware queues)
            if ((e = hipStreamCreateWithFlags(&c->cand_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate(side stream)", e);
            for (int i = 0; i < 2; i++)
                if ((e = hipEventCreateWithFlags(&c->cand_ev[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
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
        if (cfg.topk_mode == FA_TOPK_CANDIDATES) {
            c->cand_bits_bytes = std::max<size_t>(c->cms_words / 8, 8) + 8;
            for (int d = 0; d < 2; d++)
                if (cfg.key_sets & (d ? FA_KEYS_DSTADDR_CMS : FA_KEYS_SRCADDR_CMS)) {
                    if ((e = hipMalloc(&c->cand_bits[d], c->cand_bits_bytes)) != hipSuccess) return bail("hipMalloc(candidate bits)", e);
                    if ((e = hipMemsetAsync(c->cand_bits[d], 0, c->cand_bits_bytes, c->stream)) != hipSuccess) return bail("memset", e);
                }
            if ((e = hipMalloc(&c->cand_chunkmax, 2 * ((sizeof(unsigned short) << c->ks_log2 >> 6) + 256))) != hipSuccess) return bail("hipMalloc(candidate scan scratch)", e);
            if ((e = hipMalloc(&c->cand_state, 2 * sizeof(CandState))) != hipSuccess) return bail("hipMalloc(candidate state)", e);
            if ((e = hipMemsetAsync(c->cand_state, 0, 2 * sizeof(CandState), c->stream)) != hipSuccess) return bail("memset", e);
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
    (void)hipFree(c->cut_buf);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->cand_stream) (void)hipStreamDestroy(c->cand_stream);
    for (int i = 0; i < 2; i++)
        if (c->cand_ev[i]) (void)hipEventDestroy(c->cand_ev[i]);
    (void)hipFree(c->m_out[0]);
    (void)hipFree(c->m_out[1]);
    (void)hipFree(c->fs_scratch);
    (void)hipFree(c->fs_off);
    (void)hipFree(c->wl_scratch);
    (void)hipFree(c->part_buf);
    (void)hipFree(c->part_cnt);
    (void)hipFree(c->xch_buf);
    if (c->h_rows) (void)hipHostFree(c->h_rows);
    (void)hipFree(c->cms_src);
    (void)hipFree(c->cms_dst);
    (void)hipFree(c->cms_src_m);
    (void)hipFree(c->cms_dst_m);
    (void)hipFree(c->ks_src);
    (void)hipFree(c->ks_dst);
    (void)hipFree(c->cand_bits[0]);
    (void)hipFree(c->cand_bits[1]);
    (void)hipFree(c->cand_state);
    (void)hipFree(c->cand_chunkmax);
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

#include "maintain_host.inc"
#include "ingest_host.inc"
#include "decode_host.inc"
#include "format_host.inc"
#include "state_host.inc"

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


#include "rows_host.inc"
#include "group_host.inc"

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
    if (c->cand_state && rc == FA_OK) {
        CandState h[2];
        if (hipMemcpy(h, c->cand_state, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            for (int d = 0; d < 2; d++) {
                c->stats.topk_theta[d] = h[d].theta;
                c->stats.topk_candidates[d] = h[d].sel[2];
            }
    }
    *out = c->stats;
    return rc;
}

// ---- synthetic producer ------------------------------------------------------------------------
// (the offsets are 32-bit: the lengths are summed in 64 bits first, a call that would write 4 GiB or more is refused)
__global__ __launch_bounds__(256) void mock_len_sum_kernel(const uint32_t* len, uint32_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += len[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += (unsigned long long)__shfl_xor((long long)s, o);
    if (__lane_id() == 0 && s) atomicAdd(out, s);
}
extern "C" int fa_mock_generate_device(fa_ctx* c, const fa_mock_params* g, uint64_t i0, uint64_t n, void* d_buf,
                                       size_t cap, void* d_off, uint64_t* bytes_out) {
    FA_ON_DEVICE(c);
    FA_ON_DEVICE(c);
    if (!c || !g || !d_buf || !d_off || n == 0 || n >= (1ull << 31)) return FA_ERR_ARG;
    uint32_t* off = (uint32_t*)d_off;
    uint32_t* len = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    HIPCHK(c, hipMalloc(&len, (n + 1) * sizeof(uint32_t) + 16));
    int rc = FA_OK;
    bool too_big = false;
    do {
        hipLaunchKernelGGL(gen_len_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, *g, i0, (uint32_t)n, len);
        if (hipMemsetAsync(len + n, 0, 4, c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        unsigned long long* sum64 = reinterpret_cast<unsigned long long*>(len + ((n + 1 + 1) & ~(uint64_t)1));  // (8-byte aligned, behind the lengths)
        unsigned long long h_sum = 0;
        if (hipMemsetAsync(sum64, 0, 8, c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        hipLaunchKernelGGL(mock_len_sum_kernel, dim3(1024), dim3(256), 0, c->stream, (const uint32_t*)len, (uint32_t)n, sum64);
        if (hipMemcpyAsync(&h_sum, sum64, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = FA_ERR_HIP; break; }
        if (h_sum >= (1ull << 32) - 64) { rc = FA_ERR_ARG; too_big = true; if (bytes_out) *bytes_out = h_sum; break; }
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
    if (too_big) c->err = "fa_mock_generate_device: 4 GiB or more of records in one call (device offsets are 32-bit): generate fewer records per call";
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
