/*
 * flowagg.h - C-ABI of libflowagg: the MI355X (gfx950) flow-aggregation stage.
 *
 * The reference (cloudflare/flow-pipeline) has no plugin/FFI interface; this ABI
 * is what a Go Kafka consumer binds through cgo to replace the
 * ClickHouse path `flows -> flows_raw -> flows_5m`
 * (compose/clickhouse/create.sh:5-110) while keeping the inserter's shape
 * (inserter/inserter.go:113-196: buffer -> flush -> sink).
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; device pointers travel as
 *    `const void*` / `void*` and are documented as such;
 *  - every function returns 0 (FA_OK) or a negative fa_status; text via
 *    fa_last_error();
 *  - the library never keeps a caller pointer after a call returns (cgo rule):
 *    fa_ingest copies into library-owned pinned staging before returning;
 *  - one fa_ctx per (Kafka partition, GPU).  A ctx is not thread-safe; distinct
 *    ctxs are independent (sarama runs one ConsumeClaim goroutine per claimed
 *    partition, inserter.go:176);
 *  - malformed records are counted and dropped, never fatal
 *    (inserter.go:125-126); sink/resource errors are returned (the reference
 *    log.Fatal()s on them, inserter.go:102-105);
 *  - there is NO CPU fallback: without a HIP device fa_create fails.
 */
#ifndef FLOWAGG_H
#define FLOWAGG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FA_ABI_VERSION 8

typedef struct fa_ctx fa_ctx;

typedef enum {
    FA_OK = 0,
    FA_ERR_ARG = -1,        /* bad argument */
    FA_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error at create */
    FA_ERR_HIP = -3,        /* HIP runtime error (text in fa_last_error) */
    FA_ERR_NOMEM = -4,
    FA_ERR_TABLE_FULL = -5, /* a table cannot grow further (2^30 slots) or the distinct-address set behind fa_topk is
                               full.  Aggregates are never dropped on the way there: updates that meet a full table
                               are parked (the spill buffers hold everything the host may have in flight: 2 x
                               max_batch_records per table) and replayed after the table has grown - the host looks
                               at the device counters before every launch. */
    FA_ERR_CAPACITY = -6,   /* caller's output buffer too small; *n_out = required */
    FA_ERR_FRAMING = -7,    /* offsets==NULL and the stream is not a chain of framed records */
    FA_ERR_UNSUPPORTED = -8
} fa_status;

/* key_sets bitmask */
enum {
    FA_KEYS_AS_PAIR = 1u,     /* flows_5m: (Date,Timeslot,SrcAS,DstAS,EType) - create.sh:92-110 */
    FA_KEYS_SRCADDR_CMS = 2u, /* Count-Min sketch over SrcAddr, weight Bytes*SamplingRate (viz-ch.json:233) */
    FA_KEYS_DSTADDR_CMS = 4u, /* same over DstAddr (viz-ch.json:479) */
    FA_KEYS_ADDR_PORT_PROTO = 8u, /* exact (Date,Timeslot,SrcAddr,DstPort,Proto) -> sum(Bytes), sum(Packets), count():
                                     second concurrent key set of BASELINE config 5, same Date/Timeslot rule as
                                     flows_5m (create.sh:92-110); rows via fa_*_window_app */
    FA_KEYS_PORT_HIST = 16u,      /* exact sum(Bytes*SamplingRate), count() GROUP BY SrcPort and GROUP BY DstPort
                                     (viz-ch.json:358,604); dense below 65536, hashed above (UInt32 column) */
    FA_KEYS_MINUTE_SERIES = 32u   /* sum(Bytes*SamplingRate) GROUP BY toStartOfMinute(TimeFlowStart) (viz-ch.json:74) */
};

typedef struct {
    int32_t device;               /* HIP device ordinal */
    uint32_t window_secs;         /* 300 = toStartOfFiveMinute (create.sh:96); must divide 86400 */
    uint32_t subwindow_secs;      /* 0, or a divisor of window_secs (60 = toStartOfMinute, viz-ch.json:74):
                                     rows are kept per sub-bucket so sliding windows can be formed */
    uint32_t table_capacity_log2; /* slots of the device group-by table (64 B each); 0 -> 20 */
    uint32_t cms_depth;           /* 0 -> 4 */
    uint32_t cms_width_log2;      /* 0 -> 20 */
    uint64_t cms_seed;
    uint32_t key_sets;            /* 0 -> FA_KEYS_AS_PAIR */
    int32_t framed;               /* 1: each record is varint(len)||payload (-proto.fixedlen=true,
                                     mocker.go:98-101); 0: bare payload (mocker.go:96-97) */
    uint32_t max_batch_records;   /* upper bound on n per ingest call; 0 -> 1<<24; at most (1<<25)-1 (a call whose records
                                     leave as wide tuples is split into launches of <= 1<<24) */
    uint32_t topk_capacity_log2;  /* slots of each distinct-address set behind fa_topk (32 B each); 0 -> 20 */
    uint32_t wide_capacity_log2;  /* slots of the wide-key table (64 B each) behind FA_KEYS_ADDR_PORT_PROTO /
                                     PORT_HIST / MINUTE_SERIES; 0 -> 20; grows by itself like the flows_5m table */
    uint32_t topk_mode;           /* ABI 7.  FA_TOPK_EXACT (0): fa_topk ranks EVERY address ever ingested (the ranking the dashboards
                                     compute, viz-ch.json:233,479) - the distinct-address sets hold every key (2^(universe + 1) slots)
                                     and the ingest path looks each non-repeating address up in them.
                                     FA_TOPK_CANDIDATES (1): the standard Count-Min heavy-hitter contract, made deterministic per
                                     ingest launch ("batch" t = one launch, fa_stats.kernel_launches):
                                       R_t = R_(t-1) u { x in batch t : estimate_(t-1)(x) >= theta_(t-1) },
                                       theta_t = max(lower edge of the estimate bin (64 octaves x 32 steps) that holds rank
                                                 topk_track among R_t's estimates [0 while |R_t| < topk_track],
                                                 total weight_t >> (topk_capacity_log2 - 2), 1);
                                     (Launch boundaries are the library's: one fa_ingest / fa_ingest_device call is ONE launch
                                     unless it exceeds max_batch_records or 2^24 wide-tuple records, or is the first large call
                                     of a ctx with FA_KEYS_ADDR_PORT_PROTO - those are split.  R is therefore a function of the
                                     stream, the caller's batches AND the configuration; what is guaranteed for every split is
                                     the containment below, and bit-for-bit reproducibility for a fixed configuration.)
                                     fa_topk ranks R - every key whose estimate stood at or above the running topk_track-th
                                     estimate at a batch boundary and that occurred again afterwards: on a stream whose heavy
                                     hitters recur in every batch the same rows as the exact mode (BASELINE config 3: checked),
                                     without the 2 x 2 GiB of sets in the ingest path.  Nothing is admitted during the first
                                     batch of a ctx.  Restated in oracle/pyoracle.py (topk_candidates).  Sizing: the weight term
                                     of theta bounds R by the keys whose ESTIMATE reaches total >> (topk_capacity_log2 - 2) - at
                                     most 2^(topk_capacity_log2 - 2) true ones plus what the sketch's noise (about
                                     total >> cms_width_log2 per counter) lifts over it: keep cms_width_log2 >= topk_capacity_log2
                                     - 1, or a stream of many light keys fills R and fa_topk reports FA_ERR_TABLE_FULL. */
    uint32_t topk_track;          /* candidates mode: the rank the threshold follows (fa_topk serves k <= topk_track); 0 -> 256.
                                     Addresses above the threshold take the slower path through the set: the cost grows with it
                                     (BASELINE config 3: 1.02 / 1.16 / 1.20 ms per launch at 16 / 128 / 1024) */
    uint32_t reserved[1];
} fa_config;
enum { FA_TOPK_EXACT = 0, FA_TOPK_CANDIDATES = 1 };

/* One flows_5m row, scalar columns (create.sh:70-90). */
typedef struct {
    uint32_t date;     /* toDate(TimeReceived), days since 1970-01-01 UTC (create.sh:66) */
    uint32_t timeslot; /* start of the (sub)window, seconds (create.sh:96) */
    uint32_t src_as, dst_as, etype, _pad;
    uint64_t bytes, packets, count; /* sum(Bytes), sum(Packets), count() - wrap mod 2^64 */
} fa_row5m;

/* One row of the (SrcAddr,DstPort,Proto) rollup (FA_KEYS_ADDR_PORT_PROTO). */
typedef struct {
    uint32_t date;        /* toDate(TimeReceived) */
    uint32_t timeslot;    /* start of the (sub)window, seconds */
    uint8_t src_addr[16]; /* FixedString(16) (create.sh:12) */
    uint32_t dst_port, proto;
    uint64_t bytes, packets, count;
} fa_row_app;

/* GROUP BY SrcPort / DstPort (viz-ch.json:358,604). */
typedef struct {
    uint32_t port, _pad;
    uint64_t weight; /* sum(Bytes*SamplingRate), wraps mod 2^64 */
    uint64_t count;  /* rows with this port (a port whose rows all have weight 0 is still a group) */
} fa_port_row;

/* GROUP BY toStartOfMinute(TimeFlowStart) (viz-ch.json:74). */
typedef struct {
    uint32_t minute; /* start of the minute, seconds (TimeFlowStart narrowed to DateTime, create.sh:40) */
    uint32_t _pad;
    uint64_t weight; /* sum(Bytes*SamplingRate) */
    uint64_t count;
} fa_minute_row;

/* One row of the `flows` table (create.sh:7-27): the 15 projected columns. */
typedef struct {
    uint64_t time_received, time_flow_start, sampling_rate, bytes, packets;
    uint32_t sequence_num, src_as, dst_as, etype, proto, src_port, dst_port;
    uint32_t status; /* 0 = ok, 1 = malformed record (all other members 0) */
    uint8_t sampler_address[16], src_addr[16], dst_addr[16]; /* FixedString(16) */
} fa_flow_row;

/* Device-resident struct-of-arrays projection (the flows_raw analogue in HBM).
 * Every member is a DEVICE pointer to an array of n elements, owned by the ctx
 * and valid until the next fa_decode_device / fa_destroy on that ctx. */
typedef struct {
    const uint64_t *time_received, *time_flow_start, *sampling_rate, *bytes, *packets;
    const uint32_t *sequence_num, *src_as, *dst_as, *etype, *proto, *src_port, *dst_port;
    const uint8_t *sampler_address, *src_addr, *dst_addr; /* 16 B per record */
    const uint8_t* status;                                /* 0 ok, 1 malformed */
} fa_columns;

typedef struct {
    uint64_t records_ok;     /* decoded and aggregated */
    uint64_t records_bad;    /* malformed, dropped (inserter.go:125-126) */
    uint64_t records_slow;   /* handled by the generic (non-LDS) device parser */
    uint64_t bytes_in;       /* wire bytes ingested */
    uint64_t batches;
    uint64_t table_used;     /* occupied group-by slots */
    uint64_t table_capacity;
    uint64_t kernel_ns;        /* device time of the last ingest kernel launch (hipEvent) */
    uint64_t kernel_ns_total;  /* summed over every ingest kernel launch */
    uint64_t kernel_launches;
    uint64_t batch_ns_total;   /* tile kernel .. aggregation kernel, summed over every ingest launch */
    uint64_t records_direct;   /* records that took the direct device-wide-table path */
    uint64_t records_retried;  /* records decoded by the order-free second-chance parser */
    uint64_t wide_used;        /* occupied slots of the wide-key table */
    uint64_t wide_capacity;
    uint64_t wave_tile_launches; /* ingest launches that ran the wave-tile kernel (the rest: workgroup-tile kernel) */
    /* ABI 4 */
    uint64_t compact_tuple_launches; /* ... of those, launches that wrote compact 8-byte tuples (the rest: 16-byte ones) */
    uint64_t records_misfit_compact; /* records of compact-tuple launches whose values only a wide tuple holds */
    uint64_t decode_ns_total;        /* fa_decode_device: device time, summed over its launches (hipEvent) */
    uint64_t decode_launches;
    /* ABI 6 */
    uint64_t records_late;           /* records whose time bucket lies below a flows_5m window that fa_close_window /
                                        fa_drop_window(FA_ROWS_5M, timeslot) had already closed when they arrived.  They are
                                        aggregated like any other record (a later read of that timeslot shows them - what a
                                        late INSERT into flows_5m does, create.sh:70-90); the count tells the consumer that a
                                        closed window has been re-opened. */
    /* the wide log (FA_KEYS_ADDR_PORT_PROTO on a stream that opens a row for most of its records: a launch's 32-byte
     * tuples stay in their scatter segments - a "chunk" - and are folded into the hash table only when more than
     * FA_WIDE_LOG_CHUNKS (default 8) are pending; window reads sort them together with the table's rows).  Memory bound:
     * at most FA_WIDE_LOG_CHUNKS + 1 pairs of segment buffers of 2 x 32 B x (records of the largest launch) + 6 %, i.e.
     * 9 x 1.14 GiB at max_batch_records = 2^24 - held until fa_destroy.  A pair that cannot be allocated is not an
     * error: the oldest pending chunk is folded early and its buffers are taken over (wide_log_nomem_folds). */
    uint64_t wide_log_chunks;        /* chunks pending right now */
    uint64_t wide_log_bytes;         /* device bytes of every segment buffer the log holds (pending + recycled) */
    uint64_t wide_log_records;       /* records of the pending chunks (upper bound of their live tuples) */
    uint64_t wide_log_recorded;      /* chunks recorded so far */
    uint64_t wide_log_folded;        /* ... folded into the table by the region-owned kernel */
    uint64_t wide_log_replayed;      /* ... folded through the atomic replay (the table had grown since they were scattered) */
    uint64_t wide_log_dropped;       /* ... dropped whole by a window close (nothing alive above the watermark) */
    uint64_t wide_log_watermark_moves; /* times a close moved a pending chunk's watermark instead of folding it */
    uint64_t wide_log_nomem_folds;   /* chunks folded early because another pair of segment buffers could not be allocated */
    uint64_t wide_log_mode;          /* 1: the next launch keeps its (SrcAddr,DstPort,Proto) tuples in the log */
    /* ABI 7 - candidates mode (fa_config.topk_mode = FA_TOPK_CANDIDATES), SrcAddr then DstAddr: the admission threshold of the last
     * launch boundary and the candidates held at it (0 in the exact mode) */
    uint64_t topk_theta[2];
    uint64_t topk_candidates[2];
    /* launches of the kernel variant that learns a producer's field order per wave (flows_5m alone; chosen while most records of
     * the last launches needed the order-free parser - a producer that does not marshal in field-number order) */
    uint64_t learnt_order_launches;
    /* ABI 8 - where the HOST time of fa_ingest (host buffers) goes, summed over the calls, in nanoseconds: the whole call, of it the
     * wait for a free staging slot (the transfer and the kernels of the call before the last still hold it: the GPU / PCIe side is
     * behind) and the copy into page-locked staging incl. the offsets' narrowing to 32 bits (the consumer's cores are behind) */
    uint64_t host_ingest_ns;
    uint64_t host_stage_wait_ns;
    uint64_t host_stage_copy_ns;
} fa_stats_t;

typedef struct {
    uint8_t key[16]; /* FixedString(16) address */
    uint64_t weight; /* Count-Min estimate of sum(Bytes*SamplingRate) (>= exact) */
} fa_topk_row;

uint32_t fa_abi_version(void);

int fa_create(const fa_config* cfg, fa_ctx** out);
void fa_destroy(fa_ctx*);
const char* fa_last_error(const fa_ctx*); /* ctx may be NULL: last create error */

/* ---- ingest: decode + project + aggregate (the hot path) ---------------- */
/* Host buffers.  buf[0..len) holds n records back to back; offsets (n+1
 * entries, offsets[n]==len) delimit them - one Kafka message value each
 * (mocker.go:103-106).  offsets may be NULL when cfg.framed: the stream is then
 * split on the host by walking the varint length prefixes. */
int fa_ingest(fa_ctx*, const uint8_t* buf, size_t len, const uint64_t* offsets, size_t n);

/* Device-resident buffers (inputs already in HBM).  d_buf: DEVICE pointer to
 * len bytes, 16-byte aligned, followed by >= 32 readable slack bytes; d_offsets: DEVICE pointer to
 * n+1 uint32 offsets relative to d_buf (so len < 4 GiB per call).  Asynchronous
 * on the ctx stream; fa_sync() or any result call waits.
 * ABI 6: d_offsets may be NULL when cfg.framed - the chain of varint(len)-framed records is then cut into records ON THE
 * DEVICE (per 16 KiB block the first frame start is guessed - the first of the block's 256 leading byte positions from which two
 * consecutive frames are non-empty protobuf messages with ascending field numbers that end where their length prefix says -
 * then proven by a fixed-point pass over all blocks, offsets emitted; n is ignored, the call may become several launches).
 * FA_ERR_FRAMING when the bytes are not such a chain ending at len.  A chain whose guesses do not settle in 12 rounds (records
 * longer than several blocks, producers that do not marshal in field order) is copied back and walked on the host - same
 * result, slower.  Several passes over the bytes: a sixth to a third of the rate of a call WITH offsets (which a Kafka consumer
 * has for free: one message = one record) - the path for framed dumps.  fa_ingest (host buffers, offsets == NULL, 1 MiB <= len
 * <= 1 GiB) uses it too; smaller host buffers are walked on the host. */
int fa_ingest_device(fa_ctx*, const void* d_buf, size_t len, const void* d_offsets, size_t n);

/* ABI 8, optional: allocates NOW what the first fa_ingest calls of a ctx would allocate inside the consumer's loop - both
 * page-locked staging slots and their device twins, the deferral lists - for batches of up to `bytes` wire bytes in `records`
 * records (page-locking 2 x 64 MiB costs tens of milliseconds).  A consumer calls it in Setup with its batch bound. */
int fa_reserve_ingest(fa_ctx*, size_t bytes, size_t records);

int fa_sync(fa_ctx*);

/* ---- decode + project only (flows / flows_raw columns) ------------------- */
int fa_decode(fa_ctx*, const uint8_t* buf, size_t len, const uint64_t* offsets, size_t n,
              fa_flow_row* out_rows /* host, n entries */);
int fa_decode_device(fa_ctx*, const void* d_buf, size_t len, const void* d_offsets, size_t n,
                     fa_columns* out);

/* ---- window close: emit flows_5m rows ------------------------------------ */
/* Lists the distinct timeslots currently held, ascending. */
int fa_open_timeslots(fa_ctx*, uint32_t* out, size_t cap, size_t* n_out);
/* Emits the rows of `timeslot` (0xFFFFFFFF = every row) sorted by
 * (date,timeslot,src_as,dst_as,etype) and removes them from the device table.
 * FA_ERR_CAPACITY (nothing removed) if cap is too small; *n_out = rows needed. */
int fa_close_window(fa_ctx*, uint32_t timeslot, fa_row5m* out, size_t cap, size_t* n_out);
/* Same, without removing (peek). */
int fa_read_window(fa_ctx*, uint32_t timeslot, fa_row5m* out, size_t cap, size_t* n_out);
/* Device-side window close for an exchange across GPUs (RCCL all-gather of rows): the window's rows, sorted like
 * fa_read_window's, stay in HBM.  *d_rows: DEVICE pointer to *n_out fa_row5m, owned by the ctx, valid until its next
 * window / ingest call.  With sub-windows the rows are handed out per sub-bucket (not folded).  Nothing is removed. */
int fa_window_rows_device(fa_ctx*, uint32_t timeslot, const void** d_rows, size_t* n_out);
/* Adds n rows that sit in HBM (another rank's fa_window_rows_device, gathered over RCCL) to this ctx's table. */
int fa_merge_rows_device(fa_ctx*, const void* d_rows, size_t n);

/* ---- ABI 5: device-resident window close (every read above is built from these two steps) ------------------------
 * Row kinds and their emit order (what the matching read call returns):
 *   FA_ROWS_5M        fa_row5m      ORDER BY date, timeslot, src_as, dst_as, etype          (fa_read_window)
 *   FA_ROWS_APP       fa_row_app    ORDER BY date, timeslot, src_addr bytes, dst_port, proto (fa_read_window_app)
 *   FA_ROWS_PORT_SRC / _DST  fa_port_row   ORDER BY weight DESC, port                       (fa_top_ports; viz-ch.json:358,604)
 *   FA_ROWS_MINUTE    fa_minute_row ORDER BY minute                                         (fa_minute_series; viz-ch.json:74)
 *   FA_ROWS_TOPK_SRC / _DST  fa_topk_row   ORDER BY weight DESC, key bytes                  (fa_topk; viz-ch.json:233,479)
 * Multi-GPU window close (one ctx per GPU / Kafka partition, inserter.go:176): every rank calls fa_rows_device,
 * the ranks all-gather the device buffers over RCCL (counts first, no padding), every rank calls
 * fa_rows_merge_device on the concatenation - sum is a commutative monoid, so the result equals the single-ctx
 * result bit for bit (SummingMergeTree collapse, create.sh:70-90) - and fa_drop_window removes what was closed. */
enum {
    FA_ROWS_5M = 0, FA_ROWS_APP = 1, FA_ROWS_PORT_SRC = 2, FA_ROWS_PORT_DST = 3, FA_ROWS_MINUTE = 4,
    FA_ROWS_TOPK_SRC = 5, FA_ROWS_TOPK_DST = 6
};
size_t fa_row_bytes(int kind); /* bytes of one row of the kind; 0 for an unknown kind */
/* This ctx's result for (kind, timeslot) exactly as the matching read call would return it - the sub-buckets of a
 * sliding window folded, rows in emit order, cut after k rows (k = 0: all; timeslot is ignored by the kinds that
 * are not windowed) - left in HBM.  *d_rows: DEVICE pointer to *n_out rows, owned by the ctx, valid until its next
 * call.  Nothing is removed.  For FA_ROWS_TOPK_* after fa_merge_allreduce / fa_merged_view_set(1) the estimates come
 * from the merged sketch: the union of every rank's k rows then contains the global top k (a key of the global top k
 * ranks at least as high among the keys of any rank that saw it). */
int fa_rows_device(fa_ctx*, int kind, uint32_t timeslot, size_t k, const void** d_rows, size_t* n_out);
/* Merges n rows of `kind` sitting in HBM (several ranks' fa_rows_device results back to back, any order): rows with
 * equal keys are summed (top-k rows: kept once), the result is put into emit order and cut after k rows (0: all).
 * *d_out: DEVICE pointer to *n_out rows, owned by the ctx, valid until its next call.  The ctx's own state is not
 * touched.  FA_ROWS_5M rows must lie on this ctx's bucket grid (FA_ERR_ARG otherwise).  The ctx reads d_rows on its own
 * stream: whatever wrote them (a collective, a copy on another stream) must have completed when the call is made. */
int fa_rows_merge_device(fa_ctx*, int kind, const void* d_rows, size_t n, size_t k, const void** d_out, size_t* n_out);
/* ABI 6 - hash-partitioned window close, for row sets too large to gather on every rank ((SrcAddr,DstPort,Proto): a
 * window of BASELINE config 5 is 16.6 M rows x 56 B PER RANK).  The n rows of `kind` at d_rows (DEVICE; a fa_rows_device
 * result) are regrouped by owner: rank r of `world` owns the keys with (hash(key) >> 32) * world >> 32 == r (the same function of
 * the key on every rank, independent of the row's sums).  *d_out: DEVICE pointer to the n rows, group 0 first (order
 * inside a group unspecified), owned by the ctx, valid until its next fa_rows_partition_device; counts[r] (HOST array of
 * `world` entries) = rows of group r.  The ranks then exchange the groups with ONE all-to-all (RCCL: all_to_all_single
 * with these counts), every rank calls fa_rows_merge_device on what it received - 1 / world of the keys, complete - and
 * emits or gathers only that share; fa_drop_window as usual.  world <= 1024. */
int fa_rows_partition_device(fa_ctx*, int kind, const void* d_rows, size_t n, uint32_t world, const void** d_out, size_t* counts);
/* Copies n rows of `kind` from HBM into the caller's host buffer (cap in rows).  This call and every window read: when `out` is
 * page-locked host memory (hipHostMalloc, or hipHostRegister'ed by the caller - both ends of the buffer are looked at), a
 * result of 1 MiB or more is ONE copy-engine transfer into it (the link's 57 GB/s on MI355X, no host thread involved); into
 * pageable memory the rows are relayed through the ctx's two pinned slots by host threads (54 GB/s). */
int fa_rows_fetch(fa_ctx*, int kind, const void* d_rows, size_t n, void* out, size_t cap);
/* Removes what fa_close_window / fa_close_window_app would remove after emitting `timeslot` (kind FA_ROWS_5M or
 * FA_ROWS_APP): the whole window for tumbling windows and close-all, the oldest sub-bucket when windows slide. */
int fa_drop_window(fa_ctx*, int kind, uint32_t timeslot);
/* ABI 6: removes every (sub-)bucket whose start lies in [timeslot_lo, timeslot_hi) - both on the bucket grid - in one pass:
 * a consumer of TUMBLING windows over sub-buckets closes a whole window with it (fa_drop_window removes one sub-bucket
 * per call, what a sliding consumer wants). */
int fa_drop_range(fa_ctx*, int kind, uint32_t timeslot_lo, uint32_t timeslot_hi);

/* ---- bulk-load sink: flows_5m rows as ClickHouse RowBinary ------------------- */
/* Serialises rows for `INSERT INTO flows_5m FORMAT RowBinary` with the column list of
 * create.sh:70-90: Date (UInt16 days), Timeslot (DateTime = UInt32), SrcAS, DstAS (UInt32),
 * ETypeMap.EType Array(UInt32), ETypeMap.Bytes / .Packets / .Count Array(UInt64) - one element each,
 * what flows_5m_view writes (create.sh:100-103) - then Bytes, Packets, Count (UInt64).  70 bytes per
 * row; replaces the reference's per-row db.Exec (inserter/inserter.go:100-106).  Pure host code, no ctx.
 * Returns FA_ERR_CAPACITY (and the size in *bytes_out) when cap is too small; FA_ERR_ARG for a Date
 * that does not fit UInt16. */
#define FA_ROWBINARY_ROW5M_BYTES 70
int fa_rows_to_rowbinary(const fa_row5m* rows, size_t n, uint8_t* out, size_t cap, size_t* bytes_out);

/* ---- second exact key set: (SrcAddr, DstPort, Proto) ----------------------- */
/* Same contract as fa_read_window / fa_close_window / fa_merge_rows; rows sorted by
 * (date, timeslot, src_addr bytes, dst_port, proto).  Need FA_KEYS_ADDR_PORT_PROTO. */
int fa_read_window_app(fa_ctx*, uint32_t timeslot, fa_row_app* out, size_t cap, size_t* n_out);
int fa_close_window_app(fa_ctx*, uint32_t timeslot, fa_row_app* out, size_t cap, size_t* n_out);
int fa_merge_rows_app(fa_ctx*, const fa_row_app* rows, size_t n);
/* ABI 7: ONE window's rows without the (date, timeslot) every one of them carries - 48 instead of 56 bytes per row over PCIe, which
 * is what a window close of this key set costs (BASELINE config 5: 16.6 M rows per window).  timeslot: a window start (not
 * 0xFFFFFFFF); sliding windows are folded as in fa_read_window_app; same order.  *date_out = toDate(timeslot). */
typedef struct {
    uint8_t src_addr[16];
    uint32_t dst_port, proto;
    uint64_t bytes, packets, count;
} fa_row_app48;
int fa_read_window_app48(fa_ctx*, uint32_t timeslot, fa_row_app48* out, size_t cap, size_t* n_out, uint32_t* date_out);
int fa_close_window_app48(fa_ctx*, uint32_t timeslot, fa_row_app48* out, size_t cap, size_t* n_out, uint32_t* date_out);

/* ---- dashboard read side (not windowed: the dashboard picks $timeFilter) ----- */
/* dst = 0: GROUP BY SrcPort, 1: GROUP BY DstPort.  Every port that occurred, ORDER BY weight DESC
 * (ties: port ascending), cut at k.  Needs FA_KEYS_PORT_HIST. */
int fa_top_ports(fa_ctx*, int dst, size_t k, fa_port_row* out, size_t cap, size_t* n_out);
/* Adds port rows produced by another ctx / rank. */
int fa_merge_ports(fa_ctx*, int dst, const fa_port_row* rows, size_t n);
/* The per-minute series, ORDER BY minute.  Needs FA_KEYS_MINUTE_SERIES. */
int fa_minute_series(fa_ctx*, fa_minute_row* out, size_t cap, size_t* n_out);
int fa_merge_minutes(fa_ctx*, const fa_minute_row* rows, size_t n);
/* Clears the port groups and the minute series (start of a new $timeFilter range). */
int fa_dashboard_reset(fa_ctx*);

/* ---- address rendering of the dashboards (host code, no ctx) ------------------ */
/* The string the top-talker panels group by and display (viz-ch.json:233,479; README.md:186-221):
 *   if(EType = 0x800, IPv4NumToString(reinterpretAsUInt32(substring(reverse(Addr), 13, 4))), IPv6NumToString(Addr))
 * EType 0x0800: the first four bytes of the FixedString(16) in network order, dotted ("192.168.1.1"); any
 * other EType: ClickHouse's IPv6NumToString = the BIND inet_ntop6 rules (longest run of >= 2 zero groups
 * becomes "::", lower-case hex without leading zeros, "::a.b.c.d" / "::ffff:a.b.c.d" for the encapsulated
 * IPv4 forms) - README.md:191 renders the FixedString of 192.168.1.1's little-endian UInt32 as "101:a8c0::".
 * out receives a NUL-terminated string (FA_ADDR_STRLEN bytes always suffice); FA_ERR_CAPACITY if cap is
 * too small. */
#define FA_ADDR_STRLEN 46
int fa_format_addr(const uint8_t addr[16], uint32_t etype, char* out, size_t cap);

/* ---- heavy hitters -------------------------------------------------------- */
/* key_set: FA_KEYS_SRCADDR_CMS or FA_KEYS_DSTADDR_CMS.  The k addresses with the largest Count-Min
 * estimate of sum(Bytes*SamplingRate) (viz-ch.json:233,479) among ALL distinct addresses ingested
 * since the last fa_cms_reset; rows sorted by weight descending, ties by key bytes ascending.
 * Deterministic (independent of ingestion order).  FA_ERR_TABLE_FULL if the distinct-address set
 * overflowed (cfg.topk_capacity_log2). */
int fa_topk(fa_ctx*, uint32_t key_set, size_t k, fa_topk_row* out, size_t cap, size_t* n_out);
/* Adds candidate keys found elsewhere (another GPU / Kafka partition) to this ctx's distinct-address
 * set, so that fa_topk after fa_merge_allreduce ranks the union.  keys: n * 16 bytes (host). */
int fa_topk_merge_keys(fa_ctx*, uint32_t key_set, const uint8_t* keys, size_t n);
int fa_cms_query(fa_ctx*, uint32_t key_set, const uint8_t key[16], uint64_t* weight);
/* Copies the raw sketch (depth * 2^width_log2 uint64) to host. */
int fa_cms_read(fa_ctx*, uint32_t key_set, uint64_t* out, size_t cap_words);
int fa_cms_reset(fa_ctx*, uint32_t key_set); /* sketch and distinct-address set */

/* ---- multi-GPU merge at window close (one ctx per GPU / Kafka partition) --- */
/* Exposes the device-resident mergeable state so the host can run collectives
 * on it (RCCL all-reduce for the dense sketches).  DEVICE pointers. */
typedef struct {
    void* cms_src;      /* depth*2^width_log2 uint64, or NULL */
    void* cms_dst;
    size_t cms_words;
    void* port_hist;        /* 2*65536 entries of {weight, count} uint64 (SrcPort, then DstPort), or NULL */
    size_t port_hist_words; /* 4*65536 */
    /* ABI 4: the MERGED view of the sketches (window close across GPUs): separate buffers of cms_words uint64 that
     * a collective fills with the sum over all ranks - the ctx's own sketches stay untouched, so the merge can be
     * repeated (idempotent) and ingest can go on.  NULL when the key set is off. */
    void* cms_src_merged;
    void* cms_dst_merged;
} fa_device_state;
int fa_device_state_get(fa_ctx*, fa_device_state* out);
/* Declares the merged view valid (1: the caller's collective has filled cms_*_merged with all-rank sums; fa_topk /
 * fa_cms_read / fa_cms_query then answer from it) or stale (0).  Any later ingest or fa_cms_reset makes it stale. */
int fa_merged_view_set(fa_ctx*, int valid);
/* Adds partial rows produced by another ctx/rank (e.g. gathered over RCCL)
 * into this ctx's table: sum is a commutative monoid, so the merged table
 * equals the single-shard table bit for bit. */
int fa_merge_rows(fa_ctx*, const fa_row5m* rows, size_t n);
/* In-library RCCL path: comm is an `ncclComm_t` (librccl.so is bound lazily).  ncclAllReduce(sum, uint64) of the
 * ctx's sketches INTO the merged view (out of place, on the ctx stream) and marks it valid: calling it again - or
 * after more ingest - simply recomputes the view, nothing is ever counted twice.  Sparse state (flows_5m rows, wide
 * rows, port / minute rows) travels as rows: fa_close_window + fa_merge_rows etc. */
int fa_merge_allreduce(fa_ctx*, void* rccl_comm);

int fa_stats(fa_ctx*, fa_stats_t* out);

/* ---- ABI 7: window close of a GROUP of contexts inside one process ----------------------------------------------------
 * The reference's consumer is one process with one goroutine per claimed Kafka partition (inserter.go:167-196, :176); the
 * GPU stage keeps one ctx per (partition, GPU) and ingests without any exchange.  What needs every partition is the window
 * close: the sketches, the top-k, the (SrcAddr,DstPort,Proto) rows have no SummingMergeTree behind them that would sum
 * partial results (flows_5m has: create.sh:70-90 - its rows MAY leave per partition; merged they are fewer and final).
 * A group is that close: the members' results are produced, exchanged and merged in HBM - peer copies between the GPUs
 * (hipMemcpyPeerAsync over xGMI; members on one GPU: device copies), or RCCL for the dense sketches - and one result
 * leaves.  Every result equals, bit for bit, what ONE ctx that ingested all partitions returns from the matching call - with one
 * qualification: in FA_TOPK_CANDIDATES mode every member admits candidates by ITS OWN sketch and threshold (the contract is per
 * ctx and per ingest launch), and the group ranks the UNION of the members' candidates by the merged estimate; a single ctx over
 * all partitions would admit by the merged stream's thresholds.  Both hold every key that stood above its partition's running
 * topk_track-th estimate; they are not the same set in general (equal on streams whose heavy hitters are heavy in every partition).
 * The multi-PROCESS twin (one rank per GPU under torchrun) is flow-pipeline_amd/dist.py over the same ABI 5/6 steps.
 * Threading: a group call uses every member ctx (on host threads of its own, one per member); the caller must not run any
 * other call on a member at the same time (ingest goroutines take a read lock, the closing goroutine the write lock).
 * Errors: a member's failure fails the call, nothing is dropped, and fa_last_error of EVERY member - and
 * fa_group_last_error - name the member and its error. */
typedef struct fa_group fa_group;
enum {
    FA_GROUP_PEER = 0, /* exchange by peer copies; sketches: reduce-scatter + all-gather by peer copies and an add kernel */
    FA_GROUP_RCCL = 1  /* sketches by ncclAllReduce over communicators made with ncclCommInitAll (librccl.so is bound
                          lazily; needs every member on its own GPU - FA_ERR_UNSUPPORTED otherwise); rows still by peer copies */
};
/* ctxs: n contexts (1 <= n <= 1024) with the same window / key-set / sketch / top-k configuration (FA_ERR_ARG otherwise), any
 * placement over the GPUs; a ctx is a member of at most one group at a time.  The contexts stay the caller's: destroy the group
 * first, then them.  What a close needs beyond the members' own buffers (merged sketch views, exchange and partition buffers) is
 * reserved here for members that already hold rows, and - for members that ingest afterwards - when a launch's rows come into
 * being, not inside the first close.  n - 1 host threads live as long as the group. */
int fa_group_create(fa_ctx* const* ctxs, size_t n, uint32_t flags, fa_group** out);
void fa_group_destroy(fa_group*);
const char* fa_group_last_error(const fa_group*); /* group may be NULL: last create error */
size_t fa_group_size(const fa_group*);
int fa_group_transport(const fa_group*); /* FA_GROUP_PEER or FA_GROUP_RCCL */
/* The union of the members' open flows_5m timeslots, ascending (fa_open_timeslots). */
int fa_group_open_timeslots(fa_group*, uint32_t* out, size_t cap, size_t* n_out);
/* Rows of `kind` (FA_ROWS_*) for `timeslot`, merged over the members, in the kind's emit order, cut after k rows (0: all)
 * - fa_read_window / fa_read_window_app / fa_top_ports / fa_minute_series / fa_topk of the whole topic.  `out`: host
 * buffer of cap rows of fa_row_bytes(kind).  Small row sets: every member's result is gathered on one member (they take
 * turns) and merged there.  FA_ROWS_TOPK_*: the sketches are all-reduced first (fa_group_allreduce_sketches), every member
 * ranks its distinct addresses by the MERGED estimate and hands over k rows - exact with respect to the merged sketch.
 * FA_ERR_CAPACITY: *n_out = rows needed. */
int fa_group_read_window(fa_group*, int kind, uint32_t timeslot, size_t k, void* out, size_t cap, size_t* n_out);
/* ... and removes from every member what fa_close_window / fa_close_window_app remove (kind FA_ROWS_5M or FA_ROWS_APP). */
int fa_group_close_window(fa_group*, int kind, uint32_t timeslot, void* out, size_t cap, size_t* n_out);
/* Hash-partitioned form for LARGE row sets ((SrcAddr,DstPort,Proto): 16.6 M rows x 56 B per member and window in BASELINE
 * config 5): every member regroups its rows by owner (fa_rows_partition_device with world = group size), member r receives
 * group r of every member - one peer copy per pair, every xGMI link carries 1 / n of a member's rows -, merges its share
 * (1 / n of the keys, complete) and copies it out over ITS PCIe link while the others do the same.  out receives the shares
 * back to back, share 0 first (share_rows[r] rows each - host array of fa_group_size entries, may be NULL); inside a share
 * the rows are in the kind's emit order; a key appears in exactly one share.  The union equals fa_group_read_window's rows. */
int fa_group_read_window_partitioned(fa_group*, int kind, uint32_t timeslot, void* out, size_t cap, size_t* share_rows, size_t* n_out);
int fa_group_close_window_partitioned(fa_group*, int kind, uint32_t timeslot, void* out, size_t cap, size_t* share_rows, size_t* n_out);
/* Every member's merged sketch view = the sum of all members' sketches (out of place, repeatable: fa_merge_allreduce's
 * contract); fa_topk / fa_cms_read / fa_cms_query of ANY member then answer for the whole topic until it ingests again. */
int fa_group_allreduce_sketches(fa_group*);
/* fa_topk over the whole topic (= fa_group_read_window(FA_ROWS_TOPK_SRC / _DST, 0, k, ...)). */
int fa_group_topk(fa_group*, uint32_t key_set, size_t k, fa_topk_row* out, size_t cap, size_t* n_out);
/* fa_stats summed over the members (kernel_ns: the slowest member's last launch). */
int fa_group_stats(fa_group*, fa_stats_t* out);

/* ---- synthetic producer (mocker/mocker.go:57-106 distribution) ------------ */
enum {
    FA_MOCK_MOCKER = 0,   /* mocker.go:57-91 literally: 9 (SrcAS,DstAS) groups */
    FA_MOCK_ASPAIRS = 1,  /* BASELINE config 2: 65 536 AS pairs x 2 ETypes */
    FA_MOCK_ZIPF = 2,     /* BASELINE configs 3-5: Zipf-distributed addresses */
    FA_MOCK_GOFLOW = 3,   /* ASPAIRS keys in the shape GoFlow marshals an sFlow sample: 33 of the 67 fields of
                             pb-ext/flow.pb.go:57-147 (Type, TimeFlowEnd, SamplerAddress, NextHop, NextHopAS, SrcNet,
                             DstNet, InIf, OutIf, Proto, IP/TCP details, MACs, VLANs, fragment id, flow label), ~165 B */
    FA_MOCK_DISTINCT = 4, /* ASPAIRS shape, every record its own (SrcAS,DstAS) group: table-growth / spill tests */
    FA_MOCK_REVERSED = 5  /* ASPAIRS values, fields marshalled in DESCENDING field-number order: valid proto3 that no
                             canonical-order walk accepts - every record takes the order-free second-chance parser */
};
#define FA_MOCK_MAX_RECORD 256 /* upper bound of one generated record (any mode), framed */
typedef struct {
    uint32_t mode;
    uint32_t framed;    /* -proto.fixedlen */
    uint64_t seed;
    uint64_t n_total;
    uint64_t t0;
    uint32_t span_secs;
    uint32_t per_sec;
    uint32_t zipf_log2_universe;
    uint32_t zipf_s_x100;
} fa_mock_params;
/* Generates records [i0,i0+n) straight into HBM.  d_buf: DEVICE buffer of cap
 * bytes (needs 32 B slack beyond the bytes written); d_offsets: DEVICE n+1
 * uint32.  *bytes_out = bytes written.  Synchronous. */
int fa_mock_generate_device(fa_ctx*, const fa_mock_params*, uint64_t i0, uint64_t n, void* d_buf,
                            size_t cap, void* d_offsets, uint64_t* bytes_out);
/* Host-side twin (same bytes), for feeding fa_ingest. */
int fa_mock_generate_host(const fa_mock_params*, uint64_t i0, uint64_t n, uint8_t* buf, size_t cap,
                          uint64_t* offsets, uint64_t* bytes_out);

#ifdef __cplusplus
}
#endif
#endif
