/*
 * flow_oracle.h - CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the one hot path of cloudflare/flow-pipeline that
 * this repository replaces:
 *
 *   Kafka message value (varint(len) || protobuf FlowMessage)
 *     -> proto3 decode                      (inserter/inserter.go:122-126;
 *                                            kafka_format='Protobuf', compose/clickhouse/create.sh:33-34)
 *     -> 15-column projection               (compose/clickhouse/create.sh:7-27)
 *     -> Date / DateTime narrowing          (create.sh:36-68)
 *     -> 5-minute (Date,Timeslot,SrcAS,DstAS,EType) sum/sum/count rollup
 *                                           (create.sh:70-110)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library - and only as the checker.  The product path
 * (flow-pipeline_amd/csrc, libflowagg.so) never links or calls it.
 *
 * PARITY STATUS: the reference has no tests, golden vectors or fixtures for this
 * path and its arithmetic lives in third-party servers that cannot run here
 * (ClickHouse unpinned `yandex/clickhouse-server`; Go protobuf v1.4.3).  So the
 * *rollup semantics* are "parity unpinned" (restated from the SQL).  The *decode*
 * is pinned against upb-protobuf 7.35.1 driven by the reference's own schema
 * (pb-ext/flow.proto and the descriptor embedded at pb-ext/flow.pb.go:650-714):
 * see tests/golden/make_golden.py and tests/test_oracle_vs_upb.py.
 */
#ifndef FLOW_ORACLE_H
#define FLOW_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One row of the ClickHouse `flows` table (create.sh:7-27). */
typedef struct {
    uint64_t time_received;   /* field 2  */
    uint64_t time_flow_start; /* field 38 */
    uint64_t sampling_rate;   /* field 3  */
    uint64_t bytes;           /* field 9  */
    uint64_t packets;         /* field 10 */
    uint32_t sequence_num;    /* field 4  */
    uint32_t src_as;          /* field 14 */
    uint32_t dst_as;          /* field 15 */
    uint32_t etype;           /* field 30 (proto name `Etype`, column `EType`) */
    uint32_t proto;           /* field 20 */
    uint32_t src_port;        /* field 21 */
    uint32_t dst_port;        /* field 22 */
    uint32_t _pad;
    uint8_t sampler_address[16]; /* field 11, FixedString(16): right-padded with NUL */
    uint8_t src_addr[16];        /* field 6  */
    uint8_t dst_addr[16];        /* field 7  */
} fo_row;

/* One row of flows_5m restricted to its scalar columns. */
typedef struct {
    uint32_t date;     /* toDate(TimeReceived): days since epoch, UTC */
    uint32_t timeslot; /* toStartOfFiveMinute(TimeReceived) (or the configured granule) */
    uint32_t src_as, dst_as, etype, _pad;
    uint64_t bytes, packets, count;
} fo_row5m;

enum { FO_OK = 0, FO_BAD = -1 };

/* Decode one bare FlowMessage payload.  Returns FO_OK or FO_BAD (malformed
 * record: the caller counts it and drops it, inserter.go:125-126). */
int fo_decode(const uint8_t* p, size_t n, fo_row* out);

/* Decode one message value that must hold exactly one framed record:
 * varint(len) || payload with len == remaining bytes (mocker.go:98-106). */
int fo_decode_framed(const uint8_t* p, size_t n, fo_row* out);

/* Decode n records delimited by offsets[n+1]; rows[k] is zeroed and status[k]=1
 * for a bad record.  Returns the number of bad records. */
uint64_t fo_decode_batch(const uint8_t* buf, const uint64_t* offsets, size_t n, int framed,
                         fo_row* rows, uint32_t* status);

/* Split a concatenated stream of framed records into offsets (n+1 entries).
 * Returns the number of records, or (size_t)-1 if the stream is malformed /
 * cap too small.  offsets[k] is the start of record k's length prefix. */
size_t fo_frame_split(const uint8_t* buf, size_t len, uint64_t* offsets, size_t cap);

/* ---- exact rollup (flows_5m) ------------------------------------------- */
typedef struct fo_rollup fo_rollup;
fo_rollup* fo_rollup_new(uint32_t granule_secs /* 300 */);
void fo_rollup_free(fo_rollup*);
void fo_rollup_add(fo_rollup*, const fo_row* row);
/* Ingest n records delimited by offsets[n+1] (byte offsets into buf). framed:
 * 1 = each slot holds varint(len)||payload, 0 = bare payload.  Returns the
 * number of bad (dropped) records. */
uint64_t fo_rollup_ingest(fo_rollup*, const uint8_t* buf, const uint64_t* offsets, size_t n,
                          int framed);
void fo_rollup_merge(fo_rollup* dst, const fo_rollup* src);
size_t fo_rollup_size(const fo_rollup*);
/* Emit rows sorted by (date,timeslot,src_as,dst_as,etype). timeslot_filter ==
 * 0xFFFFFFFF emits everything.  Returns rows written (<= cap). */
size_t fo_rollup_rows(const fo_rollup*, uint32_t timeslot_filter, fo_row5m* out, size_t cap);

/* ---- Count-Min sketch + exact weights (viz-ch.json:233,479 "top talkers") */
/* weight = Bytes * SamplingRate (u64 wrap), key = FixedString(16) address.   */
uint32_t fo_cms_column(const uint8_t key[16], uint64_t seed, uint32_t width_log2, uint32_t row);
void fo_cms_update(uint64_t* cms, uint32_t depth, uint32_t width_log2, uint64_t seed,
                   const uint8_t key[16], uint64_t weight);
uint64_t fo_cms_query(const uint64_t* cms, uint32_t depth, uint32_t width_log2, uint64_t seed,
                      const uint8_t key[16]);

/* ---- synthetic generator (mocker.go:57-91 distribution and BASELINE configs) */
enum { FO_GEN_MOCKER = 0, FO_GEN_ASPAIRS = 1, FO_GEN_ZIPF = 2,
       FO_GEN_GOFLOW = 3,   /* ASPAIRS keys, marshalled with the fields GoFlow fills for an sFlow sample (33 of the 67
                               fields of pb-ext/flow.pb.go:57-147) - spec: DESIGN.md "Synthetic generator" */
       FO_GEN_DISTINCT = 4, /* ASPAIRS shape, SrcAS = 1 + (i & 0xfffff), DstAS = 1 + (i >> 20 & 0xfffff): one group per record */
       FO_GEN_REVERSED = 5  /* ASPAIRS values, fields in descending field-number order (valid proto3, not canonical) */
};
#define FO_GEN_MAX_RECORD 256
typedef struct {
    uint32_t mode;
    uint32_t framed;      /* 1: varint(len) prefix (ClickHouse stacks), 0: bare (Postgres stacks) */
    uint64_t seed;
    uint64_t n_total;     /* record count the time axis is spread over */
    uint64_t t0;          /* first TimeReceived (seconds) */
    uint32_t span_secs;   /* ASPAIRS/ZIPF: TimeReceived = t0 + floor(span_secs*i/n_total) */
    uint32_t per_sec;     /* MOCKER: TimeReceived = t0 + floor(i/per_sec) */
    uint32_t zipf_log2_universe; /* ZIPF: address universe = 2^this */
    uint32_t zipf_s_x100;        /* ZIPF: exponent * 100 (110 = 1.1) */
} fo_gen_params;

/* Exact length (framed or bare per params) of record i. */
uint32_t fo_gen_record_len(const fo_gen_params*, uint64_t i);
/* Write records [i0, i0+n) back to back into out (cap bytes); offsets gets n+1
 * entries relative to out.  Returns bytes written or (size_t)-1 on overflow. */
size_t fo_gen_records(const fo_gen_params*, uint64_t i0, uint64_t n, uint8_t* out, size_t cap,
                      uint64_t* offsets);
/* The decoded truth for record i (what a correct decoder must produce). */
void fo_gen_row(const fo_gen_params*, uint64_t i, fo_row* out);

/* ---- BASELINE config 3 at full scale: both Count-Min sketches (and, optionally, the exact per-address weights) of
 * the generator stream [i0, i0+n), straight from the generator's truth rows (fo_gen_row - what a correct decoder
 * yields; decode parity is established elsewhere), on `threads` threads with private sketch copies that are summed at
 * the end.  cms_src / cms_dst: depth << width_log2 uint64 each, ADDED to.  exact_src / exact_dst (may be NULL, ZIPF
 * mode only): 2 << zipf_log2_universe uint64 each, indexed [v6][rank] = rank + (v6 << zipf_log2_universe): the exact
 * sum(Bytes*SamplingRate) GROUP BY address (viz-ch.json:233,479) - the address of (rank, v6) is fo_zipf_key(). */
void fo_cms_stream(const fo_gen_params* g, uint64_t i0, uint64_t n, int threads, uint32_t depth, uint32_t width_log2,
                   uint64_t seed, uint64_t* cms_src, uint64_t* cms_dst, uint64_t* exact_src, uint64_t* exact_dst);
/* BASELINE config 5 at full scale: a checksum of the (Date,Timeslot,SrcAddr,DstPort,Proto) rollup (create.sh:92-110's Date /
 * Timeslot rule on the second key set) that is linear in the sums, so that it can be computed over the RECORDS without
 * grouping them: per timeslot slot0 + k*gran (k < nslots)  out_sum[k] += sum h(key) * (3*Bytes + 5*Packets + 7),
 * out_cnt[k] += records.  Returns the records that fell outside the slots.  pyoracle.app_rows_checksum is the rows' side. */
uint64_t fo_app_checksum_stream(const fo_gen_params* g, uint64_t i0, uint64_t n, int threads, uint32_t gran, uint32_t slot0,
                                uint32_t nslots, uint64_t* out_sum, uint64_t* out_cnt);
/* The 16-byte address the ZIPF generator gives rank `rank` (dst = 0: SrcAddr, 1: DstAddr; v6 = 0: 4-byte IPv4 form). */
void fo_zipf_key(uint64_t rank, int dst, int v6, uint8_t out[16]);

/* ---- cpu_baseline helper: generate + decode + rollup on `threads` threads.
 * Records [i0,i0+n) are generated up front (untimed), then decode+rollup is
 * timed.  Returns seconds of the timed region; *rows_out/bad_out optional. */
double fo_bench_rollup(const fo_gen_params*, uint64_t i0, uint64_t n, int threads,
                       uint64_t* wire_bytes_out, uint64_t* groups_out, uint64_t* bad_out,
                       uint64_t* checksum_out);

/* Same job with the harness details exposed: `groups_hint` = expected distinct groups of the range (0: unknown)
 * sizes the shard and part tables, which are allocated and first-touched BEFORE the start barrier; the result
 * carries the per-thread decode+rollup times and what is left for the merge; rows (optional, cap rows_cap) receives
 * the merged rows sorted by (date,timeslot,src_as,dst_as,etype).  Returns FO_BAD when rows_cap is too small. */
typedef struct {
    double seconds;      /* timed region: first thread's start to last thread's end */
    double decode_min, decode_max, decode_mean; /* per-thread decode + shard rollup, seconds */
    double merge_seconds; /* seconds - decode_max: partition sort + fold */
    uint64_t wire_bytes, groups, bad, checksum, rows;
    uint32_t threads, _pad;
} fo_bench_result;
int fo_bench_rollup_ex(const fo_gen_params*, uint64_t i0, uint64_t n, int threads, uint64_t groups_hint,
                       fo_bench_result* res, fo_row5m* rows, size_t rows_cap);

#ifdef __cplusplus
}
#endif
#endif
