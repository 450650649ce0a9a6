/*
 * flow_oracle.c - CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See flow_oracle.h for scope and parity status ("rollup parity unpinned",
 * decode pinned against upb-protobuf with the reference's schema).
 *
 * Decoder semantics = canonical proto3 parsing as exercised at
 * inserter/inserter.go:122-126 (proto.Unmarshal) and by ClickHouse's Protobuf
 * input format (create.sh:33-34), restated from the protobuf wire-format spec
 * and pinned case-by-case against upb (SURVEY.md Appendix A.2 plus
 * tests/golden/edge_cases.json):
 *   - fields in any order; unknown fields skipped by wire type;
 *   - duplicates: last one wins;  absent: 0 / empty;
 *   - tag = varint32: at most 5 bytes, value <= 0xFFFFFFFF, field number != 0;
 *   - varint value: at most 10 bytes, bits above 64 dropped; uint32 columns keep
 *     the low 32 bits;
 *   - LEN size = varint32 (<= 5 bytes) and must not run past the record;
 *   - wire types 6,7 are errors; groups (3/4) are skipped with matching END
 *     tags, nesting limit 100; a known field inside a group is NOT applied and
 *     field number 0 is tolerated there (upb's skipper does not check it);
 *   - a known field carried with the wrong wire type is treated as unknown.
 * Projection (create.sh:7-27): bytes -> FixedString(16) right-padded with NUL;
 * a value longer than 16 bytes is a bad record (ClickHouse raises; policy in
 * SURVEY.md 8(a)-4).
 */
#include "flow_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>

/* ------------------------------------------------------------------ decode */

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
} rd;

/* varint with a byte limit; returns 0 on success */
static int rd_varint(rd* r, int max_bytes, uint64_t* out) {
    uint64_t v = 0;
    for (int i = 0; i < max_bytes; i++) {
        if (r->p >= r->end) return -1;
        uint8_t b = *r->p++;
        if (i < 9)
            v |= (uint64_t)(b & 0x7f) << (7 * i);
        else
            v |= (uint64_t)(b & 0x01) << 63; /* 10th byte: only bit 63 survives */
        if (!(b & 0x80)) {
            *out = v;
            return 0;
        }
    }
    return -1; /* too long */
}

/* tag inside an unknown group: upb's group skipper does not reject field 0 */
static int rd_tag_raw(rd* r, uint32_t* field, uint32_t* wt) {
    uint64_t t;
    if (rd_varint(r, 5, &t)) return -1;
    if (t > 0xFFFFFFFFull) return -1;
    *field = (uint32_t)(t >> 3);
    *wt = (uint32_t)(t & 7);
    return 0;
}

/* top-level tag: field number 0 is an error */
static int rd_tag(rd* r, uint32_t* field, uint32_t* wt) {
    if (rd_tag_raw(r, field, wt)) return -1;
    if (*field == 0) return -1;
    return 0;
}

static int rd_size(rd* r, uint64_t* sz) {
    if (rd_varint(r, 5, sz)) return -1;
    if (*sz > 0x7FFFFFFFull) return -1;
    if (*sz > (uint64_t)(r->end - r->p)) return -1;
    return 0;
}

/* skip a value of wire type wt belonging to `field` (needed for group END match) */
static int rd_skip(rd* r, uint32_t field, uint32_t wt, int depth) {
    uint64_t v;
    switch (wt) {
    case 0: return rd_varint(r, 10, &v);
    case 1:
        if (r->end - r->p < 8) return -1;
        r->p += 8;
        return 0;
    case 2:
        if (rd_size(r, &v)) return -1;
        r->p += v;
        return 0;
    case 5:
        if (r->end - r->p < 4) return -1;
        r->p += 4;
        return 0;
    case 3: {
        if (depth >= 100) return -1;
        for (;;) {
            uint32_t f2, w2;
            if (rd_tag_raw(r, &f2, &w2)) return -1; /* also catches running off the end */
            if (w2 == 4) return f2 == field ? 0 : -1;
            if (rd_skip(r, f2, w2, depth + 1)) return -1;
        }
    }
    default: return -1; /* 4 (stray END), 6, 7 */
    }
}

static int take_addr(rd* r, uint8_t dst[16]) {
    uint64_t sz;
    if (rd_size(r, &sz)) return -1;
    if (sz > 16) return -1; /* FixedString(16) overflow -> bad record */
    memset(dst, 0, 16);
    memcpy(dst, r->p, sz);
    r->p += sz;
    return 0;
}

int fo_decode(const uint8_t* p, size_t n, fo_row* out) {
    rd r = {p, p + n};
    fo_row row;
    memset(&row, 0, sizeof row);
    while (r.p < r.end) {
        uint32_t field, wt;
        uint64_t v;
        if (rd_tag(&r, &field, &wt)) return FO_BAD;
        if (wt == 0) {
            if (rd_varint(&r, 10, &v)) return FO_BAD;
            switch (field) {
            case 2: row.time_received = v; break;
            case 3: row.sampling_rate = v; break;
            case 4: row.sequence_num = (uint32_t)v; break;
            case 9: row.bytes = v; break;
            case 10: row.packets = v; break;
            case 14: row.src_as = (uint32_t)v; break;
            case 15: row.dst_as = (uint32_t)v; break;
            case 20: row.proto = (uint32_t)v; break;
            case 21: row.src_port = (uint32_t)v; break;
            case 22: row.dst_port = (uint32_t)v; break;
            case 30: row.etype = (uint32_t)v; break;
            case 38: row.time_flow_start = v; break;
            default: break;
            }
        } else if (wt == 2 && (field == 6 || field == 7 || field == 11)) {
            uint8_t* dst = field == 6 ? row.src_addr : field == 7 ? row.dst_addr : row.sampler_address;
            if (take_addr(&r, dst)) return FO_BAD;
        } else {
            if (rd_skip(&r, field, wt, 0)) return FO_BAD;
        }
    }
    *out = row;
    return FO_OK;
}

int fo_decode_framed(const uint8_t* p, size_t n, fo_row* out) {
    rd r = {p, p + n};
    uint64_t len;
    if (rd_varint(&r, 10, &len)) return FO_BAD;
    if (len != (uint64_t)(r.end - r.p)) return FO_BAD;
    return fo_decode(r.p, (size_t)len, out);
}

uint64_t fo_decode_batch(const uint8_t* buf, const uint64_t* off, size_t n, int framed, fo_row* rows,
                         uint32_t* status) {
    uint64_t bad = 0;
    for (size_t k = 0; k < n; k++) {
        const uint8_t* p = buf + off[k];
        size_t len = (size_t)(off[k + 1] - off[k]);
        memset(&rows[k], 0, sizeof rows[k]);
        int rc = framed ? fo_decode_framed(p, len, &rows[k]) : fo_decode(p, len, &rows[k]);
        status[k] = rc == FO_OK ? 0 : 1;
        if (rc != FO_OK) {
            memset(&rows[k], 0, sizeof rows[k]);
            bad++;
        }
    }
    return bad;
}

size_t fo_frame_split(const uint8_t* buf, size_t len, uint64_t* offsets, size_t cap) {
    rd r = {buf, buf + len};
    size_t n = 0;
    while (r.p < r.end) {
        uint64_t l;
        if (n + 1 >= cap) return (size_t)-1;
        offsets[n] = (uint64_t)(r.p - buf);
        if (rd_varint(&r, 10, &l)) return (size_t)-1;
        if (l > (uint64_t)(r.end - r.p)) return (size_t)-1;
        r.p += l;
        n++;
    }
    if (n >= cap) return (size_t)-1;
    offsets[n] = len;
    return n;
}

/* ------------------------------------------------------------------ rollup */
/* flows_raw_view (create.sh:64-68): Date = toDate(TimeReceived), TimeReceived
 * narrowed UInt64 -> DateTime (u32 seconds).  flows_5m_view (create.sh:92-110):
 * key (Date, toStartOfFiveMinute(TimeReceived), SrcAS, DstAS, [EType]),
 * sum(Bytes), sum(Packets), count().  Server TZ = UTC.  Parity domain:
 * 65536 <= TimeReceived < 2^32; outside it this oracle just
 * applies the same arithmetic to the low 32 bits. */

typedef struct {
    uint32_t used;
    uint32_t timeslot, src_as, dst_as, etype;
    uint64_t bytes, packets, count;
} slot;

struct fo_rollup {
    uint32_t gran;
    size_t cap, n;
    slot* t;
};

static uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27;
    z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}

fo_rollup* fo_rollup_new(uint32_t gran) {
    fo_rollup* r = (fo_rollup*)calloc(1, sizeof *r);
    r->gran = gran ? gran : 300;
    r->cap = 1024;
    r->t = (slot*)calloc(r->cap, sizeof(slot));
    return r;
}
void fo_rollup_free(fo_rollup* r) {
    if (!r) return;
    free(r->t);
    free(r);
}
size_t fo_rollup_size(const fo_rollup* r) { return r->n; }

static void rollup_put(fo_rollup* r, uint32_t ts, uint32_t sa, uint32_t da, uint32_t et, uint64_t b,
                       uint64_t p, uint64_t c);
static slot* table_alloc(size_t cap);

static void rollup_grow(fo_rollup* r) {
    slot* old = r->t;
    size_t oc = r->cap;
    r->cap *= 2;
    r->n = 0;
    r->t = table_alloc(r->cap);
    for (size_t i = 0; i < oc; i++)
        if (old[i].used)
            rollup_put(r, old[i].timeslot, old[i].src_as, old[i].dst_as, old[i].etype, old[i].bytes,
                       old[i].packets, old[i].count);
    free(old);
}

static void rollup_put(fo_rollup* r, uint32_t ts, uint32_t sa, uint32_t da, uint32_t et, uint64_t b,
                       uint64_t p, uint64_t c) {
    if ((r->n + 1) * 2 > r->cap) rollup_grow(r);
    uint64_t h = mix64(((uint64_t)sa << 32 | da) ^ mix64((uint64_t)ts << 32 | et));
    size_t m = r->cap - 1, i = (size_t)h & m;
    for (;;) {
        slot* s = &r->t[i];
        if (!s->used) {
            s->used = 1;
            s->timeslot = ts;
            s->src_as = sa;
            s->dst_as = da;
            s->etype = et;
            s->bytes = b;
            s->packets = p;
            s->count = c;
            r->n++;
            return;
        }
        if (s->timeslot == ts && s->src_as == sa && s->dst_as == da && s->etype == et) {
            s->bytes += b; /* UInt64 sums wrap mod 2^64 */
            s->packets += p;
            s->count += c;
            return;
        }
        i = (i + 1) & m;
    }
}

void fo_rollup_add(fo_rollup* r, const fo_row* row) {
    uint32_t t = (uint32_t)row->time_received; /* UInt64 -> DateTime */
    uint32_t ts = t - t % r->gran;
    rollup_put(r, ts, row->src_as, row->dst_as, row->etype, row->bytes, row->packets, 1);
}

/* The group-by table of a many-group stream (config 2: 393 216 groups, tens of MB per shard) does not fit the caches:
 * every record would wait for one DRAM access.  Like any production hash aggregation the ingest loop therefore works
 * on a short window of decoded records: a record's home slot is prefetched when it is decoded and the record is added
 * FO_WIN records later (same results - the adds commute; a table growth in between only makes a prefetch useless). */
#define FO_WIN 16
uint64_t fo_rollup_ingest(fo_rollup* r, const uint8_t* buf, const uint64_t* off, size_t n,
                          int framed) {
    uint64_t bad = 0;
    struct { uint32_t ts, sa, da, et; uint64_t b, p; } win[FO_WIN];
    size_t head = 0, fill = 0;
    for (size_t k = 0; k < n; k++) {
        fo_row row;
        const uint8_t* p = buf + off[k];
        size_t len = (size_t)(off[k + 1] - off[k]);
        int rc = framed ? fo_decode_framed(p, len, &row) : fo_decode(p, len, &row);
        if (rc != FO_OK) {
            bad++;
            continue;
        }
        if (fill == FO_WIN) { /* retire the oldest */
            rollup_put(r, win[head].ts, win[head].sa, win[head].da, win[head].et, win[head].b, win[head].p, 1);
            fill--;
        }
        const uint32_t t = (uint32_t)row.time_received; /* UInt64 -> DateTime */
        const uint32_t ts = t - t % r->gran;
        win[head].ts = ts; win[head].sa = row.src_as; win[head].da = row.dst_as; win[head].et = row.etype;
        win[head].b = row.bytes; win[head].p = row.packets;
        {
            const uint64_t h = mix64(((uint64_t)row.src_as << 32 | row.dst_as) ^ mix64((uint64_t)ts << 32 | row.etype));
            __builtin_prefetch(&r->t[(size_t)h & (r->cap - 1)], 1, 1);
        }
        head = (head + 1) % FO_WIN;
        fill++;
    }
    for (size_t i = 0; i < fill; i++) { /* oldest first */
        const size_t j = (head + FO_WIN - fill + i) % FO_WIN;
        rollup_put(r, win[j].ts, win[j].sa, win[j].da, win[j].et, win[j].b, win[j].p, 1);
    }
    return bad;
}

void fo_rollup_merge(fo_rollup* dst, const fo_rollup* src) {
    for (size_t i = 0; i < src->cap; i++)
        if (src->t[i].used)
            rollup_put(dst, src->t[i].timeslot, src->t[i].src_as, src->t[i].dst_as, src->t[i].etype,
                       src->t[i].bytes, src->t[i].packets, src->t[i].count);
}

static int row5m_cmp(const void* a, const void* b) {
    const fo_row5m* x = (const fo_row5m*)a;
    const fo_row5m* y = (const fo_row5m*)b;
#define CMP(f) \
    if (x->f != y->f) return x->f < y->f ? -1 : 1
    CMP(date);
    CMP(timeslot);
    CMP(src_as);
    CMP(dst_as);
    CMP(etype);
#undef CMP
    return 0;
}

size_t fo_rollup_rows(const fo_rollup* r, uint32_t filter, fo_row5m* out, size_t cap) {
    size_t k = 0;
    for (size_t i = 0; i < r->cap; i++) {
        const slot* s = &r->t[i];
        if (!s->used) continue;
        if (filter != 0xFFFFFFFFu && s->timeslot != filter) continue;
        if (k >= cap) return k;
        fo_row5m* o = &out[k++];
        o->date = s->timeslot / 86400u; /* == floor(t/86400) whenever 86400 % gran == 0 */
        o->timeslot = s->timeslot;
        o->src_as = s->src_as;
        o->dst_as = s->dst_as;
        o->etype = s->etype;
        o->_pad = 0;
        o->bytes = s->bytes;
        o->packets = s->packets;
        o->count = s->count;
    }
    qsort(out, k, sizeof *out, row5m_cmp);
    return k;
}

/* --------------------------------------------------------------------- CMS */
/* There is no Count-Min sketch in the reference; the exact contract is the
 * dashboard query `GROUP BY SrcAddr ORDER BY sum(Bytes*SamplingRate) DESC`
 * (compose/grafana/dashboards/viz-ch.json:233,479).  The sketch is defined by
 * this repository (DESIGN.md "Sketch"): a PREFIX-PARTITIONED Count-Min sketch.
 *   a = mix64(lo ^ mix64(seed + phi)); h1 = mix64(a ^ hi); h2 = a | 1
 *   pbits = min(8, width_log2 - 4); sub = width_log2 - pbits
 *   prefix = h1 & (2^pbits - 1); l1 = h1 >> 32; l2 = (h2 >> 32) | 1        (32-bit double hashing, Kirsch & Mitzenmacher)
 *   column(r) = prefix << sub | (uint32)(l1 + r * l2) >> (32 - sub)
 * i.e. 2^pbits independent sketches of width 2^sub, the key picks one with bits that take no part in the row hashes:
 * all rows of a key share the top pbits of the column (so an update touches one 2^sub-wide block per row). */
uint32_t fo_cms_column(const uint8_t key[16], uint64_t seed, uint32_t wl2, uint32_t row) {
    uint64_t lo, hi;
    memcpy(&lo, key, 8);
    memcpy(&hi, key + 8, 8);
    const uint64_t a = mix64(lo ^ mix64(seed + 0x9E3779B97F4A7C15ull));
    const uint64_t h1 = mix64(a ^ hi), h2 = a | 1ull;
    const uint32_t pbits = wl2 - 4u < 8u ? wl2 - 4u : 8u, sub = wl2 - pbits;
    const uint32_t prefix = (uint32_t)h1 & ((1u << pbits) - 1u), l1 = (uint32_t)(h1 >> 32), l2 = (uint32_t)(h2 >> 32) | 1u;
    return (prefix << sub) | ((uint32_t)(l1 + row * l2) >> (32u - sub));
}
void fo_cms_update(uint64_t* cms, uint32_t depth, uint32_t wl2, uint64_t seed, const uint8_t key[16],
                   uint64_t w) {
    for (uint32_t r = 0; r < depth; r++)
        cms[((size_t)r << wl2) + fo_cms_column(key, seed, wl2, r)] += w;
}
uint64_t fo_cms_query(const uint64_t* cms, uint32_t depth, uint32_t wl2, uint64_t seed,
                      const uint8_t key[16]) {
    uint64_t best = ~0ull;
    for (uint32_t r = 0; r < depth; r++) {
        uint64_t v = cms[((size_t)r << wl2) + fo_cms_column(key, seed, wl2, r)];
        if (v < best) best = v;
    }
    return best;
}

/* --------------------------------------------------------------- generator */
/* Restates the value distribution of mocker/mocker.go:57-91 with a
 * counter-based PRNG so any record index can be regenerated independently
 * (the Go program uses an unseeded math/rand stream and is not reproducible).
 * Spec (DESIGN.md "Synthetic generator"):
 *   base(i)  = mix64(seed * 0x9E3779B97F4A7C15 + i + 1)
 *   rnd(i,j) = mix64(base(i) ^ ((j + 1) * 0xD1B54A32D192ED03))
 */
static uint64_t gen_rnd(const fo_gen_params* g, uint64_t i, uint32_t j) {
    uint64_t base = mix64(g->seed * 0x9E3779B97F4A7C15ull + i + 1);
    return mix64(base ^ ((uint64_t)(j + 1) * 0xD1B54A32D192ED03ull));
}

/* Integer-only Zipf-like rank sampler over [0, 2^L): the universe is cut into
 * octaves [2^k-1, 2^(k+1)-1); octave k is drawn with probability proportional
 * to w_k = floor(2^32 * 2^(-k*(s-1))) (the integral of x^-s over an octave
 * scales as 2^(-k(s-1))), computed in fixed point by repeated multiplication
 * with q = round(2^32 * 2^-(s-1)) taken from a small table; ranks are uniform
 * inside an octave.  Pure integer math => bit-identical on CPU and GPU. */
static uint32_t zipf_q32(uint32_t s_x100) {
    /* round(2^32 * 2^-((s_x100-100)/100)) for the exponents the configs use */
    switch (s_x100) {
    case 80: return 0xFFFFFFFFu;  /* s<1: handled by caller (weights grow) */
    case 100: return 0xFFFFFFFFu; /* ~1.0 */
    case 110: return 4007346185u; /* 2^-0.1 * 2^32 */
    case 120: return 3738986199u; /* 2^-0.2 * 2^32 */
    case 150: return 3037000500u; /* 2^-0.5 * 2^32 */
    default: return 4007346185u;
    }
}
static uint32_t zipf_g32(uint32_t s_x100) {
    /* growth factor for s < 1: round(2^30 * 2^(0.2)) for s=0.8 */
    return s_x100 == 80 ? 1233405467u : (1u << 30);
}
static uint64_t zipf_rank(const fo_gen_params* g, uint64_t r) {
    uint32_t L = g->zipf_log2_universe ? g->zipf_log2_universe : 24;
    uint32_t s = g->zipf_s_x100 ? g->zipf_s_x100 : 110;
    /* octave weights in 2.30 fixed point, cumulative in u64 */
    uint64_t w[40], tot = 0, cur = 1ull << 30;
    for (uint32_t k = 0; k < L; k++) {
        w[k] = cur ? cur : 1;
        tot += w[k];
        if (s >= 100)
            cur = (cur * (uint64_t)zipf_q32(s)) >> 32;
        else
            cur = (cur * (uint64_t)zipf_g32(s)) >> 30;
    }
    uint64_t u = (r >> 11) % tot; /* modulo bias < 2^-20, identical on both sides */
    uint32_t k = 0;
    while (u >= w[k]) {
        u -= w[k];
        k++;
    }
    uint64_t lo = (1ull << k) - 1, span = 1ull << k;
    uint64_t r2 = mix64(r ^ 0xA5A5A5A5A5A5A5A5ull);
    return lo + (r2 & (span - 1));
}
/* fixed bijective map rank -> 16-byte key (SURVEY.md 8(d) cfg 3) */
static void zipf_key(uint64_t rank, uint64_t salt, uint8_t out[16], int v4) {
    uint64_t a = mix64(rank * 0x9E3779B97F4A7C15ull + salt);
    uint64_t b = mix64(a ^ 0xD1B54A32D192ED03ull);
    memset(out, 0, 16);
    if (v4) {
        memcpy(out, &a, 4);
    } else {
        memcpy(out, &a, 8);
        memcpy(out + 8, &b, 8);
    }
}

void fo_gen_row(const fo_gen_params* g, uint64_t i, fo_row* o) {
    memset(o, 0, sizeof *o);
    uint64_t r0 = gen_rnd(g, i, 0), r1 = gen_rnd(g, i, 1), r2 = gen_rnd(g, i, 2),
             r3 = gen_rnd(g, i, 3), r4 = gen_rnd(g, i, 4), r5 = gen_rnd(g, i, 5);
    static const uint8_t pfx[15] = {0x20, 0x01, 0x0d, 0xb8, 0, 0, 0, 0x01, 0, 0, 0, 0, 0, 0, 0};
    o->sampling_rate = 1;                         /* mocker.go:77 */
    o->bytes = r0 % 1500;                         /* mocker.go:59 */
    o->packets = r1 % 100;                        /* mocker.go:60 */
    o->src_port = (uint32_t)(r5 & 0xFFFF);        /* mocker.go:87 */
    o->dst_port = (uint32_t)((r5 >> 16) & 0xFFFF);/* mocker.go:88 */
    o->sequence_num = (uint32_t)i;                /* mocker.go:89-91 */
    if (g->mode == FO_GEN_MOCKER) {
        uint32_t ps = g->per_sec ? g->per_sec : 4;
        o->time_received = g->t0 + i / ps;        /* mocker.go:57,85-86 */
        o->src_as = 65000 + (uint32_t)(r2 % 3);   /* mocker.go:61,80 */
        o->dst_as = 65000 + (uint32_t)(r3 % 3);   /* mocker.go:62,81 */
        o->etype = 0x86dd;                        /* mocker.go:82 */
        memcpy(o->src_addr, pfx, 15);             /* mocker.go:64-71 */
        memcpy(o->dst_addr, pfx, 15);
        o->src_addr[15] = (uint8_t)(r4 & 0xff);
        o->dst_addr[15] = (uint8_t)((r4 >> 8) & 0xff);
    } else {
        uint64_t nt = g->n_total ? g->n_total : 1;
        o->time_received = g->t0 + (uint64_t)g->span_secs * i / nt;
        int v6 = (int)((r2 >> 16) & 1);
        o->etype = v6 ? 0x86dd : 0x0800;
        if (g->mode == FO_GEN_ASPAIRS || g->mode == FO_GEN_GOFLOW || g->mode == FO_GEN_DISTINCT || g->mode == FO_GEN_REVERSED) {
            o->src_as = 64512 + (uint32_t)(r2 & 255);
            o->dst_as = 64512 + (uint32_t)((r2 >> 8) & 255);
            if (g->mode == FO_GEN_DISTINCT) {
                o->src_as = 1 + (uint32_t)(i & 0xfffff);
                o->dst_as = 1 + (uint32_t)((i >> 20) & 0xfffff);
            }
            if (g->mode == FO_GEN_GOFLOW) { /* the projected columns GoFlow fills besides the mocker's */
                o->sampling_rate = (r2 >> 17) & 1 ? 2048 : 1024;
                o->proto = (r2 >> 18) & 1 ? 6 : 17;
                o->sampler_address[0] = 10;
                o->sampler_address[1] = 255;
                o->sampler_address[3] = (uint8_t)(gen_rnd(g, i, 6) & 7);
            }
            if (v6) {
                memcpy(o->src_addr, pfx, 15);
                memcpy(o->dst_addr, pfx, 15);
                o->src_addr[15] = (uint8_t)(r4 & 0xff);
                o->dst_addr[15] = (uint8_t)((r4 >> 8) & 0xff);
            } else {
                o->src_addr[0] = 10; o->src_addr[1] = (uint8_t)(r4 >> 16);
                o->src_addr[2] = (uint8_t)(r4 >> 24); o->src_addr[3] = (uint8_t)(r4 & 0xff);
                o->dst_addr[0] = 10; o->dst_addr[1] = (uint8_t)(r4 >> 32);
                o->dst_addr[2] = (uint8_t)(r4 >> 40); o->dst_addr[3] = (uint8_t)((r4 >> 8) & 0xff);
            }
        } else { /* FO_GEN_ZIPF */
            uint64_t rs = zipf_rank(g, r3), rd_ = zipf_rank(g, r4);
            o->src_as = 64512 + (uint32_t)(rs & 255);
            o->dst_as = 64512 + (uint32_t)(rd_ & 255);
            zipf_key(rs, 0x1111, o->src_addr, !v6);
            zipf_key(rd_, 0x2222, o->dst_addr, !v6);
            o->sampling_rate = (r2 >> 17) & 1 ? 1000 : 1;
            o->proto = (r2 >> 18) & 1 ? 6 : 17;
        }
    }
    o->time_flow_start = o->time_received;
}

static size_t put_varint(uint8_t* p, uint64_t v) {
    size_t n = 0;
    while (v >= 0x80) {
        p[n++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    p[n++] = (uint8_t)v;
    return n;
}
static size_t put_vfield(uint8_t* p, uint32_t field, uint64_t v) {
    if (!v) return 0; /* proto3 omits zero values */
    size_t n = put_varint(p, (uint64_t)field << 3);
    return n + put_varint(p + n, v);
}
static size_t put_bfield(uint8_t* p, uint32_t field, const uint8_t* d, size_t len) {
    if (!len) return 0;
    size_t n = put_varint(p, ((uint64_t)field << 3) | 2);
    n += put_varint(p + n, len);
    memcpy(p + n, d, len);
    return n + len;
}
static size_t addr_len(const fo_row* r, const uint8_t* a) {
    /* generator emits 16-byte addresses for IPv6 rows, 4-byte for IPv4 rows */
    (void)a;
    return r->etype == 0x0800 ? 4 : 16;
}

/* golang/protobuf marshals known fields in field-number order (mocker.go:97). */
static size_t encode_row(const fo_row* r, uint8_t* p) {
    size_t n = 0;
    n += put_vfield(p + n, 2, r->time_received);
    n += put_vfield(p + n, 3, r->sampling_rate);
    n += put_vfield(p + n, 4, r->sequence_num);
    n += put_bfield(p + n, 6, r->src_addr, addr_len(r, r->src_addr));
    n += put_bfield(p + n, 7, r->dst_addr, addr_len(r, r->dst_addr));
    n += put_vfield(p + n, 9, r->bytes);
    n += put_vfield(p + n, 10, r->packets);
    n += put_vfield(p + n, 14, r->src_as);
    n += put_vfield(p + n, 15, r->dst_as);
    n += put_vfield(p + n, 20, r->proto);
    n += put_vfield(p + n, 21, r->src_port);
    n += put_vfield(p + n, 22, r->dst_port);
    n += put_vfield(p + n, 30, r->etype);
    n += put_vfield(p + n, 38, r->time_flow_start);
    return n;
}

/* FO_GEN_GOFLOW: the record as GoFlow marshals an sFlow sample - every field it fills, in field-number order
 * (pb-ext/flow.pb.go:57-147; golang/protobuf emits known fields ascending and omits zeros).  The fields outside the
 * ClickHouse projection are derived here from rnd(i,6..8); values: DESIGN.md "Synthetic generator". */
static size_t encode_goflow(const fo_gen_params* g, uint64_t i, const fo_row* r, uint8_t* p) {
    static const uint8_t pfx[15] = {0x20, 0x01, 0x0d, 0xb8, 0, 0, 0, 0x01, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t r6 = gen_rnd(g, i, 6), r7 = gen_rnd(g, i, 7), r8 = gen_rnd(g, i, 8);
    const int v6 = r->etype == 0x86dd;
    const size_t alen = v6 ? 16 : 4;
    uint8_t nh[16];
    memset(nh, 0, 16);
    if (v6) {
        memcpy(nh, pfx, 15);
        nh[15] = (uint8_t)(r6 >> 8);
    } else {
        nh[0] = 10; nh[1] = (uint8_t)(r6 >> 8); nh[2] = (uint8_t)(r6 >> 16); nh[3] = 1;
    }
    const uint32_t vlan = 100 + (uint32_t)(r8 & 15);
    size_t n = 0;
    n += put_vfield(p + n, 1, 1);                                   /* Type = SFLOW_5 */
    n += put_vfield(p + n, 2, r->time_received);
    n += put_vfield(p + n, 3, r->sampling_rate);
    n += put_vfield(p + n, 4, r->sequence_num);
    n += put_vfield(p + n, 5, r->time_received);                    /* TimeFlowEnd */
    n += put_bfield(p + n, 6, r->src_addr, alen);
    n += put_bfield(p + n, 7, r->dst_addr, alen);
    n += put_vfield(p + n, 9, r->bytes);
    n += put_vfield(p + n, 10, r->packets);
    n += put_bfield(p + n, 11, r->sampler_address, 4);
    n += put_bfield(p + n, 12, nh, alen);                           /* NextHop */
    n += put_vfield(p + n, 13, 64512 + ((r6 >> 24) & 255));         /* NextHopAS */
    n += put_vfield(p + n, 14, r->src_as);
    n += put_vfield(p + n, 15, r->dst_as);
    n += put_vfield(p + n, 16, v6 ? 48 : 24);                       /* SrcNet */
    n += put_vfield(p + n, 17, v6 ? 32 + ((r6 >> 32) & 31) : 8 + ((r6 >> 32) & 15)); /* DstNet */
    n += put_vfield(p + n, 18, 1 + ((r6 >> 40) & 63));              /* InIf */
    n += put_vfield(p + n, 19, 1 + ((r6 >> 46) & 63));              /* OutIf */
    n += put_vfield(p + n, 20, r->proto);
    n += put_vfield(p + n, 21, r->src_port);
    n += put_vfield(p + n, 22, r->dst_port);
    n += put_vfield(p + n, 23, (r7 & 3) ? 0 : 0xb8);                /* IPTos */
    n += put_vfield(p + n, 25, 32 + ((r7 >> 2) & 127));             /* IPTTL */
    n += put_vfield(p + n, 26, r->proto == 6 ? ((r7 >> 9) & 0x3f) : 0); /* TCPFlags */
    n += put_vfield(p + n, 27, 0x3cfdfe000000ull | ((r7 >> 16) & 0xffffff)); /* SrcMac */
    n += put_vfield(p + n, 28, 0xa0369f000000ull | ((r7 >> 40) & 0xffffff)); /* DstMac */
    n += put_vfield(p + n, 29, vlan);                               /* VlanId */
    n += put_vfield(p + n, 30, r->etype);
    n += put_vfield(p + n, 33, vlan);                               /* SrcVlan */
    n += put_vfield(p + n, 34, 200 + ((r8 >> 4) & 15));             /* DstVlan */
    n += put_vfield(p + n, 35, v6 ? 0 : ((r8 >> 8) & 0xffff));      /* FragmentId */
    n += put_vfield(p + n, 37, v6 ? ((r8 >> 24) & 0xfffff) : 0);    /* IPv6FlowLabel */
    n += put_vfield(p + n, 38, r->time_flow_start);
    return n;
}

/* FO_GEN_REVERSED: the same fields, last first (a producer is free to emit fields in any order) */
static size_t encode_row_reversed(const fo_row* r, uint8_t* p) {
    static const uint32_t order[14] = {38, 30, 22, 21, 20, 15, 14, 10, 9, 7, 6, 4, 3, 2};
    size_t n = 0;
    for (int k = 0; k < 14; k++) {
        switch (order[k]) {
        case 38: n += put_vfield(p + n, 38, r->time_flow_start); break;
        case 30: n += put_vfield(p + n, 30, r->etype); break;
        case 22: n += put_vfield(p + n, 22, r->dst_port); break;
        case 21: n += put_vfield(p + n, 21, r->src_port); break;
        case 20: n += put_vfield(p + n, 20, r->proto); break;
        case 15: n += put_vfield(p + n, 15, r->dst_as); break;
        case 14: n += put_vfield(p + n, 14, r->src_as); break;
        case 10: n += put_vfield(p + n, 10, r->packets); break;
        case 9: n += put_vfield(p + n, 9, r->bytes); break;
        case 7: n += put_bfield(p + n, 7, r->dst_addr, addr_len(r, r->dst_addr)); break;
        case 6: n += put_bfield(p + n, 6, r->src_addr, addr_len(r, r->src_addr)); break;
        case 4: n += put_vfield(p + n, 4, r->sequence_num); break;
        case 3: n += put_vfield(p + n, 3, r->sampling_rate); break;
        default: n += put_vfield(p + n, 2, r->time_received); break;
        }
    }
    return n;
}

static size_t gen_one(const fo_gen_params* g, uint64_t i, uint8_t* out) {
    fo_row r;
    uint8_t tmp[FO_GEN_MAX_RECORD];
    fo_gen_row(g, i, &r);
    size_t n = g->mode == FO_GEN_GOFLOW ? encode_goflow(g, i, &r, tmp) : g->mode == FO_GEN_REVERSED ? encode_row_reversed(&r, tmp) : encode_row(&r, tmp);
    size_t k = 0;
    if (g->framed) k = put_varint(out, n); /* proto.Buffer.EncodeMessage, mocker.go:99-101 */
    memcpy(out + k, tmp, n);
    return k + n;
}

uint32_t fo_gen_record_len(const fo_gen_params* g, uint64_t i) {
    uint8_t tmp[FO_GEN_MAX_RECORD + 16];
    return (uint32_t)gen_one(g, i, tmp);
}

size_t fo_gen_records(const fo_gen_params* g, uint64_t i0, uint64_t n, uint8_t* out, size_t cap,
                      uint64_t* offsets) {
    size_t pos = 0;
    uint8_t tmp[FO_GEN_MAX_RECORD + 16];
    for (uint64_t k = 0; k < n; k++) {
        size_t l = gen_one(g, i0 + k, tmp);
        if (pos + l > cap) return (size_t)-1;
        memcpy(out + pos, tmp, l);
        if (offsets) offsets[k] = pos;
        pos += l;
    }
    if (offsets) offsets[n] = pos;
    return pos;
}

/* ------------------------------------------------------------- cpu bench */
/* One shard of the record range per thread.  Untimed: every thread generates its own shard (so that the sample is
 * ready in seconds on a many-core host).  Timed, between two barriers: (1) decode + roll up the shard into a
 * private table sized for it up front, (2) merge by key partition: every thread sorts its groups by
 * hash % threads, then thread t folds partition t of every shard into the final table part t - one barrier, all
 * threads busy in both phases (the earlier tree merge left most threads idle and ran SLOWER than one core on a
 * 256-thread host).  The result is the set of `threads` disjoint table parts. */
typedef struct {
    uint32_t timeslot, src_as, dst_as, etype;
    uint64_t bytes, packets, count;
} bentry;
typedef struct job {
    const fo_gen_params* g;
    uint64_t i0;
    size_t n;         /* records of this shard */
    uint8_t* buf;     /* the shard's wire bytes (owned) */
    uint64_t* off;
    size_t wire;
    fo_rollup* r;     /* shard table */
    fo_rollup* part;  /* final table part t */
    bentry* ent;      /* this shard's groups, sorted by destination part */
    size_t* ent_off;  /* threads + 1 */
    size_t ent_cap;   /* entries `ent` holds */
    size_t groups_hint; /* expected number of distinct groups of the whole range (0: unknown) */
    uint64_t bad;
    int t, threads, failed;
    double t_start, t_end, t_ingest;
    struct job* all;
    pthread_barrier_t* bar;
} job;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static size_t pow2_at_least(size_t v) {
    size_t c = 1024;
    while (c < v) c <<= 1;
    return c;
}
/* Large tables sit on 2 MiB pages where the kernel offers them (THP "madvise"): a random probe into a table of
 * tens of MB is a TLB miss per record on 4 KiB pages, and under a hypervisor a TLB miss is a two-dimensional page
 * walk - with 8 shard tables that walk, not the cores, bounded the sweep (measured here: 8 threads 1.8x one). */
static slot* table_alloc(size_t cap) {
    const size_t bytes = cap * sizeof(slot);
    if (bytes >= ((size_t)4 << 20)) {
        void* p = NULL;
        const size_t huge = (size_t)2 << 20;
        if (posix_memalign(&p, huge, (bytes + huge - 1) & ~(huge - 1)) == 0) {
            (void)madvise(p, (bytes + huge - 1) & ~(huge - 1), MADV_HUGEPAGE);
            memset(p, 0, bytes);
            return (slot*)p;
        }
    }
    return (slot*)calloc(cap, sizeof(slot));
}
static fo_rollup* rollup_new_sized(uint32_t gran, size_t groups) {
    fo_rollup* r = (fo_rollup*)calloc(1, sizeof *r);
    r->gran = gran ? gran : 300;
    r->cap = pow2_at_least(groups * 2 + 2);
    r->t = table_alloc(r->cap);
    return r;
}
static uint64_t slot_hash(const slot* s) {
    return mix64(((uint64_t)s->src_as << 32 | s->dst_as) ^ mix64((uint64_t)s->timeslot << 32 | s->etype));
}

static void* job_run(void* a) {
    job* j = (job*)a;
    const int T = j->threads;
    /* untimed: this shard's records, and the tables - allocated AND first-touched here (round 2 allocated the shard
     * table inside the timed region: 100 MB of calloc page faults per thread, serialised in the kernel on a many-core
     * host), sized from the caller's group-count hint (config 2: 393 216 groups, whatever the shard length). */
    const size_t cap = j->n * (j->g->mode == FO_GEN_GOFLOW ? 200 : 96) + 256;
    j->buf = (uint8_t*)malloc(cap);
    j->off = (uint64_t*)malloc((j->n + 1) * sizeof(uint64_t));
    j->wire = j->buf && j->off ? fo_gen_records(j->g, j->i0, j->n, j->buf, cap, j->off) : (size_t)-1;
    j->failed = j->wire == (size_t)-1;
    {
        size_t shard_groups = j->groups_hint ? j->groups_hint : (1u << 20);
        if (shard_groups > j->n) shard_groups = j->n;
        j->r = rollup_new_sized(300, shard_groups);
        memset(j->r->t, 0, j->r->cap * sizeof(slot)); /* first touch */
        const size_t part_groups = j->groups_hint ? j->groups_hint / (size_t)T + j->groups_hint / (size_t)(4 * T) + 1024 : 1024;
        j->part = rollup_new_sized(300, part_groups);
        memset(j->part->t, 0, j->part->cap * sizeof(slot));
        j->ent_off = (size_t*)calloc((size_t)T + 1, sizeof(size_t));
        j->ent = (bentry*)malloc((shard_groups + 1) * sizeof(bentry));
        if (j->ent) memset(j->ent, 0, (shard_groups + 1) * sizeof(bentry));
        j->ent_cap = shard_groups + 1;
    }
    pthread_barrier_wait(j->bar);
    j->t_start = now_s();
    /* (1) decode + roll up */
    if (!j->failed) j->bad = fo_rollup_ingest(j->r, j->buf, j->off, j->n, (int)j->g->framed);
    j->t_ingest = now_s() - j->t_start;
    /* (2a) this shard's groups, ordered by destination part */
    if (j->r->n + 1 > j->ent_cap) { /* (only when the hint was too small) */
        free(j->ent);
        j->ent = (bentry*)malloc((j->r->n + 1) * sizeof(bentry));
        j->ent_cap = j->r->n + 1;
    }
    for (size_t i = 0; i < j->r->cap; i++)
        if (j->r->t[i].used) j->ent_off[(slot_hash(&j->r->t[i]) >> 40) % (uint64_t)T + 1]++;
    for (int p = 0; p < T; p++) j->ent_off[p + 1] += j->ent_off[p];
    {
        size_t* cur = (size_t*)malloc((size_t)T * sizeof(size_t));
        memcpy(cur, j->ent_off, (size_t)T * sizeof(size_t));
        for (size_t i = 0; i < j->r->cap; i++) {
            const slot* s = &j->r->t[i];
            if (!s->used) continue;
            bentry* e = &j->ent[cur[(slot_hash(s) >> 40) % (uint64_t)T]++];
            e->timeslot = s->timeslot; e->src_as = s->src_as; e->dst_as = s->dst_as; e->etype = s->etype;
            e->bytes = s->bytes; e->packets = s->packets; e->count = s->count;
        }
        free(cur);
    }
    pthread_barrier_wait(j->bar);
    /* (2b) fold partition t of every shard (the part table grows by itself if the hint was too small) */
    for (int u = 0; u < T; u++) {
        const job* o = &j->all[u];
        for (size_t k = o->ent_off[j->t]; k < o->ent_off[j->t + 1]; k++) {
            const bentry* e = &o->ent[k];
            rollup_put(j->part, e->timeslot, e->src_as, e->dst_as, e->etype, e->bytes, e->packets, e->count);
        }
    }
    pthread_barrier_wait(j->bar);
    j->t_end = now_s();
    return NULL;
}

int fo_bench_rollup_ex(const fo_gen_params* g, uint64_t i0, uint64_t n, int threads, uint64_t groups_hint,
                       fo_bench_result* res, fo_row5m* rows, size_t rows_cap) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n && n) threads = (int)n;
    job* jobs = (job*)calloc(threads, sizeof(job));
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads);
    for (int t = 0; t < threads; t++) {
        size_t a = (size_t)(n * t / threads), b = (size_t)(n * (t + 1) / threads);
        jobs[t].g = g;
        jobs[t].i0 = i0 + a;
        jobs[t].n = b - a;
        jobs[t].t = t;
        jobs[t].threads = threads;
        jobs[t].all = jobs;
        jobs[t].bar = &bar;
        jobs[t].groups_hint = (size_t)groups_hint;
    }
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, job_run, &jobs[t]);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    double t0 = jobs[0].t_start, t1 = jobs[0].t_end;
    uint64_t bad = 0, wire = 0, groups = 0, cs = 0;
    int failed = 0;
    size_t w = 0;
    for (int t = 0; t < threads; t++) {
        if (jobs[t].t_start < t0) t0 = jobs[t].t_start;
        if (jobs[t].t_end > t1) t1 = jobs[t].t_end;
        bad += jobs[t].bad;
        wire += jobs[t].failed ? 0 : jobs[t].wire;
        failed |= jobs[t].failed;
        groups += jobs[t].part->n;
        for (size_t i = 0; i < jobs[t].part->cap; i++) {
            const slot* s = &jobs[t].part->t[i];
            if (!s->used) continue;
            cs += mix64(((uint64_t)s->timeslot << 32 | s->etype) ^
                        mix64((uint64_t)s->src_as << 32 | s->dst_as)) *
                  (s->bytes * 3 + s->packets * 5 + s->count * 7 + 1);
            if (rows && w < rows_cap) {
                fo_row5m* o = &rows[w];
                o->date = s->timeslot / 86400u;
                o->timeslot = s->timeslot;
                o->src_as = s->src_as;
                o->dst_as = s->dst_as;
                o->etype = s->etype;
                o->_pad = 0;
                o->bytes = s->bytes;
                o->packets = s->packets;
                o->count = s->count;
            }
            w++;
        }
    }
    if (rows && w <= rows_cap) qsort(rows, w, sizeof(fo_row5m), row5m_cmp);
    const double dt = t1 - t0;
    double lo = 1e99, hi = 0, sum = 0;
    for (int t = 0; t < threads; t++) {
        if (jobs[t].t_ingest < lo) lo = jobs[t].t_ingest;
        if (jobs[t].t_ingest > hi) hi = jobs[t].t_ingest;
        sum += jobs[t].t_ingest;
    }
    if (getenv("FO_BENCH_VERBOSE"))
        fprintf(stderr, "[oracle bench] %d threads: shard decode+rollup %.3f..%.3f s, whole timed region %.3f s\n", threads, lo, hi, dt);
    pthread_barrier_destroy(&bar);
    if (res) {
        res->seconds = dt;
        res->decode_min = lo;
        res->decode_max = hi;
        res->decode_mean = sum / threads;
        res->merge_seconds = dt - hi;
        res->wire_bytes = wire;
        res->groups = groups;
        res->bad = failed ? ~0ull : bad;
        res->checksum = cs;
        res->rows = w;
        res->threads = (uint32_t)threads;
        res->_pad = 0;
    }
    for (int t = 0; t < threads; t++) {
        fo_rollup_free(jobs[t].r);
        fo_rollup_free(jobs[t].part);
        free(jobs[t].ent);
        free(jobs[t].ent_off);
        free(jobs[t].buf);
        free(jobs[t].off);
    }
    free(jobs);
    free(th);
    return (rows && w > rows_cap) ? FO_BAD : FO_OK;
}

double fo_bench_rollup(const fo_gen_params* g, uint64_t i0, uint64_t n, int threads,
                       uint64_t* wire_out, uint64_t* groups_out, uint64_t* bad_out,
                       uint64_t* checksum_out) {
    fo_bench_result r;
    fo_bench_rollup_ex(g, i0, n, threads, 0, &r, NULL, 0);
    if (wire_out) *wire_out = r.wire_bytes;
    if (groups_out) *groups_out = r.groups;
    if (bad_out) *bad_out = r.bad;
    if (checksum_out) *checksum_out = r.checksum;
    return r.seconds;
}

/* ------------------------------------------------- config 3 at full scale */
void fo_zipf_key(uint64_t rank, int dst, int v6, uint8_t out[16]) { zipf_key(rank, dst ? 0x2222 : 0x1111, out, !v6); }

typedef struct {
    const fo_gen_params* g;
    uint64_t i0, n;
    uint32_t depth, wl2;
    uint64_t seed;
    uint64_t *src, *dst;          /* private sketch copies */
    uint64_t *ex_src, *ex_dst;    /* shared exact arrays (atomic adds) or NULL */
} cjob;
static void* cjob_run(void* a) {
    cjob* j = (cjob*)a;
    const fo_gen_params* g = j->g;
    const uint32_t L = g->zipf_log2_universe ? g->zipf_log2_universe : 24;
    for (uint64_t k = 0; k < j->n; k++) {
        fo_row r;
        const uint64_t i = j->i0 + k;
        fo_gen_row(g, i, &r);
        const uint64_t w = r.bytes * r.sampling_rate; /* UInt64 wrap */
        if (w) {
            fo_cms_update(j->src, j->depth, j->wl2, j->seed, r.src_addr, w);
            fo_cms_update(j->dst, j->depth, j->wl2, j->seed, r.dst_addr, w);
        }
        if (j->ex_src && g->mode == FO_GEN_ZIPF) { /* the ranks behind the two addresses: the generator's own draws */
            const uint64_t r3 = gen_rnd(g, i, 3), r4 = gen_rnd(g, i, 4);
            const uint64_t v6 = r.etype == 0x86dd;
            __atomic_fetch_add(&j->ex_src[zipf_rank(g, r3) + (v6 << L)], w, __ATOMIC_RELAXED);
            __atomic_fetch_add(&j->ex_dst[zipf_rank(g, r4) + (v6 << L)], w, __ATOMIC_RELAXED);
        }
    }
    return NULL;
}
void fo_cms_stream(const fo_gen_params* g, uint64_t i0, uint64_t n, int threads, uint32_t depth, uint32_t wl2,
                   uint64_t seed, uint64_t* cms_src, uint64_t* cms_dst, uint64_t* exact_src, uint64_t* exact_dst) {
    if (threads < 1) threads = 1;
    const size_t words = (size_t)depth << wl2;
    cjob* jobs = (cjob*)calloc(threads, sizeof(cjob));
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        jobs[t].g = g;
        jobs[t].i0 = i0 + n * t / threads;
        jobs[t].n = n * (t + 1) / threads - n * t / threads;
        jobs[t].depth = depth;
        jobs[t].wl2 = wl2;
        jobs[t].seed = seed;
        jobs[t].src = (uint64_t*)calloc(words, 8);
        jobs[t].dst = (uint64_t*)calloc(words, 8);
        jobs[t].ex_src = exact_src;
        jobs[t].ex_dst = exact_dst;
        pthread_create(&th[t], NULL, cjob_run, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        for (size_t k = 0; k < words; k++) {
            cms_src[k] += jobs[t].src[k];
            cms_dst[k] += jobs[t].dst[k];
        }
        free(jobs[t].src);
        free(jobs[t].dst);
    }
    free(jobs);
    free(th);
}

/* ------------------------------------------------- config 5 at full scale: (SrcAddr,DstPort,Proto) rows without a group-by
 * A window of BASELINE config 5 holds 16.6 M rows of this key set; grouping 100 M records on the CPU to compare them row by
 * row takes minutes.  The rollup is a sum per key, so a checksum that is LINEAR in the sums needs no grouping:
 *   sum over rows   h(key) * (3*sum(Bytes) + 5*sum(Packets) + 7*count())                      (GPU side, over its rows)
 * = sum over records h(key of the record) * (3*Bytes + 5*Packets + 7)                          (here, over the stream)
 * mod 2^64, with key = (Timeslot, SrcAddr, DstPort, Proto) and the Date/Timeslot rule of flows_5m_view (create.sh:92-110:
 * Timeslot = t - t mod gran of the DateTime-narrowed TimeReceived, create.sh:39,66).  Together with "the GPU's rows are
 * strictly ascending by key" (every key once) equal checksums mean equal rows, up to a 64-bit hash collision.
 * out_sum / out_cnt: nslots entries, slot k = timeslot slot0 + k*gran (records outside are counted in *outside). */
static uint64_t app_key_hash(uint32_t timeslot, const uint8_t a[16], uint32_t dst_port, uint32_t proto) {
    uint64_t lo, hi;
    memcpy(&lo, a, 8);
    memcpy(&hi, a + 8, 8);
    uint64_t h = mix64(((uint64_t)timeslot << 32 | dst_port) ^ 0x9E3779B97F4A7C15ull);
    h = mix64(h ^ lo);
    h = mix64(h ^ hi);
    return mix64(h ^ proto);
}
typedef struct {
    const fo_gen_params* g;
    uint64_t i0, n;
    uint32_t gran, slot0, nslots;
    uint64_t *sum, *cnt, outside;
} ajob;
static void* ajob_run(void* a) {
    ajob* j = (ajob*)a;
    for (uint64_t k = 0; k < j->n; k++) {
        fo_row r;
        fo_gen_row(j->g, j->i0 + k, &r);
        const uint32_t t32 = (uint32_t)r.time_received;
        const uint32_t ts = t32 - t32 % j->gran;
        const uint32_t s = (ts - j->slot0) / j->gran;
        if (ts < j->slot0 || s >= j->nslots) {
            j->outside++;
            continue;
        }
        j->sum[s] += app_key_hash(ts, r.src_addr, r.dst_port, r.proto) * (3 * r.bytes + 5 * r.packets + 7);
        j->cnt[s] += 1;
    }
    return NULL;
}
uint64_t fo_app_checksum_stream(const fo_gen_params* g, uint64_t i0, uint64_t n, int threads, uint32_t gran, uint32_t slot0,
                                uint32_t nslots, uint64_t* out_sum, uint64_t* out_cnt) {
    if (threads < 1) threads = 1;
    ajob* jobs = (ajob*)calloc(threads, sizeof(ajob));
    pthread_t* th = (pthread_t*)calloc(threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        jobs[t].g = g;
        jobs[t].i0 = i0 + n * t / threads;
        jobs[t].n = n * (t + 1) / threads - n * t / threads;
        jobs[t].gran = gran;
        jobs[t].slot0 = slot0;
        jobs[t].nslots = nslots;
        jobs[t].sum = (uint64_t*)calloc(nslots, 8);
        jobs[t].cnt = (uint64_t*)calloc(nslots, 8);
        pthread_create(&th[t], NULL, ajob_run, &jobs[t]);
    }
    uint64_t outside = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        for (uint32_t k = 0; k < nslots; k++) {
            out_sum[k] += jobs[t].sum[k];
            out_cnt[k] += jobs[t].cnt[k];
        }
        outside += jobs[t].outside;
        free(jobs[t].sum);
        free(jobs[t].cnt);
    }
    free(jobs);
    free(th);
    return outside;
}
