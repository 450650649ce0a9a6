// wire.cuh - device-side proto3 FlowMessage parsers (gfx950).
//
// What is parsed: the record type of pb-ext/flow.proto:7-65 (and the 67-field
// superset of pb-ext/flow.pb.go:57-147, whose extra fields are skipped by wire
// type), projected onto the 15 ClickHouse `flows` columns
// (compose/clickhouse/create.sh:7-27).  Decode semantics are the proto3 rules
// listed in SURVEY.md Appendix A.2 (what proto.Unmarshal does at
// inserter/inserter.go:122-126).
//
// Three tiers (each one only ever answers "decoded exactly" or "not sure"; the last is the truth):
//   parse_canon<>   - speculates that the record is what proto.Marshal (mocker.go:97) / GoFlow emit:
//                     each pb-ext/flow.proto field at most once, in ascending field-number order,
//                     minimal tags, small varints.  Straight-line walk over the 27 schema fields
//                     with wave-uniform skips of the fields no lane of the wave carries
//                     (~280 VALU instructions for a mocker-shaped record, independent of order
//                     checks, wire-type dispatch or capture selects).  Anything else - other field
//                     order, duplicates, unknown fields, long varints, groups - leaves the cursor
//                     short of the record end and is answered "not sure".
//   parse_fast<>    - one record per lane, reads the record through 8-byte
//                     sliding windows built from aligned dword loads (LDS tile
//                     or global memory).  Handles every field whose tag is <= 2
//                     bytes and whose varint is <= 6 bytes (values < 2^42), LEN /
//                     fixed32 / fixed64 skips and <=16-byte address payloads.  It
//                     only ever answers "decoded exactly" or "not sure".
//   parse_generic   - byte-at-a-time parser with the complete semantics (10-byte
//                     varints, 5-byte tags, groups with a 100-deep stack, every
//                     error rule).  It is the single source of truth for "bad
//                     record"; records parse_fast is not sure about are deferred
//                     to a second kernel that runs it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

// ---- portability shims: the parsers also compile for the host so that tests can fuzz them
// against the oracle without a GPU (tests/test_host_parsers.py); the product only runs the
// device instantiations.
#define FA_HD __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define FA_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)  // wave-uniform: does any lane ...
#else
#define FA_ANY(c) (c)
#endif
FA_HD uint32_t fa_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, sh);  // uses sh[1:0]
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3)));
#endif
}
// find-first-bit-low; 0xffffffff for 0 (v_ffbl_b32's defined result, which __builtin_ctz does not promise)
FA_HD uint32_t fa_ffbl(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else
    return x ? (uint32_t)__builtin_ctz(x) : 0xffffffffu;
#endif
}

// Column selection bitmask (what a kernel variant needs; the rest is dead code).
enum : uint32_t {
    COL_TIME_RECEIVED = 1u << 0,
    COL_TIME_FLOW_START = 1u << 1,
    COL_SEQUENCE_NUM = 1u << 2,
    COL_SAMPLING_RATE = 1u << 3,
    COL_SAMPLER_ADDRESS = 1u << 4,
    COL_SRC_ADDR = 1u << 5,
    COL_DST_ADDR = 1u << 6,
    COL_SRC_AS = 1u << 7,
    COL_DST_AS = 1u << 8,
    COL_ETYPE = 1u << 9,
    COL_PROTO = 1u << 10,
    COL_SRC_PORT = 1u << 11,
    COL_DST_PORT = 1u << 12,
    COL_BYTES = 1u << 13,
    COL_PACKETS = 1u << 14,
    COL_ALL = (1u << 15) - 1,
    COLS_AS_ROLLUP = COL_TIME_RECEIVED | COL_SRC_AS | COL_DST_AS | COL_ETYPE | COL_BYTES | COL_PACKETS,
};

struct Rec {
    uint64_t time_received, time_flow_start, sampling_rate, bytes, packets;
    uint32_t sequence_num, src_as, dst_as, etype, proto, src_port, dst_port;
    uint32_t sampler[4], src[4], dst[4];  // FixedString(16), little-endian dwords
};

FA_HD void rec_clear(Rec& r) {
    r.time_received = r.time_flow_start = r.sampling_rate = r.bytes = r.packets = 0;
    r.sequence_num = r.src_as = r.dst_as = r.etype = r.proto = r.src_port = r.dst_port = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) r.sampler[i] = r.src[i] = r.dst[i] = 0;
}

// ---- byte sources ---------------------------------------------------------
struct LdsSrc {
    const uint32_t* base;  // LDS, dword aligned
    FA_HD uint32_t dw(uint32_t i) const { return base[i]; }
};
// LDS by ABSOLUTE byte address: dword i = the four bytes at LDS address 4 i.  The cursors of a walk then ARE LDS addresses
// (tile base folded in once per record) and a window load is "and -4, ds_read2, ds_read" - the add of the tile's base that
// every field step of LdsSrc pays is gone (one VALU instruction per step: 14 of a mocker-shaped record's ~420, 33 of GoFlow's).
struct LdsAbsSrc {
    FA_HD uint32_t dw(uint32_t i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const __attribute__((address_space(3))) uint32_t* lds_p;
        return *(lds_p)(i << 2);
#else
        return 0u * i;
#endif
    }
};
struct GlobalSrc {
    const uint32_t* base;  // global, dword aligned
    FA_HD uint32_t dw(uint32_t i) const { return base[i]; }
};

// 8 bytes starting at byte offset pos (little endian), from aligned dwords.
template <class Src>
FA_HD uint64_t window64(const Src& s, uint32_t pos) {
    uint32_t i = pos >> 2, sh = pos & 3;
    uint32_t d0 = s.dw(i), d1 = s.dw(i + 1), d2 = s.dw(i + 2);
    uint32_t lo = fa_alignbyte(d1, d0, sh);
    uint32_t hi = fa_alignbyte(d2, d1, sh);
    return (uint64_t)hi << 32 | lo;
}

// Varint of 1..6 bytes sitting in the low bytes of v.  Returns false when the
// stop byte is not within 6 bytes.  *len = encoded length, *val = value (< 2^42).
FA_HD bool varint6(uint64_t v, uint32_t& len, uint64_t& val) {
    const uint64_t m = ~v & 0x0000808080808080ull;
    const uint32_t stop = (uint32_t)__builtin_ctzll(m | (1ull << 63));  // 7,15,...,47 (63: none)
    len = (stop >> 3) + 1;
    const uint64_t x = v & ((2ull << stop) - 1);
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    xl = ((xl & 0x7f007f00u) >> 1) | (xl & 0x007f007fu);
    xl = ((xl & 0x3fff0000u) >> 2) | (xl & 0x00003fffu);
    xh = ((xh & 0x00007f00u) >> 1) | (xh & 0x0000007fu);
    val = (uint64_t)xl | ((uint64_t)xh << 28);
    return m != 0;
}

// Up to 16 payload bytes at byte offset pos, zero padded beyond len (<= 16).
template <class Src>
FA_HD void load_fixed16(const Src& s, uint32_t pos, uint32_t len, uint32_t out[4]) {
    uint32_t i = pos >> 2, sh = pos & 3;
    uint32_t d0 = s.dw(i), d1 = s.dw(i + 1), d2 = s.dw(i + 2), d3 = s.dw(i + 3), d4 = s.dw(i + 4);
    uint32_t w[4];
    w[0] = fa_alignbyte(d1, d0, sh);
    w[1] = fa_alignbyte(d2, d1, sh);
    w[2] = fa_alignbyte(d3, d2, sh);
    w[3] = fa_alignbyte(d4, d3, sh);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int rem = (int)len - 4 * k;  // bytes of this dword that belong to the value
        uint32_t mask = rem >= 4 ? 0xffffffffu : rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u);
        out[k] = w[k] & mask;
    }
}

// Fast parser.  [pos,end) are byte offsets of the bare payload inside src.
// Returns true iff the record was decoded exactly; false = defer to parse_generic.
//
// The loop body is straight-line: no per-field branches (proto3 zero-omission puts
// the lanes of a wave on different fields, a switch() would diverge on almost every
// iteration) and no early exit - every doubt is OR-ed into a sticky `doubt` word and
// checked once after the loop together with "the cursor landed exactly on the record
// end" (any truncated tag/varint/fixed/LEN necessarily overshoots `end`).  Wire-type
// properties come out of tiny in-register lookup tables (shifted constants) instead
// of compare/select chains.  The source must be readable ~32 bytes past `end`.
template <uint32_t COLS, class Src>
FA_HD bool parse_fast(const Src& s, uint32_t pos, uint32_t end, Rec& r) {
    constexpr bool WANT_ADDR = (COLS & (COL_SRC_ADDR | COL_DST_ADDR | COL_SAMPLER_ADDRESS)) != 0;
    uint32_t doubt = 0;
    while (pos < end) {
        const uint64_t w = window64(s, pos);
        const uint32_t w0 = (uint32_t)w;
        const uint32_t two = (w0 >> 7) & 1u;  // 1: 2-byte tag (fields 16..2047)
        const uint32_t tag = (w0 & 0x7fu) | (((w0 >> 1) & 0x3f80u) & (0u - two));
        doubt |= (w0 >> 15) & two;            // tag of 3+ bytes
        const uint32_t wt = tag & 7u;
        const uint64_t v = w >> (8u << two);  // >= 6 valid bytes after the tag
        uint32_t vl;
        uint64_t val;
        const uint32_t novar = varint6(v, vl, val) ? 0u : 1u;
        // per-wire-type facts from shifted constants: 0 varint, 1 fixed64, 2 LEN, 5 fixed32
        const uint32_t hv = (0x05u >> wt) & 1u;              // carries a varint (value or size)
        const uint32_t isl = (0x04u >> wt) & 1u;             // LEN
        const uint32_t fixw = (0x400080u >> (wt * 4u)) & 0xfu;  // fixed payload bytes
        doubt |= ((0x27u >> wt) & 1u) ^ 1u;                  // groups (3,4) and wire types 6,7
        doubt |= tag < 8u ? 1u : 0u;                         // field number 0
        doubt |= hv & novar;                                 // varint longer than 6 bytes
        doubt |= isl & (vl > 5u ? 1u : 0u);                  // LEN size is a varint32
        doubt |= isl & ((uint32_t)(val >> 20) ? 1u : 0u);    // a LEN field of 1 MiB or more: the clamp below would let the cursor land inside the payload
        // LEN payload size, clamped so the cursor can neither wrap nor loop
        const uint32_t sz = (uint32_t)(val >> 20) ? (1u << 20) : (uint32_t)val;
        const uint32_t is_addr = isl & ((((tag | 8u) == 0x3au) | (tag == 0x5au)) ? 1u : 0u);  // fields 6, 7, 11
        doubt |= is_addr & (sz > 16u ? 1u : 0u);             // FixedString(16) overflow
        if (COLS & COL_TIME_RECEIVED) r.time_received = tag == 0x10u ? val : r.time_received;
        if (COLS & COL_SAMPLING_RATE) r.sampling_rate = tag == 0x18u ? val : r.sampling_rate;
        if (COLS & COL_SEQUENCE_NUM) r.sequence_num = tag == 0x20u ? (uint32_t)val : r.sequence_num;
        if (COLS & COL_BYTES) r.bytes = tag == 0x48u ? val : r.bytes;
        if (COLS & COL_PACKETS) r.packets = tag == 0x50u ? val : r.packets;
        if (COLS & COL_SRC_AS) r.src_as = tag == 0x70u ? (uint32_t)val : r.src_as;
        if (COLS & COL_DST_AS) r.dst_as = tag == 0x78u ? (uint32_t)val : r.dst_as;
        if (COLS & COL_PROTO) r.proto = tag == 0xa0u ? (uint32_t)val : r.proto;
        if (COLS & COL_SRC_PORT) r.src_port = tag == 0xa8u ? (uint32_t)val : r.src_port;
        if (COLS & COL_DST_PORT) r.dst_port = tag == 0xb0u ? (uint32_t)val : r.dst_port;
        if (COLS & COL_ETYPE) r.etype = tag == 0xf0u ? (uint32_t)val : r.etype;
        if (COLS & COL_TIME_FLOW_START) r.time_flow_start = tag == 0x130u ? val : r.time_flow_start;
        const uint32_t npos = pos + 1u + two + (vl & (0u - hv)) + fixw;
        if (WANT_ADDR) {
            if (is_addr) {
                uint32_t a16[4];
                load_fixed16(s, npos, sz > 16u ? 16u : sz, a16);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (COLS & COL_SRC_ADDR) r.src[k] = tag == 0x32u ? a16[k] : r.src[k];
                    if (COLS & COL_DST_ADDR) r.dst[k] = tag == 0x3au ? a16[k] : r.dst[k];
                    if (COLS & COL_SAMPLER_ADDRESS) r.sampler[k] = tag == 0x5au ? a16[k] : r.sampler[k];
                }
            }
        }
        pos = npos + (sz & (0u - isl));
    }
    return doubt == 0 && pos == end;
}

// ---- canonical-order parser ---------------------------------------------------------
// Cursor over the record: pos plus the 8 bytes at pos (x = bytes pos..pos+3, y = pos+4..pos+7).
// The window is re-read only after some lane of the wave has advanced.
struct Cursor {
    uint32_t pos, x, y;
};
template <class Src>
FA_HD void cur_load(const Src& s, Cursor& c) {
    const uint32_t i = c.pos >> 2;
    const uint32_t d0 = s.dw(i), d1 = s.dw(i + 1), d2 = s.dw(i + 2);
    c.x = fa_alignbyte(d1, d0, c.pos);
    c.y = fa_alignbyte(d2, d1, c.pos);
}

// 4 varint bytes in v (little endian), stop byte's bit 7 at bit index sb (7,15,23,31) -> value (< 2^28)
FA_HD uint32_t varint28(uint32_t v, uint32_t sb) {
    const uint32_t m = v & ((2u << (sb & 31u)) - 1u) & 0x7f7f7f7fu;
    const uint32_t t = m - ((m & 0x7f007f00u) >> 1);  // two 14-bit halves
    return t - (t >> 16) * 0xC000u;                    // lo14 | hi14 << 14
}

// Varint field whose value fits 4 bytes (< 2^28).  TAG: the 1- or 2-byte tag as a little-endian
// integer.  A longer value, a truncated one or one that would cross `end` is "no match": the cursor
// stays, the final pos == end test fails and the record is deferred to the general parsers.
template <uint32_t TAG, bool WANT, class Src>
FA_HD void canon_short(const Src& s, Cursor& c, uint32_t end, uint32_t& out) {
    constexpr uint32_t TL = TAG > 0xffu ? 2u : 1u;
    const bool m = (c.x & (TL == 1 ? 0xffu : 0xffffu)) == TAG;
    if (FA_ANY(m)) {
        const uint32_t v = fa_alignbyte(c.y, c.x, TL);  // value bytes 0..3
        const uint32_t sb = fa_ffbl(~v & 0x80808080u);
        const uint32_t pn = c.pos + (sb >> 3) + (TL + 1u);  // sb = 0xffffffff (no stop) -> far beyond end
        const bool ok = m && pn <= end;
        if (WANT) {
            const uint32_t val = varint28(v, sb);
            out = ok ? val : out;
        }
        c.pos = ok ? pn : c.pos;
        cur_load(s, c);
    }
}

// Varint field read through the whole 8-byte window: value of up to 8-TL bytes (timestamps,
// sequence numbers).  Produces the full 64-bit value (< 2^49 / 2^42).
template <uint32_t TAG, bool WANT, class Src>
FA_HD void canon_long(const Src& s, Cursor& c, uint32_t end, uint64_t& out) {
    constexpr uint32_t TL = TAG > 0xffu ? 2u : 1u;
    const bool m = (c.x & (TL == 1 ? 0xffu : 0xffffu)) == TAG;
    if (FA_ANY(m)) {
        const uint32_t s0 = fa_ffbl(~c.x & (TL == 1 ? 0x80808000u : 0x80800000u));
        const uint32_t s1 = fa_ffbl(~c.y & 0x80808080u) | 32u;  // stays 0xffffffff when there is no stop
        const uint32_t sb = s0 < s1 ? s0 : s1;                  // window bit index of the stop byte's bit 7
        const uint32_t pn = c.pos + (sb >> 3) + 1u;
        const bool ok = m && pn <= end;
        if (WANT) {
            const uint32_t lo = fa_alignbyte(c.y, c.x, TL);  // value bytes 0..3
            const uint32_t hi = c.y >> (8u * TL);            // value bytes 4..
            const uint32_t sv = (sb - 8u * TL) & 63u;        // stop bit relative to the value (7..55)
            const uint64_t keep = (2ull << sv) - 1ull;
            const uint32_t a = varint28(lo & (uint32_t)keep, 31u);
            const uint32_t b = varint28(hi & (uint32_t)(keep >> 32), 31u);
            const uint64_t val = (uint64_t)a | ((uint64_t)b << 28);
            out = ok ? val : out;
        }
        c.pos = ok ? pn : c.pos;
        cur_load(s, c);
    }
}

// bytes field holding an address: one length byte <= 16 (FixedString(16), create.sh:11-13)
template <uint32_t TAG, bool WANT, class Src>
FA_HD void canon_addr(const Src& s, Cursor& c, uint32_t end, uint32_t out[4]) {
    const bool m = (c.x & 0xffu) == TAG;
    if (FA_ANY(m)) {
        const uint32_t len = (c.x >> 8) & 0xffu;
        const uint32_t pn = c.pos + 2u + len;
        const bool ok = m && len <= 16u && pn <= end;
        if (WANT) {
            uint32_t a16[4];
            load_fixed16(s, c.pos + 2u, len > 16u ? 16u : len, a16);
#pragma unroll
            for (int k = 0; k < 4; k++) out[k] = ok ? a16[k] : out[k];
        }
        c.pos = ok ? pn : c.pos;
        cur_load(s, c);
    }
}

// ---- fields outside the projection, generically (FULL walk) ------------------------------------------
// One field of any skippable shape at the cursor of the lanes with `go`: a varint of up to 10 bytes (MAC addresses
// are 7), fixed32 / fixed64, or a bytes field with a one-byte length.  Anything else (groups, longer lengths, a
// field that would cross `end`) leaves the cursor where it is - the record then fails the final pos == end test.
template <bool TWO, class Src>
FA_HD void canon_skip(const Src& s, Cursor& c, uint32_t end, bool go) {
    constexpr uint32_t TL = TWO ? 2u : 1u;
    constexpr uint32_t NONE = 0x20000000u;  // "no such length": pushes the cursor far beyond any record end
    const uint32_t wt = c.x & 7u;
    const uint32_t v0 = fa_alignbyte(c.y, c.x, TL);  // value bytes 0..3
    const uint32_t v1 = c.y >> (8u * TL);            // value bytes 4..(7-TL); zero above
    const uint32_t s0 = fa_ffbl(~v0 & 0x80808080u);
    const uint32_t s1 = fa_ffbl(~v1 & (0x80808080u >> (8u * TL))) | 32u;  // (stays 0xffffffff when there is no stop)
    const uint32_t sb = s0 < s1 ? s0 : s1;
    uint32_t vlen = sb == 0xffffffffu ? NONE : (sb >> 3) + 1u;
    if (FA_ANY(go && wt == 0u && sb == 0xffffffffu)) {  // a varint longer than the window: its tail is in the next one
        Cursor c2;
        c2.pos = c.pos + 8u;
        cur_load(s, c2);
        const uint32_t s2 = fa_ffbl(~c2.x & 0x80808080u);
        const uint32_t l2 = (s2 >> 3) + 1u;  // bytes of the varint in the second window
        const bool ok2 = s2 != 0xffffffffu && l2 <= 2u + TL;  // 10 bytes at most in all
        vlen = sb == 0xffffffffu ? (ok2 ? (8u - TL) + l2 : NONE) : vlen;
    }
    const uint32_t lb = v0 & 0xffu;  // bytes field: one length byte
    // (selects side by side, & instead of &&: as one chain of ?: with a short circuit in it this compiled into a decision tree
    // of exec-mask branches, a dozen per call)
    uint32_t body = NONE;
    body = wt == 5u ? 4u : body;
    body = wt == 1u ? 8u : body;
    body = ((wt == 2u) & (lb < 0x80u)) ? 1u + lb : body;
    body = wt == 0u ? vlen : body;
    const uint32_t pn = c.pos + TL + body;
    const bool ok = go && pn <= end;
    c.pos = ok ? pn : c.pos;
    cur_load(s, c);
}

// A run of fields outside the projection: every field whose tag lies in [LO, HI] is skipped, in any order and any
// number (they carry no projected column, so their order and multiplicity cannot change the result; a projected
// field's tag is never inside a run, so a duplicate or out-of-place projected field still fails the record).
// 1-byte tags: LO, HI < 0x80.  2-byte tags: key = b0 | b1 << 8 with b0 >= 0x80, monotonic in the field number;
// HI <= 0x7fff excludes tags of 3 and more bytes, LO >= 0x0180 excludes non-minimal ones (b1 = 0).
template <uint32_t LO, uint32_t HI, class Src>
FA_HD void canon_run(const Src& s, Cursor& c, uint32_t end) {
    constexpr bool TWO = LO >= 0x80u;
    static_assert(TWO ? (LO >= 0x0180u && HI <= 0x7fffu) : HI < 0x80u, "canon_run: tag range");
    for (int it = 0; it < 96; it++) {  // (bounded: a record of <= 16 KiB cannot hold more fields per run that matter)
        const uint32_t key = TWO ? (c.x & 0xffffu) : (c.x & 0xffu);
        bool go = (key - LO) <= (HI - LO) && c.pos < end;
        if (TWO) go = go && (c.x & 0x80u) != 0u;
        if (!FA_ANY(go)) break;
        const uint32_t before = c.pos;
        canon_skip<TWO>(s, c, end, go);
        if (!FA_ANY(go && c.pos != before)) break;  // only lanes that cannot move are left
    }
}

// r must be cleared by the caller.  Returns true iff [pos,end) is exactly a canonical encoding.
// Field list = pb-ext/flow.proto:16-64 in field-number order (what proto.Marshal emits).
// FULL: the walk also crosses every field of the 67-field message GoFlow marshals (pb-ext/flow.pb.go:57-147:
// NextHop 12, NextHopAS 13, SrcNet / DstNet 16-17, SrcMac / DstMac / VlanId 27-29, SrcVlan .. FragmentOffset 33-36,
// VRF ids, encapsulation, MPLS, PPP 39-64, country / ASDB 100-103) and any other field up to number 2047 - as
// generic runs between the projected fields.  The lean walk costs nothing extra on flow.proto-shaped records; a
// wave switches to the full one when its records need it (lane_work).
template <uint32_t COLS, bool FULL = false, class Src>
FA_HD bool parse_canon(const Src& s, uint32_t pos, uint32_t end, Rec& r) {
    Cursor c;
    c.pos = pos;
    cur_load(s, c);
    uint32_t d32 = 0;  // discarded captures
    uint64_t d64 = 0;
    uint32_t sr32 = 0, by32 = 0, pk32 = 0;
    canon_short<0x08u, false>(s, c, end, d32);  //  1 Type
    canon_long<0x10u, (COLS & COL_TIME_RECEIVED) != 0>(s, c, end, r.time_received);         //  2 TimeReceived
    canon_short<0x18u, (COLS & COL_SAMPLING_RATE) != 0>(s, c, end, sr32);                   //  3 SamplingRate
    {
        uint64_t seq = 0;
        canon_long<0x20u, (COLS & COL_SEQUENCE_NUM) != 0>(s, c, end, seq);                  //  4 SequenceNum
        r.sequence_num = (uint32_t)seq;
    }
    canon_long<0x28u, false>(s, c, end, d64);                                               //  5 TimeFlowEnd
    canon_addr<0x32u, (COLS & COL_SRC_ADDR) != 0>(s, c, end, r.src);                        //  6 SrcAddr
    canon_addr<0x3au, (COLS & COL_DST_ADDR) != 0>(s, c, end, r.dst);                        //  7 DstAddr
    if (FULL) canon_run<0x40u, 0x47u>(s, c, end);                                           //  8 (unassigned)
    canon_short<0x48u, (COLS & COL_BYTES) != 0>(s, c, end, by32);                           //  9 Bytes
    canon_short<0x50u, (COLS & COL_PACKETS) != 0>(s, c, end, pk32);                         // 10 Packets
    canon_addr<0x5au, (COLS & COL_SAMPLER_ADDRESS) != 0>(s, c, end, r.sampler);             // 11 SamplerAddress
    if (FULL) canon_run<0x60u, 0x6fu>(s, c, end);                                           // 12 NextHop, 13 NextHopAS
    canon_short<0x70u, (COLS & COL_SRC_AS) != 0>(s, c, end, r.src_as);                      // 14 SrcAS
    canon_short<0x78u, (COLS & COL_DST_AS) != 0>(s, c, end, r.dst_as);                      // 15 DstAS
    if (FULL) canon_run<0x0180u, 0x018fu>(s, c, end);                                       // 16 SrcNet, 17 DstNet
    // Runs of fields that flow exporters rarely fill are guarded by ONE tag-range test per run (wave-uniform)
    // instead of one test per field: 18..20, 23..26, 31..37.  (A lane can only be inside a run if its tag was
    // in the run's range when the run started - fields come in ascending order.)
    if (FA_ANY(((c.x & 0xffffu) - 0x0190u) <= 0x0010u)) {
        canon_short<0x0190u, false>(s, c, end, d32);  // 18 InIf
        canon_short<0x0198u, false>(s, c, end, d32);  // 19 OutIf
        canon_short<0x01a0u, (COLS & COL_PROTO) != 0>(s, c, end, r.proto);                  // 20 Proto
    }
    canon_short<0x01a8u, (COLS & COL_SRC_PORT) != 0>(s, c, end, r.src_port);                // 21 SrcPort
    canon_short<0x01b0u, (COLS & COL_DST_PORT) != 0>(s, c, end, r.dst_port);                // 22 DstPort
    if (FA_ANY(((c.x & 0xffffu) - 0x01b8u) <= 0x0018u)) {
        canon_short<0x01b8u, false>(s, c, end, d32);  // 23 IPTos
        canon_short<0x01c0u, false>(s, c, end, d32);  // 24 ForwardingStatus
        canon_short<0x01c8u, false>(s, c, end, d32);  // 25 IPTTL
        canon_short<0x01d0u, false>(s, c, end, d32);  // 26 TCPFlags
    }
    if (FULL) canon_run<0x01d8u, 0x01efu>(s, c, end);                                       // 27 SrcMac, 28 DstMac, 29 VlanId
    canon_short<0x01f0u, (COLS & COL_ETYPE) != 0>(s, c, end, r.etype);                      // 30 Etype
    if (FULL) {
        canon_run<0x01f8u, 0x02afu>(s, c, end);                                             // 31 IcmpType .. 37 IPv6FlowLabel
    } else if (FA_ANY(((c.x & 0xffffu) - 0x01f8u) <= 0x00b0u)) {
        canon_short<0x01f8u, false>(s, c, end, d32);  // 31 IcmpType
        canon_short<0x0280u, false>(s, c, end, d32);  // 32 IcmpCode
        canon_short<0x02a8u, false>(s, c, end, d32);  // 37 IPv6FlowLabel
    }
    canon_long<0x02b0u, (COLS & COL_TIME_FLOW_START) != 0>(s, c, end, r.time_flow_start);   // 38 TimeFlowStart
    if (FULL) canon_run<0x02b8u, 0x7fffu>(s, c, end);                                       // 39 .. 2047 (VRF, 42 FlowDirection, encap, MPLS, PPP, country, ASDB)
    else canon_short<0x02d0u, false>(s, c, end, d32);  // 42 FlowDirection
    r.sampling_rate = sr32;
    r.bytes = by32;
    r.packets = pk32;
    return c.pos == end;
}

// ---- template walks --------------------------------------------------------------------------------------
// The canonical walk above asks, for each of the 27 schema fields, "does any lane carry it?" - on the streams the
// reference's own producers emit, 13 of those questions are always answered "no" and the other 14 "yes, every lane".
// A template walk is the same walk compiled for ONE producer's field list: no wave-level question per field (every
// lane executes every step; a lane whose record omits the field - proto3 zero omission - simply does not move), no
// step for fields the producer never sets, and per field the cheapest exact step for its shape:
//   tw_time5   a varint of exactly 5 bytes (a Unix timestamp between 1978 and 3058) - tag and continuation bits checked
//              with two masked compares;
//   tw_short   a varint whose value fits 4 bytes (< 2^28);            tw_skip   a varint that ends inside the window;
//   tw_addr    a bytes field with a one-byte length <= 16;            tw_skip7 for 7-byte MAC varints.
// Same contract as every tier: a step only moves the cursor over a field it has validated completely, steps come in
// ascending field order, so "cursor == end" means the record is exactly a sub-sequence of the template's fields - decoded
// exactly - and anything else (another field, another order, a longer varint, a timestamp outside 1978..3058) is "not
// sure" and goes to the general tiers.  Shapes: what mocker/mocker.go:76-91 sets (+ Proto, which BASELINE's Zipf
// configurations set), and the 33 fields GoFlow fills for an sFlow sample (pb-ext/flow.pb.go:57-147).
FA_HD uint32_t fa_ubfe(uint32_t v, uint32_t off, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, off, width);  // uses off[4:0], width[4:0]
#else
    off &= 31u;
    width &= 31u;
    return width ? (v >> off) & ((1u << width) - 1u) : 0u;
#endif
}
// value (< 2^28) of a varint of <= 4 bytes: v = its bytes (little endian, anything above), sb = bit index of the stop
// byte's bit 7 (7, 15, 23, 31)
FA_HD uint32_t varint28b(uint32_t v, uint32_t sb) {
    const uint32_t m = fa_ubfe(v, 0u, sb) & 0x7f7f7f7fu;  // (sb = 31: bits 0..30; bit 31 is the stop byte's clear bit 7)
    const uint32_t t = m - ((m & 0x7f007f00u) >> 1);      // two 14-bit halves
    return t - (t >> 16) * 0xC000u;                        // lo14 | hi14 << 14
}

template <uint32_t TAG, bool WANT, bool LAST = false, class Src>
FA_HD void tw_short(const Src& s, Cursor& c, uint32_t end, uint32_t& out) {
    constexpr uint32_t TL = TAG > 0xffu ? 2u : 1u;
    const bool m = (c.x & (TL == 1 ? 0xffu : 0xffffu)) == TAG;
    const uint32_t v = fa_alignbyte(c.y, c.x, TL);  // value bytes 0..3
    const uint32_t sb = fa_ffbl(~v & 0x80808080u);
    const uint32_t pn = c.pos + (sb >> 3) + (TL + 1u);  // sb = 0xffffffff (no stop) -> far beyond end
    const bool ok = m && pn <= end;
    if (WANT) {
        const uint32_t val = varint28b(v, sb);
        out = ok ? val : out;
    }
    c.pos = ok ? pn : c.pos;
    if (!LAST) cur_load(s, c);
}
// a varint outside the projection that ends inside the window (value of <= 8 - TL bytes)
template <uint32_t TAG, bool LAST = false, class Src>
FA_HD void tw_skip(const Src& s, Cursor& c, uint32_t end) {
    constexpr uint32_t TL = TAG > 0xffu ? 2u : 1u;
    const bool m = (c.x & (TL == 1 ? 0xffu : 0xffffu)) == TAG;
    const uint32_t s0 = fa_ffbl(~c.x & (TL == 1 ? 0x80808000u : 0x80800000u));
    const uint32_t s1 = fa_ffbl(~c.y & 0x80808080u) | 32u;  // stays 0xffffffff when there is no stop
    const uint32_t sb = s0 < s1 ? s0 : s1;
    const uint32_t pn = c.pos + (sb >> 3) + 1u;
    const bool ok = m && pn <= end;
    c.pos = ok ? pn : c.pos;
    if (!LAST) cur_load(s, c);
}
// a varint outside the projection of up to 7 bytes behind a 2-byte tag (GoFlow's MAC addresses: uint64 fields holding 48 bits):
// the stop byte is looked for in the window and in the one byte behind it that the cursor's three dwords always hold
// (12 - (pos & 3) >= 9 bytes from pos).  Branch-free - the general canon_skip in this place (any wire type, a second window
// behind a wave-level question) compiled into a decision tree of a dozen exec-mask branches per field.  A longer varint leaves
// the cursor where it is: not sure, the general tiers take the record.
template <uint32_t TAG, class Src>
FA_HD void tw_skip7(const Src& s, Cursor& c, uint32_t end) {
    static_assert(TAG > 0xffu, "tw_skip7: 2-byte tags");
    const bool m = (c.x & 0xffffu) == TAG;
    const uint32_t z = fa_alignbyte(0u, s.dw((c.pos >> 2) + 2u), c.pos);  // byte 8 (.. 11 - (pos & 3)), zero above
    const uint32_t s0 = fa_ffbl(~c.x & 0x80800000u);
    const uint32_t s1 = fa_ffbl(~c.y & 0x80808080u) | 32u;  // (stays 0xffffffff when there is no stop)
    const uint32_t s2 = fa_ffbl(~z & 0x80u) | 64u;
    const uint32_t s01 = s0 < s1 ? s0 : s1;
    const uint32_t sb = s01 < s2 ? s01 : s2;
    const uint32_t pn = c.pos + (sb >> 3) + 1u;
    const bool ok = m && pn <= end;
    c.pos = ok ? pn : c.pos;
    cur_load(s, c);
}
// a varint of exactly 5 bytes (value in [2^28, 2^35)): timestamps
template <uint32_t TAG, bool WANT, bool LAST = false, class Src>
FA_HD void tw_time5(const Src& s, Cursor& c, uint32_t end, uint64_t& out) {
    constexpr uint32_t TL = TAG > 0xffu ? 2u : 1u;
    // tag bytes equal, value bytes 0..3 with the continuation bit, byte 4 without: two masked compares
    const bool m = TL == 1 ? ((c.x & 0x808080ffu) == (0x80808000u | TAG) && (c.y & 0x00008080u) == 0x00000080u)
                           : ((c.x & 0x8080ffffu) == (0x80800000u | TAG) && (c.y & 0x00808080u) == 0x00008080u);
    const uint32_t pn = c.pos + TL + 5u;
    const bool ok = m && pn <= end;
    if (WANT) {
        const uint32_t v = fa_alignbyte(c.y, c.x, TL) & 0x7f7f7f7fu;  // value bytes 0..3, payload bits
        const uint32_t t = v - ((v & 0x7f007f00u) >> 1);
        const uint32_t lo28 = t - (t >> 16) * 0xC000u;
        const uint32_t b4 = fa_ubfe(c.y, 8u * TL, 7u);
        const uint64_t val = (uint64_t)lo28 | ((uint64_t)b4 << 28);
        out = ok ? val : out;
    }
    c.pos = ok ? pn : c.pos;
    if (!LAST) cur_load(s, c);
}
template <uint32_t TAG, bool WANT, class Src>
FA_HD void tw_addr(const Src& s, Cursor& c, uint32_t end, uint32_t out[4]) {
    const bool m = (c.x & 0xffu) == TAG;
    const uint32_t len = fa_ubfe(c.x, 8u, 8u);
    const uint32_t pn = c.pos + 2u + len;
    const bool ok = m && len <= 16u && pn <= end;
    if (WANT) {
        uint32_t a16[4];
        load_fixed16(s, c.pos + 2u, len > 16u ? 16u : len, a16);
#pragma unroll
        for (int k = 0; k < 4; k++) out[k] = ok ? a16[k] : out[k];
    }
    c.pos = ok ? pn : c.pos;
    cur_load(s, c);
}

enum : int { SHAPE_MOCKER = 0, SHAPE_GOFLOW = 1 };
// r must be cleared by the caller.  Returns true iff [pos,end) is exactly an encoding of (a sub-sequence of) the shape's
// fields, in field-number order.
template <uint32_t COLS, int SHAPE, class Src>
FA_HD bool parse_tmpl(const Src& s, uint32_t pos, uint32_t end, Rec& r) {
    constexpr bool GF = SHAPE == SHAPE_GOFLOW;
    Cursor c;
    c.pos = pos;
    cur_load(s, c);
    uint32_t d32 = 0, sr32 = 0, by32 = 0, pk32 = 0;
    uint64_t d64 = 0;
    if (GF) tw_short<0x08u, false>(s, c, end, d32);  //  1 Type
    tw_time5<0x10u, (COLS & COL_TIME_RECEIVED) != 0>(s, c, end, r.time_received);              //  2 TimeReceived
    tw_short<0x18u, (COLS & COL_SAMPLING_RATE) != 0>(s, c, end, sr32);                         //  3 SamplingRate
    if (COLS & COL_SEQUENCE_NUM) {                                                             //  4 SequenceNum (uint32: <= 5 bytes)
        uint64_t seq = 0;
        canon_long<0x20u, true>(s, c, end, seq);
        r.sequence_num = (uint32_t)seq;
    } else {
        tw_skip<0x20u>(s, c, end);
    }
    if (GF) tw_time5<0x28u, false>(s, c, end, d64);                                            //  5 TimeFlowEnd
    tw_addr<0x32u, (COLS & COL_SRC_ADDR) != 0>(s, c, end, r.src);                              //  6 SrcAddr
    tw_addr<0x3au, (COLS & COL_DST_ADDR) != 0>(s, c, end, r.dst);                              //  7 DstAddr
    tw_short<0x48u, (COLS & COL_BYTES) != 0>(s, c, end, by32);                                 //  9 Bytes
    tw_short<0x50u, (COLS & COL_PACKETS) != 0>(s, c, end, pk32);                               // 10 Packets
    if (GF) {
        tw_addr<0x5au, (COLS & COL_SAMPLER_ADDRESS) != 0>(s, c, end, r.sampler);               // 11 SamplerAddress
        uint32_t nh[4];
        tw_addr<0x62u, false>(s, c, end, nh);                                                  // 12 NextHop
        tw_short<0x68u, false>(s, c, end, d32);  // 13 NextHopAS
    }
    tw_short<0x70u, (COLS & COL_SRC_AS) != 0>(s, c, end, r.src_as);                            // 14 SrcAS
    tw_short<0x78u, (COLS & COL_DST_AS) != 0>(s, c, end, r.dst_as);                            // 15 DstAS
    if (GF) {
        tw_short<0x0180u, false>(s, c, end, d32);  // 16 SrcNet
        tw_short<0x0188u, false>(s, c, end, d32);  // 17 DstNet
        tw_short<0x0190u, false>(s, c, end, d32);  // 18 InIf
        tw_short<0x0198u, false>(s, c, end, d32);  // 19 OutIf
        tw_short<0x01a0u, (COLS & COL_PROTO) != 0>(s, c, end, r.proto);                        // 20 Proto
    } else if (FA_ANY((c.x & 0xffffu) == 0x01a0u)) {  // (mocker.go never sets it; BASELINE's Zipf configurations do)
        tw_short<0x01a0u, (COLS & COL_PROTO) != 0>(s, c, end, r.proto);
    }
    tw_short<0x01a8u, (COLS & COL_SRC_PORT) != 0>(s, c, end, r.src_port);                      // 21 SrcPort
    tw_short<0x01b0u, (COLS & COL_DST_PORT) != 0>(s, c, end, r.dst_port);                      // 22 DstPort
    if (GF) {
        tw_short<0x01b8u, false>(s, c, end, d32);  // 23 IPTos
        tw_short<0x01c8u, false>(s, c, end, d32);  // 25 IPTTL
        tw_short<0x01d0u, false>(s, c, end, d32);  // 26 TCPFlags
        tw_skip7<0x01d8u>(s, c, end);              // 27 SrcMac (7-byte varint)
        tw_skip7<0x01e0u>(s, c, end);              // 28 DstMac
        tw_short<0x01e8u, false>(s, c, end, d32);  // 29 VlanId
    }
    tw_short<0x01f0u, (COLS & COL_ETYPE) != 0>(s, c, end, r.etype);                            // 30 Etype
    if (GF) {
        tw_short<0x0288u, false>(s, c, end, d32);  // 33 SrcVlan
        tw_short<0x0290u, false>(s, c, end, d32);  // 34 DstVlan
        tw_short<0x0298u, false>(s, c, end, d32);  // 35 FragmentId
        tw_short<0x02a8u, false>(s, c, end, d32);  // 37 IPv6FlowLabel
    }
    tw_time5<0x02b0u, (COLS & COL_TIME_FLOW_START) != 0, true>(s, c, end, r.time_flow_start);  // 38 TimeFlowStart
    r.sampling_rate = sr32;
    r.bytes = by32;
    r.packets = pk32;
    (void)d32;
    return c.pos == end;
}

// ---- learnt field order -----------------------------------------------------------------------------------
// A producer that does not marshal in field-number order (proto3 allows any order) still marshals every record the same way.
// seq_learn takes the field list of ONE record the order-free parser was sure about; parse_seq is the template walk over that
// list, with the tags at run time instead of at compile time: per step one tag compare, the cheapest exact decode for the
// field's shape (the canonical walk's steps: canon_short / canon_long / canon_addr / canon_skip), the cursor moves only over a
// field validated completely - a record is "sure" iff it is exactly a sub-sequence of the learnt list (zero-valued fields are
// omitted by proto3: their step simply does not move), every field at most once.  Anything else - another field, another
// order, a duplicate, a value outside the step's range - is "not sure" and goes to parse_fast, like from every other tier.
// A step word: tag (1 or 2 bytes, little endian) | kind << 16 | column << 20.
constexpr uint32_t SEQ_MAX = 24;
enum : uint32_t { SQ_SHORT = 0, SQ_LONG = 1, SQ_ADDR = 2, SQ_SKIP = 3 };
enum : uint32_t {
    SQC_NONE = 0, SQC_TIME_RECEIVED, SQC_SAMPLING_RATE, SQC_SEQUENCE_NUM, SQC_SRC_ADDR, SQC_DST_ADDR, SQC_BYTES, SQC_PACKETS,
    SQC_SAMPLER, SQC_SRC_AS, SQC_DST_AS, SQC_PROTO, SQC_SRC_PORT, SQC_DST_PORT, SQC_ETYPE, SQC_TIME_FLOW_START
};
FA_HD uint32_t fa_uniform(uint32_t v) {  // a value every lane of the wave holds: into a scalar register
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
// the step of a tag: the shapes parse_canon uses for the schema's projected fields (pb-ext/flow.proto:16-64), a skip for any other
FA_HD uint32_t seq_step_of(uint32_t tag) {
    switch (tag) {
    case 0x10u: return tag | SQ_LONG << 16 | SQC_TIME_RECEIVED << 20;
    case 0x18u: return tag | SQ_SHORT << 16 | SQC_SAMPLING_RATE << 20;
    case 0x20u: return tag | SQ_LONG << 16 | SQC_SEQUENCE_NUM << 20;
    case 0x32u: return tag | SQ_ADDR << 16 | SQC_SRC_ADDR << 20;
    case 0x3au: return tag | SQ_ADDR << 16 | SQC_DST_ADDR << 20;
    case 0x48u: return tag | SQ_SHORT << 16 | SQC_BYTES << 20;
    case 0x50u: return tag | SQ_SHORT << 16 | SQC_PACKETS << 20;
    case 0x5au: return tag | SQ_ADDR << 16 | SQC_SAMPLER << 20;
    case 0x70u: return tag | SQ_SHORT << 16 | SQC_SRC_AS << 20;
    case 0x78u: return tag | SQ_SHORT << 16 | SQC_DST_AS << 20;
    case 0x01a0u: return tag | SQ_SHORT << 16 | SQC_PROTO << 20;
    case 0x01a8u: return tag | SQ_SHORT << 16 | SQC_SRC_PORT << 20;
    case 0x01b0u: return tag | SQ_SHORT << 16 | SQC_DST_PORT << 20;
    case 0x01f0u: return tag | SQ_SHORT << 16 | SQC_ETYPE << 20;
    case 0x02b0u: return tag | SQ_LONG << 16 | SQC_TIME_FLOW_START << 20;
    default: return tag | SQ_SKIP << 16 | SQC_NONE << 20;
    }
}
// The field list of the record at [pos, end) -> steps[0 .. n); 0: nothing learnt (more than SEQ_MAX fields, a tag twice, a tag
// of 3 bytes, a field that is not a plain varint / fixed / bytes field with a one-byte length, a record that does not end on a
// field boundary).  Every lane of a wave runs this on the SAME record (pos and end are wave-uniform): a scalar loop.
template <class Src>
FA_HD uint32_t seq_learn(const Src& s, uint32_t pos, uint32_t end, uint32_t* steps) {
    uint32_t n = 0;
    while (pos < end) {
        if (n == SEQ_MAX) return 0u;
        const uint64_t w = window64(s, pos);
        const uint32_t w0 = (uint32_t)w;
        const uint32_t two = (w0 >> 7) & 1u;
        if (two && ((w0 >> 15) & 1u)) return 0u;          // a tag of 3+ bytes
        if (two && ((w0 >> 8) & 0x7fu) == 0u) return 0u;  // a non-minimal 2-byte tag
        const uint32_t tag = two ? (w0 & 0xffffu) : (w0 & 0xffu);  // (as the walks compare it: the tag's bytes, little endian)
        const uint32_t wt = tag & 7u;
        if ((tag >> 3) == 0u && !two) return 0u;  // field number 0
        const uint64_t v = w >> (8u << two);
        uint32_t vl;
        uint64_t val;
        const bool var_ok = varint6(v, vl, val);
        uint32_t body;
        if (wt == 0u) {
            if (!var_ok) return 0u;
            body = vl;
        } else if (wt == 1u) {
            body = 8u;
        } else if (wt == 5u) {
            body = 4u;
        } else if (wt == 2u) {
            if (!var_ok || vl != 1u) return 0u;
            body = 1u + (uint32_t)val;
        } else {
            return 0u;
        }
        for (uint32_t k = 0; k < n; k++)
            if ((steps[k] & 0xffffu) == tag) return 0u;
        steps[n++] = seq_step_of(tag);
        pos += 1u + two + body;
    }
    return pos == end ? n : 0u;
}
// runtime-tag forms of the canonical walk's steps (same checks, same value ranges)
template <class Src>
FA_HD bool sq_short(const Src& s, Cursor& c, uint32_t end, uint32_t tag, uint32_t tl, uint32_t& val) {
    const bool m = (c.x & (tl == 1u ? 0xffu : 0xffffu)) == tag;
    const uint32_t v = fa_alignbyte(c.y, c.x, tl);  // value bytes 0..3
    const uint32_t sb = fa_ffbl(~v & 0x80808080u);
    const uint32_t pn = c.pos + (sb >> 3) + (tl + 1u);  // sb = 0xffffffff (no stop) -> far beyond end
    const bool ok = m && pn <= end;
    val = varint28(v, sb);
    c.pos = ok ? pn : c.pos;
    cur_load(s, c);
    return ok;
}
template <class Src>
FA_HD bool sq_long(const Src& s, Cursor& c, uint32_t end, uint32_t tag, uint32_t tl, uint64_t& val) {
    const bool m = (c.x & (tl == 1u ? 0xffu : 0xffffu)) == tag;
    const uint32_t s0 = fa_ffbl(~c.x & (tl == 1u ? 0x80808000u : 0x80800000u));
    const uint32_t s1 = fa_ffbl(~c.y & 0x80808080u) | 32u;  // stays 0xffffffff when there is no stop
    const uint32_t sb = s0 < s1 ? s0 : s1;
    const uint32_t pn = c.pos + (sb >> 3) + 1u;
    const bool ok = m && pn <= end;
    const uint32_t lo = fa_alignbyte(c.y, c.x, tl);
    const uint32_t hi = c.y >> (8u * tl);
    const uint32_t sv = (sb - 8u * tl) & 63u;
    const uint64_t keep = (2ull << sv) - 1ull;
    const uint32_t a = varint28(lo & (uint32_t)keep, 31u);
    const uint32_t b = varint28(hi & (uint32_t)(keep >> 32), 31u);
    val = (uint64_t)a | ((uint64_t)b << 28);
    c.pos = ok ? pn : c.pos;
    cur_load(s, c);
    return ok;
}
template <class Src>
FA_HD bool sq_addr(const Src& s, Cursor& c, uint32_t end, uint32_t tag, uint32_t a16[4]) {
    const bool m = (c.x & 0xffu) == tag;
    const uint32_t len = (c.x >> 8) & 0xffu;
    const uint32_t pn = c.pos + 2u + len;
    const bool ok = m && len <= 16u && pn <= end;
    load_fixed16(s, c.pos + 2u, len > 16u ? 16u : len, a16);
    c.pos = ok ? pn : c.pos;
    cur_load(s, c);
    return ok;
}
// r must be cleared by the caller.  steps / n: what seq_learn left (wave-uniform).
template <uint32_t COLS, class Src>
FA_HD bool parse_seq(const Src& s, uint32_t pos, uint32_t end, Rec& r, const uint32_t* steps, uint32_t n) {
    Cursor c;
    c.pos = pos;
    cur_load(s, c);
    uint32_t next = n ? steps[0] : 0u;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t st = fa_uniform(next);
        next = steps[i + 1 < n ? i + 1 : i];  // (the next step's word is on its way while this one runs)
        const uint32_t tag = st & 0xffffu, kind = (st >> 16) & 7u, col = st >> 20;
        const uint32_t tl = tag > 0xffu ? 2u : 1u;
        if (kind == SQ_SHORT) {
            uint32_t v;
            const bool ok = sq_short(s, c, end, tag, tl, v);
            switch (col) {
            case SQC_SAMPLING_RATE: if (COLS & COL_SAMPLING_RATE) r.sampling_rate = ok ? (uint64_t)v : r.sampling_rate; break;
            case SQC_BYTES: if (COLS & COL_BYTES) r.bytes = ok ? (uint64_t)v : r.bytes; break;
            case SQC_PACKETS: if (COLS & COL_PACKETS) r.packets = ok ? (uint64_t)v : r.packets; break;
            case SQC_SRC_AS: if (COLS & COL_SRC_AS) r.src_as = ok ? v : r.src_as; break;
            case SQC_DST_AS: if (COLS & COL_DST_AS) r.dst_as = ok ? v : r.dst_as; break;
            case SQC_PROTO: if (COLS & COL_PROTO) r.proto = ok ? v : r.proto; break;
            case SQC_SRC_PORT: if (COLS & COL_SRC_PORT) r.src_port = ok ? v : r.src_port; break;
            case SQC_DST_PORT: if (COLS & COL_DST_PORT) r.dst_port = ok ? v : r.dst_port; break;
            case SQC_ETYPE: if (COLS & COL_ETYPE) r.etype = ok ? v : r.etype; break;
            default: break;
            }
        } else if (kind == SQ_LONG) {
            uint64_t v;
            const bool ok = sq_long(s, c, end, tag, tl, v);
            switch (col) {
            case SQC_TIME_RECEIVED: if (COLS & COL_TIME_RECEIVED) r.time_received = ok ? v : r.time_received; break;
            case SQC_SEQUENCE_NUM: if (COLS & COL_SEQUENCE_NUM) r.sequence_num = ok ? (uint32_t)v : r.sequence_num; break;
            case SQC_TIME_FLOW_START: if (COLS & COL_TIME_FLOW_START) r.time_flow_start = ok ? v : r.time_flow_start; break;
            default: break;
            }
        } else if (kind == SQ_ADDR) {
            uint32_t a16[4];
            const bool ok = sq_addr(s, c, end, tag, a16);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((COLS & COL_SRC_ADDR) && col == SQC_SRC_ADDR) r.src[k] = ok ? a16[k] : r.src[k];
                if ((COLS & COL_DST_ADDR) && col == SQC_DST_ADDR) r.dst[k] = ok ? a16[k] : r.dst[k];
                if ((COLS & COL_SAMPLER_ADDRESS) && col == SQC_SAMPLER) r.sampler[k] = ok ? a16[k] : r.sampler[k];
            }
        } else {
            const bool go = (c.x & (tl == 1u ? 0xffu : 0xffffu)) == tag;
            if (tl == 1u) canon_skip<false>(s, c, end, go);
            else canon_skip<true>(s, c, end, go);
        }
    }
    return c.pos == end;
}

// ---- generic parser (complete semantics) -----------------------------------
struct ByteRd {
    const uint8_t* p;
    const uint8_t* end;
};

FA_HD bool g_varint(ByteRd& r, int max_bytes, uint64_t& out) {
    uint64_t v = 0;
    for (int i = 0; i < max_bytes; i++) {
        if (r.p >= r.end) return false;
        uint32_t b = *r.p++;
        if (i < 9)
            v |= (uint64_t)(b & 0x7f) << (7 * i);
        else
            v |= (uint64_t)(b & 1) << 63;
        if (!(b & 0x80)) {
            out = v;
            return true;
        }
    }
    return false;
}

#define FA_MAX_GROUP_DEPTH 100

// Returns true = record OK (r filled), false = malformed.
__host__ __device__ __noinline__ bool parse_generic(const uint8_t* p, const uint8_t* end, Rec& r) {
    uint32_t stack[FA_MAX_GROUP_DEPTH];
    int depth = 0;
    ByteRd rd{p, end};
    rec_clear(r);
    while (rd.p < rd.end) {
        uint64_t t;
        if (!g_varint(rd, 5, t)) return false;
        if (t > 0xFFFFFFFFull) return false;
        uint32_t field = (uint32_t)(t >> 3), wt = (uint32_t)(t & 7);
        if (depth == 0 && field == 0) return false;
        uint64_t v;
        switch (wt) {
        case 0:
            if (!g_varint(rd, 10, v)) return false;
            if (depth == 0) {
                switch (field) {
                case 2: r.time_received = v; break;
                case 3: r.sampling_rate = v; break;
                case 4: r.sequence_num = (uint32_t)v; break;
                case 9: r.bytes = v; break;
                case 10: r.packets = v; break;
                case 14: r.src_as = (uint32_t)v; break;
                case 15: r.dst_as = (uint32_t)v; break;
                case 20: r.proto = (uint32_t)v; break;
                case 21: r.src_port = (uint32_t)v; break;
                case 22: r.dst_port = (uint32_t)v; break;
                case 30: r.etype = (uint32_t)v; break;
                case 38: r.time_flow_start = v; break;
                default: break;
                }
            }
            break;
        case 1:
            if (rd.end - rd.p < 8) return false;
            rd.p += 8;
            break;
        case 5:
            if (rd.end - rd.p < 4) return false;
            rd.p += 4;
            break;
        case 2: {
            if (!g_varint(rd, 5, v)) return false;
            if (v > 0x7FFFFFFFull || v > (uint64_t)(rd.end - rd.p)) return false;
            if (depth == 0 && (field == 6 || field == 7 || field == 11)) {
                if (v > 16) return false;
                uint32_t* dst = field == 6 ? r.src : field == 7 ? r.dst : r.sampler;
                uint8_t tmp[16];
                for (int i = 0; i < 16; i++) tmp[i] = i < (int)v ? rd.p[i] : 0;
                for (int k = 0; k < 4; k++)
                    dst[k] = (uint32_t)tmp[4 * k] | (uint32_t)tmp[4 * k + 1] << 8 |
                             (uint32_t)tmp[4 * k + 2] << 16 | (uint32_t)tmp[4 * k + 3] << 24;
            }
            rd.p += v;
            break;
        }
        case 3:
            if (depth >= FA_MAX_GROUP_DEPTH) return false;
            stack[depth++] = field;
            break;
        case 4:
            if (depth == 0) return false;
            if (stack[--depth] != field) return false;
            break;
        default: return false;
        }
    }
    return depth == 0;
}

// Strip the frame prefix of a framed record: varint(len) || payload with
// len == remaining bytes (mocker.go:98-106: one record per Kafka message).
// Fast form on a window; returns false when not sure.
FA_HD bool frame_fast(uint64_t w, uint32_t rec_len, uint32_t& prefix_len) {
    uint32_t vl;
    uint64_t val;
    if (!varint6(w, vl, val)) return false;
    if (vl > rec_len || val != (uint64_t)(rec_len - vl)) return false;
    prefix_len = vl;
    return true;
}

FA_HD bool frame_generic(const uint8_t*& p, const uint8_t* end) {
    ByteRd rd{p, end};
    uint64_t len;
    if (!g_varint(rd, 10, len)) return false;
    if (len != (uint64_t)(rd.end - rd.p)) return false;
    p = rd.p;
    return true;
}

}  // namespace fa
