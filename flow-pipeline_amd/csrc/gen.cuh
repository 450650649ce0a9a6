// gen.cuh - synthetic FlowMessage producer (host + device), the HBM-resident
// stand-in for mocker/mocker.go:53-106.
//
// Same value distribution as the mocker (mocker.go:57-91): Bytes ~ U[0,1500),
// Packets ~ U[0,100), SrcAS/DstAS = 65000 + U{0,1,2}, 2001:db8:0:1::XX
// addresses, 16 random port bits, Etype 0x86dd, SamplingRate 1,
// SequenceNum = i, TimeReceived = TimeFlowStart; fields marshalled in
// field-number order with proto3 zero omission (proto.Marshal, mocker.go:97) and
// the optional varint length prefix (proto.Buffer.EncodeMessage, mocker.go:98-101).
// math/rand is replaced by a counter-based generator so that record i can be
// produced independently by any lane (spec in DESIGN.md "Synthetic generator"):
//   base(i)  = mix64(seed * 0x9E3779B97F4A7C15 + i + 1)
//   rnd(i,j) = mix64(base(i) ^ ((j + 1) * 0xD1B54A32D192ED03))
// Modes ASPAIRS / ZIPF widen the key space for BASELINE.json configs 2-5.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/flowagg.h"
#include "table.cuh"  // mix64

namespace fa {

struct GenRow {
    uint64_t time_received, sampling_rate, bytes, packets;
    uint32_t sequence_num, src_as, dst_as, etype, proto, src_port, dst_port, addr_len;
    uint8_t src[16], dst[16];
    // FA_MOCK_GOFLOW only: the fields outside the projection (SamplerAddress is projected: create.sh:11)
    uint8_t sampler[4], next_hop[16];
    uint64_t src_mac, dst_mac;
    uint32_t next_hop_as, src_net, dst_net, in_if, out_if, ip_tos, ip_ttl, tcp_flags, vlan_id, dst_vlan, fragment_id, flow_label;
};

__host__ __device__ inline uint64_t gen_rnd(const fa_mock_params& g, uint64_t i, uint32_t j) {
    uint64_t base = mix64(g.seed * 0x9E3779B97F4A7C15ull + i + 1);
    return mix64(base ^ ((uint64_t)(j + 1) * 0xD1B54A32D192ED03ull));
}

// Integer-only Zipf-like rank over [0, 2^L): octave k = [2^k-1, 2^(k+1)-1) is
// drawn with weight ~ 2^(-k(s-1)) (fixed point), uniform inside the octave.
__host__ __device__ inline uint64_t gen_zipf_rank(const fa_mock_params& g, uint64_t r) {
    uint32_t L = g.zipf_log2_universe ? g.zipf_log2_universe : 24;
    uint32_t s = g.zipf_s_x100 ? g.zipf_s_x100 : 110;
    uint64_t mul;
    uint32_t shift;
    if (s >= 100) {
        shift = 32;
        mul = s == 100 ? 0xFFFFFFFFull : s == 110 ? 4007346185ull : s == 120 ? 3738986199ull
              : s == 150 ? 3037000500ull : 4007346185ull;
    } else {
        shift = 30;
        mul = s == 80 ? 1233405467ull : (1ull << 30);
    }
    if (L > 40) L = 40;
    uint64_t w[40], tot = 0, cur = 1ull << 30;
    for (uint32_t k = 0; k < L; k++) {
        w[k] = cur ? cur : 1;
        tot += w[k];
        cur = (cur * mul) >> shift;
    }
    uint64_t u = (r >> 11) % tot;
    uint32_t k = 0;
    while (u >= w[k]) {
        u -= w[k];
        k++;
    }
    uint64_t lo = (1ull << k) - 1, span = 1ull << k;
    uint64_t r2 = mix64(r ^ 0xA5A5A5A5A5A5A5A5ull);
    return lo + (r2 & (span - 1));
}

__host__ __device__ inline void put_le(uint8_t* p, uint64_t v, int n) {
    for (int i = 0; i < n; i++) p[i] = (uint8_t)(v >> (8 * i));
}

__host__ __device__ inline void gen_zipf_key(uint64_t rank, uint64_t salt, uint8_t out[16], bool v4) {
    uint64_t a = mix64(rank * 0x9E3779B97F4A7C15ull + salt);
    uint64_t b = mix64(a ^ 0xD1B54A32D192ED03ull);
    for (int i = 0; i < 16; i++) out[i] = 0;
    if (v4) {
        put_le(out, a, 4);
    } else {
        put_le(out, a, 8);
        put_le(out + 8, b, 8);
    }
}

__host__ __device__ inline void gen_row(const fa_mock_params& g, uint64_t i, GenRow& o) {
    const uint8_t pfx[15] = {0x20, 0x01, 0x0d, 0xb8, 0, 0, 0, 0x01, 0, 0, 0, 0, 0, 0, 0};
    uint64_t r0 = gen_rnd(g, i, 0), r1 = gen_rnd(g, i, 1), r2 = gen_rnd(g, i, 2),
             r3 = gen_rnd(g, i, 3), r4 = gen_rnd(g, i, 4), r5 = gen_rnd(g, i, 5);
    o.sampling_rate = 1;
    o.bytes = r0 % 1500;
    o.packets = r1 % 100;
    o.src_port = (uint32_t)(r5 & 0xFFFF);
    o.dst_port = (uint32_t)((r5 >> 16) & 0xFFFF);
    o.sequence_num = (uint32_t)i;
    o.proto = 0;
    for (int k = 0; k < 16; k++) o.src[k] = o.dst[k] = 0;
    if (g.mode == FA_MOCK_MOCKER) {
        uint32_t ps = g.per_sec ? g.per_sec : 4;
        o.time_received = g.t0 + i / ps;
        o.src_as = 65000 + (uint32_t)(r2 % 3);
        o.dst_as = 65000 + (uint32_t)(r3 % 3);
        o.etype = 0x86dd;
        o.addr_len = 16;
        for (int k = 0; k < 15; k++) o.src[k] = o.dst[k] = pfx[k];
        o.src[15] = (uint8_t)(r4 & 0xff);
        o.dst[15] = (uint8_t)((r4 >> 8) & 0xff);
    } else {
        uint64_t nt = g.n_total ? g.n_total : 1;
        o.time_received = g.t0 + (uint64_t)g.span_secs * i / nt;
        bool v6 = (r2 >> 16) & 1;
        o.etype = v6 ? 0x86dd : 0x0800;
        o.addr_len = v6 ? 16 : 4;
        if (g.mode == FA_MOCK_ASPAIRS || g.mode == FA_MOCK_GOFLOW || g.mode == FA_MOCK_DISTINCT || g.mode == FA_MOCK_REVERSED) {
            o.src_as = 64512 + (uint32_t)(r2 & 255);
            o.dst_as = 64512 + (uint32_t)((r2 >> 8) & 255);
            if (g.mode == FA_MOCK_DISTINCT) {  // record i is the only member of its group (i < 2^40)
                o.src_as = 1 + (uint32_t)(i & 0xfffff);
                o.dst_as = 1 + (uint32_t)((i >> 20) & 0xfffff);
            }
            if (v6) {
                for (int k = 0; k < 15; k++) o.src[k] = o.dst[k] = pfx[k];
                o.src[15] = (uint8_t)(r4 & 0xff);
                o.dst[15] = (uint8_t)((r4 >> 8) & 0xff);
            } else {
                o.src[0] = 10; o.src[1] = (uint8_t)(r4 >> 16); o.src[2] = (uint8_t)(r4 >> 24);
                o.src[3] = (uint8_t)(r4 & 0xff);
                o.dst[0] = 10; o.dst[1] = (uint8_t)(r4 >> 32); o.dst[2] = (uint8_t)(r4 >> 40);
                o.dst[3] = (uint8_t)((r4 >> 8) & 0xff);
            }
            if (g.mode == FA_MOCK_GOFLOW) {
                // what GoFlow fills in for an sFlow sample besides the mocker's fields (pb-ext/flow.pb.go:57-147)
                const uint64_t r6 = gen_rnd(g, i, 6), r7 = gen_rnd(g, i, 7), r8 = gen_rnd(g, i, 8);
                o.sampling_rate = ((r2 >> 17) & 1) ? 2048 : 1024;
                o.proto = ((r2 >> 18) & 1) ? 6 : 17;
                o.sampler[0] = 10; o.sampler[1] = 255; o.sampler[2] = 0; o.sampler[3] = (uint8_t)(r6 & 7);
                for (int k = 0; k < 16; k++) o.next_hop[k] = 0;
                if (v6) {
                    for (int k = 0; k < 15; k++) o.next_hop[k] = pfx[k];
                    o.next_hop[15] = (uint8_t)(r6 >> 8);
                } else {
                    o.next_hop[0] = 10; o.next_hop[1] = (uint8_t)(r6 >> 8); o.next_hop[2] = (uint8_t)(r6 >> 16); o.next_hop[3] = 1;
                }
                o.next_hop_as = 64512 + (uint32_t)((r6 >> 24) & 255);
                o.src_net = v6 ? 48 : 24;
                o.dst_net = v6 ? 32 + (uint32_t)((r6 >> 32) & 31) : 8 + (uint32_t)((r6 >> 32) & 15);
                o.in_if = 1 + (uint32_t)((r6 >> 40) & 63);
                o.out_if = 1 + (uint32_t)((r6 >> 46) & 63);
                o.ip_tos = (r7 & 3) ? 0 : 0xb8;
                o.ip_ttl = 32 + (uint32_t)((r7 >> 2) & 127);
                o.tcp_flags = o.proto == 6 ? (uint32_t)((r7 >> 9) & 0x3f) : 0;
                o.src_mac = 0x3cfdfe000000ull | ((r7 >> 16) & 0xffffff);  // 48 bits: 7-byte varints
                o.dst_mac = 0xa0369f000000ull | ((r7 >> 40) & 0xffffff);
                o.vlan_id = 100 + (uint32_t)(r8 & 15);
                o.dst_vlan = 200 + (uint32_t)((r8 >> 4) & 15);
                o.fragment_id = v6 ? 0 : (uint32_t)((r8 >> 8) & 0xffff);
                o.flow_label = v6 ? (uint32_t)((r8 >> 24) & 0xfffff) : 0;
            }
        } else {
            uint64_t rs = gen_zipf_rank(g, r3), rdst = gen_zipf_rank(g, r4);
            o.src_as = 64512 + (uint32_t)(rs & 255);
            o.dst_as = 64512 + (uint32_t)(rdst & 255);
            gen_zipf_key(rs, 0x1111, o.src, !v6);
            gen_zipf_key(rdst, 0x2222, o.dst, !v6);
            o.sampling_rate = ((r2 >> 17) & 1) ? 1000 : 1;
            o.proto = ((r2 >> 18) & 1) ? 6 : 17;
        }
    }
}

__host__ __device__ inline uint32_t enc_varint(uint8_t* p, uint64_t v) {
    uint32_t n = 0;
    while (v >= 0x80) {
        p[n++] = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    p[n++] = (uint8_t)v;
    return n;
}
__host__ __device__ inline uint32_t enc_vfield(uint8_t* p, uint32_t field, uint64_t v) {
    if (!v) return 0;
    uint32_t n = enc_varint(p, (uint64_t)field << 3);
    return n + enc_varint(p + n, v);
}
__host__ __device__ inline uint32_t enc_bfield(uint8_t* p, uint32_t field, const uint8_t* d, uint32_t len) {
    if (!len) return 0;
    uint32_t n = enc_varint(p, ((uint64_t)field << 3) | 2);
    n += enc_varint(p + n, len);
    for (uint32_t i = 0; i < len; i++) p[n + i] = d[i];
    return n + len;
}

// Encodes record i into out (>= FA_MOCK_MAX_RECORD bytes); returns the record's length.
__host__ __device__ inline uint32_t gen_encode(const fa_mock_params& g, uint64_t i, uint8_t* out) {
    GenRow r;
    gen_row(g, i, r);
    uint8_t tmp[FA_MOCK_MAX_RECORD];
    uint32_t n = 0;
    if (g.mode == FA_MOCK_GOFLOW) {  // field-number order, proto3 zero omission (golang/protobuf, like mocker.go:97)
        n += enc_vfield(tmp + n, 1, 1);  // Type = SFLOW_5
        n += enc_vfield(tmp + n, 2, r.time_received);
        n += enc_vfield(tmp + n, 3, r.sampling_rate);
        n += enc_vfield(tmp + n, 4, r.sequence_num);
        n += enc_vfield(tmp + n, 5, r.time_received);  // TimeFlowEnd
        n += enc_bfield(tmp + n, 6, r.src, r.addr_len);
        n += enc_bfield(tmp + n, 7, r.dst, r.addr_len);
        n += enc_vfield(tmp + n, 9, r.bytes);
        n += enc_vfield(tmp + n, 10, r.packets);
        n += enc_bfield(tmp + n, 11, r.sampler, 4);
        n += enc_bfield(tmp + n, 12, r.next_hop, r.addr_len);
        n += enc_vfield(tmp + n, 13, r.next_hop_as);
        n += enc_vfield(tmp + n, 14, r.src_as);
        n += enc_vfield(tmp + n, 15, r.dst_as);
        n += enc_vfield(tmp + n, 16, r.src_net);
        n += enc_vfield(tmp + n, 17, r.dst_net);
        n += enc_vfield(tmp + n, 18, r.in_if);
        n += enc_vfield(tmp + n, 19, r.out_if);
        n += enc_vfield(tmp + n, 20, r.proto);
        n += enc_vfield(tmp + n, 21, r.src_port);
        n += enc_vfield(tmp + n, 22, r.dst_port);
        n += enc_vfield(tmp + n, 23, r.ip_tos);
        n += enc_vfield(tmp + n, 25, r.ip_ttl);
        n += enc_vfield(tmp + n, 26, r.tcp_flags);
        n += enc_vfield(tmp + n, 27, r.src_mac);
        n += enc_vfield(tmp + n, 28, r.dst_mac);
        n += enc_vfield(tmp + n, 29, r.vlan_id);
        n += enc_vfield(tmp + n, 30, r.etype);
        n += enc_vfield(tmp + n, 33, r.vlan_id);  // SrcVlan
        n += enc_vfield(tmp + n, 34, r.dst_vlan);
        n += enc_vfield(tmp + n, 35, r.fragment_id);
        n += enc_vfield(tmp + n, 37, r.flow_label);
        n += enc_vfield(tmp + n, 38, r.time_received);  // TimeFlowStart
        uint32_t k = 0;
        if (g.framed) k = enc_varint(out, n);
        for (uint32_t j = 0; j < n; j++) out[k + j] = tmp[j];
        return k + n;
    }
    if (g.mode == FA_MOCK_REVERSED) {
        n += enc_vfield(tmp + n, 38, r.time_received);
        n += enc_vfield(tmp + n, 30, r.etype);
        n += enc_vfield(tmp + n, 22, r.dst_port);
        n += enc_vfield(tmp + n, 21, r.src_port);
        n += enc_vfield(tmp + n, 20, r.proto);
        n += enc_vfield(tmp + n, 15, r.dst_as);
        n += enc_vfield(tmp + n, 14, r.src_as);
        n += enc_vfield(tmp + n, 10, r.packets);
        n += enc_vfield(tmp + n, 9, r.bytes);
        n += enc_bfield(tmp + n, 7, r.dst, r.addr_len);
        n += enc_bfield(tmp + n, 6, r.src, r.addr_len);
        n += enc_vfield(tmp + n, 4, r.sequence_num);
        n += enc_vfield(tmp + n, 3, r.sampling_rate);
        n += enc_vfield(tmp + n, 2, r.time_received);
        uint32_t k = 0;
        if (g.framed) k = enc_varint(out, n);
        for (uint32_t j = 0; j < n; j++) out[k + j] = tmp[j];
        return k + n;
    }
    n += enc_vfield(tmp + n, 2, r.time_received);
    n += enc_vfield(tmp + n, 3, r.sampling_rate);
    n += enc_vfield(tmp + n, 4, r.sequence_num);
    n += enc_bfield(tmp + n, 6, r.src, r.addr_len);
    n += enc_bfield(tmp + n, 7, r.dst, r.addr_len);
    n += enc_vfield(tmp + n, 9, r.bytes);
    n += enc_vfield(tmp + n, 10, r.packets);
    n += enc_vfield(tmp + n, 14, r.src_as);
    n += enc_vfield(tmp + n, 15, r.dst_as);
    n += enc_vfield(tmp + n, 20, r.proto);
    n += enc_vfield(tmp + n, 21, r.src_port);
    n += enc_vfield(tmp + n, 22, r.dst_port);
    n += enc_vfield(tmp + n, 30, r.etype);
    n += enc_vfield(tmp + n, 38, r.time_received);  // TimeFlowStart = TimeReceived (mocker.go:85-86)
    uint32_t k = 0;
    if (g.framed) k = enc_varint(out, n);
    for (uint32_t j = 0; j < n; j++) out[k + j] = tmp[j];
    return k + n;
}

}  // namespace fa
