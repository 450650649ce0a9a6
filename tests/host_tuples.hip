// host_tuples.hip - host instantiation of the scatter-sink tuple formats (table.cuh): round trips, the partition
// bijection of the compact format and its balance.  TEST INFRASTRUCTURE (tests/test_host_parsers.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../flow-pipeline_amd/csrc/table.cuh"

using namespace fa;

static uint64_t st = 0x9876543;
static uint64_t rnd() {
    st += 0x9E3779B97F4A7C15ull;
    return mix64(st);
}

int main() {
    uint64_t fails = 0;
    const uint32_t etypes[] = {0, 0x0800, 0x86dd, 0x0806, 0x8847, 1, 0xffff, 0x10000};
    // 1. random values around the format limits: fits() <=> lossless round trip
    for (int it = 0; it < 4000000; it++) {
        const uint32_t asbits = (uint32_t)(rnd() % 33), asbits2 = (uint32_t)(rnd() % 33);
        const uint32_t src = asbits ? (uint32_t)(rnd() & ((1ull << asbits) - 1)) : 0, dst = asbits2 ? (uint32_t)(rnd() & ((1ull << asbits2) - 1)) : 0;
        const uint64_t b = rnd() & ((1ull << (rnd() % 34)) - 1), p = rnd() & ((1ull << (rnd() % 20)) - 1);
        const uint32_t tbr = (uint32_t)(rnd() % 20), et = etypes[rnd() % 8];
        TupleVals v;
        if (tup16_fits(tbr, b, p, et)) {
            tup16_unpack(tup16_pack(src, dst, (uint32_t)b, (uint32_t)p, tbr, et), v);
            if (v.src_as != src || v.dst_as != dst || v.bytes != b || v.packets != p || v.tbr != tbr || v.etype != et) fails++;
        }
        const bool f8 = t8_fits(src, dst, tbr, b, p, et);
        const bool want8 = tbr < 16 && src < (1u << 20) && dst < (1u << 20) && b < (1u << 17) && p < (1u << 9) &&
                           (et == 0 || et == 0x800 || et == 0x86dd || et == 0x806);
        if (f8 != want8) fails++;
        if (f8) {
            uint32_t part = 0;
            const uint32_t tb_base = (uint32_t)(rnd() & 0x3ffffff);  // any batch position: the tuple stores (tb_base + tbr) & 15
            const uint2 t = t8_pack(src, dst, (uint32_t)b, (uint32_t)p, tbr, et, tb_base, part);
            if (part > 255) fails++;
            // the partition is a function of the KEY alone (absolute bucket): the same group from a launch with another
            // tb_base lands in the same partition, and that partition is the key's region of the device table
            if (part != t8_part(src, dst, tb_base + tbr, et)) fails++;
            uint64_t k0, k1;
            pack_key(tb_base + tbr, src, dst, et, k0, k1);
            if (as_region8(k0, k1, key_hash(k0, k1)) != part) fails++;
            const uint32_t home = as_home(k0, k1, key_hash(k0, k1), (1u << 20) - 1, as_rlog2(20));
            if ((home >> 12) != part || as_next(home | 0xfff, (1u << 20) - 1, 8) != (home & ~0xfffu)) fails++;
            t8_unpack(t, part, tb_base, v);
            if (v.src_as != src || v.dst_as != dst || v.bytes != b || v.packets != p || v.tbr != tbr || v.etype != et) {
                if (fails++ < 5) printf("t8 round trip: src %u dst %u b %llu p %llu tbr %u et %x -> %u %u %u %u %u %x (part %u)\n", src, dst,
                                        (unsigned long long)b, (unsigned long long)p, tbr, et, v.src_as, v.dst_as, v.bytes, v.packets, v.tbr, v.etype, part);
            }
        }
    }
    // 1b. the branch-free EType dictionary, every value of a 17-bit range (and some beyond): code and its inverse
    for (uint32_t et = 0; et < (1u << 17) + 64; et++) {
        const uint32_t e = et < (1u << 17) ? et : (uint32_t)rnd();
        const uint32_t want = e == 0x0800u ? 1u : e == 0x86ddu ? 2u : e == 0x0806u ? 3u : 0u;
        if (t8_etcode(e) != want) fails++;
    }
    if (t8_etype(0) != 0 || t8_etype(1) != 0x0800u || t8_etype(2) != 0x86ddu || t8_etype(3) != 0x0806u) fails++;
    // 2. balance of the partition over key populations: only SrcAS[7:0] varies / only the rest varies / config 2 / mocker
    auto balance = [&](const char* name, auto gen, uint32_t nkeys) {
        std::vector<uint32_t> cnt(256, 0);
        for (uint32_t k = 0; k < nkeys; k++) {
            uint32_t src, dst, tbr, et;
            gen(k, src, dst, tbr, et);
            uint32_t part;
            (void)t8_pack(src, dst, 1, 1, tbr, et, 5666666u, part);
            cnt[part]++;
        }
        uint32_t mx = 0;
        for (uint32_t c : cnt) mx = c > mx ? c : mx;
        const double mean = nkeys / 256.0;
        printf("%-40s keys=%u  max/mean=%.3f\n", name, nkeys, mx / mean);
        if (nkeys >= 65536 && mx > 1.35 * mean) fails++;
    };
    balance("config 2 (256 x 256 AS pairs x 2 x 3)", [](uint32_t k, uint32_t& s, uint32_t& d, uint32_t& t, uint32_t& e) {
        s = 64512 + (k & 255); d = 64512 + ((k >> 8) & 255); e = (k >> 16) & 1 ? 0x86dd : 0x800; t = 2 + ((k >> 17) % 3); }, 65536 * 6);
    balance("only SrcAS[7:0] varies", [](uint32_t k, uint32_t& s, uint32_t& d, uint32_t& t, uint32_t& e) {
        s = 65000 + (k & 255) - (65000 & 255); d = 65000; e = 0x86dd; t = 2; (void)k; }, 65536);
    balance("only DstAS varies", [](uint32_t k, uint32_t& s, uint32_t& d, uint32_t& t, uint32_t& e) {
        s = 13335; d = k; e = 0x86dd; t = 2; }, 1 << 20);
    balance("random public ASNs", [&](uint32_t k, uint32_t& s, uint32_t& d, uint32_t& t, uint32_t& e) {
        s = (uint32_t)(rnd() % 400000); d = (uint32_t)(rnd() % 400000); e = 0x800; t = 2; (void)k; }, 1 << 20);
    // 3. keys outside the compact format: region = top bits of the hash; tiny tables have fewer regions of >= 64 slots
    for (int it = 0; it < 200000; it++) {
        const uint32_t src = (uint32_t)rnd(), dst = (uint32_t)rnd() | (1u << 20), tb = (uint32_t)(rnd() & 0x3ffffff), et = etypes[rnd() % 8];
        uint64_t k0, k1;
        pack_key(tb, src, dst, et, k0, k1);
        const uint32_t h = key_hash(k0, k1);
        if (as_region8(k0, k1, h) != (h >> 24)) fails++;
        for (uint32_t lg = 6; lg <= 22; lg += 4) {
            const uint32_t mask = (1u << lg) - 1, rl = as_rlog2(lg), home = as_home(k0, k1, h, mask, rl);
            if (home > mask || (rl && (home >> (lg - rl)) != ((h >> 24) >> (8 - rl))) || ((mask >> rl) + 1) < 64) fails++;
            uint32_t i = home;
            for (int k = 0; k < 70; k++) i = as_next(i, mask, rl);
            if ((i >> (lg - rl)) != (home >> (lg - rl))) fails++;  // probing never leaves the region
        }
    }
    printf(fails ? "FAILED (%llu)\n" : "OK\n", (unsigned long long)fails);
    return fails ? 1 : 0;
}
