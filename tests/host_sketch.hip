// host_sketch.hip - host instantiation of the sketch definition the kernels use (sinks.cuh: cms_hash2, cms_key, cms_column,
// cms_low) against the oracle's restatement (oracle/flow_oracle.c: fo_cms_column), and the invariants the scatter sink
// relies on.  TEST INFRASTRUCTURE (tests/test_host_parsers.py).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../flow-pipeline_amd/csrc/sinks.cuh"
extern "C" {
#include "../oracle/flow_oracle.h"
}

using namespace fa;

static uint64_t st = 0x5EED5EED;
static uint64_t rnd() {
    st += 0x9E3779B97F4A7C15ull;
    return mix64(st);
}

int main() {
    uint64_t fails = 0, checked = 0;
    for (int it = 0; it < 400000; it++) {
        uint8_t key[16];
        uint64_t lo = rnd(), hi = (it % 3) ? rnd() : 0;  // IPv4-shaped keys: bytes 4..15 zero
        if (it % 5 == 0) lo &= 0xffffffffull, hi = 0;
        memcpy(key, &lo, 8);
        memcpy(key + 8, &hi, 8);
        const uint64_t seed = (it % 7) ? rnd() : 0x5EED;
        uint64_t h1, h2;
        cms_hash2(lo, hi, seed, h1, h2);
        for (uint32_t wl2 = 4; wl2 <= 28; wl2 += (it % 2) ? 3 : 4) {
            const CmsKey k = cms_key(h1, h2, wl2);
            const uint32_t pbits = cms_pbits(wl2), sub = wl2 - pbits;
            if (pbits > 8 || sub < 4 || k.prefix >> pbits) fails++;
            for (uint32_t r = 0; r < 8; r++) {
                const uint32_t col = cms_column(k, r, wl2), want = fo_cms_column(key, seed, wl2, r);
                checked++;
                if (col != want || col >> wl2) {
                    if (fails++ < 5) printf("column: wl2 %u row %u: kernel %u oracle %u\n", wl2, r, col, want);
                }
                // what the scatter sink relies on: every row's column lies in the key's partition, and the tuple
                // {l1, l2} alone gives the column inside the partition
                if ((col >> sub) != k.prefix || (col & ((1u << sub) - 1u)) != cms_low(k.l1, k.l2, r, sub)) fails++;
            }
        }
    }
    // the default geometry goes through the sink, 5 x 2^20 and narrow sketches do not
    if (!cms_scatterable(4, 20) || !cms_scatterable(4, 14) || !cms_scatterable(3, 12) || cms_scatterable(5, 20) || cms_scatterable(4, 11) ||
        !cms_scatterable(16, 18) || cms_scatterable(16, 19))
        fails++;
    // fa_topk's pre-selection bins: monotone over the whole u64 range, below TK_BINS, exact below 64
    {
        uint32_t prev = 0;
        for (unsigned long long v = 0; v < 70000; v++) {
            const uint32_t b = topk_bin(v);
            if (b < prev || b >= TK_BINS || (v < 64 && b != v)) fails++;
            prev = b;
        }
        for (int e = 6; e < 64; e++)
            for (int d = -3; d <= 3; d++)
                for (int m = 0; m < 32; m++) {
                    const unsigned long long v = (1ull << e) + ((unsigned long long)m << (e - 5)) + (unsigned long long)(long long)d;
                    if (v < (1ull << e) && e == 63) continue;
                    if (topk_bin(v - 1) > topk_bin(v) || topk_bin(v) >= TK_BINS) fails++;
                    checked++;
                }
        if (topk_bin(~0ull) != 1919u || topk_bin(64) != 64u || topk_bin(63) != 63u) fails++;
        for (uint32_t b = 0; b < 1920u; b++) {  // the lower edge of every bin (the candidates mode's threshold)
            const unsigned long long f = topk_bin_floor(b);
            if (topk_bin(f) != b || (f && topk_bin(f - 1) >= b)) fails++;
        }
        for (int it = 0; it < 2000000; it++) {
            unsigned long long a = rnd() >> (rnd() & 63), b = rnd() >> (rnd() & 63);
            if (a > b) { const unsigned long long t = a; a = b; b = t; }
            if (topk_bin(a) > topk_bin(b)) fails++;
        }
    }
    printf("checked=%llu\n", (unsigned long long)checked);
    printf(fails ? "FAILED (%llu)\n" : "OK\n", (unsigned long long)fails);
    return fails ? 1 : 0;
}
