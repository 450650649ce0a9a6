// rowplan.cuh - sort keys of the window close reduced to the bits that differ (merge.cuh; host test: tests/host_rowplan.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

constexpr int RP_MAX_WORDS = 4, RP_MAX_SEGS = 24;
struct RowPlan {
    // segment s copies `width` bits of key word `src` (from bit `sshift` up) to packed word `dst` at bit `dshift`
    uint8_t src[RP_MAX_SEGS], sshift[RP_MAX_SEGS], width[RP_MAX_SEGS], dst[RP_MAX_SEGS], dshift[RP_MAX_SEGS];
    uint32_t nseg, nwords;
    uint32_t bits[RP_MAX_WORDS];  // bits in use of packed word p
};
// packed word p of a key whose words are kw[0..nw)
__host__ __device__ __forceinline__ unsigned long long rowplan_pack(const RowPlan& plan, const unsigned long long* kw, int nw, uint32_t p) {
    unsigned long long v = 0;
    for (uint32_t s = 0; s < plan.nseg; s++) {
        if (plan.dst[s] != p) continue;
        unsigned long long word = 0;
        for (int w = 0; w < nw; w++) word = plan.src[s] == w ? kw[w] : word;
        const unsigned long long m = plan.width[s] >= 64 ? ~0ull : ((1ull << plan.width[s]) - 1ull);
        v |= ((word >> plan.sshift[s]) & m) << plan.dshift[s];
    }
    return v;
}

// host: the packing plan for the OR / AND masks of `nw` key words (wbits[w]: bits of word w that belong to the key).
// Runs of varying bits, least significant word first; when there are more runs than the plan has segments the two
// neighbours with the smallest gap are joined (the constant bits in between ride along: more bits, same order).
inline void row_plan_build(const unsigned long long* orm, const unsigned long long* andm, const int* wbits, int nw, RowPlan& plan) {
    struct Run {
        int w, lo, len;
    };
    Run runs[RP_MAX_WORDS * 64];
    int nr = 0;
    for (int w = 0; w < nw; w++) {
        unsigned long long var = orm[w] ^ andm[w];
        if (wbits[w] < 64) var &= (1ull << wbits[w]) - 1ull;
        for (int b = 0; b < 64;) {
            if (!((var >> b) & 1ull)) {
                b++;
                continue;
            }
            int e = b;
            while (e < 64 && ((var >> e) & 1ull)) e++;
            runs[nr++] = Run{w, b, e - b};
            b = e;
        }
    }
    while (nr > RP_MAX_SEGS - RP_MAX_WORDS) {
        int best = -1, gap = 1 << 30;
        for (int i = 0; i + 1 < nr; i++)
            if (runs[i].w == runs[i + 1].w && runs[i + 1].lo - (runs[i].lo + runs[i].len) < gap) {
                gap = runs[i + 1].lo - (runs[i].lo + runs[i].len);
                best = i;
            }
        // (more than 20 runs over at most 4 words: two of them share a word)
        runs[best].len = runs[best + 1].lo + runs[best + 1].len - runs[best].lo;
        for (int i = best + 1; i + 1 < nr; i++) runs[i] = runs[i + 1];
        nr--;
    }
    plan.nseg = 0;
    uint32_t pos = 0;
    for (int i = 0; i < nr; i++) {
        int rest = runs[i].len, src = runs[i].lo;
        while (rest > 0) {
            const uint32_t p = pos >> 6, ds = pos & 63u;
            const int take = rest < (int)(64u - ds) ? rest : (int)(64u - ds);
            const uint32_t s = plan.nseg++;
            plan.src[s] = (uint8_t)runs[i].w;
            plan.sshift[s] = (uint8_t)src;
            plan.width[s] = (uint8_t)take;
            plan.dst[s] = (uint8_t)p;
            plan.dshift[s] = (uint8_t)ds;
            pos += (uint32_t)take;
            src += take;
            rest -= take;
        }
    }
    plan.nwords = (pos + 63u) >> 6;
    for (uint32_t p = 0; p < (uint32_t)RP_MAX_WORDS; p++) plan.bits[p] = p < plan.nwords ? (pos - 64u * p < 64u ? pos - 64u * p : 64u) : 0u;
}
}  // namespace fa
