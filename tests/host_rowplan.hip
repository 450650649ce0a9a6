// host_rowplan.hip - host check of the window close's packed sort keys (flow-pipeline_amd/csrc/rowplan.cuh): for random key
// sets - few or many varying bits, scattered or in runs, one to four key words, word widths below 64 - ordering the rows by
// the packed words must be ordering them by the full key, and equal packed keys must mean equal full keys.
// TEST INFRASTRUCTURE (tests/test_host_parsers.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../flow-pipeline_amd/csrc/rowplan.cuh"

using namespace fa;

static uint64_t st = 0x1234567;
static uint64_t rnd() {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

struct Key {
    unsigned long long w[4];
};

int main() {
    uint64_t fails = 0, cases = 0, max_segs = 0, one_word = 0;
    for (int it = 0; it < 3000; it++) {
        const int nw = 1 + (int)(rnd() % 4);
        int wbits[4];
        unsigned long long vary[4], fixed[4];
        for (int w = 0; w < nw; w++) {
            wbits[w] = (rnd() % 3) ? 64 : 8 + (int)(rnd() % 56);
            const unsigned long long wm = wbits[w] < 64 ? (1ull << wbits[w]) - 1ull : ~0ull;
            switch (rnd() % 5) {
            case 0: vary[w] = 0; break;                                   // a constant word (one timeslot, zero address half)
            case 1: vary[w] = ((1ull << (1 + rnd() % 20)) - 1ull) << (rnd() % 40); break;   // one run
            case 2: vary[w] = rnd() & rnd() & rnd(); break;               // sparse scattered bits (many runs)
            case 3: vary[w] = rnd(); break;                               // half of the bits, scattered: forces run merging
            default: vary[w] = ~0ull; break;
            }
            vary[w] &= wm;
            fixed[w] = rnd() & wm & ~vary[w];
        }
        const int n = 2 + (int)(rnd() % 300);
        std::vector<Key> rows(n);
        unsigned long long orm[4] = {0, 0, 0, 0}, andm[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        for (auto& r : rows)
            for (int w = 0; w < nw; w++) {
                // (a few duplicates of whole keys: every 7th row repeats an earlier one)
                r.w[w] = fixed[w] | (rnd() & vary[w]);
            }
        for (int i = 7; i < n; i += 7) rows[i] = rows[rnd() % i];
        for (const auto& r : rows)
            for (int w = 0; w < nw; w++) {
                orm[w] |= r.w[w];
                andm[w] &= r.w[w];
            }
        RowPlan plan;
        row_plan_build(orm, andm, wbits, nw, plan);
        if (plan.nseg > (uint32_t)RP_MAX_SEGS || plan.nwords > (uint32_t)RP_MAX_WORDS) {
            fails++;
            continue;
        }
        max_segs = std::max<uint64_t>(max_segs, plan.nseg);
        one_word += plan.nwords <= 1;
        std::vector<Key> packed(n);
        for (int i = 0; i < n; i++)
            for (uint32_t p = 0; p < 4; p++) {
                packed[i].w[p] = p < plan.nwords ? rowplan_pack(plan, rows[i].w, nw, p) : 0ull;
                if (p < plan.nwords && plan.bits[p] < 64 && (packed[i].w[p] >> plan.bits[p]) != 0ull) fails++;  // (bits beyond end_bit)
            }
        auto cmp = [&](const Key& a, const Key& b, int words) {  // most significant word last
            for (int w = words - 1; w >= 0; w--)
                if (a.w[w] != b.w[w]) return a.w[w] < b.w[w] ? -1 : 1;
            return 0;
        };
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++) {
                cases++;
                if (cmp(rows[i], rows[j], nw) != cmp(packed[i], packed[j], 4)) fails++;
            }
    }
    printf("cases=%llu fails=%llu max_segs=%llu one_word_plans=%llu\n", (unsigned long long)cases, (unsigned long long)fails, (unsigned long long)max_segs,
           (unsigned long long)one_word);
    puts(fails ? "FAIL" : "OK");
    return fails ? 1 : 0;
}
