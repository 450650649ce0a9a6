#!/usr/bin/env python3
"""Candidates mode, batch by batch against its restatement: thresholds (fa_stats.topk_theta_*), candidates held, and the first
batch at which the library's set and the oracle's differ (debugging aid; GPU)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
fa, po = _pkg.load(), _pkg.load_oracle()
n, nb, zl, wl2, track, cap = [int(x) for x in (sys.argv[1:7] if len(sys.argv) > 6 else (60000, 12, 12, 12, 32, 10))]
depth, seed = 4, 0x5EED
gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=1301 + nb, n_total=n, zipf_log2_universe=zl)
step = n // nb
batches, raw = [], []
for b in range(nb):
    buf, off = po.gen_records(gp, b * step, step)
    rows = po.gen_rows(gp, b * step, step)
    raw.append((buf, off))
    with np.errstate(over="ignore"):
        batches.append((rows["src_addr"], rows["bytes"] * rows["sampling_rate"]))
with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=cap, topk_mode=fa.TOPK_CANDIDATES, topk_track=track) as agg:
    for b in range(nb):
        agg.ingest(*raw[b])
        st = agg.stats()
        sk, cand, est, thetas = po.topk_candidates(batches[:b + 1], depth, wl2, seed, track=track, capacity_log2=cap)
        got = agg.topk(fa.FA_KEYS_SRCADDR_CMS, 1 << 20)
        gk = {bytes(r["key"]) for r in got}
        ok = {bytes(k) for k in cand}
        print("batch %2d: theta gpu %d oracle %d | held gpu %d (stats %d) oracle %d | missing on gpu %d, extra on gpu %d | sketch equal %s" % (
            b, st["topk_theta_src"], thetas[-1], len(gk), st["topk_candidates_src"], len(ok), len(ok - gk), len(gk - ok),
            np.array_equal(agg.cms_read(fa.FA_KEYS_SRCADDR_CMS).reshape(-1), sk)))
        if ok != gk and b > 0:
            # which of the oracle's admissions of THIS batch are missing, and how often did they occur in it
            keys_b = [bytes(k) for k in np.ascontiguousarray(batches[b][0])]
            for k in sorted(ok - gk)[:6]:
                print("   missing %s: occurrences in this batch %d, in the batch before %d" % (k.hex(), keys_b.count(k), [bytes(x) for x in np.ascontiguousarray(batches[b - 1][0])].count(k)))
