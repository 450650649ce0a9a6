#!/usr/bin/env python3
"""Stress: the (SrcAddr,DstPort,Proto) rows of one ctx against the numpy restatement, many times (a flaky window-close test:
which side is wrong, and on which wide path).  python tools/stress_app.py [iterations]   (FA_WIDE=scatter|atomic pins the path)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import _pkg
fa = _pkg.load(); po = _pkg.load_oracle()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ks = int(os.environ.get("STRESS_KS", "63"))
n = 300_000
gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=91, n_total=n, zipf_log2_universe=14, span_secs=900)
buf, off = po.gen_records(gp, 0, n)
rows, status = po.decode_batch(buf, off, 1)
bad = 0
for it in range(iters):
    kw = dict(framed=True, key_sets=ks, cms_width_log2=14, topk_capacity_log2=16, subwindow_secs=60)
    with fa.FlowAgg(**kw) as agg:
        half = n // 2
        if it % 2:
            agg.ingest(buf, off)
        else:  # two launches
            agg.ingest(buf[: int(off[half])], off[: half + 1])
            agg.ingest(buf[int(off[half]):], off[half:] - off[half])
        slots = agg.open_timeslots()
        for ts in (fa.ALL_TIMESLOTS, int(slots[0]), int(slots[0]) + 60, int(slots[0]) + 240):
            got = agg.read_window_app(ts)
            if ts == fa.ALL_TIMESLOTS:
                want = po.rollup_app(rows, status, 60).astype(fa.ROW_APP_DTYPE)
            else:
                want = po.rollup_app(rows, status, 60, window=300, timeslot=ts).astype(fa.ROW_APP_DTYPE)
            if got.tobytes() != want.tobytes():
                bad += 1
                gk = {bytes(r.tobytes()[:32]): r for r in got}
                wk = {bytes(r.tobytes()[:32]): r for r in want}
                extra = [k for k in gk if k not in wk]; missing = [k for k in wk if k not in gk]
                print("MISMATCH it", it, "ts", ts, "rows", len(got), len(want), "extra", len(extra), "missing", len(missing),
                      "count sums", int(got["count"].sum()), int(want["count"].sum()), "dup keys in got", len(got) - len(gk), flush=True)
                for k in extra[:3]:
                    print("  extra", gk[k])
                for k in missing[:3]:
                    print("  missing", wk[k])
        st = agg.stats()
print("done", iters, "iterations, mismatches", bad, {k: st[k] for k in ("wave_tile_launches", "records_ok")})
