#!/usr/bin/env python3
"""Diagnostic (GPU box): DISTINCT-mode batch through several library configurations, rows diffed against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg
import torch
fa = _pkg.load(); po = _pkg.load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
gp = po.gen_params(mode=4, framed=1, seed=3, n_total=n, span_secs=600)
buf, off = po.gen_records(gp, 0, n)
ref = po.Rollup(300); ref.ingest(buf, off, 1)
want = ref.rows()
dev = torch.device("cuda", 0)
for name, env, tab in (("t8+agg8 tab16", {}, 16), ("t8+generic tab16", {"FA_AGG": "generic"}, 16), ("t16 tab16", {"FA_TUPLE": "16"}, 16),
                       ("t8+agg8 tab23", {}, 23), ("t16 tab23", {"FA_TUPLE": "16"}, 23), ("direct tab16", {"FA_SINK": "direct"}, 16)):
    for k in ("FA_AGG", "FA_TUPLE", "FA_SINK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with fa.FlowAgg(framed=True, table_capacity_log2=tab, max_batch_records=n) as agg:
        mp = fa.mock_params(mode=4, framed=1, seed=3, n_total=n, span_secs=600)
        cap = n * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev); d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        w = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), cap, d_off.data_ptr())
        agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), n)
        st = agg.stats()
        got = agg.read_window()
    same = got.tobytes() == want.tobytes()
    print("%-18s rows %d/%d same=%s direct=%d retried=%d table=%d/%d" % (name, len(got), len(want), same, st["records_direct"], st["records_retried"], st["table_used"], st["table_capacity"]))
    if not same and len(got) == len(want):
        keys = ("date", "timeslot", "src_as", "dst_as", "etype")
        keq = np.ones(len(got), dtype=bool)
        for k in keys:
            keq &= got[k] == want[k]
        veq = (got["bytes"] == want["bytes"]) & (got["packets"] == want["packets"]) & (got["count"] == want["count"])
        print("   key-equal rows %d, value-equal rows %d, both %d" % (keq.sum(), veq.sum(), (keq & veq).sum()))
        bad = np.nonzero(~(keq & veq))[0][:6]
        for i in bad:
            print("   got ", got[i], "\n   want", want[i])
        print("   sums: bytes %d vs %d  packets %d vs %d" % (got["bytes"].sum(), want["bytes"].sum(), got["packets"].sum(), want["packets"].sum()))
