import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import _pkg, torch, numpy as np
fa = _pkg.load()
n, chunk, span = 100_000_000, 16_666_667, 1800
dev = torch.device("cuda", 0)
mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=5, n_total=n, span_secs=span, zipf_log2_universe=24, zipf_s_x100=80)
with fa.FlowAgg(framed=True, key_sets=9, window_secs=300, subwindow_secs=60, wide_capacity_log2=28, table_capacity_log2=23, max_batch_records=chunk) as agg:
    cap = chunk * 96 + 4096
    d_buf = torch.empty(cap, dtype=torch.uint8, device=dev); d_off = torch.empty(chunk + 1, dtype=torch.int32, device=dev)
    for i0 in range(0, n, chunk):
        m = min(chunk, n - i0)
        w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
        agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
    agg.sync()
    st = agg.stats(); print({k: st[k] for k in st if "wide" in k or "spill" in k or "table" in k})
    for k in range(6):
        t = time.perf_counter(); r = agg.read_window(fa.T0 + 300 * k); print("read", k, len(r), "%.2f ms" % ((time.perf_counter() - t) * 1e3))
    st = agg.stats(); print({k: st[k] for k in st if "wide" in k or "spill" in k or "table" in k})
