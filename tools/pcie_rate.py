#!/usr/bin/env python3
"""PCIe-inclusive ingest rate: host buffers through fa_ingest (copy into pinned staging + H2D + kernels), the
path the Kafka consumer uses.  A side measurement for DESIGN.md - never bench.py's `value` (that one starts with
the inputs resident in HBM)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

fa = _pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
mp = fa.mock_params(mode=fa.MOCK_ASPAIRS, framed=1, seed=2, n_total=n, span_secs=900)
buf, off = fa.mock_generate_host(mp, 0, n)
with fa.FlowAgg(framed=True, max_batch_records=1 << 22) as agg:
    agg.ingest(buf, off)  # warm-up: staging buffers, tables
    agg.sync()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        agg.ingest(buf, off)
    agg.sync()
    dt = (time.perf_counter() - t0) / reps
    rows = agg.read_window()
    assert int(rows["count"].sum()) == n * (reps + 1)
print("fa_ingest (host buffers, %d records, %.1f MB wire): %.3f s per pass = %.2f M records/s = %.2f GB/s wire" % (
    n, buf.nbytes / 1e6, dt, n / dt / 1e6, buf.nbytes / dt / 1e9))
