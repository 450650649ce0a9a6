#!/usr/bin/env python3
"""PCIe-inclusive ingest rate: host buffers through fa_ingest (copy into pinned staging + H2D + kernels), the
path the Kafka consumer uses - and what bounds it: the same bytes through a bare pinned H2D copy (the link), a
single-thread host copy (what one staging thread moves), and fa_ingest at several staging-thread counts.
A side measurement for DESIGN.md - never bench.py's `value` (that one starts with the inputs resident in HBM)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

import torch  # noqa: E402

fa = _pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
mp = fa.mock_params(mode=fa.MOCK_ASPAIRS, framed=1, seed=2, n_total=n, span_secs=900)
buf, off = fa.mock_generate_host(mp, 0, n)
out = {"records": n, "wire_MB": buf.nbytes / 1e6, "cpus_usable": len(os.sched_getaffinity(0))}
try:
    with open("/sys/fs/cgroup/cpu.max") as f:
        q, per = f.read().split()[:2]
        out["cgroup_cpu_quota"] = None if q == "max" else float(q) / float(per)
except OSError:
    pass
# the link: pinned host memory -> HBM, same number of bytes
pin = torch.empty(buf.nbytes, dtype=torch.uint8).pin_memory()
dev = torch.empty(buf.nbytes, dtype=torch.uint8, device="cuda")
dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
out["h2d_pinned_GBps"] = buf.nbytes * 5 / (time.perf_counter() - t0) / 1e9
# one host thread copying pageable -> pinned (what a staging thread does, minus the offset narrowing)
dst = pin.numpy()
np.copyto(dst, buf)
t0 = time.perf_counter()
for _ in range(3):
    np.copyto(dst, buf)
out["one_thread_copy_GBps"] = buf.nbytes * 3 / (time.perf_counter() - t0) / 1e9
del pin, dev
rates = {}
for th in (4, 8, 12, 16, 24):
    os.environ["FA_STAGE_THREADS"] = str(th)
    with fa.FlowAgg(framed=True, max_batch_records=1 << 22) as agg:
        agg.ingest(buf, off)  # warm-up: staging buffers, tables
        agg.sync()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            agg.ingest(buf, off)
        agg.sync()
        dt = (time.perf_counter() - t0) / reps
        assert int(agg.read_window()["count"].sum()) == n * (reps + 1)
    rates[str(th)] = {"M_records_per_s": n / dt / 1e6, "wire_GBps": buf.nbytes / dt / 1e9}
out["fa_ingest_by_staging_threads"] = rates
print(json.dumps(out))
