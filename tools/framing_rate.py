#!/usr/bin/env python3
"""Rate of the offsets-free device path (fa_ingest_device with offsets == NULL: framing.cuh cuts the framed chain into
records on the GPU) beside the same batch WITH offsets - BASELINE config 2's records, 33.3 M per call.  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402


def main():
    import torch
    fa = _pkg.load()
    fa.build()
    dev = torch.device("cuda", 0)
    n = 33_333_334
    mp = fa.mock_params(mode=fa.MOCK_ASPAIRS, framed=1, seed=2, n_total=100_000_000, span_secs=900, per_sec=400_000)
    out = {"workload": "one 33.3 M-record launch of BASELINE configs[1] (framed, 71.9 B per record)"}
    with fa.FlowAgg(framed=True, max_batch_records=n) as agg:
        cap = n * fa.mock_record_cap(fa.MOCK_ASPAIRS) + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        w = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), cap, d_off.data_ptr())
        for name, off in (("with_offsets", d_off.data_ptr()), ("offsets_null_device_framing", 0)):
            for _ in range(2):
                agg.ingest_device(d_buf.data_ptr(), w, off, n)
            agg.sync()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                agg.ingest_device(d_buf.data_ptr(), w, off, n)
            agg.sync()
            dt = (time.perf_counter() - t0) / reps
            out[name] = {"ms_per_call": dt * 1e3, "records_per_s": n / dt, "wire_GBps": w / dt / 1e9}
        rows = agg.read_window()
        assert int(rows["count"].sum()) == n * 14
    out["ratio"] = out["offsets_null_device_framing"]["ms_per_call"] / out["with_offsets"]["ms_per_call"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
