#!/usr/bin/env python3
"""BASELINE config 5, single-GPU shape, at scale: two concurrent key sets - flows_5m
(SrcAS,DstAS) and (SrcAddr,DstPort,Proto) - over 60-second sub-buckets, 5-minute windows tumbling AND sliding by
60 s, Zipf-0.8 addresses, seed 5; records regenerated in HBM chunk by chunk.  Checks (CPU side = oracle/):
  * every 5-minute-aligned window of flows_5m: rows bit-exact (order-independent checksum over keys and sums,
    row count) against the C oracle's rollup of the same records;
  * one sliding window (start not 5-minute aligned): rows byte-identical to the oracle rollup of exactly the records
    whose TimeReceived falls into [start, start+300);
  * (SrcAddr,DstPort,Proto): count() and sum(Bytes) over all rows == the stream's totals, one row set per window.
Prints one JSON line (commit it under profiles/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402
from bench import effective_cpus  # noqa: E402  (CPUs this process can really use: affinity cut by the cgroup quota)


def mix64(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def checksum(rows):
    with np.errstate(over="ignore"):
        a = (rows["timeslot"].astype(np.uint64) << np.uint64(32)) | rows["etype"].astype(np.uint64)
        b = (rows["src_as"].astype(np.uint64) << np.uint64(32)) | rows["dst_as"].astype(np.uint64)
        h = mix64(a ^ mix64(b))
        v = rows["bytes"] * np.uint64(3) + rows["packets"] * np.uint64(5) + rows["count"] * np.uint64(7) + np.uint64(1)
        return int((h * v).sum(dtype=np.uint64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=100_000_000)
    ap.add_argument("--chunk", type=int, default=16_666_667)
    ap.add_argument("--span", type=int, default=1800, help="seconds of event time the stream covers (6 windows)")
    ap.add_argument("--wide-log2", type=int, default=26, help="slots of the (SrcAddr,DstPort,Proto) table, log2 (round 3 ran 2^28 for the scatter sink; with the log only the first launch's 16.6 M rows ever reach the table)")
    ap.add_argument("--universe-log2", type=int, default=24)
    ap.add_argument("--rows48", action="store_true", help="window reads through fa_read_window_app48: 48-byte rows (the date / timeslot a window's rows share stay behind)")
    ap.add_argument("--pinned-out", action="store_true", help="the consumer's row buffer is page-locked: window reads are one copy-engine transfer")
    ap.add_argument("--table-log2", type=int, default=24, help="slots of the flows_5m table, log2 (3.9 M groups at the default span)")
    args = ap.parse_args()
    import torch
    fa = _pkg.load()
    po = _pkg.load_oracle()
    fa.build()
    dev = torch.device("cuda", 0)
    n = args.records
    threads = min(64, effective_cpus()[0])
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=5, n_total=n, span_secs=args.span, zipf_log2_universe=args.universe_log2, zipf_s_x100=80)
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=5, n_total=n, span_secs=args.span, zipf_log2_universe=args.universe_log2, zipf_s_x100=80)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    out = {"config": "BASELINE configs[4], single-GPU shape: (SrcAS,DstAS) + (SrcAddr,DstPort,Proto), 60-s sub-buckets, 5-min windows "
                     "(tumbling + sliding by 60 s), %d framed FlowMessages, Zipf-0.8, seed 5, %d s of event time" % (n, args.span)}
    with fa.FlowAgg(framed=True, key_sets=ks, window_secs=300, subwindow_secs=60, wide_capacity_log2=args.wide_log2, table_capacity_log2=args.table_log2,
                    max_batch_records=args.chunk) as agg:
        cap = args.chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(args.chunk + 1, dtype=torch.int32, device=dev)
        wire = 0
        i0 = 0
        st0 = agg.stats()
        t_ing = 0.0
        while i0 < n:
            m = min(args.chunk, n - i0)
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            t1 = time.perf_counter()
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            agg.sync()
            t_ing += time.perf_counter() - t1
            wire += w
            i0 += m
        st1 = agg.stats()
        launches = st1["kernel_launches"] - st0["kernel_launches"]
        path_s = (st1["batch_ns_total"] - st0["batch_ns_total"]) * 1e-9
        out.update({"records": n, "wire_bytes": wire, "launches": int(launches), "path_ms_per_launch": path_s / launches * 1e3,
                    "records_per_s_device_path": n / path_s, "roofline_frac_path": wire / path_s / 8e12,
                    "ingest_wall_s_with_sync_per_chunk": t_ing, "flows_5m_groups_in_table": int(st1["table_used"]),
                    "wide_rows_in_table": int(st1["wide_used"]), "wide_table_capacity": int(st1["wide_capacity"])})
        assert st1["records_ok"] == n and st1["records_bad"] == 0
        # ---- flows_5m, every aligned window (peek: nothing removed)
        t0 = fa.T0
        aligned = [t0 + 300 * k for k in range((args.span + 299) // 300)]
        wins, tms = [], []
        for ts in aligned:
            tw = time.perf_counter()
            wins.append(agg.read_window(ts))
            tms.append((time.perf_counter() - tw) * 1e3)
        # (the first read of a ctx allocates the sort's temporary storage and the row buffers: reported apart)
        out["read_first_aligned_window_ms"] = tms[0]
        out["read_aligned_windows_ms_median"] = float(np.median(tms[1:])) if len(tms) > 1 else tms[0]
        allrows = np.concatenate(wins)
        ref = po.bench_rollup(gp, 0, n, threads)
        out["flows_5m_rows"] = int(len(allrows))
        out["flows_5m_aligned_windows_bit_exact"] = bool(ref["bad"] == 0 and ref["groups"] == len(allrows) and checksum(allrows) == ref["checksum"]
                                                       and int(allrows["count"].sum()) == n)
        # ---- one sliding window: [t0 + 420, t0 + 720)
        start = t0 + 420
        got = agg.read_window(start)
        ia = -(-(start - t0) * n // args.span)          # first record with TimeReceived >= start  (t = t0 + span*i // n)
        ib = -(-(start + 300 - t0) * n // args.span)
        rr = po.Rollup(300)
        step = 8_000_000
        for a in range(ia, ib, step):
            buf, off = po.gen_records(gp, a, min(step, ib - a))
            assert rr.ingest(buf, off, 1) == 0
        want = rr.rows()
        # fold the oracle's two aligned timeslots into the window (the window start is the row's timeslot)
        key = np.stack([want[c].astype(np.uint64) for c in ("src_as", "dst_as", "etype")], axis=1)
        order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
        want = want[order]
        key = key[order]
        first = np.ones(len(want), dtype=bool)
        first[1:] = (key[1:] != key[:-1]).any(axis=1)
        starts = np.nonzero(first)[0]
        folded = want[starts].copy()
        for c in ("bytes", "packets", "count"):
            folded[c] = np.add.reduceat(want[c], starts)
        folded["timeslot"] = start
        folded["date"] = start // 86400
        out["sliding_window_rows"] = int(len(got))
        out["sliding_window_bit_exact"] = bool(got.tobytes() == folded.tobytes())
        # ---- (SrcAddr,DstPort,Proto): totals over all windows
        cnt = by = 0
        nrows = 0
        app_ms = []
        # (a consumer keeps ONE row buffer per kind: fresh pages - 930 MB per window - cost more than the copy into them)
        nreuse = int(st1["wide_used"] + st1["wide_log_records"]) // max(len(aligned) - 1, 1) + (1 << 20)
        reuse = fa.FlowAgg.pinned_rows(fa.ROWS_APP, nreuse) if args.pinned_out else np.empty(nreuse, dtype=fa.ROW_APP_DTYPE)
        if args.rows48:
            reuse = reuse.view(np.uint8)[:nreuse * 48].view(fa.ROW_APP48_DTYPE)
            out["row_format"] = "fa_row_app48 (48 bytes: the window's date and timeslot are returned once)"
        reuse.view(np.uint8)[::4096] = 0

        def read_app(ts):
            return agg.read_window_app48(ts, out=reuse)[0] if args.rows48 else agg.read_window_app(ts, out=reuse)
        out["row_buffer"] = "page-locked (one copy-engine transfer per read)" if args.pinned_out else "pageable (relayed through the ctx's pinned slots)"
        sums = []
        for ts in aligned:
            tw = time.perf_counter()
            app = read_app(ts)
            app_ms.append((time.perf_counter() - tw) * 1e3)
            sums.append((int(app["count"].sum()), int(app["bytes"].sum(dtype=np.uint64)), len(app)))
            cnt += sums[-1][0]
            by += sums[-1][1]
            nrows += len(app)
        out["app_rows"] = nrows
        out["read_app_windows_ms"] = [round(x, 1) for x in app_ms]  # (SrcAddr,DstPort,Proto) rows of one aligned window each: collect + device merge + copy out (reused host buffer)
        # ---- real closes, oldest window first: read + drop of its five 60-s sub-buckets in one pass (the log's watermark moves;
        # the table's rows of the window are zeroed in place)
        close_ms, closes_ok = [], True
        for i, ts in enumerate(aligned):
            tw = time.perf_counter()
            app = read_app(ts)
            agg.drop_range(fa.ROWS_APP, ts, ts + 300)  # (a tumbling consumer: the window's five sub-buckets in one pass)
            close_ms.append((time.perf_counter() - tw) * 1e3)
            closes_ok = closes_ok and (int(app["count"].sum()), int(app["bytes"].sum(dtype=np.uint64)), len(app)) == sums[i]
        stc = agg.stats()
        out["close_app_windows_ms"] = [round(x, 1) for x in close_ms]
        out["closes_return_the_windows_read_before"] = bool(closes_ok)
        out["app_rows_left_after_all_closes"] = int(len(agg.read_window_app()))
        out["wide_log_after_closes"] = {k: int(stc[k]) for k in ("wide_log_chunks", "wide_log_folded", "wide_log_replayed", "wide_log_dropped", "wide_log_watermark_moves")}
        ingest_ms_per_window = path_s * 1e3 / max(len(aligned), 1)
        out["per_window_ms"] = {"ingest_device_path": round(ingest_ms_per_window, 2), "close_median": round(float(np.median(close_ms)), 1),
                                "ingest_plus_close": round(ingest_ms_per_window + float(np.median(close_ms)), 1)}
        out["roofline_frac_ingest_plus_close"] = wire / ((path_s + sum(close_ms) * 1e-3) * 8e12)
        out["wide_mode"] = os.environ.get("FA_WIDE", "adaptive")
        out["app_count_equals_records"] = bool(cnt == n)
        out["app_sum_bytes_equals_flows_5m"] = bool(by == int(allrows["bytes"].sum(dtype=np.uint64)))
    print(json.dumps(out))
    ok = (out["flows_5m_aligned_windows_bit_exact"] and out["sliding_window_bit_exact"] and out["app_count_equals_records"] and out["app_sum_bytes_equals_flows_5m"]
          and out["closes_return_the_windows_read_before"] and out["app_rows_left_after_all_closes"] == 0)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
