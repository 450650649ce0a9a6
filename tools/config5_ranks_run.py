#!/usr/bin/env python3
"""BASELINE config 5 (configs[4]) on ONE GPU: "8xMI355X: concurrent multi-key rollup (SrcAS,DstAS)+(SrcAddr,DstPort,Proto)
with 1-min sliding windows, Zipf-0.8".  Only one MI355X is reachable from the build environment, so the 8 ranks share it:
eight processes (torch.distributed.run), one context each, exchange over gloo - the per-rank kernels, the device-side
window-close merges - flows_5m rows all-gathered (fa_rows_device -> gathered device buffers -> fa_rows_merge_device), the
(SrcAddr,DstPort,Proto) rows HASH-PARTITIONED (fa_rows_partition_device -> one all-to-all -> every rank merges and keeps 1 / ranks
of the keys; flow-pipeline_amd/dist.py) - and the checks are the ones an 8-GPU node runs; what this run canNOT show is xGMI
bandwidth or scaling.
Partition p = the chunks c of the stream with c % ranks == p (every partition spans the whole time range).
Checks (CPU side = oracle/, rank 0):
  * flows_5m, every 5-minute-aligned window closed across ranks: rows == the C oracle's rollup of the whole stream
    (row count, order-independent checksum over keys and sums, count() total);
  * one SLIDING window (start on a 60-s sub-bucket, not 5-minute aligned) of both key sets, merged across ranks:
    flows_5m rows byte-identical to the oracle rollup of exactly the records inside [start, start + 300), folded;
    (SrcAddr,DstPort,Proto) rows byte-identical to the numpy restatement over the same records;
  * (SrcAddr,DstPort,Proto) over all aligned windows: count() == records, sum(Bytes) == flows_5m's;
  * every rank holds byte-identical merged flows_5m results, and of the partitioned (SrcAddr,DstPort,Proto) rows exactly the keys it owns.
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P tools/config5_ranks_run.py
Prints one JSON line on rank 0 (commit it under profiles/)."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _pkg  # noqa: E402
from bench import effective_cpus  # noqa: E402  (CPUs this process can really use: affinity cut by the cgroup quota)
from config5_run import checksum  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=100_000_000)
    ap.add_argument("--chunk", type=int, default=4_166_667)
    ap.add_argument("--span", type=int, default=1800)
    ap.add_argument("--wide-log2", type=int, default=25)
    ap.add_argument("--table-log2", type=int, default=22)
    ap.add_argument("--universe-log2", type=int, default=24)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # every rank on the box's only GPU; the exchange's transport is gloo (host memory)
    dist.init_process_group("gloo")
    fa = _pkg.load()
    po = _pkg.load_oracle()
    dev = torch.device("cuda", 0)
    n = args.records
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=5, n_total=n, span_secs=args.span, zipf_log2_universe=args.universe_log2, zipf_s_x100=80)
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=5, n_total=n, span_secs=args.span, zipf_log2_universe=args.universe_log2, zipf_s_x100=80)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    nchunks = (n + args.chunk - 1) // args.chunk
    out = {"config": "BASELINE configs[4] on one GPU: %d ranks (one context each, gloo transport) x partitions of a %d-record Zipf-0.8 stream (seed 5, %d s of "
                     "event time), key sets (SrcAS,DstAS) + (SrcAddr,DstPort,Proto), 60-s sub-buckets, 5-min windows tumbling and sliding; window close merged "
                     "on the device (fa_rows_device -> gathered buffers -> fa_rows_merge_device)" % (world, n, args.span)}
    with fa.FlowAgg(framed=True, key_sets=ks, window_secs=300, subwindow_secs=60, wide_capacity_log2=args.wide_log2, table_capacity_log2=args.table_log2,
                    max_batch_records=args.chunk) as agg:
        cap = args.chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(args.chunk + 1, dtype=torch.int32, device=dev)
        mine = wire = 0
        st0 = agg.stats()
        t0w = time.perf_counter()
        for c in range(rank, nchunks, world):
            i0 = c * args.chunk
            m = min(args.chunk, n - i0)
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            mine += m
            wire += w
        agg.sync()
        t_ingest = time.perf_counter() - t0w
        st1 = agg.stats()
        assert st1["records_ok"] == mine and st1["records_bad"] == 0
        path_s = (st1["batch_ns_total"] - st0["batch_ns_total"]) * 1e-9
        del d_buf, d_off
        torch.cuda.empty_cache()
        t0 = fa.T0
        aligned = [t0 + 300 * k for k in range((args.span + 299) // 300)]
        dist.barrier()
        # ---- a sliding window first (peek semantics: rows_merged does not remove anything)
        start = t0 + 420
        tw = time.perf_counter()
        slide5m = fa.dist.rows_merged(agg, fa.ROWS_5M, start)
        slide_app = fa.dist.rows_merged_partitioned(agg, fa.ROWS_APP, start)  # (this rank's share)
        t_slide = time.perf_counter() - tw
        # ---- then every aligned window, closed across ranks (sliding semantics: a close drops the oldest sub-bucket only)
        tw = time.perf_counter()
        wins = [fa.dist.rows_merged(agg, fa.ROWS_5M, ts) for ts in aligned]
        t_aligned = time.perf_counter() - tw
        tw = time.perf_counter()
        cnt = by = nrows = 0
        owned_ok = True
        for ts in aligned:
            app = fa.dist.rows_merged_partitioned(agg, fa.ROWS_APP, ts)  # this rank's 1 / ranks of the window's keys
            cnt += int(app["count"].sum())
            by += int(app["bytes"].sum(dtype=np.uint64))
            nrows += len(app)
            owned_ok = owned_ok and bool((fa.dist.partition_rows_host(app[::97], fa.ROWS_APP, world) == rank).all())
        t_app = time.perf_counter() - tw
        tw = time.perf_counter()
        closed = fa.dist.close_window_merged(agg, aligned[0])       # a real close: removes the oldest sub-bucket on every rank
        closed_app = fa.dist.close_window_app_partitioned(agg, aligned[0])
        after = fa.dist.rows_merged(agg, fa.ROWS_5M, aligned[0] + 60)   # the next sliding window still reads complete
        t_close = time.perf_counter() - tw
    allrows = np.concatenate(wins)
    h = hashlib.sha256()
    for a in (slide5m, allrows, closed, after):  # (what every rank holds whole: the all-gathered kinds)
        h.update(np.ascontiguousarray(a).tobytes())
    tot = torch.tensor([cnt, by, nrows, len(closed_app), 0 if owned_ok else 1], dtype=torch.int64)
    dist.all_reduce(tot)
    cnt, by, nrows, closed_app_rows, not_owned = (int(v) for v in tot.tolist())
    # the sliding window's shares go to rank 0 for the byte-for-byte check (the sink of a sharded close would not gather them)
    slide_parts = fa.dist.allgather_struct(slide_app, fa.ROW_APP_DTYPE, device="cpu")
    slide_app = fa.dist.merge_rows_app_host(slide_parts) if rank == 0 else slide_app
    del slide_parts
    digest = np.frombuffer(h.digest(), dtype=np.uint8).copy()
    digests = fa.dist.allgather_bytes(digest, device="cpu")
    per_rank = torch.tensor([float(mine), float(wire), path_s, t_ingest, t_slide, t_aligned, t_app, t_close], dtype=torch.float64)
    gathered = [torch.zeros_like(per_rank) for _ in range(world)]
    dist.all_gather(gathered, per_rank)
    ok = True
    if rank == 0:
        g = np.stack([x.numpy() for x in gathered])
        out.update({
            "ranks": world, "records": int(g[:, 0].sum()), "wire_bytes": int(g[:, 1].sum()),
            "device_path_seconds_per_rank": [round(float(x), 4) for x in g[:, 2]],
            "note_on_rates": "the ranks time-share one GPU: a rank's device-path time includes waiting for the others' kernels",
            "ingest_wall_s_slowest_rank": float(g[:, 3].max()),
            "merge_sliding_window_both_key_sets_s": float(g[:, 4].max()), "merge_aligned_windows_flows_5m_s": float(g[:, 5].max()),
            "merge_aligned_windows_app_s": float(g[:, 6].max()), "close_and_reread_s": float(g[:, 7].max()),
            "all_ranks_hold_identical_merged_results": bool(all(bytes(d) == bytes(digests[0]) for d in digests)),
            "flows_5m_rows": int(len(allrows)), "app_rows": int(nrows), "sliding_window_rows": int(len(slide5m)), "sliding_window_app_rows": int(len(slide_app)),
            "app_exchange": "hash-partitioned all-to-all: every rank merges and keeps the keys it owns", "app_rows_all_on_their_owner": not_owned == 0,
            "closed_app_rows": closed_app_rows,
        })
        assert out["records"] == n
        threads = min(64, effective_cpus()[0])
        t0c = time.perf_counter()
        ref = po.bench_rollup(gp, 0, n, threads)
        out["flows_5m_aligned_windows_bit_exact"] = bool(ref["bad"] == 0 and ref["groups"] == len(allrows) and checksum(allrows) == ref["checksum"]
                                                       and int(allrows["count"].sum()) == n)
        out["closed_window_equals_first_aligned_window"] = bool(closed.tobytes() == wins[0].tobytes())
        # the records inside the sliding window, straight from the generator
        ia = -(-(start - t0) * n // args.span)
        ib = -(-(start + 300 - t0) * n // args.span)
        rr = po.Rollup(300)
        app_parts = []
        step = 4_000_000
        for a in range(ia, ib, step):
            buf, off = po.gen_records(gp, a, min(step, ib - a))
            assert rr.ingest(buf, off, 1) == 0
            rows_, st_ = po.decode_batch(buf, off, 1)
            app_parts.append(po.rollup_app(rows_, st_, 60, window=300, timeslot=start).astype(fa.ROW_APP_DTYPE))
        want = rr.rows()
        key = np.stack([want[c].astype(np.uint64) for c in ("src_as", "dst_as", "etype")], axis=1)
        order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
        want, key = want[order], key[order]
        first = np.ones(len(want), dtype=bool)
        first[1:] = (key[1:] != key[:-1]).any(axis=1)
        starts = np.nonzero(first)[0]
        folded = want[starts].copy()
        for c in ("bytes", "packets", "count"):
            folded[c] = np.add.reduceat(want[c], starts)
        folded["timeslot"] = start
        folded["date"] = start // 86400
        out["sliding_window_bit_exact"] = bool(slide5m.tobytes() == folded.tobytes())
        want_app = fa.dist.merge_rows_app_host(app_parts)
        out["sliding_window_app_bit_exact"] = bool(slide_app.tobytes() == want_app.tobytes())
        out["app_count_equals_records"] = bool(cnt == n)
        out["app_sum_bytes_equals_flows_5m"] = bool(by == int(allrows["bytes"].sum(dtype=np.uint64)))
        out["cpu_oracle_seconds"] = time.perf_counter() - t0c
        print(json.dumps(out), flush=True)
        ok = all(out[k] for k in ("flows_5m_aligned_windows_bit_exact", "sliding_window_bit_exact", "sliding_window_app_bit_exact", "app_count_equals_records",
                                  "app_sum_bytes_equals_flows_5m", "all_ranks_hold_identical_merged_results", "closed_window_equals_first_aligned_window",
                                  "app_rows_all_on_their_owner"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
