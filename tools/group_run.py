#!/usr/bin/env python3
"""BASELINE configs[3] / [4] through the C-ABI group close (fa_group_*, ABI 7): ONE process, 8 contexts - one per Kafka
partition, all on the box's only MI355X (the peer-copy transport degrades to device copies; what this run cannot show is xGMI
bandwidth or scaling) - each ingests its partition of a Zipf stream with the flows_5m rollup, both Count-Min sketches and the
(SrcAddr,DstPort,Proto) key set; then the windows of the whole topic are closed through the group:
  * flows_5m rows of every aligned window, merged over the partitions (gathered on one member, merged in HBM);
  * (SrcAddr,DstPort,Proto) rows of every aligned window, hash-partitioned (every member merges and copies out 1 / 8 of the keys);
  * sketches reduced into every member's merged view (reduce-scatter + all-gather by copies), top-100 of both.
Checks (CPU side = oracle/): flows_5m rows of all windows == the C oracle's rollup of the WHOLE stream (row count, order-
independent checksum); (SrcAddr,DstPort,Proto): count() == records and sum(Bytes) == flows_5m's, every key in the share of its
owner (sampled); merged sketches bit-exact against the CPU sketch of the whole stream; top-100 == the ranking of the whole
address universe by the CPU sketch.  Prints one JSON line (commit it under profiles/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import _pkg  # noqa: E402
from bench import effective_cpus, rows_checksum  # noqa: E402
from config3_run import estimates, universe_keys  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=200_000_000)
    ap.add_argument("--members", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=8_333_334)
    ap.add_argument("--span", type=int, default=1800)
    ap.add_argument("--universe-log2", type=int, default=24)
    ap.add_argument("--topk-mode", default="exact", choices=["exact", "candidates"])
    args = ap.parse_args()
    import torch
    fa = _pkg.load()
    po = _pkg.load_oracle()
    dev = torch.device("cuda", 0)
    n, nm, L = args.records, args.members, args.universe_log2
    depth, wl2, seed = 4, 20, 0x5EED
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=3, n_total=n, span_secs=args.span, zipf_log2_universe=L, zipf_s_x100=110)
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=3, n_total=n, span_secs=args.span, zipf_log2_universe=L, zipf_s_x100=110)
    cand = args.topk_mode == "candidates"
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_SRCADDR_CMS | fa.FA_KEYS_DSTADDR_CMS | fa.FA_KEYS_ADDR_PORT_PROTO
    out = {"config": "BASELINE configs[3] + [4] shapes through fa_group_* on one GPU: %d contexts (one per Kafka partition) in one process, %d-record Zipf-1.1 "
                     "stream (seed 3, %d s of event time), key sets flows_5m + both Count-Min sketches (%d x 2^%d) + (SrcAddr,DstPort,Proto); top-k mode %s"
                     % (nm, n, args.span, depth, wl2, args.topk_mode)}
    kw = dict(framed=True, key_sets=ks, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, max_batch_records=args.chunk, wide_capacity_log2=24, table_capacity_log2=22,
              topk_capacity_log2=16 if cand else L + 1, topk_mode=fa.TOPK_CANDIDATES if cand else fa.TOPK_EXACT)
    members = [fa.FlowAgg(**kw) for _ in range(nm)]
    try:
        cap = args.chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(args.chunk + 1, dtype=torch.int32, device=dev)
        nchunks = (n + args.chunk - 1) // args.chunk
        wire = 0
        t0w = time.perf_counter()
        for c in range(nchunks):  # chunk c belongs to partition c % members (every partition spans the whole time range)
            i0 = c * args.chunk
            m = min(args.chunk, n - i0)
            agg = members[c % nm]
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            agg.sync()  # (the one generator buffer is reused)
            wire += w
        out["ingest_wall_s_with_generation"] = time.perf_counter() - t0w
        del d_buf, d_off
        torch.cuda.empty_cache()
        with fa.FlowGroup(members) as g:
            st = g.stats()
            assert st["records_ok"] == n and st["records_bad"] == 0, st
            out.update({"records": n, "wire_bytes": wire, "members": nm, "transport": "peer copies (one GPU: device copies)" if g.transport == fa.GROUP_PEER else "rccl",
                        "device_path_ms_sum_over_members": st["batch_ns_total"] * 1e-6})
            slots = [int(t) for t in g.open_timeslots()]
            out["windows"] = len(slots)
            # ---- flows_5m, merged
            t = time.perf_counter()
            wins = [g.read_window(fa.ROWS_5M, ts, cap=1 << 20) for ts in slots]
            out["read_5m_windows_merged_ms"] = [round(1e3 * (time.perf_counter() - t) / max(len(slots), 1), 2), "per window, %d windows" % len(slots)]
            allrows = np.concatenate(wins)
            # ---- (SrcAddr,DstPort,Proto), partitioned
            cnt = by = nrows = 0
            owned_ok = True
            app_ms = []
            # (rows of a window <= its records; a buffer that is too small costs a whole close: the library reports the size it needs
            # only after collect + exchange + merge - round 5's "first close 2x the median" was mostly this retry)
            buf_rows = 2 * n // max(len(slots), 1) + (1 << 20)
            for ts in slots:
                t = time.perf_counter()
                rows, shares = g.read_window_partitioned(fa.ROWS_APP, ts, cap=buf_rows)
                app_ms.append(round(1e3 * (time.perf_counter() - t), 1))
                buf_rows = max(buf_rows, len(rows) + (1 << 16))
                cnt += int(rows["count"].sum())
                by += int(rows["bytes"].sum(dtype=np.uint64))
                nrows += len(rows)
                samp = rows[::211]
                owner = fa.dist.partition_rows_host(samp, fa.ROWS_APP, nm)
                bounds = np.cumsum([0] + shares)
                owned_ok = owned_ok and bool((np.searchsorted(bounds, np.arange(len(rows))[::211], side="right") - 1 == owner).all())
            out["read_app_windows_partitioned_ms"] = app_ms
            out["app_rows"] = nrows
            # ---- sketches + top-k
            t = time.perf_counter()
            g.allreduce_sketches()
            out["allreduce_both_sketches_ms"] = round(1e3 * (time.perf_counter() - t), 2)
            t = time.perf_counter()
            tops = [g.topk(k, 100) for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]
            out["topk100_both_sketches_first_ms"] = round(1e3 * (time.perf_counter() - t), 2)
            t = time.perf_counter()
            tops2 = [g.topk(k, 100) for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]
            out["topk100_both_sketches_again_ms"] = round(1e3 * (time.perf_counter() - t), 2)
            sk = [members[nm - 1].cms_read(k).reshape(-1).copy() for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]  # any member: the merged view
            # ---- a real close of the oldest window
            t = time.perf_counter()
            closed = g.close_window(fa.ROWS_5M, slots[0], cap=1 << 20)
            closed_app, _ = g.close_window_partitioned(fa.ROWS_APP, slots[0], cap=buf_rows)
            out["close_oldest_window_both_key_sets_ms"] = round(1e3 * (time.perf_counter() - t), 1)
            left = g.read_window(fa.ROWS_5M, cap=1 << 22)
    finally:
        for m in members:
            m.close()
    # ---- CPU side
    threads = min(64, effective_cpus()[0])
    t = time.perf_counter()
    ref = po.bench_rollup_ex(gp, 0, n, threads, groups_hint=len(allrows))
    c_src = np.zeros(depth << wl2, dtype=np.uint64)
    c_dst = np.zeros(depth << wl2, dtype=np.uint64)
    po.cms_stream(gp, 0, n, threads, depth, wl2, seed, c_src, c_dst)
    out["cpu_oracle_seconds"] = time.perf_counter() - t
    out["flows_5m_rows"] = int(len(allrows))
    out["flows_5m_merged_equals_oracle_rollup_of_all_partitions"] = bool(ref["bad"] == 0 and ref["groups"] == len(allrows) and ref["checksum"] == rows_checksum(allrows)
                                                                           and int(allrows["count"].sum()) == n)
    out["closed_window_equals_its_read"] = bool(closed.tobytes() == wins[0].tobytes() and int(left["count"].sum()) == n - int(closed["count"].sum()))
    out["app_count_equals_records"] = bool(cnt == n)
    out["app_sum_bytes_equals_flows_5m"] = bool(by == int(allrows["bytes"].sum(dtype=np.uint64)))
    out["app_rows_in_their_owners_share"] = bool(owned_ok)
    out["closed_app_rows"] = int(len(closed_app))
    out["merged_sketches_bit_exact"] = bool(np.array_equal(sk[0], c_src) and np.array_equal(sk[1], c_dst))
    ok_top = tops[0].tobytes() == tops2[0].tobytes() and tops[1].tobytes() == tops2[1].tobytes()
    for dst, (cms, top) in enumerate(zip((c_src, c_dst), tops)):
        lo, hi = universe_keys(L, dst)
        est = estimates(cms, lo, hi, depth, wl2, seed)
        c400 = np.argpartition(est, len(est) - 400)[-400:]
        uniq = {}
        for i in c400:
            uniq[lo[i].tobytes() + hi[i].tobytes()] = int(est[i])
        want = sorted(uniq.items(), key=lambda kv: (-kv[1], kv[0]))[:100]
        ok_top = ok_top and [(bytes(r["key"]), int(r["weight"])) for r in top] == want
    out["top100_equals_ranking_of_the_whole_universe"] = bool(ok_top)
    print(json.dumps(out))
    ok = all(out[k] for k in ("flows_5m_merged_equals_oracle_rollup_of_all_partitions", "closed_window_equals_its_read", "app_count_equals_records",
                              "app_sum_bytes_equals_flows_5m", "app_rows_in_their_owners_share", "merged_sketches_bit_exact", "top100_equals_ranking_of_the_whole_universe"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
