#!/usr/bin/env python3
"""The C++ host program (flow-pipeline_amd/host/inserter_gpu: the reference consumer's shape, one thread per claimed partition, ONE
process, the window close of the whole topic through fa_group_*) at scale on one GPU: 8 partition logs of a Zipf stream with the
flows_5m rollup, both sketches and (SrcAddr,DstPort,Proto); its RowBinary output == the C oracle's rollup of ALL partitions, the
(SrcAddr,DstPort,Proto) rows add up to the stream, its top-k == the exhaustive ranking by the CPU sketch.  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
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
    ap.add_argument("--records", type=int, default=48_000_000)
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--flush", type=int, default=262144)
    ap.add_argument("--universe-log2", type=int, default=20)
    ap.add_argument("--topk-mode", default="auto", choices=["auto", "exact", "candidates"], help="auto = the program's default: candidates with more than one partition")
    args = ap.parse_args()
    fa = _pkg.load()
    po = _pkg.load_oracle()
    fa.build()
    host = os.path.join(ROOT, "flow-pipeline_amd", "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    n, nparts, L = args.records, args.partitions, args.universe_log2
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=9, n_total=n, span_secs=900, zipf_log2_universe=L, zipf_s_x100=110)
    out = {"config": "inserter_gpu (C++ host, one process, %d partition threads, group close) on one GPU: %d-record Zipf-1.1 stream, key sets flows_5m + both sketches + "
                     "(SrcAddr,DstPort,Proto), flush.count %d" % (nparts, n, args.flush)}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        t = time.perf_counter()
        step = 4_000_000
        files = [open(os.path.join(tmp, "p%d.log" % p), "wb") for p in range(nparts)]
        for c, i0 in enumerate(range(0, n, step)):  # chunk c of the stream is partition c % partitions' next stretch of messages
            buf, _ = po.gen_records(gp, i0, min(step, n - i0))
            files[c % nparts].write(bytes(buf))
        for f in files:
            f.close()
        out["generate_logs_s"] = time.perf_counter() - t
        paths = [os.path.join(tmp, "p%d.log" % p) for p in range(nparts)]
        rb, app, topk, met = (os.path.join(tmp, x) for x in ("flows_5m.rowbinary", "app.rows", "topk.tsv", "metrics.txt"))
        t = time.perf_counter()
        r = subprocess.run([os.path.join(host, "inserter_gpu"), "-input.files=" + ",".join(paths), "-flush.count=%d" % args.flush, "-flush.dur=1h", "-key.sets=15",
                            "-out.rowbinary=" + rb, "-out.app=" + app, "-out.topk=" + topk, "-topk.k=100", "-metrics.dump=" + met, "-gpu.devices=1", "-gpu.wide.log2=25", "-loglevel=info",
                            "-topk.mode=" + args.topk_mode] + (["-gpu.keyset.log2=22"] if args.topk_mode == "exact" else []),
                           capture_output=True, text=True)
        out["host_wall_s"] = time.perf_counter() - t
        if r.returncode != 0:
            print(json.dumps(dict(out, error=r.stderr[-2000:])))
            sys.exit(1)
        out["host_phases"] = [l.split('msg="')[1].rstrip('"') for l in r.stderr.splitlines() if "phases:" in l]
        metrics = {l.split()[0]: int(l.split()[1]) for l in open(met) if not l.startswith("#")}
        rows = fa.rowbinary_to_rows(open(rb, "rb").read()) if os.path.getsize(rb) < (1 << 28) else None
        app_rows = np.fromfile(app, dtype=fa.ROW_APP_DTYPE)
        lines = [l.split("\t") for l in open(topk).read().splitlines()]
    out["insert_count"] = metrics["insert_count"]
    out["topk_mode"] = args.topk_mode
    cand = args.topk_mode != "exact"  # (auto: more than one partition -> candidates)
    # device tables of one partition's ctx: distinct-address sets 2 x 32 B x 2^log2, sketches 2 x 4 x 2^20 x 8 B (+ their merged views), flows_5m table 64 B x 2^20, wide 64 B x 2^25
    out["table_bytes_per_partition"] = {"distinct_address_sets": 2 * 32 << (16 if cand else 22), "sketches_and_merged_views": 4 * 4 * 8 << 20, "flows_5m_table": 64 << 20, "wide_table": 64 << 25}
    out["distinct_address_sets_bytes_all_partitions"] = nparts * (2 * 32 << (16 if cand else 22))
    out["records_per_s_wall_including_file_reads"] = n / out["host_wall_s"]
    threads = min(64, effective_cpus()[0])
    ref = po.bench_rollup_ex(gp, 0, n, threads, groups_hint=len(rows))
    out["flows_5m_rows"] = int(len(rows))
    out["rowbinary_equals_oracle_rollup_of_all_partitions"] = bool(ref["bad"] == 0 and ref["groups"] == len(rows) and ref["checksum"] == rows_checksum(rows) and int(rows["count"].sum()) == n)
    out["app_rows"] = int(len(app_rows))
    out["app_count_equals_records"] = bool(int(app_rows["count"].sum()) == n)
    out["app_sum_bytes_equals_flows_5m"] = bool(int(app_rows["bytes"].sum(dtype=np.uint64)) == int(rows["bytes"].sum(dtype=np.uint64)))
    depth, wl2, seed = 4, 20, 0
    c_src = np.zeros(depth << wl2, dtype=np.uint64)
    c_dst = np.zeros(depth << wl2, dtype=np.uint64)
    po.cms_stream(gp, 0, n, threads, depth, wl2, seed, c_src, c_dst)
    ok_top = True
    for dst, (cms, tag) in enumerate(((c_src, "src"), (c_dst, "dst"))):
        lo, hi = universe_keys(L, dst)
        est = estimates(cms, lo, hi, depth, wl2, seed)
        c400 = np.argpartition(est, len(est) - 400)[-400:]
        uniq = {}
        for i in c400:
            uniq[lo[i].tobytes() + hi[i].tobytes()] = int(est[i])
        want = sorted(uniq.items(), key=lambda kv: (-kv[1], kv[0]))[:100]
        mine = [(bytes.fromhex(k), int(v)) for tg, k, v in lines if tg == tag]
        ok_top = ok_top and mine == want
    out["top100_equals_ranking_of_the_whole_universe"] = bool(ok_top)
    print(json.dumps(out))
    ok = all(out[k] for k in ("rowbinary_equals_oracle_rollup_of_all_partitions", "app_count_equals_records", "app_sum_bytes_equals_flows_5m", "top100_equals_ranking_of_the_whole_universe"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
