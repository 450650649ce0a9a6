#!/usr/bin/env python3
"""BASELINE config 4 (configs[3]) on ONE GPU: "8 Kafka partitions sharded 1-per-GPU, per-GPU sketch + RCCL all-reduce merge at
window close, 1 B flows".  Only one MI355X is reachable from the build environment, so the 8 ranks of this run share it:
eight processes (torch.distributed.run), one context each, exchange over gloo - the per-rank kernels, the merge code
(flow-pipeline_amd/dist.py) and the checks are the ones an 8-GPU node runs; what this run canNOT show is xGMI bandwidth
or scaling (the driver's bench.py --gpus 8 does).  Partition p = the 4 M-record chunks c of the stream with c % 8 == p
(every partition spans the whole time range, as Kafka partitions of one topic do).
Checks at window close (CPU side = oracle/, on rank 0):
  * merged sketches (all-reduce of the 8 ranks' sketches into each rank's merged view) BIT-EXACT against the CPU sketch
    of the whole 1 B-record stream, identical on every rank;
  * merged top-100 (every rank ranks ITS distinct addresses by the merged estimate and sends its first 100 rows; the
    union is merged on the device - exact, see dist.topk_merged) == the ranking of the whole address universe by the CPU
    sketch, identical on every rank;
  * merged flows_5m rows (device buffers gathered, merged by fa_rows_merge_device) == the C oracle's rollup of the whole
    stream (row count, order-independent checksum over keys and sums), identical on every rank.
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P tools/config4_run.py
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
from config3_run import estimates, universe_keys  # noqa: E402
from config5_run import checksum  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1_000_000_000)
    ap.add_argument("--chunk", type=int, default=4_166_667)
    ap.add_argument("--universe-log2", type=int, default=24)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # every rank on the box's only GPU; the exchange goes over gloo (host memory)
    dist.init_process_group("gloo")
    fa = _pkg.load()
    po = _pkg.load_oracle()
    dev = torch.device("cuda", 0)
    n, L = args.records, args.universe_log2
    depth, wl2, seed = 4, 20, 0x5EED
    KS = (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=3, n_total=n, span_secs=900, zipf_log2_universe=L, zipf_s_x100=110)
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=3, n_total=n, span_secs=900, zipf_log2_universe=L, zipf_s_x100=110)
    nchunks = (n + args.chunk - 1) // args.chunk
    out = {"config": "BASELINE configs[3] on one GPU: %d ranks (one context each, gloo exchange) x partitions of a %d-record Zipf-1.1 stream "
                     "(2^%d addresses, chunks of %d records dealt round-robin), key sets flows_5m + both sketches (depth %d x 2^%d); "
                     "merge at window close: sketches all-reduced into the merged view, each rank's top rows and flows_5m rows gathered and merged on the device"
                     % (world, n, L, args.chunk, depth, wl2)}
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=L + 2,
                    max_batch_records=args.chunk) as agg:
        cap = args.chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(args.chunk + 1, dtype=torch.int32, device=dev)
        mine = 0
        wire = 0
        st0 = agg.stats()
        t0 = time.perf_counter()
        for c in range(rank, nchunks, world):
            i0 = c * args.chunk
            m = min(args.chunk, n - i0)
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            mine += m
            wire += w
        agg.sync()
        t_ingest = time.perf_counter() - t0
        st1 = agg.stats()
        assert st1["records_ok"] == mine and st1["records_bad"] == 0
        path_s = (st1["batch_ns_total"] - st0["batch_ns_total"]) * 1e-9
        # ---- window close: merges (every rank ends up with the same merged state)
        dist.barrier()
        t0 = time.perf_counter()
        tops = [fa.dist.topk_merged(agg, k, 100, candidates_per_rank=None, device="cpu") for k in KS]
        t_topk = time.perf_counter() - t0
        merged_sk = [agg.cms_read(k).reshape(-1).copy() for k in KS]
        t0 = time.perf_counter()
        rows = fa.dist.close_window_merged(agg, fa.ALL_TIMESLOTS, device="cpu")
        t_rows = time.perf_counter() - t0
    # ---- every rank holds the same merged results?
    h = hashlib.sha256()
    for a in merged_sk:
        h.update(a.tobytes())
    for t in tops:
        h.update(t.tobytes())
    h.update(rows.tobytes())
    digest = np.frombuffer(h.digest(), dtype=np.uint8).copy()
    digests = fa.dist.allgather_bytes(digest, device="cpu")
    per_rank = torch.tensor([float(mine), float(wire), path_s, t_ingest, t_topk, t_rows], dtype=torch.float64)
    gathered = [torch.zeros_like(per_rank) for _ in range(world)]
    dist.all_gather(gathered, per_rank)
    if rank == 0:
        g = np.stack([x.numpy() for x in gathered])
        out.update({
            "ranks": world, "records": int(g[:, 0].sum()), "wire_bytes": int(g[:, 1].sum()),
            "device_path_seconds_per_rank": [round(float(x), 4) for x in g[:, 2]],
            "records_per_s_per_rank_device_path": [round(float(a / b)) for a, b in zip(g[:, 0], g[:, 2])],
            "note_on_rates": "the ranks time-share one GPU: a rank's device-path time (hipEvents around its own launches) includes waiting for the others' kernels",
            "ingest_wall_s_slowest_rank": float(g[:, 3].max()),
            "merge_topk_both_sketches_s": float(g[:, 4].max()), "merge_rows_s": float(g[:, 5].max()),
            "all_ranks_hold_identical_merged_results": bool(all(bytes(d) == bytes(digests[0]) for d in digests)),
            "flows_5m_rows": int(len(rows)),
        })
        assert out["records"] == n
        threads = min(64, effective_cpus()[0])
        t0 = time.perf_counter()
        words = depth << wl2
        c_src = np.zeros(words, dtype=np.uint64)
        c_dst = np.zeros(words, dtype=np.uint64)
        po.cms_stream(gp, 0, n, threads, depth, wl2, seed, c_src, c_dst)
        out["merged_sketches_bit_exact_vs_whole_stream"] = bool(np.array_equal(merged_sk[0], c_src) and np.array_equal(merged_sk[1], c_dst))
        ok_top = True
        for dst, (cms, top) in enumerate(zip((c_src, c_dst), tops)):
            lo, hi = universe_keys(L, dst)
            est = estimates(cms, lo, hi, depth, wl2, seed)
            cand = np.argpartition(est, len(est) - 400)[-400:]
            uniq = {}
            for i in cand:
                uniq[lo[i].tobytes() + hi[i].tobytes()] = int(est[i])
            want = sorted(uniq.items(), key=lambda kv: (-kv[1], kv[0]))[:100]
            got = [(bytes(r["key"]), int(r["weight"])) for r in top]
            ok_top = ok_top and got == want
        out["merged_top100_equals_ranking_of_the_whole_universe"] = bool(ok_top)
        ref = po.bench_rollup(gp, 0, n, min(8, threads))
        out["merged_rows_equal_oracle_rollup"] = bool(ref["bad"] == 0 and ref["groups"] == len(rows) and checksum(rows) == ref["checksum"]
                                                     and int(rows["count"].sum()) == n)
        out["cpu_oracle_seconds"] = time.perf_counter() - t0
        print(json.dumps(out), flush=True)
        ok = (out["merged_sketches_bit_exact_vs_whole_stream"] and ok_top and out["merged_rows_equal_oracle_rollup"]
              and out["all_ranks_hold_identical_merged_results"])
    else:
        ok = True
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
