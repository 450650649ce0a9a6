#!/usr/bin/env python3
"""BASELINE config 3 as specified: 1xMI355X, Count-Min heavy hitters over SrcAddr / DstAddr,
1 B framed FlowMessages with Zipf-1.1 addresses (universe 2^24), regenerated in 16.67 M-record chunks in HBM and
ingested with key_sets = flows_5m rollup + both sketches.  Checks (the CPU side is oracle/, test infrastructure):
  * both sketches BIT-EXACT against the CPU sketch of the same stream - on the first 100 M records and on the full stream;
  * top-100 by estimate == the ranking of every address of the universe by the CPU sketch's estimate;
  * on the 100 M prefix: estimates never below the exact GROUP BY weight, and within eps * total weight (eps = e / width)
    for at least a 1 - e^-depth share of the addresses.
Prints one JSON line (commit it under profiles/)."""
import argparse
import json
import math
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


def universe_keys(L, dst):
    """(lo, hi) of the generator's address for every (v6, rank): index = rank + (v6 << L)  (gen.cuh gen_zipf_key)."""
    rank = np.arange(1 << L, dtype=np.uint64)
    with np.errstate(over="ignore"):
        a = mix64(rank * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x2222 if dst else 0x1111))
        b = mix64(a ^ np.uint64(0xD1B54A32D192ED03))
    lo = np.concatenate([a & np.uint64(0xFFFFFFFF), a])   # v4: the low 4 bytes of a, rest zero; v6: a || b
    hi = np.concatenate([np.zeros(1 << L, dtype=np.uint64), b])
    return lo, hi


def columns(a, h1, wl2, row):
    """(oracle/pyoracle.py cms_columns, restated: the tools do not import the oracle's numpy helpers at module level)"""
    pbits = min(8, wl2 - 4)
    sub = wl2 - pbits
    with np.errstate(over="ignore"):
        prefix = (h1 & np.uint64((1 << pbits) - 1)).astype(np.int64)
        l1 = (h1 >> np.uint64(32)).astype(np.uint32)
        l2 = ((a | np.uint64(1)) >> np.uint64(32)).astype(np.uint32) | np.uint32(1)
        low = ((l1 + np.uint32(row) * l2) >> np.uint32(32 - sub)).astype(np.int64)
    return (prefix << sub) | low


def estimates(cms, lo, hi, depth, wl2, seed):
    with np.errstate(over="ignore"):
        s0 = mix64(np.array([(seed + 0x9E3779B97F4A7C15) & (2**64 - 1)], dtype=np.uint64))[0]
        a = mix64(lo ^ s0)
        h1 = mix64(a ^ hi)
        best = np.full(len(lo), np.uint64(2**64 - 1), dtype=np.uint64)
        for r in range(depth):
            best = np.minimum(best, cms[columns(a, h1, wl2, r) + (r << wl2)])
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1_000_000_000)
    ap.add_argument("--chunk", type=int, default=33_333_334, help="records per launch (until round 6: 16 666 667, the wide tuples' limit of round 2; see profiles/r06_exp_config3_launch_size.jsonl)")
    ap.add_argument("--prefix", type=int, default=100_000_000)
    ap.add_argument("--universe-log2", type=int, default=24)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--topk-log2", type=int, default=0, help="slots of each distinct-address set (default: universe + 2)")
    ap.add_argument("--timing-only", action="store_true", help="A/B runs: ingest and print the path numbers, skip the CPU-side checks")
    ap.add_argument("--topk-track", type=int, default=0, help="candidates mode: the rank the admission threshold follows (0: the library's default)")
    ap.add_argument("--no-assert", action="store_true", help="ablation builds (FA_DEBUG_FLAGS: results are wrong by design): no result checks")
    ap.add_argument("--topk-mode", default="exact", choices=["exact", "candidates"],
                    help="exact: every address is kept and ranked (2^(universe + 2) slots per set); candidates: the Count-Min heavy-hitter "
                         "contract (include/flowagg.h, fa_config.topk_mode) - sets of 2^16 slots, nothing of them in the ingest path")
    args = ap.parse_args()
    import torch
    fa = _pkg.load()
    po = _pkg.load_oracle()
    fa.build()
    dev = torch.device("cuda", 0)
    n, L = args.records, args.universe_log2
    depth, wl2, seed = 4, 20, 0x5EED
    threads = args.threads or min(64, effective_cpus()[0])
    KS = (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=3, n_total=n, span_secs=900, zipf_log2_universe=L, zipf_s_x100=110)
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=3, n_total=n, span_secs=900, zipf_log2_universe=L, zipf_s_x100=110)
    out = {"config": "BASELINE configs[2]: 1xMI355X, Count-Min heavy hitters SrcAddr/DstAddr, %d framed FlowMessages, Zipf-1.1 over 2^%d addresses, "
                     "regenerated in %d-record chunks; sketch depth %d x 2^%d x u64 per key set" % (n, L, args.chunk, depth, wl2)}
    cand = args.topk_mode == "candidates"
    out["topk_mode"] = args.topk_mode
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=args.topk_log2 or (16 if cand else L + 2),
                    max_batch_records=args.chunk, topk_mode=fa.TOPK_CANDIDATES if cand else fa.TOPK_EXACT, topk_track=args.topk_track) as agg:
        cap = args.chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(args.chunk + 1, dtype=torch.int32, device=dev)
        prefix_sk = None
        wire = 0
        t_gen = t_ing = 0.0
        i0 = 0
        st0 = agg.stats()
        series, prev_ns = [], st0["batch_ns_total"]
        while i0 < n:
            m = min(args.chunk, n - i0)
            if prefix_sk is None and i0 >= args.prefix:
                prefix_sk = [agg.cms_read(k).reshape(-1).copy() for k in KS]
                out["prefix_records"] = i0
            t0 = time.perf_counter()
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            t1 = time.perf_counter()
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            agg.sync()
            t_gen += t1 - t0
            t_ing += time.perf_counter() - t1
            if args.timing_only:  # (per launch: device-path time of this chunk - the sets fill up during the first launches)
                ns = agg.stats()["batch_ns_total"]
                series.append(round((ns - prev_ns) * 1e-6, 4))
                prev_ns = ns
            wire += w
            i0 += m
        st1 = agg.stats()
        full_sk = [agg.cms_read(k).reshape(-1).copy() for k in KS]
        tops = [agg.topk(k, 100) for k in KS]
        tk = []
        for k in KS:  # fa_topk(k = 100) on the full sets: wall time per call (the first call above paid the buffers)
            for _ in range(3):
                t0 = time.perf_counter()
                agg.topk(k, 100)
                tk.append((time.perf_counter() - t0) * 1e3)
        out["topk100_ms_per_call"] = [round(v, 3) for v in tk]
        out["addresses_held"] = [int(len(agg.topk(k, 1 << 30))) for k in KS] if (cand or not args.timing_only) else None
        rows = agg.read_window()
    assert args.no_assert or (int(rows["count"].sum()) == n and st1["records_ok"] == n and st1["records_bad"] == 0)
    launches = st1["kernel_launches"] - st0["kernel_launches"]
    path_s = (st1["batch_ns_total"] - st0["batch_ns_total"]) * 1e-9
    out.update({
        "records": n, "wire_bytes": wire, "launches": int(launches),
        "path_ms_per_launch": path_s / launches * 1e3,
        "records_per_s_device_path": n / path_s,
        "roofline_frac_path": wire / path_s / 8e12,
        "ingest_wall_s_with_sync_per_chunk": t_ing, "generator_wall_s": t_gen,
        "records_direct_path": int(st1["records_direct"]), "flows_5m_rows": int(len(rows)),
    })
    if series:
        out["path_ms_series"] = series
        out["path_ms_last_third_mean"] = float(np.mean(series[-max(len(series) // 3, 1):]))
    if args.timing_only:
        out["checks"] = "skipped (--timing-only)"
        print(json.dumps(out))
        return
    # ---- CPU side (oracle): the same stream, sketches + exact weights of the prefix
    t0 = time.perf_counter()
    words = depth << wl2
    c_src = np.zeros(words, dtype=np.uint64)
    c_dst = np.zeros(words, dtype=np.uint64)
    ex_src = np.zeros(2 << L, dtype=np.uint64)
    ex_dst = np.zeros(2 << L, dtype=np.uint64)
    npre = out.get("prefix_records", 0)
    if npre:
        po.cms_stream(gp, 0, npre, threads, depth, wl2, seed, c_src, c_dst, ex_src, ex_dst)
    p_src, p_dst = c_src.copy(), c_dst.copy()
    po.cms_stream(gp, npre, n - npre, threads, depth, wl2, seed, c_src, c_dst)
    out["cpu_oracle_seconds"] = time.perf_counter() - t0
    out["cpu_oracle_threads"] = threads
    out["sketch_bit_exact_full_stream"] = bool(np.array_equal(full_sk[0], c_src) and np.array_equal(full_sk[1], c_dst))
    if npre:
        out["sketch_bit_exact_prefix"] = bool(np.array_equal(prefix_sk[0], p_src) and np.array_equal(prefix_sk[1], p_dst))
    # ---- top-100: GPU ranking vs the ranking of the whole universe by the CPU sketch
    ok_top = True
    for dst, (cms, top) in enumerate(zip((c_src, c_dst), tops)):
        lo, hi = universe_keys(L, dst)
        est = estimates(cms, lo, hi, depth, wl2, seed)
        # order: weight descending, then key bytes ascending (memcmp order = big-endian value of (lo, hi) byte strings)
        cand = np.argpartition(est, len(est) - 400)[-400:]
        # (two ranks may share a 4-byte IPv4 form - 2^24 ranks into 32 bits: one address, listed once)
        uniq = {}
        for i in cand:
            uniq[lo[i].tobytes() + hi[i].tobytes()] = int(est[i])
        want = sorted(uniq.items(), key=lambda kv: (-kv[1], kv[0]))[:100]
        got = [(bytes(r["key"]), int(r["weight"])) for r in top]
        ok_top = ok_top and got == want
        if got != want:
            k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
            out["top100_first_difference_%s" % ("dst" if dst else "src")] = [k, got[k][0].hex() if k >= 0 else None, got[k][1] if k >= 0 else None,
                                                                            want[k][0].hex() if k >= 0 else None, want[k][1] if k >= 0 else None, len(got), len(want)]
        if dst == 0:
            out["top3_src"] = [(k.hex(), w) for k, w in got[:3]]
    out["top100_equals_ranking_of_the_whole_universe"] = bool(ok_top)
    # ---- error bound on the prefix (overestimate-only; eps = e / width, confidence 1 - e^-depth)
    if npre:
        for dst, (cms, ex) in enumerate(zip((p_src, p_dst), (ex_src, ex_dst))):
            lo, hi = universe_keys(L, dst)
            seen = np.nonzero(ex)[0]
            est = estimates(cms, lo[seen], hi[seen], depth, wl2, seed)
            total_w = int(ex.sum(dtype=np.uint64))
            eps_w = math.e / (1 << wl2) * total_w
            never_below = bool((est >= ex[seen]).all())
            within = float(((est - ex[seen]).astype(np.float64) <= eps_w).mean())
            tag = "dst" if dst else "src"
            out["prefix_%s_addresses" % tag] = int(len(seen))
            out["prefix_%s_never_underestimates" % tag] = never_below
            out["prefix_%s_share_within_eps" % tag] = within
            out["prefix_%s_required_share" % tag] = 1 - math.exp(-depth)
    print(json.dumps(out))
    ok = out["sketch_bit_exact_full_stream"] and out.get("sketch_bit_exact_prefix", True) and ok_top
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
