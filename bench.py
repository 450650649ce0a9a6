#!/usr/bin/env python3
"""bench.py - FlowMessages/s aggregated into flows_5m on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (decode -> project -> (SrcAS,DstAS) 5-minute
rollup) over one batch of synthetic input that is already resident in HBM:
BASELINE config 2 = 100 M mocker-shaped framed FlowMessages, 64 k SrcAS/DstAS
pairs, 3 five-minute windows, 2 ETypes, per GPU (weak scaling: every rank is one
Kafka partition with its own 100 M records).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the contract in the task description) with
`roofline` (wire bytes / tile-kernel time vs 8 TB/s HBM peak, hipEvent-timed on
the library's own stream) and `cpu_baseline` (the C oracle on this box's cores
on a bounded sample of the same workload; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per launch from rocprofv3 PMC passes of THIS command line (tools/profile.sh; PMC counters
# cannot be read from inside the process).  Only quoted when the workload is the default one profiled.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01_traffic.json")


def pmc_traffic(default_workload, kernel_key="tile_kernel"):
    if not default_workload or not os.path.exists(TRAFFIC_FILE):
        return None, None
    with open(TRAFFIC_FILE) as f:
        t = json.load(f)
    k = t.get("kernels", {})
    return k.get(kernel_key, {}).get("traffic_bytes"), k


def mix64(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def rows_checksum(rows):
    """Same order-independent checksum the oracle's bench helper computes."""
    with np.errstate(over="ignore"):
        a = (rows["timeslot"].astype(np.uint64) << np.uint64(32)) | rows["etype"].astype(np.uint64)
        b = (rows["src_as"].astype(np.uint64) << np.uint64(32)) | rows["dst_as"].astype(np.uint64)
        h = mix64(a ^ mix64(b))
        v = rows["bytes"] * np.uint64(3) + rows["packets"] * np.uint64(5) + rows["count"] * np.uint64(7) + np.uint64(1)
        return int((h * v).sum(dtype=np.uint64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=int, default=100_000_000, help="records per GPU per step")
    ap.add_argument("--chunk", type=int, default=16_666_667, help="records per ingest launch (<= 2^24)")
    ap.add_argument("--mode", default="aspairs", choices=["mocker", "aspairs", "zipf"])
    ap.add_argument("--cpu-sample", type=int, default=16_000_000, help="records timed on the CPU oracle (0 = skip)")
    ap.add_argument("--key-sets", type=int, default=1, help="fa key_sets mask (must include 1 = flows_5m rollup); 9 = config 5's "
                    "two concurrent key sets; side measurements only - the default is the BASELINE metric")
    ap.add_argument("--zipf-s", type=int, default=110, help="zipf exponent x100 for --mode zipf")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--no-assert", action="store_true", help="ablation runs (FA_DEBUG_FLAGS): skip result checks")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run --nproc-per-node N")
    # FA_BENCH_BACKEND=gloo + FA_BENCH_SHARE_GPU=1: dry run of the N>1 code path on a 1-GPU box (all ranks on
    # device 0, exchange over gloo) - for testing the harness only, never a reported configuration
    backend = os.environ.get("FA_BENCH_BACKEND", "nccl")
    if os.environ.get("FA_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where the window-close exchange tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    fa = _pkg.load()
    fa.build()
    mode = {"mocker": fa.MOCK_MOCKER, "aspairs": fa.MOCK_ASPAIRS, "zipf": fa.MOCK_ZIPF}[args.mode]
    n_rec = args.records
    # every rank = one Kafka partition with its own stream (seed 2 = config 2, + rank)
    mp = fa.mock_params(mode=mode, framed=1, seed=2 + rank, n_total=n_rec, span_secs=900, per_sec=400_000,
                        zipf_s_x100=args.zipf_s)
    assert args.key_sets & fa.FA_KEYS_AS_PAIR, "--key-sets must include the flows_5m rollup"

    agg = fa.FlowAgg(device=local_rank, framed=True, table_capacity_log2=20, key_sets=args.key_sets,
                     max_batch_records=args.chunk, wide_capacity_log2=26 if args.key_sets & 8 else 0,
                     topk_capacity_log2=25 if args.key_sets & 6 else 0)  # distinct addresses: 2^24 in the zipf / aspairs generators
    chunks = []
    wire_bytes = 0
    i0 = 0
    while i0 < n_rec:
        m = min(args.chunk, n_rec - i0)
        cap = m * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(m + 1, dtype=torch.int32, device=dev)
        w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
        chunks.append((d_buf, d_off, w, m))
        wire_bytes += w
        i0 += m

    def step():
        for d_buf, d_off, w, m in chunks:
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)

    def fence():
        agg.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    fence()
    st0 = agg.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st1 = agg.stats()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    launches = st1["kernel_launches"] - st0["kernel_launches"]
    kern_s = (st1["kernel_ns_total"] - st0["kernel_ns_total"]) * 1e-9
    batch_s = (st1["batch_ns_total"] - st0["batch_ns_total"]) * 1e-9  # tile + retry + exotic + agg kernels
    bytes_per_launch = wire_bytes / len(chunks)
    avg_launch_s = kern_s / max(launches, 1)
    avg_batch_s = batch_s / max(launches, 1)
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    achieved_batch = bytes_per_launch / avg_batch_s / 1e9 if avg_batch_s > 0 else 0.0

    # window close across ranks (the only exchange step): gather + merge flows_5m rows
    t_merge = time.perf_counter()
    if world > 1:
        merged = fa.dist.close_window_merged(agg, fa.ALL_TIMESLOTS, device=xdev)
    else:
        merged = agg.close_window(fa.ALL_TIMESLOTS)
    topk_rows = None
    if args.key_sets & fa.FA_KEYS_SRCADDR_CMS:
        # BASELINE configs[3] shape: per-GPU sketches, RCCL all-reduce at window close, top-k over the union of the
        # ranks' candidates (dist.topk_merged); single rank: the local ranking
        if world > 1 and backend == "nccl":
            topk_rows = fa.dist.topk_merged(agg, fa.FA_KEYS_SRCADDR_CMS, 100, candidates_per_rank=1000, device=xdev)
        else:
            topk_rows = agg.topk(fa.FA_KEYS_SRCADDR_CMS, 100)
    merge_ms = (time.perf_counter() - t_merge) * 1e3
    total_steps = args.warmup + args.steps
    ok_total = int(merged["count"].sum())
    expect = n_rec * total_steps * world
    assert args.no_assert or ok_total == expect, "merged count() %d != records ingested %d" % (ok_total, expect)

    value = n_rec * args.steps * world / elapsed
    traffic, traffic_all = pmc_traffic(args.records == 100_000_000 and args.chunk == 16_666_667 and args.mode == "aspairs"
                                       and not os.environ.get("FA_DEBUG_FLAGS") and args.key_sets == 1,
                                       "wtile_kernel" if st1["wave_tile_launches"] > st0["wave_tile_launches"] else "tile_kernel")
    out = {
        "metric": "FlowMessages/sec aggregated into flows_5m",
        "value": value,
        "unit": "FlowMessages/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: 1xMI355X per rank, %d mocker-shaped framed FlowMessages, "
                        "64k SrcAS/DstAS pairs x 2 ETypes x 3 five-minute windows, sum(Bytes,Packets)+count() group-by"
                        % n_rec,
            "records_per_gpu_per_step": n_rec,
            "wire_bytes_per_gpu_per_step": wire_bytes,
            "bytes_per_record": wire_bytes / n_rec,
            "generator": args.mode,
            "key_sets": args.key_sets,
            "launches_per_step": len(chunks),
            "partitioning": "one Kafka partition per GPU, no data-path collective; rows all-gathered at window close",
            "window_close_merge_ms": merge_ms,
            "groups": int(len(merged)),
            "topk_src_addr_rows": None if topk_rows is None else int(len(topk_rows)),
            "records_direct_path": int(st1["records_direct"] - st0["records_direct"]),
            "records_second_chance_parser": int(st1["records_retried"] - st0["records_retried"]),
            "wire_GBps_whole_job": wire_bytes * args.steps * world / elapsed / 1e9,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": "profiles/r01_traffic.json: rocprofv3 --pmc, 2*FETCH_SIZE (gfx950 wide-read "
                              "correction) + WRITE_SIZE, bytes per launch" if traffic else None,
            "kernel": ("fa::wtile_kernel<%s>" if st1["wave_tile_launches"] > st0["wave_tile_launches"] else "fa::tile_kernel<MODE_INGEST, %s>")
                      % ("AS_PAIR" if args.key_sets == 1 else "KS_ALL"),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "avg_launch_ms": avg_launch_s * 1e3,
            "launches_timed": int(launches),
            # every kernel of a batch (tile + second-chance parsers + tuple aggregation), same bytes
            "all_kernels_avg_ms": avg_batch_s * 1e3,
            "all_kernels_achieved": achieved_batch,
            "all_kernels_frac": achieved_batch / HBM_PEAK_GBS,
            "agg_kernel_traffic": (traffic_all or {}).get("agg_kernel", {}).get("traffic_bytes"),
        },
    }

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        po = _pkg.load_oracle()
        sample = min(args.cpu_sample, n_rec)
        gp = po.gen_params(mode=mode, framed=1, seed=2, n_total=n_rec, span_secs=900, per_sec=400_000)
        cores = os.cpu_count() or 1
        res = po.bench_rollup(gp, 0, sample, cores)
        one = po.bench_rollup(gp, 0, min(sample, 2_000_000), 1)  # the same code on ONE core (scalar port), for scale
        out["cpu_baseline"] = {
            "value": sample / res["seconds"],
            "unit": "FlowMessages/s",
            "cores": cores,
            "kind": "port",
            "sample": "first %d records of the same workload (%.2f GB wire), C oracle restatement "
                      "(decode+project+hash rollup), one shard per thread + pairwise tree merge of the shard tables; "
                      "the Go inserter + ClickHouse cannot run in this image" % (sample, res["wire_bytes"] / 1e9),
            "seconds": res["seconds"],
            "single_core_value": min(sample, 2_000_000) / one["seconds"],
        }
        if not args.no_verify:
            # parity on the same sample: GPU rows checksum == oracle rows checksum
            d_buf, d_off, w, m = chunks[0]
            s = min(sample, m)
            check = fa.FlowAgg(device=local_rank, framed=True, table_capacity_log2=20, max_batch_records=args.chunk)
            nbytes = int(d_off[s].item())
            check.ingest_device(d_buf.data_ptr(), nbytes, d_off.data_ptr(), s)
            rows = check.read_window()
            check.close()
            if s == sample:
                out["parity_sample_ok"] = bool(rows_checksum(rows) == res["checksum"] and res["bad"] == 0)
                assert out["parity_sample_ok"], "GPU rows differ from the oracle on the CPU sample"
    if rank == 0:
        print(json.dumps(out))
    agg.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
