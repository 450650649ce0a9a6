#!/usr/bin/env python3
"""bench.py - FlowMessages/s aggregated into flows_5m on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (decode -> project -> (SrcAS,DstAS) 5-minute
rollup) over one batch of synthetic input that is already resident in HBM:
BASELINE config 2 = 100 M mocker-shaped framed FlowMessages, 64 k SrcAS/DstAS
pairs, 3 five-minute windows, 2 ETypes, per GPU (weak scaling: every rank is one
Kafka partition with its own 100 M records).

    python bench.py --gpus N --steps K --warmup W      (N > 1 without a torchrun environment: starts its own ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the contract in the task description):
  roofline      wire bytes / hipEvent time of the WHOLE decode+aggregate path of a launch (ingest kernel +
                second-chance parsers + tuple aggregation; SURVEY.md 8(d): t_kernel = decode + aggregate) vs the
                8 TB/s HBM peak; the ingest kernel alone is reported beside it as `dominant_kernel`;
  cpu_baseline  the C oracle on this box's usable CPUs (affinity mask cut by the cgroup quota), thread sweep with
                per-thread decode times, best run; N=1 only;
  parity        every record of the step verified against the oracle (per-chunk row checksums), outside the
                timed region, on every rank; N>1: also the rows merged across ranks at window close against the
                oracle's rollup of all partitions;
  host_fed      the PCIe-inclusive rate through fa_ingest (host buffers) - a secondary figure, never `value`.
  ms_per_step_all / ms_per_step_median / settle
                the headline's K steps are timed in ONE bracket (the contract: `value`, `ms_per_step`); before it an untimed
                settle phase runs steps until two consecutive ones agree within 1 % (a fresh box starts at sclk 155 MHz), and
                behind it the same K steps run once more, each with its own fence: per-step times and their median.
  secondary     (N = 1, default workload) the other BASELINE configurations and the consumer on the SAME driver clock, each with
                its own parity booleans against the oracle and `ok`: config3_exact / config3_candidates (configs[2], 200 M
                records), config5 (configs[4] single-GPU shape, 100 M records: ingest fraction AND close time of one run),
                group8 (8 contexts in one process through fa_group_*), host_consume (the C++ consumer, 8 partition logs of
                config 2's stream).  --no-secondary skips them; they never change metric / value / config.
  group_preflight
                (N > 1) rank 0 creates one small ctx on EVERY visible device and runs the in-process group close (fa_group_*:
                peer copies AND ncclCommInitAll) on known partitions against the oracle - reported, never fatal.
Other workloads (side measurements, their JSON goes to profiles/): --mode zipf --key-sets 7 (config 3 shape),
--key-sets 9 (config 5 shape), --mode goflow / reversed (67-field producer / order-free parser), --stage decode
(projection only).
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# HBM bytes per launch from rocprofv3 PMC passes of THIS command line (tools/profile.sh; PMC counters cannot be
# read from inside the process).  Quoted only for the default workload AND when the file was produced by the very
# sources this process runs (source_hash stamp).
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r06_traffic.json")
SOA_BYTES_PER_RECORD = 5 * 8 + 7 * 4 + 3 * 16 + 1  # fa_columns: 15 columns + status byte


def pmc_traffic(fa, default_workload):
    """-> (path traffic bytes per launch, per-kernel dict, note)"""
    if not default_workload:
        return None, None, "not the profiled workload"
    if not os.path.exists(TRAFFIC_FILE):
        return None, None, "no PMC profile committed"
    with open(TRAFFIC_FILE) as f:
        t = json.load(f)
    if t.get("source_hash") != fa.source_hash():
        return None, None, "profiles/r06_traffic.json was measured on other sources (%s != %s)" % (t.get("source_hash"), fa.source_hash())
    k = t.get("kernels", {})
    total = sum(v.get("traffic_bytes", 0.0) for v in k.values())
    return total or None, k, None


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


def effective_cpus():
    """CPUs this process can really use: the affinity mask, cut by a cgroup CPU quota when there is one (a
    barrier-synchronised job with more threads than the quota allows only measures the scheduler)."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:  # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    if quota is None:
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0 and per > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = aff if quota is None else max(1, min(aff, int(quota)))
    return eff, aff, quota


def cpu_baseline(po, gp, per_thread, n_rec, groups_hint, reps=3):
    """The C oracle (decode + 15-column projection + hash rollup, one shard per thread, merge by key partition) on
    the CPUs this process may really use (affinity mask cut by the cgroup quota).  Thread sweep with the SAME number of
    records per thread at every point (`per_thread`, 6 M by default: 96 M records and >= 0.3 s of timed region at 16
    threads - round 3 timed 0.08 s and the figure moved by 20 % from run to run), every point run `reps` times and
    reported as the MEDIAN; shard tables are sized from the group count, allocated and first-touched before the start
    barrier, and sit on 2 MiB pages."""
    cores, aff, quota = effective_cpus()
    sweep, detail = {}, {}
    points = sorted({1, 8, 32, max(1, cores // 2), cores} & set(range(1, cores + 1)) | {1})
    for th in points:
        n = min(n_rec, per_thread * th)
        runs = []
        for _ in range(reps):
            r = po.bench_rollup_ex(gp, 0, n, th, groups_hint)
            assert r["bad"] == 0
            runs.append(r)
        runs.sort(key=lambda r: r["seconds"])
        r = runs[len(runs) // 2]
        sweep[th] = n / r["seconds"]
        detail[th] = {"records": n, "seconds": r["seconds"], "seconds_all_runs": [x["seconds"] for x in runs],
                      "decode_seconds_min": r["decode_seconds_min"], "decode_seconds_max": r["decode_seconds_max"],
                      "decode_seconds_mean": r["decode_seconds_mean"], "merge_seconds": r["merge_seconds"], "wire_bytes": r["wire_bytes"],
                      "_run": r}
    th = max(sweep, key=lambda k: sweep[k])
    rate, n_used, r = sweep[th], detail[th]["records"], detail[th]["_run"]
    for d in detail.values():
        d.pop("_run")
    single = sweep[1]
    eff = rate / (th * single) if single > 0 else 0.0
    spread = max(detail[th]["seconds_all_runs"]) / min(detail[th]["seconds_all_runs"]) - 1.0
    # what bounds the best point: the slowest shard's decode + rollup, or the merge behind it
    limiter = ("per-shard decode + hash rollup (slowest thread %.3f s of %.3f s; the rest is the key-partitioned merge)"
               % (r["decode_seconds_max"], r["seconds"]))
    if th > 1 and eff < 0.7:
        limiter += ("; %.2f of linear scaling: threads share the memory system - per-thread decode time grows from %.0f ns "
                    "per record at 1 thread to %.0f ns at %d" % (eff, 1e9 * detail[1]["decode_seconds_max"] / detail[1]["records"],
                                                                 1e9 * r["decode_seconds_max"] / (n_used / th), th))
    out = {
        "value": rate,
        "unit": "FlowMessages/s",
        "cores": th,
        "cores_available": cores,
        "cores_affinity": aff,
        "cgroup_cpu_quota": quota,
        "kind": "port",
        "sample": "first %d records of the same workload (%.2f GB wire; %d per thread at every point of the sweep), C oracle "
                  "restatement (generic protobuf walk + 15-column projection + open-addressing rollup), one shard per thread, "
                  "shard tables sized from the group count and first-touched before the start barrier (2 MiB pages), merged by "
                  "key partition; median of %d runs per point, best point of the sweep; the Go inserter + ClickHouse cannot run "
                  "in this image" % (n_used, r["wire_bytes"] / 1e9, per_thread, reps),
        "seconds": r["seconds"],
        "runs_per_point": reps,
        "run_to_run_spread_at_best_point": spread,
        "thread_sweep_records_per_s": {str(k): v for k, v in sorted(sweep.items())},
        "thread_sweep_detail": {str(k): v for k, v in sorted(detail.items())},
        "single_core_value": single,
        "scaling_efficiency_vs_single_core": eff,
        "limiter": limiter,
    }
    return out


def gpu_clocks(device):
    """sclk / mclk of `device` as rocm-smi reports them right now (MHz; None when rocm-smi is missing or says nothing): the
    ingest kernel follows the core clock (same box, a minute apart: 7 %), so the line records what it ran at."""
    import re
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        js = json.loads(r.stdout[r.stdout.index("{"):]) if "{" in r.stdout else {}
    except (OSError, ValueError, subprocess.TimeoutExpired):
        return None
    out = {}
    for card in js.values():
        if not isinstance(card, dict):
            continue
        for k, v in card.items():
            m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
            for name in ("sclk", "mclk", "fclk", "socclk"):
                if name in k.lower() and m:
                    out[name + "_mhz"] = int(m.group(1))
    return out or None


def preflight(fa, torch, dist, rank, world, local_rank, backend, xdev, strict=False):
    """world > 1, before anything is timed: the three exchanges of a window close on tiny KNOWN data - the uneven all-gather of
    device row buffers, the all-to-all with split sizes, the in-library ncclAllReduce through a communicator made here with
    ncclCommInitRank - each checked against numpy.  The first run on an 8-GPU node can then tell a failing collective from a
    failing kernel: a mismatch or an exception here names the exchange, with the tail of this rank's NCCL_DEBUG=WARN log."""
    import ctypes as C
    t0 = time.perf_counter()
    res = {"ok": False}
    import socket
    dbg = os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", socket.gethostname()).replace("%p", str(os.getpid()))
    try:
        dev = torch.device("cuda", local_rank)
        ids = [torch.zeros(2, dtype=torch.int64, device=xdev) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([local_rank, rank], dtype=torch.int64, device=xdev))
        res["device_ids"] = [int(t[0].item()) for t in ids]
        res["rccl_ranks_seen"] = len({int(t[1].item()) for t in ids})
        res["backend"] = backend
        rb = fa.ROW5M_DTYPE.itemsize

        def rows_for(src, n, dst_as):
            r = np.zeros(n, dtype=fa.ROW5M_DTYPE)
            r["date"], r["timeslot"] = fa.T0 // 86400, fa.T0
            r["src_as"], r["dst_as"], r["etype"] = np.arange(n), dst_as, 0x800
            r["bytes"], r["packets"], r["count"] = src + 1, 2 * (src + 1), 1
            return r

        with fa.FlowAgg(device=local_rank, framed=True, key_sets=7, cms_width_log2=12, topk_capacity_log2=12) as pf:
            # 1. uneven all-gather: rank r brings r + 1 rows (keys 0..r) -> key i is held by the ranks r >= i
            mine = torch.from_numpy(rows_for(rank, rank + 1, 7).view(np.uint8).copy()).to(dev)
            buf, total = fa.dist.allgather_device_rows(mine.data_ptr(), rank + 1, rb)
            ptr, m = pf.rows_merge_device(fa.ROWS_5M, buf.data_ptr(), total)
            got = pf.rows_fetch(fa.ROWS_5M, ptr, m)
            want = rows_for(0, world, 7)
            for i in range(world):
                held = np.arange(i, world)
                want["bytes"][i], want["packets"][i], want["count"][i] = (held + 1).sum(), 2 * (held + 1).sum(), len(held)
            res["allgather_uneven"] = bool(total == world * (world + 1) // 2 and got.tobytes() == want.tobytes())
            # 2. all-to-all with split sizes: rank s sends s + d + 1 rows to rank d (dst_as = d names the receiver)
            counts = [rank + d + 1 for d in range(world)]
            send = np.concatenate([rows_for(rank, rank + d + 1, d) for d in range(world)])
            sbuf = torch.from_numpy(send.view(np.uint8).copy()).to(dev)
            rbuf, rtot = fa.dist.alltoall_device_rows(sbuf.data_ptr(), counts, rb)
            recv = np.frombuffer(rbuf.cpu().numpy().tobytes(), dtype=fa.ROW5M_DTYPE)
            res["all_to_all_split"] = bool(rtot == sum(s + rank + 1 for s in range(world)) and
                                           recv.tobytes() == np.concatenate([rows_for(s, s + rank + 1, rank) for s in range(world)]).tobytes())
            # 3. fa_merge_allreduce (ncclAllReduce inside libflowagg) through a communicator of our own: every rank's sketch of its
            #    own tiny stream -> merged view == the sum of all ranks' sketches (gathered over torch.distributed, summed in numpy)
            mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=900 + rank, n_total=4096, zipf_log2_universe=10)
            hb, ho = fa.mock_generate_host(mp, 0, 4096)
            pf.ingest(hb, ho)
            own = pf.cms_read(fa.FA_KEYS_SRCADDR_CMS).reshape(-1).astype(np.int64)
            parts = [torch.zeros(own.size, dtype=torch.int64, device=xdev) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(own).to(xdev))
            want_sk = sum(p.cpu().numpy().astype(np.uint64) for p in parts)
            if backend == "nccl":
                rccl = None
                for name in ("librccl.so", "librccl.so.1", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"):
                    try:
                        rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
                        break
                    except OSError:
                        pass
                assert rccl is not None, "librccl.so not found"

                class UniqueId(C.Structure):
                    _fields_ = [("internal", C.c_char * 128)]
                uid = UniqueId()
                if rank == 0:
                    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0, "ncclGetUniqueId failed"
                t = torch.from_numpy(np.frombuffer(bytes(uid), dtype=np.uint8).copy()).to(xdev)
                dist.broadcast(t, src=0)
                C.memmove(C.byref(uid), t.cpu().numpy().tobytes(), 128)
                comm = C.c_void_p()
                rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
                assert rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0, "ncclCommInitRank failed"
                pf.merge_allreduce(comm.value)
                got_sk = pf.cms_read(fa.FA_KEYS_SRCADDR_CMS).reshape(-1)
                rccl.ncclCommDestroy.argtypes = [C.c_void_p]
                rccl.ncclCommDestroy(comm)
                res["in_library_allreduce"] = bool(np.array_equal(got_sk, want_sk))
            else:
                fa.dist.allreduce_sketches(pf)  # (ranks share a GPU: RCCL cannot put two ranks on one device - the gloo twin)
                res["in_library_allreduce"] = "skipped (backend %s: ranks share a GPU); torch.distributed twin: %s" % (
                    backend, bool(np.array_equal(pf.cms_read(fa.FA_KEYS_SRCADDR_CMS).reshape(-1), want_sk)))
        flags = [res["allgather_uneven"], res["all_to_all_split"], res["in_library_allreduce"] is True or
                 (isinstance(res["in_library_allreduce"], str) and res["in_library_allreduce"].endswith("True"))]
        t = torch.tensor([1 if all(flags) else 0], dtype=torch.int64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        res["ok"] = bool(int(t.item()))
    except Exception as e:  # noqa: BLE001 - whatever failed, say which exchange and show the RCCL log
        res["error"] = "%s: %s" % (type(e).__name__, e)
    res["seconds"] = time.perf_counter() - t0
    if not res["ok"]:
        tail = ""
        if dbg and os.path.exists(dbg):
            with open(dbg, errors="replace") as f:
                tail = "".join(f.readlines()[-40:])
        sys.stderr.write("bench.py preflight FAILED on rank %d: %s\n--- NCCL_DEBUG=WARN tail (%s) ---\n%s\n" % (rank, json.dumps(res), dbg or "stderr", tail))
        if strict:
            raise SystemExit(3)
        # (not fatal by default: the line then carries preflight.ok = false with the exchange that failed, and the run's own
        # parity leg - merged rows against the oracle's rollup of all partitions - still decides whether the result stands)
    return res


# ---------------------------------------------------------------------------------------------------------------------------
# secondary blocks (N = 1): the other BASELINE configurations on the driver's clock.  Every block: its own contexts, its own
# parity booleans against the oracle (oracle/ = the checker, never the thing measured), `ok` = all of them.
def _universe_keys(L, dst):
    """(lo, hi) of the Zipf generator's address for every (v6, rank): index = rank + (v6 << L)  (csrc/gen.cuh gen_zipf_key)."""
    rank = np.arange(1 << L, dtype=np.uint64)
    with np.errstate(over="ignore"):
        a = mix64(rank * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x2222 if dst else 0x1111))
        b = mix64(a ^ np.uint64(0xD1B54A32D192ED03))
    lo = np.concatenate([a & np.uint64(0xFFFFFFFF), a])   # v4: the low 4 bytes of a, rest zero; v6: a || b
    hi = np.concatenate([np.zeros(1 << L, dtype=np.uint64), b])
    return lo, hi


def _sketch_columns(a, h1, wl2, row):
    pbits = min(8, wl2 - 4)
    sub = wl2 - pbits
    with np.errstate(over="ignore"):
        prefix = (h1 & np.uint64((1 << pbits) - 1)).astype(np.int64)
        l1 = (h1 >> np.uint64(32)).astype(np.uint32)
        l2 = ((a | np.uint64(1)) >> np.uint64(32)).astype(np.uint32) | np.uint32(1)
        low = ((l1 + np.uint32(row) * l2) >> np.uint32(32 - sub)).astype(np.int64)
    return (prefix << sub) | low


def _estimates(cms, lo, hi, depth, wl2, seed):
    with np.errstate(over="ignore"):
        s0 = mix64(np.array([(seed + 0x9E3779B97F4A7C15) & (2**64 - 1)], dtype=np.uint64))[0]
        a = mix64(lo ^ s0)
        h1 = mix64(a ^ hi)
        best = np.full(len(lo), np.uint64(2**64 - 1), dtype=np.uint64)
        for r in range(depth):
            best = np.minimum(best, cms[_sketch_columns(a, h1, wl2, r) + (r << wl2)])
    return best


def _want_top100(cms, L, dst, depth, wl2, seed):
    """The first 100 rows of the ranking of EVERY address of the universe by the CPU sketch's estimate (weight DESC, key bytes)."""
    lo, hi = _universe_keys(L, dst)
    est = _estimates(cms, lo, hi, depth, wl2, seed)
    c400 = np.argpartition(est, len(est) - 400)[-400:]
    uniq = {}
    for i in c400:  # (two ranks may share a 4-byte IPv4 form: one address, listed once)
        uniq[lo[i].tobytes() + hi[i].tobytes()] = int(est[i])
    return sorted(uniq.items(), key=lambda kv: (-kv[1], kv[0]))[:100]


class _Zipf3:
    """The CPU side of BASELINE configs[2] / [3] (seed 3, Zipf-1.1 over 2^24 addresses), computed once for every block that
    ingests this stream: both Count-Min sketches of the whole stream and the top-100 of the whole universe."""

    def __init__(self, po, n, L=24, depth=4, wl2=20, seed=0x5EED):
        self.n, self.L, self.depth, self.wl2, self.seed = n, L, depth, wl2, seed
        self.threads = max(1, min(64, effective_cpus()[0]))
        t = time.perf_counter()
        gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=3, n_total=n, span_secs=1800, zipf_log2_universe=L, zipf_s_x100=110)
        self.gp = gp
        self.c_src = np.zeros(depth << wl2, dtype=np.uint64)
        self.c_dst = np.zeros(depth << wl2, dtype=np.uint64)
        po.cms_stream(gp, 0, n, self.threads, depth, wl2, seed, self.c_src, self.c_dst)
        self.want = [_want_top100(c, L, d, depth, wl2, seed) for d, c in enumerate((self.c_src, self.c_dst))]
        self.seconds = time.perf_counter() - t

    def tops_ok(self, tops):
        return all([(bytes(r["key"]), int(r["weight"])) for r in top] == want for top, want in zip(tops, self.want))


def sec_config3(fa, po, torch, dev, ref, cand, chunk=33_333_334):
    """BASELINE configs[2]: Count-Min heavy hitters over SrcAddr / DstAddr beside the flows_5m rollup (key_sets 7), `ref.n` framed
    records regenerated in HBM chunk by chunk; both sketches bit-exact against the CPU sketch of the whole stream, top-100 == the
    ranking of the whole universe, count() of the rollup == records (viz-ch.json:233,479).
    Launches of 33.3 M records like the headline's (compact tuples: up to 2^25 - 1 per launch).  Until round 6 this block ran 16.67 M
    per launch, the wide tuples' limit of round 2; what a launch costs whatever its size - the flush of both sketches, the launch
    boundary's two scans in the candidates mode, the kernels' tails - then weighs twice as much: same box, 3 x, exact 0.134 - 0.135 at
    16.67 M against 0.145 - 0.151 at 33.3 M, candidates 0.157 - 0.162 against 0.173 - 0.174 (profiles/r06_exp_config3_launch_size.jsonl)."""
    n, L = ref.n, ref.L
    KS = (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=3, n_total=n, span_secs=1800, zipf_log2_universe=L, zipf_s_x100=110)
    out = {"workload": "BASELINE configs[2]: %d framed FlowMessages, Zipf-1.1 over 2^%d addresses, key sets flows_5m + both Count-Min sketches "
                       "(%d x 2^%d x u64), top-k mode %s" % (n, L, ref.depth, ref.wl2, "candidates" if cand else "exact")}
    t_all = time.perf_counter()
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=ref.depth, cms_width_log2=ref.wl2, cms_seed=ref.seed, topk_capacity_log2=16 if cand else L + 2,
                    max_batch_records=chunk, topk_mode=fa.TOPK_CANDIDATES if cand else fa.TOPK_EXACT) as agg:
        cap = chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(chunk + 1, dtype=torch.int32, device=dev)
        wire = 0
        series, prev = [], 0
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            agg.sync()  # (the one generator buffer is reused)
            ns = agg.stats()["batch_ns_total"]
            series.append((ns - prev) * 1e-6)
            prev = ns
            wire += w
        st = agg.stats()
        sk = [agg.cms_read(k).reshape(-1) for k in KS]
        tops = [agg.topk(k, 100) for k in KS]  # (the first read pays the buffers)
        tk = []
        for k in KS:
            for _ in range(3):
                t = time.perf_counter()
                agg.topk(k, 100)
                tk.append((time.perf_counter() - t) * 1e3)
        rows = agg.read_window()
    launches = int(st["kernel_launches"])
    path_s = st["batch_ns_total"] * 1e-9
    steady = series[len(series) // 2:]  # (the sets fill up during the first launches)
    out.update({
        "records": n, "wire_bytes": wire, "launches": launches, "records_per_launch": min(chunk, n),
        "path_ms_per_launch": path_s / launches * 1e3, "frac": wire / path_s / (HBM_PEAK_GBS * 1e9),
        "path_ms_per_launch_second_half": float(np.mean(steady)), "frac_second_half": wire / launches / (float(np.mean(steady)) * 1e-3) / (HBM_PEAK_GBS * 1e9),
        "records_per_s_device_path": n / path_s, "path_ms_series": [round(x, 4) for x in series],
        "topk100_ms_per_call": [round(v, 3) for v in tk], "topk100_ms_median": float(np.median(tk)),
        "sketches_bit_exact": bool(np.array_equal(sk[0], ref.c_src) and np.array_equal(sk[1], ref.c_dst)),
        "top100_equals_ranking_of_the_whole_universe": bool(ref.tops_ok(tops)),
        "rollup_count_equals_records": bool(int(rows["count"].sum()) == n and st["records_ok"] == n and st["records_bad"] == 0),
    })
    out["ok"] = bool(out["sketches_bit_exact"] and out["top100_equals_ranking_of_the_whole_universe"] and out["rollup_count_equals_records"])
    out["seconds"] = time.perf_counter() - t_all
    return out


def sec_config5(fa, po, torch, dev, n=100_000_000, chunk=16_666_667, span=1800):
    """BASELINE configs[4], single-GPU shape: flows_5m + (SrcAddr,DstPort,Proto) over 60-s sub-buckets, Zipf-0.8, seed 5.  Ingest
    fraction AND close times of the SAME run (48-byte rows into a page-locked buffer - what the consumer keeps).  Parity: every
    aligned flows_5m window against the C oracle's rollup (checksum of keys and sums, row count); one SLIDING window byte for
    byte; every (SrcAddr,DstPort,Proto) window: strictly ascending keys (every key once) and the linear checksum of its rows ==
    the oracle's over the window's records (oracle/flow_oracle.h fo_app_checksum_stream), count() == the window's records."""
    threads = max(1, min(64, effective_cpus()[0]))
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=5, n_total=n, span_secs=span, zipf_log2_universe=24, zipf_s_x100=80)
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=5, n_total=n, span_secs=span, zipf_log2_universe=24, zipf_s_x100=80)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    out = {"workload": "BASELINE configs[4], single-GPU shape: (SrcAS,DstAS) + (SrcAddr,DstPort,Proto), 60-s sub-buckets, 5-min windows, %d framed "
                       "FlowMessages, Zipf-0.8, seed 5, %d s of event time" % (n, span)}
    t_all = time.perf_counter()
    with fa.FlowAgg(framed=True, key_sets=ks, window_secs=300, subwindow_secs=60, wide_capacity_log2=26, table_capacity_log2=24, max_batch_records=chunk) as agg:
        cap = chunk * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(chunk + 1, dtype=torch.int32, device=dev)
        wire = 0
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            agg.sync()
            wire += w
        st = agg.stats()
        del d_buf, d_off
        launches = int(st["kernel_launches"])
        path_s = st["batch_ns_total"] * 1e-9
        out.update({"records": n, "wire_bytes": wire, "launches": launches, "path_ms_per_launch": path_s / launches * 1e3,
                    "frac": wire / path_s / (HBM_PEAK_GBS * 1e9), "records_per_s_device_path": n / path_s})
        ok_ing = st["records_ok"] == n and st["records_bad"] == 0
        aligned = [fa.T0 + 300 * k for k in range((span + 299) // 300)]
        wins = [agg.read_window(ts) for ts in aligned]
        allrows = np.concatenate(wins)
        ref = po.bench_rollup(gp, 0, n, threads)
        out["flows_5m_rows"] = int(len(allrows))
        out["flows_5m_aligned_windows_bit_exact"] = bool(ref["bad"] == 0 and ref["groups"] == len(allrows) and rows_checksum(allrows) == ref["checksum"]
                                                       and int(allrows["count"].sum()) == n and ok_ing)
        # one sliding window [t0 + 420, t0 + 720): byte for byte against the oracle's rollup of exactly its records
        start = fa.T0 + 420
        got = agg.read_window(start)
        ia = -(-(start - fa.T0) * n // span)
        ib = -(-(start + 300 - fa.T0) * n // span)
        part = po.bench_rollup_ex(gp, ia, ib - ia, threads, want_rows=True)["rows"]  # rows per ALIGNED timeslot: folded into the window below
        key = np.stack([part[c].astype(np.uint64) for c in ("src_as", "dst_as", "etype")], axis=1)
        order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
        part, key = part[order], key[order]
        first = np.ones(len(part), dtype=bool)
        first[1:] = (key[1:] != key[:-1]).any(axis=1)
        starts = np.nonzero(first)[0]
        folded = part[starts].copy()
        with np.errstate(over="ignore"):
            for c in ("bytes", "packets", "count"):
                folded[c] = np.add.reduceat(part[c], starts)
        folded["timeslot"], folded["date"] = start, start // 86400
        out["sliding_window_rows"] = int(len(got))
        out["sliding_window_bit_exact"] = bool(got.tobytes() == folded.tobytes())
        # (SrcAddr,DstPort,Proto): read every aligned window, then close them oldest first - 48-byte rows, page-locked buffer
        want_sum, want_cnt, outside = po.app_checksum_stream(gp, 0, n, threads, 300, fa.T0, len(aligned))
        nreuse = int(st["wide_used"] + st["wide_log_records"]) // max(len(aligned) - 1, 1) + (1 << 20)
        reuse = fa.FlowAgg.pinned_rows(fa.ROWS_APP, nreuse).view(np.uint8)[:nreuse * 48].view(fa.ROW_APP48_DTYPE)
        read_ms, close_ms, app_ok, nrows = [], [], outside == 0, 0
        for i, ts in enumerate(aligned):
            t = time.perf_counter()
            app, date = agg.read_window_app48(ts, out=reuse)
            read_ms.append((time.perf_counter() - t) * 1e3)
            nrows += len(app)
            app_ok = app_ok and date == ts // 86400 and int(app["count"].sum()) == int(want_cnt[i]) and po.app_rows_checksum(app, timeslot=ts) == int(want_sum[i]) \
                and po.app_rows_strictly_ascending(app)
        sums = []
        for i, ts in enumerate(aligned):
            t = time.perf_counter()
            app, _ = agg.read_window_app48(ts, out=reuse)
            agg.drop_range(fa.ROWS_APP, ts, ts + 300)  # (a tumbling consumer: the window's five sub-buckets in one pass)
            close_ms.append((time.perf_counter() - t) * 1e3)
            sums.append((len(app), int(app["count"].sum())))
            app_ok = app_ok and int(app["count"].sum()) == int(want_cnt[i])
        left = len(agg.read_window_app())
    out.update({"app_rows": nrows, "app_windows_bit_exact_by_linear_checksum_and_strict_order": bool(app_ok),
                "read_app_window_ms": [round(x, 2) for x in read_ms], "close_app_window_ms": [round(x, 2) for x in close_ms],
                "close_ms_median": float(np.median(close_ms)), "close_ms_first": close_ms[0],
                "row_format": "fa_row_app48 into a page-locked buffer (one copy-engine transfer per half window)",
                "app_rows_left_after_all_closes": int(left)})
    out["ok"] = bool(out["flows_5m_aligned_windows_bit_exact"] and out["sliding_window_bit_exact"] and app_ok and left == 0)
    out["seconds"] = time.perf_counter() - t_all
    return out


def sec_group8(fa, po, torch, dev, ref, members=8, chunk=8_333_334, cand=True):
    """8 contexts in ONE process (the reference consumer's shape: a goroutine per claimed partition, inserter.go:167-196) on this
    box's one GPU, each its partition of the configs[2]/[3] stream with flows_5m + both sketches + (SrcAddr,DstPort,Proto); the
    windows of the whole topic closed through fa_group_*: flows_5m merged, (SrcAddr,DstPort,Proto) hash-partitioned, sketches
    all-reduced, top-100.  What one GPU cannot show: xGMI (the peer copies are device copies here)."""
    n, nm, L = ref.n, members, ref.L
    threads = ref.threads
    mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=3, n_total=n, span_secs=1800, zipf_log2_universe=L, zipf_s_x100=110)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_SRCADDR_CMS | fa.FA_KEYS_DSTADDR_CMS | fa.FA_KEYS_ADDR_PORT_PROTO
    out = {"workload": "BASELINE configs[3] + [4] shapes through fa_group_* on one GPU: %d contexts in one process, %d-record Zipf-1.1 stream (seed 3, 1800 s), "
                       "key sets flows_5m + both sketches + (SrcAddr,DstPort,Proto); top-k mode %s" % (nm, n, "candidates" if cand else "exact")}
    t_all = time.perf_counter()
    kw = dict(framed=True, key_sets=ks, cms_depth=ref.depth, cms_width_log2=ref.wl2, cms_seed=ref.seed, max_batch_records=chunk, wide_capacity_log2=24, table_capacity_log2=22,
              topk_capacity_log2=16 if cand else L + 1, topk_mode=fa.TOPK_CANDIDATES if cand else fa.TOPK_EXACT)
    ms = [fa.FlowAgg(**kw) for _ in range(nm)]
    try:
        with fa.FlowGroup(ms) as g:  # (the group exists while its members ingest: they reserve their exchange buffers as their rows come into being)
            cap = chunk * 96 + 4096
            d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_off = torch.empty(chunk + 1, dtype=torch.int32, device=dev)
            wire = 0
            for c, i0 in enumerate(range(0, n, chunk)):  # chunk c belongs to partition c % members (every partition spans the whole time range)
                m = min(chunk, n - i0)
                agg = ms[c % nm]
                w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
                agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
                agg.sync()
                wire += w
            del d_buf, d_off
            st = g.stats()
            ok_ing = st["records_ok"] == n and st["records_bad"] == 0
            out.update({"records": n, "wire_bytes": wire, "members": nm, "device_path_ms_sum_over_members": st["batch_ns_total"] * 1e-6,
                        "frac_ingest_sum_over_members": wire / (st["batch_ns_total"] * 1e-9) / (HBM_PEAK_GBS * 1e9)})
            slots = [int(t) for t in g.open_timeslots()]
            t = time.perf_counter()
            wins = [g.read_window(fa.ROWS_5M, ts, cap=1 << 20) for ts in slots]
            out["read_5m_window_merged_ms_mean"] = 1e3 * (time.perf_counter() - t) / max(len(slots), 1)
            allrows = np.concatenate(wins)
            # (SrcAddr,DstPort,Proto), hash-partitioned: first every window TIMED, back to back into one reused buffer (the checks
            # below are seconds of numpy per window - with them in between every read would start from an idle GPU at 160 MHz),
            # then every window once more for the checks
            import ctypes as C
            nslot = (slots[-1] - slots[0]) // 300 + 1
            want_sum, want_cnt, outside = po.app_checksum_stream(ref.gp, 0, n, threads, 300, slots[0], nslot)
            buf_rows = int(want_cnt.max()) + (1 << 16)  # (rows <= records of the window)
            reuse = fa.FlowAgg.pinned_rows(fa.ROWS_APP, buf_rows)  # (page-locked, like config 5's: a consumer keeps its row buffer; a pageable one costs 5 - 30 ms more per window, box by box)
            out["row_buffer"] = "page-locked"
            shares_c = (C.c_size_t * nm)()
            nr = C.c_size_t()
            app_ms = []
            for ts in slots:
                t = time.perf_counter()
                rc = g._L.fa_group_read_window_partitioned(g._h, fa.ROWS_APP, ts, reuse.ctypes.data, buf_rows, shares_c, C.byref(nr))
                app_ms.append((time.perf_counter() - t) * 1e3)
                g._chk(rc)
            app_ok, nrows = outside == 0, 0
            for ts in slots:
                rows, shares = g.read_window_partitioned(fa.ROWS_APP, ts, cap=buf_rows)
                nrows += len(rows)
                k = (ts - slots[0]) // 300
                bounds = np.cumsum([0] + shares)
                samp = np.arange(len(rows))[::211]
                owner = fa.dist.partition_rows_host(rows[::211], fa.ROWS_APP, nm)
                app_ok = app_ok and int(rows["count"].sum()) == int(want_cnt[k]) and po.app_rows_checksum(rows) == int(want_sum[k]) \
                    and all(po.app_rows_strictly_ascending(rows[bounds[r]:bounds[r + 1]]) for r in range(nm)) \
                    and bool((np.searchsorted(bounds, samp, side="right") - 1 == owner).all())
                del rows
            out["read_app_window_partitioned_ms"] = [round(x, 1) for x in app_ms]
            out["read_app_window_partitioned_ms_median"] = float(np.median(app_ms))
            out["read_app_first_over_median"] = app_ms[0] / float(np.median(app_ms))
            t = time.perf_counter()
            g.allreduce_sketches()
            out["allreduce_both_sketches_ms"] = 1e3 * (time.perf_counter() - t)
            sk = [ms[nm - 1].cms_read(k).reshape(-1) for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]  # any member: the merged view
            t = time.perf_counter()
            tops = [g.topk(k, 100) for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]
            out["topk100_both_sketches_first_ms"] = 1e3 * (time.perf_counter() - t)
            t = time.perf_counter()
            tops2 = [g.topk(k, 100) for k in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)]
            out["topk100_both_sketches_again_ms"] = 1e3 * (time.perf_counter() - t)
            t = time.perf_counter()
            closed = g.close_window(fa.ROWS_5M, slots[0], cap=1 << 20)
            closed_app, _ = g.close_window_partitioned(fa.ROWS_APP, slots[0], cap=buf_rows)
            out["close_oldest_window_both_key_sets_ms"] = 1e3 * (time.perf_counter() - t)
            left = g.read_window(fa.ROWS_5M, cap=1 << 22)
            out["table_bytes_per_member"] = {"distinct_address_sets": 2 * 32 << (16 if cand else L + 1), "sketches": 2 * 8 * ref.depth << ref.wl2}
    finally:
        for m in ms:
            m.close()
    r5 = po.bench_rollup_ex(ref.gp, 0, n, threads, groups_hint=len(allrows))
    out["flows_5m_rows"], out["app_rows"] = int(len(allrows)), nrows
    out["flows_5m_merged_equals_oracle_rollup_of_all_partitions"] = bool(r5["bad"] == 0 and r5["groups"] == len(allrows) and r5["checksum"] == rows_checksum(allrows)
                                                                           and int(allrows["count"].sum()) == n and ok_ing)
    out["closed_window_equals_its_read"] = bool(closed.tobytes() == wins[0].tobytes() and int(left["count"].sum()) == n - int(closed["count"].sum())
                                                and int(closed_app["count"].sum()) == int(want_cnt[0]))
    out["app_windows_bit_exact_by_linear_checksum_and_strict_order"] = bool(app_ok)
    out["merged_sketches_bit_exact"] = bool(np.array_equal(sk[0], ref.c_src) and np.array_equal(sk[1], ref.c_dst))
    out["top100_equals_ranking_of_the_whole_universe"] = bool(ref.tops_ok(tops) and tops[0].tobytes() == tops2[0].tobytes() and tops[1].tobytes() == tops2[1].tobytes())
    out["ok"] = bool(out["flows_5m_merged_equals_oracle_rollup_of_all_partitions"] and out["closed_window_equals_its_read"] and app_ok
                     and out["merged_sketches_bit_exact"] and out["top100_equals_ranking_of_the_whole_universe"])
    out["seconds"] = time.perf_counter() - t_all
    return out


def sec_host_consume(fa, po, torch, dev, n=64_000_000, nparts=8, flush_count=262144, extra=()):
    """The C++ consumer (flow-pipeline_amd/host/inserter_gpu: the reference's ConsumeClaim / flush / MarkMessage shape,
    inserter.go:113-196, one thread per claimed partition, ONE process) on BASELINE config 2's stream: 8 partition logs, key set
    flows_5m.  `value` = records / the consume phase (every partition thread from its first message to its last flush; log
    mapping, context setup and the last window close are reported beside it).  Parity: its RowBinary output decoded == the C
    oracle's rollup of ALL partitions (row count, checksum of keys and sums), insert_count == records."""
    import shutil
    import subprocess
    import tempfile
    host = os.path.join(ROOT, "flow-pipeline_amd", "host")
    exe = os.path.join(host, "inserter_gpu")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    out = {"workload": "inserter_gpu (C++ host, one process, %d partition threads) on BASELINE configs[1]'s stream: %d framed FlowMessages in %d partition logs, "
                       "key set flows_5m, -flush.count %d" % (nparts, n, nparts, flush_count)}
    t_all = time.perf_counter()
    mp = fa.mock_params(mode=fa.MOCK_ASPAIRS, framed=1, seed=2, n_total=n, span_secs=900, per_sec=400_000)
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=2, n_total=n, span_secs=900, per_sec=400_000)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > n * 96 + (1 << 30) else "/tmp"
    tmp = tempfile.mkdtemp(prefix="fa_bench_logs_", dir=base)
    try:
        # partition p = records [p * n / parts, (p + 1) * n / parts): generated in HBM (the product's generator), written as the log
        # of message values back to back (-proto.fixedlen=true: every value carries its length prefix, mocker.go:98-101)
        per = n // nparts
        paths = []
        wire = 0
        t = time.perf_counter()
        with fa.FlowAgg(device=dev.index or 0, framed=True) as gen:
            cap = per * 96 + 4096
            d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_off = torch.empty(per + 1, dtype=torch.int32, device=dev)
            for p in range(nparts):
                m = per if p < nparts - 1 else n - per * (nparts - 1)
                w = gen.mock_generate_device(mp, p * per, m, d_buf.data_ptr(), cap, d_off.data_ptr())
                path = os.path.join(tmp, "p%d.log" % p)
                d_buf[:w].cpu().numpy().tofile(path)
                paths.append(path)
                wire += w
            del d_buf, d_off
        out["generate_logs_s"] = time.perf_counter() - t
        out["logs_dir"] = base
        rb, met, ph = (os.path.join(tmp, x) for x in ("flows_5m.rowbinary", "metrics.txt", "phases.json"))
        t = time.perf_counter()
        r = subprocess.run([exe, "-input.files=" + ",".join(paths), "-flush.count=%d" % flush_count, "-flush.dur=1h", "-key.sets=1", "-out.rowbinary=" + rb,
                            "-metrics.dump=" + met, "-phases.out=" + ph, "-gpu.devices=1", "-gpu.table.log2=20", "-loglevel=info"] + list(extra), capture_output=True, text=True)
        out["host_wall_s"] = time.perf_counter() - t
        if r.returncode != 0:
            out.update({"ok": False, "error": r.stderr[-1500:]})
            return out
        phases = json.load(open(ph))
        metrics = {l.split()[0]: int(l.split()[1]) for l in open(met) if not l.startswith("#")}
        rows = fa.rowbinary_to_rows_fast(np.fromfile(rb, dtype=np.uint8))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    threads = max(1, min(64, effective_cpus()[0]))
    ref = po.bench_rollup_ex(gp, 0, n, threads, groups_hint=len(rows))
    parts = phases["partitions"]
    out.update({
        "records": n, "wire_bytes": wire, "value": n / phases["consume_s"], "unit": "FlowMessages/s (consume phase)",
        "wire_GBps_consume": wire / phases["consume_s"] / 1e9, "consume_s": phases["consume_s"], "setup_s": phases["setup_s"], "finish_s": phases["finish_s"],
        "records_per_s_wall_of_the_process": n / out["host_wall_s"],
        "phases_mean_per_partition_thread_s": {k: float(np.mean([p[k] for p in parts])) for k in ("take_s", "fa_ingest_s", "lock_wait_s", "mark_s", "close_s",
                                                                                                   "lib_stage_wait_s", "lib_stage_copy_s", "device_path_s")},
        "take_ns_per_record": float(np.mean([p["take_s"] / max(p["records"], 1) for p in parts])) * 1e9,
        "batches": int(sum(p["batches"] for p in parts)), "copied_bytes": int(sum(p["copied_bytes"] for p in parts)),
        "insert_count_equals_records": bool(metrics.get("insert_count") == n),
        "rowbinary_equals_oracle_rollup_of_all_partitions": bool(ref["bad"] == 0 and ref["groups"] == len(rows) and ref["checksum"] == rows_checksum(rows)
                                                                     and int(rows["count"].sum()) == n),
        "flows_5m_rows": int(len(rows)), "phases_by_partition": parts,
    })
    out["ok"] = bool(out["insert_count_equals_records"] and out["rowbinary_equals_oracle_rollup_of_all_partitions"])
    out["seconds"] = time.perf_counter() - t_all
    return out


def run_secondary(fa, po, torch, dev, args):
    """Every block on its own: a failing block reports {"ok": false, "error": ...}, the others still run."""
    import traceback
    sec = {}
    t0 = time.perf_counter()

    def guard(name, fn, *a, **kw):
        if time.perf_counter() - t0 > args.secondary_budget:
            sec[name] = {"ok": False, "skipped": "the secondary blocks' time budget (%d s) was spent" % args.secondary_budget}
            return
        try:
            sec[name] = fn(*a, **kw)
        except Exception as e:  # noqa: BLE001 - reported in the line, never fatal for the headline
            sec[name] = {"ok": False, "error": "%s: %s" % (type(e).__name__, e), "traceback": traceback.format_exc()[-1200:]}
        torch.cuda.empty_cache()

    guard("host_consume", sec_host_consume, fa, po, torch, dev, n=args.secondary_host_records)
    guard("config5", sec_config5, fa, po, torch, dev, n=args.secondary_config5_records)
    ref = None
    try:
        ref = _Zipf3(po, args.secondary_config3_records)
        sec["zipf_stream_cpu_side_seconds"] = ref.seconds
    except Exception as e:  # noqa: BLE001
        sec["zipf_stream_cpu_side_error"] = "%s: %s" % (type(e).__name__, e)
    if ref is not None:
        guard("config3_exact", sec_config3, fa, po, torch, dev, ref, False)
        guard("config3_candidates", sec_config3, fa, po, torch, dev, ref, True)
        guard("group8", sec_group8, fa, po, torch, dev, ref)
    else:
        for k in ("config3_exact", "config3_candidates", "group8"):
            sec[k] = {"ok": False, "error": "the CPU side of the Zipf stream failed"}
    sec["seconds"] = time.perf_counter() - t0
    sec["ok"] = all(sec[k].get("ok") for k in ("config3_exact", "config3_candidates", "config5", "group8", "host_consume"))
    return sec


def group_preflight_body():
    """Runs in a process of its own (`bench.py --group-preflight-only`, started by rank 0 at N > 1 with a timeout): ONE process, a
    small ctx on EVERY visible device - the shape of the consumer the reference defines (inserter.go:167-196, :176; shard unit
    compose/docker-compose-clickhouse-mock.yml:18) - known partitions in, and the in-process window close of the whole topic
    through fa_group_* with both transports (peer copies over xGMI; ncclCommInitAll + grouped ncclAllReduce) against the oracle:
    all-reduced sketches, top-k, flows_5m rows merged, (SrcAddr,DstPort,Proto) rows hash-partitioned, a real close."""
    import torch
    fa = _pkg.load()
    po = _pkg.load_oracle()
    ndev = torch.cuda.device_count()
    res = {"devices": ndev, "ok": False}
    res["can_access_peer"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(ndev)] for i in range(ndev)]
    n, L, depth, wl2, seed = 30_000 * max(ndev, 1), 12, 4, 14, 0x5EED
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=77, n_total=n, zipf_log2_universe=L, span_secs=600)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)
    parts = []
    for p in range(ndev):  # record i belongs to Kafka partition i % devices
        recs = [raw[int(off[k]):int(off[k + 1])] for k in range(p, n, ndev)]
        o = np.zeros(len(recs) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(r) for r in recs])
        parts.append((np.frombuffer(b"".join(recs), dtype=np.uint8), o))
    rows, status = po.decode_batch(buf, off, 1)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    want5 = ref.rows()
    want_app = po.rollup_app(rows, status)
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    want_sk, want_top = {}, {}
    for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
        sk = po.cms_sketch_numpy(rows[col], w, depth, wl2, seed)
        keys = np.unique(np.ascontiguousarray(rows[col]), axis=0)
        est = po.cms_estimates_numpy(sk, keys, depth, wl2, seed)
        want_sk[ks] = sk.reshape(-1)
        want_top[ks] = sorted(zip((-est.astype(object)).tolist(), [bytes(k) for k in keys]))[:50]

    def sorted_app(r):
        addr = np.ascontiguousarray(r["src_addr"])
        hi = addr[:, :8].copy().view(">u8").reshape(-1)
        lo = addr[:, 8:].copy().view(">u8").reshape(-1)
        return r[np.lexsort((r["proto"], r["dst_port"], lo, hi, r["timeslot"], r["date"]))]

    all_ok = True
    for name, transport in (("peer", fa.GROUP_PEER), ("rccl", fa.GROUP_RCCL)):
        t0 = time.perf_counter()
        r = {}
        members = []
        try:
            members = [fa.FlowAgg(device=d, framed=True, key_sets=15, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=L + 3) for d in range(ndev)]
            for m, (b, o) in zip(members, parts):
                m.ingest(b, o)
            with fa.FlowGroup(members, transport) as g:
                r["transport"] = "rccl" if g.transport == fa.GROUP_RCCL else "peer"
                g.allreduce_sketches()
                r["allreduce_sketches"] = bool(all(np.array_equal(m.cms_read(ks).reshape(-1), want_sk[ks]) for m in members for ks in want_sk))
                r["topk"] = bool(all([(bytes(x["key"]), int(x["weight"])) for x in g.topk(ks, 50)] == [(k, -e) for e, k in want_top[ks]] for ks in want_top))
                r["read_window_5m"] = bool(g.read_window(fa.ROWS_5M).tobytes() == want5.tobytes())
                got, shares = g.read_window_partitioned(fa.ROWS_APP)
                owner = fa.dist.partition_rows_host(got, fa.ROWS_APP, ndev)
                r["read_window_app_partitioned"] = bool(sorted_app(got).tobytes() == want_app.tobytes() and owner.tolist() == np.repeat(np.arange(ndev), shares).tolist())
                ts0 = int(g.open_timeslots()[0])
                r["close_window_5m"] = bool(g.close_window(fa.ROWS_5M, ts0).tobytes() == want5[want5["timeslot"] == ts0].tobytes()
                                            and g.read_window(fa.ROWS_5M).tobytes() == want5[want5["timeslot"] != ts0].tobytes())
                got, _ = g.close_window_partitioned(fa.ROWS_APP, ts0)
                r["close_window_app_partitioned"] = bool(sorted_app(got).tobytes() == want_app[want_app["timeslot"] == ts0].tobytes()
                                                         and sorted_app(g.read_window_partitioned(fa.ROWS_APP)[0]).tobytes() == want_app[want_app["timeslot"] != ts0].tobytes())
            r["ok"] = all(v for k, v in r.items() if k != "transport")
        except Exception as e:  # noqa: BLE001 - reported: which transport, which step
            r["ok"] = False
            r["error"] = "%s: %s" % (type(e).__name__, e)
            if name == "rccl" and ndev < 2 and "UNSUPPORTED" in str(e):
                r["ok"] = None  # (one device: FA_GROUP_RCCL needs a GPU per member - nothing to check here)
        finally:
            for m in members:
                m.close()
        r["seconds"] = time.perf_counter() - t0
        res[name] = r
        all_ok = all_ok and r["ok"] is not False
    res["ok"] = bool(all_ok)
    print("GROUP_PREFLIGHT " + json.dumps(res))


def group_preflight(timeout_s=240):
    """Rank 0, N > 1: the in-process group close on every visible device, in a process of its own (a hang or a crash of the first
    multi-device run of fa_group_* must not take the bench with it).  Reported in the line, never fatal."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    if os.environ.get("FA_BENCH_SHARE_GPU"):
        env["FA_GROUP_PREFLIGHT_NOTE"] = "dry run on a shared GPU"
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--group-preflight-only"], env=env, capture_output=True, text=True, timeout=timeout_s)
        for line in r.stdout.splitlines():
            if line.startswith("GROUP_PREFLIGHT "):
                res = json.loads(line[len("GROUP_PREFLIGHT "):])
                break
        else:
            res = {"ok": False, "error": "no result (exit code %d)" % r.returncode, "stderr_tail": r.stderr[-1500:]}
    except subprocess.TimeoutExpired as e:
        res = {"ok": False, "error": "timed out after %d s (killed)" % timeout_s, "stderr_tail": (e.stderr or b"")[-1500:].decode(errors="replace") if isinstance(e.stderr, bytes) else str(e.stderr)[-1500:]}
    res["seconds_with_process_start"] = time.perf_counter() - t0
    return res


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (the driver's N=1
    command shape with another N must not die at argument parsing).  One rank per GPU over RCCL; on a box with fewer
    GPUs FA_BENCH_SHARE_GPU=1 puts every rank on device 0 and moves the window-close exchange to gloo - a dry run of
    the N>1 code path, never a reported configuration."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        if not env.get("FA_BENCH_SHARE_GPU"):
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible (FA_BENCH_SHARE_GPU=1 runs every rank on "
                             "device 0 over gloo: harness dry run only)" % (args.gpus, ndev))
        env.setdefault("FA_BENCH_BACKEND", "gloo")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--records", type=int, default=100_000_000, help="records per GPU per step")
    ap.add_argument("--chunk", type=int, default=33_333_334, help="records per ingest launch (<= 2^25 - 1: compact tuples; launches of wide tuples are split at 2^24)")
    ap.add_argument("--mode", default="aspairs", choices=["mocker", "aspairs", "zipf", "goflow", "reversed", "distinct"])
    ap.add_argument("--stage", default="ingest", choices=["ingest", "decode"],
                    help="decode: the projection stage alone (wire bytes -> 15 SoA columns in HBM, fa_decode_device)")
    ap.add_argument("--cpu-sample", type=int, default=6_000_000, help="records PER THREAD timed on the CPU oracle at every point of the thread sweep (0 = skip)")
    ap.add_argument("--key-sets", type=int, default=1, help="fa key_sets mask (must include 1 = flows_5m rollup); 9 = config 5's "
                    "two concurrent key sets; side measurements only - the default is the BASELINE metric")
    ap.add_argument("--zipf-s", type=int, default=110, help="zipf exponent x100 for --mode zipf")
    ap.add_argument("--zipf-universe-log2", type=int, default=24, help="address universe of --mode zipf (side measurements)")
    ap.add_argument("--no-verify", action="store_true", help="skip the full-step parity check against the oracle")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the PCIe-inclusive fa_ingest measurement")
    ap.add_argument("--no-assert", action="store_true", help="ablation runs (FA_DEBUG_FLAGS): skip result checks")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --records is the WHOLE job's step, split evenly over the ranks "
                    "(default: weak - every rank is a Kafka partition with --records of its own)")
    ap.add_argument("--strict-preflight", action="store_true", help="world > 1: a failing preflight aborts the run (default: reported in the line, the run goes on)")
    ap.add_argument("--no-preflight", action="store_true", help="world > 1: skip the check of the three window-close exchanges on known data")
    ap.add_argument("--group-preflight-only", action="store_true", help="(internal) run the in-process group close on every visible device and print its result")
    ap.add_argument("--no-group-preflight", action="store_true", help="world > 1: skip rank 0's in-process fa_group_* check on every visible device")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the secondary blocks (configs 3 and 5, the 8-context group close, the C++ consumer)")
    ap.add_argument("--secondary-budget", type=int, default=240, help="seconds after which no further secondary block is started")
    ap.add_argument("--secondary-config3-records", type=int, default=200_000_000)
    ap.add_argument("--secondary-config5-records", type=int, default=100_000_000)
    ap.add_argument("--secondary-host-records", type=int, default=64_000_000)
    ap.add_argument("--settle-max-steps", type=int, default=400, help="untimed settle phase: at most this many steps (it ends when two consecutive steps agree within 1 %%)")
    args = ap.parse_args()
    if args.group_preflight_only:
        group_preflight_body()
        return

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)  # (does not return)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # FA_BENCH_BACKEND=gloo + FA_BENCH_SHARE_GPU=1: dry run of the N>1 code path on a 1-GPU box (all ranks on
    # device 0, exchange over gloo) - for testing the harness only, never a reported configuration
    backend = os.environ.get("FA_BENCH_BACKEND", "gloo" if os.environ.get("FA_BENCH_SHARE_GPU") else "nccl")
    if os.environ.get("FA_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where the window-close exchange tensors live
    gpre = None
    if world > 1 and rank == 0 and not args.no_group_preflight:
        # the product's OTHER multi-GPU form - one process, a ctx per device, fa_group_* - has never run on more than one device:
        # rank 0 runs it on every visible device in a process of its own BEFORE it joins the process group (the other ranks wait in
        # init_process_group's rendezvous, on the CPU - not inside a collective that spins on their GPUs)
        gpre = group_preflight()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")  # (quiet unless something is wrong; the preflight shows its tail on failure)
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/fa_bench_rccl.%h.%p.log")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(minutes=30))

    fa = _pkg.load()
    fa.build()
    mode = {"mocker": fa.MOCK_MOCKER, "aspairs": fa.MOCK_ASPAIRS, "zipf": fa.MOCK_ZIPF, "goflow": fa.MOCK_GOFLOW,
            "reversed": fa.MOCK_REVERSED, "distinct": fa.MOCK_DISTINCT}[args.mode]
    n_rec = args.records // world if args.strong else args.records
    clocks = {"start": gpu_clocks(local_rank)} if rank == 0 else None
    pre = None
    if world > 1 and not args.no_preflight:
        pre = preflight(fa, torch, dist, rank, world, local_rank, backend, xdev, strict=args.strict_preflight)
    # every rank = one Kafka partition with its own stream (seed 2 = config 2, + rank)
    mp = fa.mock_params(mode=mode, framed=1, seed=2 + rank, n_total=n_rec, span_secs=900, per_sec=400_000,
                        zipf_s_x100=args.zipf_s, zipf_log2_universe=args.zipf_universe_log2)
    assert args.key_sets & fa.FA_KEYS_AS_PAIR, "--key-sets must include the flows_5m rollup"

    def new_ctx():
        return fa.FlowAgg(device=local_rank, framed=True, table_capacity_log2=20, key_sets=args.key_sets,
                          max_batch_records=args.chunk, wide_capacity_log2=26 if args.key_sets & 8 else 0,
                          topk_capacity_log2=max(args.zipf_universe_log2 + 1, 25 if args.zipf_universe_log2 == 24 else args.zipf_universe_log2 + 2) if args.key_sets & 6 else 0)  # distinct addresses: the universe in v4 and v6 form (2^25 slots at the default universe, as in rounds 1-2)

    agg = new_ctx()
    chunks = []
    wire_bytes = 0
    i0 = 0
    rec_cap = fa.mock_record_cap(mode)
    # (device offsets are 32-bit: a chunk stays below 4 GiB of wire bytes - GoFlow-shaped records are twice mocker's)
    args.chunk = min(args.chunk, ((1 << 32) - (1 << 24)) // rec_cap)
    while i0 < n_rec:
        m = min(args.chunk, n_rec - i0)
        cap = m * rec_cap + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(m + 1, dtype=torch.int32, device=dev)
        w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), cap, d_off.data_ptr())
        chunks.append((d_buf, d_off, w, m, i0))
        wire_bytes += w
        i0 += m

    decode_stage = args.stage == "decode"

    def step():
        for d_buf, d_off, w, m, _ in chunks:
            if decode_stage:
                agg.decode_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            else:
                agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)

    def fence():
        agg.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed_step():
        t = time.perf_counter()
        step()
        fence()
        return (time.perf_counter() - t) * 1e3

    # Untimed settle phase: a fresh box idles at sclk 155 MHz and the timed region is 40 ms - steps run (each with its own
    # fence) until two consecutive ones agree within 1 % (and at least 8: the first steps also size segment buffers and learn
    # the stream), at most --settle-max-steps.  N > 1: every rank runs the same number (the slowest rank's criterion).
    settle_ms = []
    for i in range(max(args.settle_max_steps, 0)):
        settle_ms.append(timed_step())
        done = len(settle_ms) >= 8 and abs(settle_ms[-1] - settle_ms[-2]) <= 0.01 * settle_ms[-1]
        if world > 1:
            t = torch.tensor([0 if done else 1], dtype=torch.int64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            done = int(t.item()) == 0
        if done:
            break
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
    # the same K steps once more, each with its own fence (sync overhead included: ~20 us of a 2 ms step) - not `value`
    per_step_ms = [timed_step() for _ in range(args.steps)]
    extra_steps = len(settle_ms) + len(per_step_ms)
    if clocks is not None:
        clocks["end"] = gpu_clocks(local_rank)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bytes_per_launch = wire_bytes / len(chunks)
    value = n_rec * args.steps * world / elapsed
    common = {
        "metric": "FlowMessages/sec aggregated into flows_5m" if not decode_stage else "FlowMessages/sec decoded and projected into SoA columns",
        "value": value,
        "unit": "FlowMessages/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_median": float(np.median(per_step_ms)) if per_step_ms else None,
        "ms_per_step_all": [round(x, 4) for x in per_step_ms],
        "settle": {"steps": len(settle_ms), "first_ms": round(settle_ms[0], 4) if settle_ms else None, "last_ms": round(settle_ms[-1], 4) if settle_ms else None,
                   "rule": "untimed steps until two consecutive ones agree within 1 % (>= 8, <= --settle-max-steps)"},
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
    }

    if decode_stage:
        launches = st1["decode_launches"] - st0["decode_launches"]
        dec_s = (st1["decode_ns_total"] - st0["decode_ns_total"]) * 1e-9 / max(launches, 1)
        alg = bytes_per_launch + SOA_BYTES_PER_RECORD * (n_rec / len(chunks))
        out = dict(common)
        out["config"] = {
            "workload": "projection stage of BASELINE configs[1]: %d framed FlowMessages -> 15 SoA columns + status in HBM "
                        "(fa_decode_device; tile_kernel<MODE_DECODE>), generator %s" % (n_rec, args.mode),
            "records_per_gpu_per_step": n_rec, "wire_bytes_per_gpu_per_step": wire_bytes, "launches_per_step": len(chunks),
            "soa_bytes_per_record": SOA_BYTES_PER_RECORD,
        }
        out["roofline"] = {
            "bound": "hbm", "achieved": alg / dec_s / 1e9 if dec_s else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / dec_s / 1e9 / HBM_PEAK_GBS if dec_s else 0.0, "traffic": None,
            "kernel": "fa::tile_kernel<MODE_DECODE, 1> + fa::deferred_kernel<MODE_DECODE, 1>",
            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_model": "wire bytes read once + %d B of columns written per record (SURVEY.md 8(d))" % SOA_BYTES_PER_RECORD,
            "avg_launch_ms": dec_s * 1e3, "launches_timed": int(launches),
        }
        if rank == 0:
            print(json.dumps(out))
        agg.close()
        if world > 1:
            dist.destroy_process_group()
        return

    launches = st1["kernel_launches"] - st0["kernel_launches"]
    # (the library splits a call into several launches when its records leave as wide tuples: > 2^24 per launch)
    bytes_per_launch = (st1["bytes_in"] - st0["bytes_in"]) / max(launches, 1)
    kern_s = (st1["kernel_ns_total"] - st0["kernel_ns_total"]) * 1e-9
    batch_s = (st1["batch_ns_total"] - st0["batch_ns_total"]) * 1e-9  # ingest kernel + second-chance parsers + tuple aggregation
    avg_launch_s = kern_s / max(launches, 1)
    avg_batch_s = batch_s / max(launches, 1)
    achieved_kernel = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    achieved_path = bytes_per_launch / avg_batch_s / 1e9 if avg_batch_s > 0 else 0.0
    wave = st1["wave_tile_launches"] > st0["wave_tile_launches"]
    compact = st1["compact_tuple_launches"] - st0["compact_tuple_launches"]

    # window close across ranks (the only exchange step): gather + merge flows_5m rows
    t_merge = time.perf_counter()
    if world > 1:
        merged = fa.dist.close_window_merged(agg, fa.ALL_TIMESLOTS, device=xdev)
    else:
        merged = agg.close_window(fa.ALL_TIMESLOTS)
    merge_ms = (time.perf_counter() - t_merge) * 1e3
    topk_rows = None
    topk_note = None
    if args.key_sets & fa.FA_KEYS_SRCADDR_CMS:
        # BASELINE configs[3] shape: per-GPU sketches, RCCL all-reduce at window close, top-k over the union of EVERY
        # rank's distinct addresses (exact w.r.t. the merged sketch); single rank: the local ranking
        if world > 1:
            topk_rows = fa.dist.topk_merged(agg, fa.FA_KEYS_SRCADDR_CMS, 100, candidates_per_rank=None, device=xdev)
            topk_note = "exact ranking of the merged sketch: every rank's distinct addresses are candidates"
        else:
            topk_rows = agg.topk(fa.FA_KEYS_SRCADDR_CMS, 100)
            topk_note = "exact ranking of this rank's sketch over its distinct-address set"
    total_steps = args.warmup + args.steps + extra_steps
    ok_total = int(merged["count"].sum())
    expect = n_rec * total_steps * world
    assert args.no_assert or ok_total == expect, "merged count() %d != records ingested %d" % (ok_total, expect)

    default_workload = (args.records == 100_000_000 and args.chunk == 33_333_334 and args.mode == "aspairs"
                        and not os.environ.get("FA_DEBUG_FLAGS") and args.key_sets == 1 and wave)
    traffic, traffic_all, traffic_note = pmc_traffic(fa, default_workload)
    fmt = "compact8" if compact == launches else "wide16" if compact == 0 else "mixed (%d of %d launches compact)" % (compact, launches)
    ks_name = "AS_PAIR" if args.key_sets == 1 else "KS_ALL" if args.key_sets > 7 else str(args.key_sets)
    out = dict(common)
    if clocks is not None:
        out["clocks"] = clocks
    if pre is not None:
        out["preflight"] = pre
    if gpre is not None:
        out["group_preflight"] = gpre
    out["config"] = {
        "workload": ("BASELINE configs[1]: 1xMI355X per rank, %d mocker-shaped framed FlowMessages, "
                     "64k SrcAS/DstAS pairs x 2 ETypes x 3 five-minute windows, sum(Bytes,Packets)+count() group-by" % n_rec)
                    if args.mode == "aspairs" and args.key_sets == 1 else
                    "side measurement: generator %s, key_sets %d, %d framed FlowMessages per rank" % (args.mode, args.key_sets, n_rec),
        "records_per_gpu_per_step": n_rec,
        "wire_bytes_per_gpu_per_step": wire_bytes,
        "bytes_per_record": wire_bytes / n_rec,
        "generator": args.mode,
        "key_sets": args.key_sets,
        "launches_per_step": int(launches // max(args.steps, 1)),
        "tuple_format": fmt,
        "partitioning": "one Kafka partition per GPU, no data-path collective; rows all-gathered at window close",
        "window_close_merge_ms": merge_ms,
        "groups": int(len(merged)),
        "topk_src_addr_rows": None if topk_rows is None else int(len(topk_rows)),
        "topk": topk_note,
        "records_direct_path": int(st1["records_direct"] - st0["records_direct"]),
        "records_second_chance_parser": int(st1["records_retried"] - st0["records_retried"]),
        "records_generic_parser": int(st1["records_slow"] - st0["records_slow"]),
        "wire_GBps_whole_job": wire_bytes * args.steps * world / elapsed / 1e9,
    }
    t8 = "true" if compact else "false"
    # (compact tuples are folded by agg8_kernel unless FA_AGG=generic forces the two-word-key kernel)
    agg_name = "fa::agg8_kernel" if compact and os.environ.get("FA_AGG") != "generic" else "fa::agg_kernel<%s>" % t8
    out["roofline"] = {
        "bound": "hbm",
        # t_kernel = decode + aggregate -> every kernel of a launch, first event to last
        "achieved": achieved_path,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved_path / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_source": ("profiles/r06_traffic.json (sources %s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per launch, summed over the path's kernels; "
                           "FETCH_SIZE calibrated on known byte counts (tools/micro/fetch_calib.hip): streaming reads x2, random 64-byte requests x1" % fa.source_hash()) if traffic else traffic_note,
        "kernel": ("decode+aggregate path of one launch: fa::wtile_kernel<%s, %s> + fa::deferred_kernel + %s" % (ks_name, t8, agg_name)) if wave
                  else "decode+aggregate path of one launch: fa::tile_kernel<MODE_INGEST, %s> + fa::deferred_kernel" % ks_name,
        "algorithmic_bytes_per_launch": bytes_per_launch,
        "avg_launch_ms": avg_batch_s * 1e3,
        "launches_timed": int(launches),
        "dominant_kernel": {
            "name": ("fa::wtile_kernel<%s, %s>" % (ks_name, t8)) if wave else "fa::tile_kernel<MODE_INGEST, %s>" % ks_name,
            "avg_launch_ms": avg_launch_s * 1e3,
            "achieved": achieved_kernel,
            "frac": achieved_kernel / HBM_PEAK_GBS,
            "traffic": ((traffic_all or {}).get("wtile_kernel") or {}).get("traffic_bytes"),
        },
        "traffic_by_kernel": {k: v.get("traffic_bytes") for k, v in (traffic_all or {}).items()} or None,
    }

    if world > 1:  # every rank's own path time beside rank 0's roofline block
        t = torch.tensor([avg_batch_s * 1e3, avg_launch_s * 1e3], dtype=torch.float64, device=xdev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        out["roofline"]["per_rank_path_ms"] = [float(x[0].item()) for x in allt]
        out["roofline"]["per_rank_dominant_kernel_ms"] = [float(x[1].item()) for x in allt]
        out["roofline"]["note"] = "achieved / frac are rank 0's; value is the whole job over the slowest rank's wall clock"

    if not args.no_verify and not args.no_assert:
        # parity on the WHOLE step, every rank its own partition: every chunk again through a fresh ctx, its rows against
        # the oracle's rows for the same records (order-independent checksum of (key, sums); u64 sums commute, so equal
        # chunk results = equal step).  N > 1 additionally: the rows the ranks merged at window close (RCCL all-gather +
        # device merge) against the oracle's rollup of ALL partitions (every rank's oracle rows gathered, merged with
        # numpy, sums x the number of steps ingested) - byte for byte.
        po = _pkg.load_oracle()
        gp = po.gen_params(mode=mode, framed=1, seed=2 + rank, n_total=n_rec, span_secs=900, per_sec=400_000, zipf_s_x100=args.zipf_s,
                           zipf_log2_universe=args.zipf_universe_log2)
        threads = max(1, min(effective_cpus()[0] // world, 64))
        ok = True
        verified = 0
        oracle_parts = []
        t_v = time.perf_counter()
        for d_buf, d_off, w, m, first in chunks:
            check = fa.FlowAgg(device=local_rank, framed=True, table_capacity_log2=20, max_batch_records=args.chunk)
            check.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            rows = check.read_window()
            cst = check.stats()
            check.close()
            ref = po.bench_rollup_ex(gp, first, m, threads, groups_hint=len(rows), want_rows=True)
            ok = ok and rows_checksum(rows) == ref["checksum"] and ref["bad"] == 0 and cst["records_bad"] == 0 \
                and ref["wire_bytes"] == w and ref["groups"] == len(rows)
            oracle_parts.append(ref["rows"])
            verified += m
        parity = {"ok": bool(ok), "records_verified": int(verified), "of_records_per_step": n_rec,
                  "how": "per launch: GPU flows_5m rows vs C-oracle rows of the same records (row count, wire bytes, "
                         "order-independent checksum over keys and sums)"}
        mine = fa.dist.merge_rows_host(oracle_parts)
        if world == 1:
            # the TIMED ctx itself: the rows it handed out at window close == the oracle's rollup of the step, sums x the steps
            # it ingested (warm-up included) - byte for byte (the per-launch check above runs through fresh contexts)
            want = mine.copy()
            with np.errstate(over="ignore"):
                for f in ("bytes", "packets", "count"):
                    want[f] = want[f] * np.uint64(total_steps)
            parity["timed_ctx_rows_equal_oracle"] = bool(want.tobytes() == np.ascontiguousarray(merged).tobytes())
            parity["ok"] = parity["ok"] and parity["timed_ctx_rows_equal_oracle"]
            parity["how"] += "; the timed context's own window-close rows vs the oracle's rollup x %d steps (byte-identical)" % total_steps
        if world > 1:
            want = fa.dist.merge_rows_host(fa.dist.allgather_struct(mine, fa.dist.ROW5M_DTYPE, device=xdev))
            with np.errstate(over="ignore"):
                for f in ("bytes", "packets", "count"):
                    want[f] = want[f] * np.uint64(total_steps)
            merged_ok = want.tobytes() == np.ascontiguousarray(merged).tobytes()
            flags = torch.tensor([1 if ok else 0, 1 if merged_ok else 0, verified], dtype=torch.int64, device=xdev)
            allf = [torch.zeros_like(flags) for _ in range(world)]
            dist.all_gather(allf, flags)
            parity["ok"] = all(int(f[0].item()) == 1 and int(f[1].item()) == 1 for f in allf)
            parity["ranks_ok"] = [bool(int(f[0].item())) for f in allf]
            parity["merged_rows_equal_oracle_rollup_of_all_partitions"] = [bool(int(f[1].item())) for f in allf]
            parity["records_verified"] = int(sum(int(f[2].item()) for f in allf))
            parity["of_records_per_step"] = n_rec * world
            parity["merged_rows"] = int(len(merged))
            parity["how"] += ("; N>1: the rows merged across ranks at window close vs the oracle's rollup of all %d partitions "
                              "(byte-identical rows, on every rank)" % world)
        parity["seconds"] = time.perf_counter() - t_v
        out["parity"] = parity
        assert parity["ok"], "GPU rows differ from the oracle: %r" % (parity,)

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        po = _pkg.load_oracle()
        gp = po.gen_params(mode=mode, framed=1, seed=2, n_total=n_rec, span_secs=900, per_sec=400_000, zipf_s_x100=args.zipf_s,
                           zipf_log2_universe=args.zipf_universe_log2)
        out["cpu_baseline"] = cpu_baseline(po, gp, min(args.cpu_sample, n_rec), n_rec, int(len(merged)))

    if rank == 0 and world == 1 and not args.no_host_fed and not args.no_assert:
        # the path a Kafka consumer uses: host buffers -> pinned staging -> H2D -> the same kernels (never `value`)
        nh = min(8_000_000, n_rec)
        hb, ho = fa.mock_generate_host(mp, 0, nh)
        with fa.FlowAgg(device=local_rank, framed=True, max_batch_records=1 << 22) as hagg:
            hagg.ingest(hb, ho)
            hagg.sync()
            th = time.perf_counter()
            reps = 3
            for _ in range(reps):
                hagg.ingest(hb, ho)
            hagg.sync()
            dt = (time.perf_counter() - th) / reps
            assert int(hagg.read_window()["count"].sum()) == nh * (reps + 1)
        out["host_fed"] = {"value": nh / dt, "unit": "FlowMessages/s", "wire_GBps": hb.nbytes / dt / 1e9,
                           "what": "fa_ingest from pageable host memory (%d records per call): multi-threaded copy into pinned "
                                   "staging + H2D over PCIe + kernels; link ceiling ~63 GB/s" % nh}
    agg.close()
    del chunks
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and default_workload and not (args.no_secondary or args.no_assert or args.no_verify or args.cpu_sample <= 0):  # (measurement runs of the tools skip them)
        out["secondary"] = run_secondary(fa, _pkg.load_oracle(), torch, dev, args)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
