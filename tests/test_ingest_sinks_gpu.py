"""Ingest sinks (all through the C-ABI, bit-exact against the oracle): compact / wide scatter tuples
and the adaptive switch between them, the 67-field producer on the canonical fast path, the order-free parser in
place, lossless table growth from a tiny table, the in-library RCCL merge (world 1), a 2-rank window close on
one GPU over gloo (idempotent merges), and the bench harness's N>1 path."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mix64(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def _checksum(rows):
    with np.errstate(over="ignore"):
        a = (rows["timeslot"].astype(np.uint64) << np.uint64(32)) | rows["etype"].astype(np.uint64)
        b = (rows["src_as"].astype(np.uint64) << np.uint64(32)) | rows["dst_as"].astype(np.uint64)
        h = _mix64(a ^ _mix64(b))
        v = rows["bytes"] * np.uint64(3) + rows["packets"] * np.uint64(5) + rows["count"] * np.uint64(7) + np.uint64(1)
        return int((h * v).sum(dtype=np.uint64))


def _device_batch(fa, agg, mode, seed, n, i0=0, n_total=None, **kw):
    import torch
    dev = torch.device("cuda", 0)
    mp = fa.mock_params(mode=mode, framed=1, seed=seed, n_total=n_total or n, span_secs=900, per_sec=50_000, **kw)
    cap = n * fa.mock_record_cap(mode) + 4096
    d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    w = agg.mock_generate_device(mp, i0, n, d_buf.data_ptr(), cap, d_off.data_ptr())
    return d_buf, d_off, w


@pytest.mark.parametrize("fmt", ["8", "16", ""])
@pytest.mark.parametrize("mode", [1, 0, 2])
def test_tuple_formats_agree_with_oracle(gpu_lib, fa, po, monkeypatch, fmt, mode):
    """The same 3 M-record batch through compact 8-byte tuples, wide 16-byte tuples and the adaptive default."""
    if fmt:
        monkeypatch.setenv("FA_TUPLE", fmt)
    else:
        monkeypatch.delenv("FA_TUPLE", raising=False)
    n = 3_000_000
    gp = po.gen_params(mode=mode, framed=1, seed=91 + mode, n_total=n, span_secs=900, per_sec=50_000)
    want = po.bench_rollup(gp, 0, n, 8)
    with fa.FlowAgg(framed=True, max_batch_records=n) as agg:
        d_buf, d_off, w = _device_batch(fa, agg, mode, 91 + mode, n)
        assert w == want["wire_bytes"]
        agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), n)
        agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), n)
        rows = agg.read_window()
        st = agg.stats()
    assert len(rows) == want["groups"] and st["records_ok"] == 2 * n and st["records_bad"] == 0
    for col in ("bytes", "packets", "count"):
        assert (rows[col] % np.uint64(2) == 0).all()
        rows[col] //= np.uint64(2)
    assert _checksum(rows) == want["checksum"]
    assert st["wave_tile_launches"] == 2
    assert st["compact_tuple_launches"] == (0 if fmt == "16" else 2)  # generator values fit the compact format
    assert st["records_misfit_compact"] == 0


def _enc(fa, fields):
    out = bytearray()
    for f, v in fields:
        if isinstance(v, (bytes, bytearray)):
            out += fa.schema.encode_varint((f << 3) | 2) + fa.schema.encode_varint(len(v)) + bytes(v)
        else:
            out += fa.schema.encode_varint(f << 3) + fa.schema.encode_varint(int(v))
    return bytes(out)


def _records_with_wide_values(fa, po, n, seed, frac_big):
    """Generator truth rows re-encoded; a fraction carries values only a WIDE tuple holds (32-bit ASNs, Bytes >= 2^17,
    Packets >= 2^9, unusual ETypes) and a few values no tuple holds."""
    gp = po.gen_params(mode=1, framed=0, seed=seed, n_total=n)
    rows = po.gen_rows(gp, 0, n)
    rng = np.random.default_rng(seed)
    big = rng.random(n) < frac_big
    kind = rng.integers(0, 6, n)
    recs = []
    for i in range(n):
        r = rows[i]
        sa, da, by, pk, et = int(r["src_as"]), int(r["dst_as"]), int(r["bytes"]), int(r["packets"]), int(r["etype"])
        if big[i]:
            k = int(kind[i])
            if k == 0:
                sa = 4_200_000_000 + (i % 1000)          # private 32-bit ASN
            elif k == 1:
                da = (1 << 20) + (i % 77)
            elif k == 2:
                by = (1 << 17) + i
            elif k == 3:
                pk = 512 + (i % 3000)
            elif k == 4:
                et = 0x8847                               # MPLS: not in the compact dictionary
            else:
                by = (1 << 28) + i                        # no tuple at all: direct path in either format
        alen = 16 if int(r["etype"]) == 0x86dd else 4
        fields = [(2, int(r["time_received"])), (3, 1), (4, int(r["sequence_num"])), (6, bytes(r["src_addr"][:alen])),
                  (7, bytes(r["dst_addr"][:alen])), (9, by), (10, pk), (14, sa), (15, da), (21, int(r["src_port"])),
                  (22, int(r["dst_port"])), (30, et), (38, int(r["time_flow_start"]))]
        recs.append(fa.schema.frame(_enc(fa, [(f, v) for f, v in fields if isinstance(v, bytes) or v != 0])))
    lens = np.fromiter((len(r) for r in recs), dtype=np.uint64, count=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    return np.frombuffer(b"".join(recs), dtype=np.uint8), off


@pytest.mark.parametrize("fmt", ["8", "16"])
def test_values_outside_the_compact_format_stay_exact(gpu_lib, fa, po, monkeypatch, fmt):
    monkeypatch.setenv("FA_TUPLE", fmt)
    monkeypatch.setenv("FA_SINK", "scatter")
    n = 60000
    buf, off = _records_with_wide_values(fa, po, n, 5, 0.3)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(buf, off)
        got = agg.read_window()
        st = agg.stats()
    assert got.tobytes() == ref.rows().tobytes()
    assert st["records_ok"] == n and st["wave_tile_launches"] == 1
    if fmt == "8":
        assert 0.15 * n < st["records_misfit_compact"] < 0.35 * n and st["records_direct"] >= st["records_misfit_compact"]
    else:
        assert st["records_misfit_compact"] == 0 and 0 < st["records_direct"] < 0.1 * n


def test_tuple_format_adapts_to_the_stream(gpu_lib, fa, po, monkeypatch):
    """A stream whose values do not fit compact tuples: the first launch pays (misfits take the direct path), the
    counter snapshot switches the ctx to wide tuples; results are exact throughout."""
    monkeypatch.delenv("FA_TUPLE", raising=False)
    monkeypatch.setenv("FA_SINK", "scatter")
    n = 40000
    buf, off = _records_with_wide_values(fa, po, n, 6, 0.5)
    ref = po.Rollup(300)
    with fa.FlowAgg(framed=True) as agg:
        for _ in range(6):
            agg.ingest(buf, off)
            agg.sync()
            assert ref.ingest(buf, off, 1) == 0
        got = agg.read_window()
        st = agg.stats()
    assert got.tobytes() == ref.rows().tobytes()
    assert st["wave_tile_launches"] == 6 and 1 <= st["compact_tuple_launches"] <= 2, st


@pytest.mark.parametrize("mode,name", [(3, "goflow"), (5, "reversed")])
def test_producer_shapes_rollup_and_decode(gpu_lib, fa, po, mode, name):
    """GOFLOW: the 67-field producer (MACs, VLANs, nets, NextHop ...) stays on the canonical fast path - no record
    reaches the second-chance or generic parsers.  REVERSED: every record takes the order-free parser, in place."""
    n = 400_000
    gp = po.gen_params(mode=mode, framed=1, seed=17, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(buf, off)
        got = agg.read_window()
        st = agg.stats()
    assert got.tobytes() == ref.rows().tobytes()
    assert st["records_ok"] == n and st["records_slow"] == 0 and st["wave_tile_launches"] == 1
    assert st["records_retried"] == (0 if mode == 3 else n), st
    m = 50_000
    want, wstatus = po.decode_batch(buf[:int(off[m])], off[:m + 1], 1)
    with fa.FlowAgg(framed=True) as agg:
        dec = agg.decode(buf[:int(off[m])], off[:m + 1])
    assert not wstatus.any() and not dec["status"].any()
    for col in want.dtype.names:
        if col != "_pad":
            assert np.array_equal(dec[col], want[col]), col


def test_goflow_device_generator_matches_oracle(gpu_lib, fa, po):
    import torch
    n = 100_000
    for mode in (3, 4, 5):
        gp = po.gen_params(mode=mode, framed=1, seed=23, n_total=n, span_secs=900)
        buf, off = po.gen_records(gp, 0, n)
        with fa.FlowAgg(framed=True) as agg:
            d_buf, d_off, w = _device_batch(fa, agg, mode, 23, n)
            assert w == len(buf)
            assert bytes(d_buf[:w].cpu().numpy()) == bytes(buf)
            assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), off)
        torch.cuda.synchronize()


@pytest.mark.parametrize("sync_between,cap_log2", [(True, 16), (False, 16), (False, 10)])
def test_all_distinct_groups_into_a_tiny_table_lose_nothing(gpu_lib, fa, po, sync_between, cap_log2):
    """8 M records, every one its own (SrcAS,DstAS) group, into a 2^16-slot table: the aggregation kernel's LDS
    tables overflow, the device table fills up, updates are parked and replayed after the table has grown - no
    aggregate is dropped, with or without a sync between the launches.  From 2^10 slots the table also walks
    through every region geometry (16 regions of 64 slots -> 256 regions: below 2^14 slots a region is shared by several
    partitions and the aggregation kernel keeps the atomic protocol; from there on it owns its region)."""
    n, parts = 8_000_000, 4
    m = n // parts
    gp = po.gen_params(mode=4, framed=1, seed=3, n_total=n, span_secs=900)  # (_device_batch spreads the records over 900 s)
    want = po.bench_rollup(gp, 0, n, 8)
    assert want["groups"] >= n  # (>= : a group may straddle two 5-minute windows - it does not here)
    with fa.FlowAgg(framed=True, table_capacity_log2=cap_log2, max_batch_records=m) as agg:
        bufs = [_device_batch(fa, agg, 4, 3, m, i0=k * m, n_total=n) for k in range(parts)]
        for d_buf, d_off, w in bufs:
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
            if sync_between:
                agg.sync()
        st = agg.stats()
        assert st["records_ok"] == n and st["table_used"] == want["groups"] and st["table_capacity"] >= 2 * n
        rows = agg.close_window()
    assert len(rows) == want["groups"] and int(rows["count"].sum()) == n
    assert _checksum(rows) == want["checksum"]


@pytest.mark.parametrize("path", ["scatter", "atomic"])
@pytest.mark.parametrize("depth,wl2", [(4, 20), (4, 14), (3, 12), (5, 20)])
def test_count_min_scatter_sink_is_bit_exact(gpu_lib, fa, po, monkeypatch, path, depth, wl2):
    """Both sketches through the scatter sink (tuples -> LDS bins -> cms_agg_kernel, no atomics) and through the
    memory-side atomics, at the default 32 MiB geometry, small sketches, a depth that is no power of two and one whose
    slices exceed the LDS array (5 x 2^20: atomics again): always the CPU sketch, counter for counter.  Heavy hitters
    (Zipf 1.1: one address carries 8 %% of the records) overflow their slices' segments into the atomic fallback."""
    import torch
    if path == "atomic":
        monkeypatch.setenv("FA_CMS", "atomic")
    else:
        monkeypatch.delenv("FA_CMS", raising=False)
    n = 2_000_000
    gp = po.gen_params(mode=2, framed=1, seed=33, n_total=n, zipf_log2_universe=18, zipf_s_x100=110)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    seed = 0xC0FFEE
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=20,
                    max_batch_records=n) as agg:
        d_buf, d_off, wb = _device_batch(fa, agg, 2, 33, n, zipf_log2_universe=18, zipf_s_x100=110)
        assert wb == len(buf)
        agg.ingest_device(d_buf.data_ptr(), wb, d_off.data_ptr(), n)
        agg.ingest_device(d_buf.data_ptr(), wb, d_off.data_ptr(), n)
        for col, key_set in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            want = po.cms_sketch_numpy(rows[col], w, depth, wl2, seed) * np.uint64(2)
            assert np.array_equal(agg.cms_read(key_set).reshape(-1), want), col
        ref = po.Rollup(300)
        ref.ingest(buf, off, 1)
        got = agg.read_window()
        st = agg.stats()
    want_rows = ref.rows()
    for c in ("bytes", "packets", "count"):
        want_rows[c] *= np.uint64(2)
    assert got.tobytes() == want_rows.tobytes() and st["wave_tile_launches"] == 2


RCCL_WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import _pkg
fa = _pkg.load(); po = _pkg.load_oracle()
torch.cuda.set_device(0)
rccl = None
for name in ("librccl.so", "librccl.so.1", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so"):
    try:
        rccl = C.CDLL(name, mode=C.RTLD_GLOBAL); break
    except OSError:
        pass
assert rccl is not None, "librccl.so not found"
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
n = 60000
gp = po.gen_params(mode=2, framed=1, seed=71, n_total=n, zipf_log2_universe=12)
buf, off = po.gen_records(gp, 0, n)
KS = (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS)
with fa.FlowAgg(framed=True, key_sets=7, cms_width_log2=12, topk_capacity_log2=14) as agg:
    agg.ingest(buf, off)
    before = [agg.cms_read(k).copy() for k in KS]
    top = agg.topk(KS[0], 40)
    agg.merge_allreduce(comm.value)            # ncclAllReduce(sum, u64) inside libflowagg, world 1: identity
    agg.merge_allreduce(comm.value)            # ... and idempotent: the ctx's own sketches are never reduced in place
    assert all(np.array_equal(agg.cms_read(k), b) for k, b in zip(KS, before))
    assert agg.topk(KS[0], 40).tobytes() == top.tobytes()
    agg.ingest(buf, off)                       # more ingest: the merged view is stale, readers see the local sketch
    assert all(np.array_equal(agg.cms_read(k), b * np.uint64(2)) for k, b in zip(KS, before))
    agg.merge_allreduce(comm.value)
    assert all(np.array_equal(agg.cms_read(k), b * np.uint64(2)) for k, b in zip(KS, before))
rccl.ncclCommDestroy.argtypes = [C.c_void_p]
rccl.ncclCommDestroy(comm)
print("RCCL_OK")
'''


def test_in_library_rccl_allreduce_world1(gpu_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", RCCL_WORKER % {"root": ROOT}], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


TWO_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import _pkg
fa = _pkg.load(); po = _pkg.load_oracle()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)   # both ranks share the box's only GPU; the exchange goes over gloo
dist.init_process_group("gloo")
n, nparts = 200000, 8
gp = po.gen_params(mode=2, framed=1, seed=81, n_total=n, zipf_log2_universe=14)
buf, off = po.gen_records(gp, 0, n)
raw = bytes(buf)
def shard(parts):
    idx = np.concatenate([np.arange(p, n, nparts) for p in parts]); idx.sort()
    recs = [raw[int(off[k]):int(off[k + 1])] for k in idx]
    o = np.zeros(len(recs) + 1, dtype=np.uint64); o[1:] = np.cumsum([len(r) for r in recs])
    return np.frombuffer(b"".join(recs), dtype=np.uint8), o
ks = 63
kw = dict(framed=True, key_sets=ks, cms_width_log2=14, topk_capacity_log2=16)
with fa.FlowAgg(**kw) as agg, fa.FlowAgg(**kw) as whole:
    b, o = shard(fa.dist.partitions_of(rank, world, nparts))
    agg.ingest(b, o)
    whole.ingest(buf, off)                               # the single-GPU answer
    dev = "cpu"
    # the order the advisor flagged: top-k of one sketch, then the other, then ports - nothing may be counted twice
    for key_set in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS, fa.FA_KEYS_SRCADDR_CMS):
        got = fa.dist.topk_merged(agg, key_set, 100, candidates_per_rank=None, device=dev)
        assert got.tobytes() == whole.topk(key_set, 100).tobytes(), key_set
        assert np.array_equal(agg.cms_read(key_set), whole.cms_read(key_set))
    for d in (0, 1):
        assert fa.dist.top_ports_merged(agg, d, device=dev).tobytes() == whole.top_ports(d).tobytes()
    assert fa.dist.minute_series_merged(agg, device=dev).tobytes() == whole.minute_series().tobytes()
    assert fa.dist.close_window_app_merged(agg, fa.ALL_TIMESLOTS, device=dev).tobytes() == whole.close_window_app().tobytes()
    assert fa.dist.close_window_merged(agg, fa.ALL_TIMESLOTS, device=dev).tobytes() == whole.close_window().tobytes()
# sliding windows over 60-s sub-buckets across ranks: the same device path (round 2 sent these through host memory)
kw2 = dict(framed=True, key_sets=9, subwindow_secs=60)
with fa.FlowAgg(**kw2) as agg, fa.FlowAgg(**kw2) as whole:
    b, o = shard(fa.dist.partitions_of(rank, world, nparts))
    agg.ingest(b, o)
    whole.ingest(buf, off)
    ts = int(whole.open_timeslots()[0])
    for t in (ts, ts + 60, ts + 120):
        assert fa.dist.close_window_merged(agg, t).tobytes() == whole.close_window(t).tobytes(), t
        assert fa.dist.close_window_app_merged(agg, t).tobytes() == whole.close_window_app(t).tobytes(), t
    assert fa.dist.close_window_merged(agg, fa.ALL_TIMESLOTS).tobytes() == whole.close_window().tobytes()
dist.destroy_process_group()
print("TWO_RANK_OK", rank)
'''


def _torchrun(nproc, args, env_extra, timeout=900):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_two_ranks_one_gpu_window_close_equals_single_rank(gpu_lib, tmp_path):
    script = tmp_path / "two_rank_worker.py"
    script.write_text(TWO_RANK_WORKER % {"root": ROOT})
    r = _torchrun(2, [str(script)], {})
    assert r.returncode == 0 and r.stdout.count("TWO_RANK_OK") == 2, (r.stdout[-1000:], r.stderr[-3000:])


def test_bench_harness_two_ranks_dry_run(gpu_lib):
    """bench.py's N>1 path (per-rank partitions, barrier + max-over-ranks timing, window close merged across ranks,
    merged top-k) on the 1-GPU box: both ranks on device 0, exchange over gloo.  Never a reported configuration."""
    r = _torchrun(2, ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--records", "2000000", "--chunk", "1000000",
                      "--mode", "zipf", "--key-sets", "3", "--cpu-sample", "0"],
                  {"FA_BENCH_BACKEND": "gloo", "FA_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["topk_src_addr_rows"] == 100
    assert out["roofline"]["frac"] > 0 and "cpu_baseline" not in out
    assert out["parity"]["ok"] and all(out["parity"]["merged_rows_equal_oracle_rollup_of_all_partitions"])


def test_bench_launches_its_own_ranks(gpu_lib):
    """`python bench.py --gpus 2` without torchrun (the shape of the driver's N=1 command with another N) starts its
    ranks itself; on the 1-GPU box FA_BENCH_SHARE_GPU=1 puts both on device 0 (exchange over gloo).  The line carries
    a parity statement: every rank's partition verified and the merged rows == the oracle's rollup of both partitions."""
    env = dict(os.environ, FA_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--records", "3000000",
                        "--chunk", "1000000"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    par = out["parity"]
    assert par["ok"] and par["ranks_ok"] == [True, True] and par["records_verified"] == 6_000_000
    assert par["merged_rows_equal_oracle_rollup_of_all_partitions"] == [True, True]
    assert len(out["roofline"]["per_rank_path_ms"]) == 2 and "cpu_baseline" not in out
    # rank 0's in-process group close on every visible device (one here), in a process of its own: both transports against the oracle
    gp = out["group_preflight"]
    assert gp["ok"] and gp["devices"] >= 1 and gp["peer"]["ok"] and gp["peer"]["topk"] and gp["peer"]["close_window_app_partitioned"], gp
    assert out["ms_per_step_all"] and len(out["ms_per_step_all"]) == 2 and out["settle"]["steps"] >= 1
    # without the sharing switch the same command refuses to oversubscribe the GPU instead of hanging in RCCL
    env.pop("FA_BENCH_SHARE_GPU")
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                           timeout=300, env=env, cwd=ROOT)
        assert r.returncode != 0 and "FA_BENCH_SHARE_GPU" in (r.stderr + r.stdout)


def test_reserve_ingest_and_host_phase_stats(gpu_lib, fa, po):
    """fa_reserve_ingest (ABI 8) sizes the staging a consumer's batches need up front; the rows do not depend on it, and the
    library accounts for the host time of its fa_ingest calls (fa_stats_t.host_*_ns)."""
    n = 300_000
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=31, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    with fa.FlowAgg(framed=True) as a, fa.FlowAgg(framed=True) as b:
        a.reserve_ingest(len(buf) + 1024, n)
        with pytest.raises(fa.FlowAggError):
            a.reserve_ingest(1 << 33, n)  # a batch is < 4 GiB
        for agg in (a, b):
            for lo in range(0, n, 100_000):
                o = off[lo:lo + 100_001]
                agg.ingest(buf[int(o[0]):int(o[-1])], o - o[0])
            assert agg.read_window().tobytes() == ref.rows().tobytes()
            st = agg.stats()
            assert st["host_ingest_ns"] > 0 and st["host_stage_copy_ns"] > 0 and st["host_ingest_ns"] >= st["host_stage_copy_ns"] + st["host_stage_wait_ns"]


def test_bench_side_measurements_run(gpu_lib):
    """The side-measurement modes of bench.py produce a line (small sizes; numbers are not judged here)."""
    for extra in (["--stage", "decode"], ["--mode", "goflow"], ["--mode", "reversed"]):
        r = subprocess.run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--records", "2000000", "--chunk", "1000000",
                            "--cpu-sample", "0", "--no-host-fed"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, (extra, r.stdout[-1000:], r.stderr[-3000:])
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["roofline"]["frac"] > 0, extra
        if extra == ["--mode", "goflow"]:
            assert out["config"]["records_second_chance_parser"] == 0 and out["parity"]["ok"]
        if extra == ["--mode", "reversed"]:
            assert out["config"]["records_second_chance_parser"] == 2_000_000 and out["parity"]["ok"]


@pytest.mark.parametrize("cap_log2", [10, 12, 13, 14])
@pytest.mark.parametrize("mode", [0, 1])
def test_small_tables_every_region_geometry(gpu_lib, fa, po, cap_log2, mode):
    """flows_5m into tables of 2^10 .. 2^14 slots (16 .. 256 regions: below 256 the aggregation kernel shares regions and
    keeps atomics; the direct, second-chance and merge paths always go through as_home): mocker's 9 groups stay inside
    the small table, config 2's 393 k groups make it grow through every geometry; rows == oracle either way, also
    after a window was closed (rebuild) and ingest went on."""
    n = 600_000
    gp = po.gen_params(mode=mode, framed=1, seed=90 + mode, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    ref.ingest(buf, off, 1)
    want = ref.rows()
    h = n // 2
    with fa.FlowAgg(framed=True, table_capacity_log2=cap_log2, max_batch_records=h) as agg:
        agg.ingest(buf[:int(off[h])], off[:h + 1])
        first_slot = int(want["timeslot"].min())
        closed = agg.close_window(first_slot)  # everything of the first window seen so far leaves the table
        agg.ingest(buf[int(off[h]):], off[h:] - off[h])
        rest = agg.read_window()
        st = agg.stats()
        assert st["records_ok"] == n and st["records_bad"] == 0
    both = fa.dist.merge_rows_host([closed, rest])
    assert both.tobytes() == want.tobytes()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FA_FUZZ_SEEDS", "24"))))  # (soak runs: FA_FUZZ_SEEDS=200)
def test_random_configurations_against_the_oracle(gpu_lib, fa, po, seed):
    """Differential run over randomly drawn configurations - key-set mask, generator mode, table sizes (every region
    geometry), window / sub-bucket length, launch sizes, host-fed vs device-resident input - everything the library
    produces against the oracle restatements of the same records: flows_5m rows, (SrcAddr,DstPort,Proto) rows, port
    group-bys, per-minute series, both sketches, the top-k of either contract; host-fed runs carry a few mutated records (a byte
    of the payload overwritten: whatever the parsers make of them, both sides make the same)."""
    import torch
    rng = np.random.default_rng(1000 + seed)
    ks = int(rng.choice([1, 3, 7, 9, 25, 41, 63]))
    mode = int(rng.choice([0, 1, 2, 3, 5]))
    n = int(rng.integers(150_000, 900_000))
    sub = int(rng.choice([60, 300]))
    kw = dict(mode=mode, framed=1, seed=500 + seed, n_total=n, span_secs=int(rng.choice([600, 900, 1500])),
              zipf_log2_universe=int(rng.integers(8, 19)), zipf_s_x100=int(rng.choice([80, 110, 140])))
    gp = po.gen_params(**kw)
    buf, off = po.gen_records(gp, 0, n)
    device_fed = bool(rng.integers(0, 2))
    candidates = bool(rng.integers(0, 2))
    track = int(rng.choice([16, 64, 256]))
    if not device_fed:
        buf = np.array(buf, dtype=np.uint8, copy=True)
        for r in rng.choice(n, size=int(rng.integers(0, 9)), replace=False):
            a, b = int(off[r]), int(off[r + 1])
            if b - a > 3:
                buf[int(rng.integers(a + 2, b))] = int(rng.integers(0, 256))  # (behind the length prefix: the framing stays what the offsets say)
    rows, status = po.decode_batch(buf, off, 1)
    n_bad = int((status != 0).sum())
    assert n_bad <= 8
    depth, wl2, cseed = int(rng.integers(2, 6)), int(rng.integers(10, 17)), int(rng.integers(1, 1 << 30))
    if candidates:
        wl2 = max(wl2, 15)  # (include/flowagg.h: cms_width_log2 >= topk_capacity_log2 - 1, or the sketch's noise fills the candidate set)
    cuts = np.sort(rng.choice(np.arange(1, n), size=int(rng.integers(1, 5)), replace=False))
    bounds = [0] + [int(c) for c in cuts] + [n]
    tk_cap = 16 if candidates else 20
    cfg = dict(framed=True, key_sets=ks, window_secs=300, subwindow_secs=sub, table_capacity_log2=int(rng.integers(10, 21)),
               wide_capacity_log2=int(rng.integers(8, 19)), cms_depth=depth, cms_width_log2=wl2, cms_seed=cseed, topk_capacity_log2=tk_cap,
               topk_mode=fa.TOPK_CANDIDATES if candidates else fa.TOPK_EXACT, topk_track=track,
               max_batch_records=max(b - a for a, b in zip(bounds, bounds[1:])))
    with fa.FlowAgg(**cfg) as agg:
        for a, b in zip(bounds, bounds[1:]):
            if device_fed:
                mp = fa.mock_params(**kw)
                d_buf = torch.empty((b - a) * 256 + 4096, dtype=torch.uint8, device="cuda")
                d_off = torch.empty(b - a + 1, dtype=torch.int32, device="cuda")
                w = agg.mock_generate_device(mp, a, b - a, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
                assert w == int(off[b] - off[a])
                agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), b - a)
            else:
                agg.ingest(buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a])
        st = agg.stats()
        assert st["records_ok"] == n - n_bad and st["records_bad"] == n_bad, (cfg, kw)
        # flows_5m: every window of the granule the ctx aggregates at (sub-buckets folded by the library at read time)
        ref = po.Rollup(sub)
        ref.ingest(buf, off, 1)
        want = ref.rows()
        if sub == 300:
            assert agg.read_window().tobytes() == want.tobytes(), (cfg, kw)
        else:
            assert int(agg.read_window()["count"].sum()) == n - n_bad
        if ks & fa.FA_KEYS_ADDR_PORT_PROTO:
            if sub == 300:
                got_app, want_app = agg.read_window_app(), po.rollup_app(rows, status, 300)
            else:  # 60-second sub-buckets: the 5-minute window that starts one minute into the stream (sliding)
                got_app = agg.read_window_app(po.T0 + 60)
                want_app = po.rollup_app(rows, status, 60, window=300, timeslot=po.T0 + 60)
            assert len(got_app) == len(want_app), (len(got_app), len(want_app), cfg, kw)
            for c in ("timeslot", "src_addr", "dst_port", "proto", "bytes", "packets", "count"):
                assert np.array_equal(got_app[c], want_app[c]), (c, cfg, kw)
        if ks & fa.FA_KEYS_PORT_HIST:
            for d in (0, 1):
                g, w_ = agg.top_ports(d), po.top_ports(rows, status, d)
                assert all(np.array_equal(g[c], w_[c]) for c in ("port", "weight", "count")), (d, cfg, kw)
        if ks & fa.FA_KEYS_MINUTE_SERIES:
            g, w_ = agg.minute_series(), po.minute_series(rows, status)
            assert all(np.array_equal(g[c], w_[c]) for c in ("minute", "weight", "count")), (cfg, kw)
        good = status == 0
        with np.errstate(over="ignore"):
            wgt = rows["bytes"] * rows["sampling_rate"]
        for col, key_set in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            if ks & key_set:
                sk = po.cms_sketch_numpy(rows[col][good], wgt[good], depth, wl2, cseed)
                assert np.array_equal(agg.cms_read(key_set).reshape(-1), sk), (col, cfg, kw)
                if candidates:  # every ingest call above is one launch = one batch of the contract
                    batches = [(rows[col][a:b][good[a:b]], wgt[a:b][good[a:b]]) for a, b in zip(bounds, bounds[1:])]
                    _, keys, est, _ = po.topk_candidates(batches, depth, wl2, cseed, track=track, capacity_log2=tk_cap)
                else:
                    keys = np.unique(np.ascontiguousarray(rows[col][good]), axis=0)
                    est = po.cms_estimates_numpy(sk, keys, depth, wl2, cseed)
                want_top = sorted(zip((-est.astype(object)).tolist(), [bytes(k) for k in keys]))
                k_read = int(rng.choice([10, 300, 1 << 20]))
                got_top = agg.topk(key_set, k_read)
                assert [(bytes(r["key"]), int(r["weight"])) for r in got_top] == [(k, -e) for e, k in want_top[:k_read]], (col, k_read, cfg, kw)


def test_learnt_field_order_kernel_is_chosen_by_the_counters_and_exact(gpu_lib, fa, po):
    """A producer that marshals in DESCENDING field order (valid proto3 no ordered walk accepts): after the first launches the
    library runs the kernel variant whose waves learn the field order of a tile's longest record and walk the next tiles with it
    (ingest.cuh tier 4; order-free parser only for what that refuses) - rows == the oracle's all along; canonical records again:
    back to the common kernel; records of BOTH orders mixed in one stream and a few mutated ones: still the oracle's rows."""
    n = 400_000
    gp = po.gen_params(mode=po.GEN_REVERSED, framed=1, seed=321, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    gq = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=322, n_total=n, span_secs=900)
    bufq, offq = po.gen_records(gq, 0, n)
    ref = po.Rollup(300)
    with fa.FlowAgg(framed=True, table_capacity_log2=18) as agg:
        for _ in range(4):
            agg.ingest(buf, off)
            ref.ingest(buf, off, 1)
            agg.sync()
        st = agg.stats()
        assert st["records_ok"] == 4 * n and st["records_bad"] == 0 and st["learnt_order_launches"] >= 2, st
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        # interleaved: every other record canonical (a tile's waves meet both; whatever the learnt list refuses takes the order-free parser)
        raw, rawq = bytes(buf), bytes(bufq)
        recs = []
        for k in range(60_000):
            recs.append(raw[int(off[k]):int(off[k + 1])] if k & 1 else rawq[int(offq[k]):int(offq[k + 1])])
        rng = np.random.default_rng(9)
        mixed = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
        offm = np.zeros(len(recs) + 1, dtype=np.uint64)
        offm[1:] = np.cumsum([len(r) for r in recs])
        for r in rng.choice(len(recs), size=40, replace=False):
            a, b = int(offm[r]), int(offm[r + 1])
            mixed[int(rng.integers(a + 2, b))] = int(rng.integers(0, 256))
        agg.ingest(mixed, offm)
        ref.ingest(mixed, offm, 1)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        before = agg.stats()["learnt_order_launches"]
        for _ in range(70):  # canonical records: the verdict runs out after 64 launches
            agg.ingest(bufq[:int(offq[20_000])], offq[:20_001])
            ref.ingest(bufq[:int(offq[20_000])], offq[:20_001], 1)
        agg.sync()
        agg.ingest(bufq, offq)
        ref.ingest(bufq, offq, 1)
        st = agg.stats()
        assert st["learnt_order_launches"] - before <= 66, st
        assert agg.read_window().tobytes() == ref.rows().tobytes()


@pytest.mark.parametrize("force", ["1", "0"])
def test_learnt_field_order_kernel_forced_on_every_generator_mode(gpu_lib, fa, po, monkeypatch, force):
    """FA_SEQ=1: the learnt-order variant from the first launch on, on every producer shape (on canonical streams its tier never
    runs; on the reversed one it does); FA_SEQ=0: never.  Rows == the oracle's either way."""
    monkeypatch.setenv("FA_SEQ", force)
    n = 150_000
    for mode in (po.GEN_MOCKER, po.GEN_ASPAIRS, po.GEN_GOFLOW, po.GEN_REVERSED):
        gp = po.gen_params(mode=mode, framed=1, seed=400 + mode, n_total=n, span_secs=900)
        buf, off = po.gen_records(gp, 0, n)
        ref = po.Rollup(300)
        ref.ingest(buf, off, 1)
        ref.ingest(buf, off, 1)
        with fa.FlowAgg(framed=True, table_capacity_log2=18) as agg:
            agg.ingest(buf, off)
            agg.ingest(buf, off)
            st = agg.stats()
            assert st["records_ok"] == 2 * n and (st["learnt_order_launches"] == 2) == (force == "1"), (mode, st)
            assert agg.read_window().tobytes() == ref.rows().tobytes(), mode
