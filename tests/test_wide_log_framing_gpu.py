"""The wide log, the partitioned close and device-side framing: the adaptive wide log end to end (enter - leave - drain through both folds), its behaviour when
segment buffers run out, watermark closes of real timeslots, late records, the hash-partitioned window close (device
partition == numpy restatement; two ranks on the one GPU), and window reads that leave through the pinned buffer in
pieces.  All through the C-ABI, bit-exact against the oracle restatements."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream(po, n, seed, universe_log2=22, zs=80, span=900):
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=seed, n_total=n, zipf_log2_universe=universe_log2, zipf_s_x100=zs, span_secs=span)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    return buf, off, rows, status


def _ingest(agg, buf, off, n, step, sync=True):
    for a in range(0, n, step):
        b = min(n, a + step)
        agg.ingest(buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a])
        if sync:
            agg.sync()  # (the counter feedback then sees exactly this launch: the state machine's moves are deterministic)


def _scaled(rows, k):
    out = rows.copy()
    with np.errstate(over="ignore"):
        for c in ("bytes", "packets", "count"):
            out[c] = out[c] * np.uint64(k)
    return out


def test_adaptive_wide_log_enters_leaves_and_drains_both_ways(gpu_lib, fa, po, monkeypatch):
    """The library's own choices (FA_WIDE unset) on 400 k-record launches, the log bounded at two pending chunks:
    A  a stream that opens a (SrcAddr,DstPort,Proto) row per record -> the counter feedback switches to the log;
    B  the same records again and again: chunks are recorded, the folds of the oldest ones stop opening rows -> back to the table;
    C  more of it: what is still pending drains through the region-owned fold, one chunk per launch;
    D  NEW keys: log mode again, and now the folds make the table grow - chunks scattered for the old geometry go through the
       atomic replay.
    After every phase the rows are the oracle's."""
    monkeypatch.setenv("FA_WLOG_RANGE_CHECK", "1")
    monkeypatch.delenv("FA_WIDE", raising=False)
    monkeypatch.setenv("FA_WIDE_LOG_CHUNKS", "2")
    step = 400_000
    b1, o1, r1, s1 = _stream(po, 3 * step, seed=401)
    b2, o2, r2, s2 = _stream(po, 20 * step, seed=402)
    parts1 = [po.rollup_app(r1[i * step:(i + 1) * step], s1[i * step:(i + 1) * step], 300).astype(fa.ROW_APP_DTYPE) for i in range(3)]
    assert sum(len(p) for p in parts1) > 0.9 * 3 * step  # (random ports: a row per record)
    times1 = [0, 0, 0]
    new_parts = []
    launches = [0]

    def old(agg):  # the next third of stream 1 (keys the table knows after phase A)
        i = launches[0] % 3
        _ingest(agg, b1[int(o1[i * step]):int(o1[(i + 1) * step])], o1[i * step:(i + 1) * step + 1] - o1[i * step], step, step)
        times1[i] += 1
        launches[0] += 1

    def new(agg):  # the next 400 k records of stream 2: keys nobody has seen
        i = len(new_parts)
        _ingest(agg, b2[int(o2[i * step]):int(o2[(i + 1) * step])], o2[i * step:(i + 1) * step + 1] - o2[i * step], step, step)
        new_parts.append(po.rollup_app(r2[i * step:(i + 1) * step], s2[i * step:(i + 1) * step], 300).astype(fa.ROW_APP_DTYPE))

    def check(agg):
        want = fa.dist.merge_rows_app_host([_scaled(parts1[i], times1[i]) for i in range(3) if times1[i]] + new_parts)
        assert agg.read_window_app().tobytes() == want.tobytes()

    with fa.FlowAgg(framed=True, key_sets=9, wide_capacity_log2=16, table_capacity_log2=20, max_batch_records=step) as agg:
        # A: three launches = 1.2 M records > the 2^20 the feedback wants to see
        for _ in range(3):
            old(agg)
        st = agg.stats()
        assert st["wide_log_mode"] == 1 and st["wide_log_recorded"] == 0, st
        check(agg)
        # B: log mode - chunks recorded, the third one pushes the oldest into the table
        for _ in range(3):
            old(agg)
        st = agg.stats()
        assert st["wide_log_recorded"] == 3 and st["wide_log_chunks"] == 2 and st["wide_log_folded"] + st["wide_log_replayed"] == 1, st
        assert st["wide_log_bytes"] > 0 and st["wide_log_records"] == 2 * step
        check(agg)
        for _ in range(8):  # ... until more than 2^20 folded tuples have opened no row
            old(agg)
            if agg.stats()["wide_log_mode"] == 0:
                break
        st = agg.stats()
        assert st["wide_log_mode"] == 0 and st["wide_log_chunks"] >= 1, st
        check(agg)
        # C: the pending chunks drain, one per launch, through the region-owned fold (the table has not grown since)
        pending, folded = st["wide_log_chunks"], st["wide_log_folded"]
        for k in range(pending):
            old(agg)
            st = agg.stats()
            assert st["wide_log_chunks"] == pending - k - 1 and st["wide_log_folded"] == folded + k + 1 and st["wide_log_mode"] == 0, st
        check(agg)
        # D: new keys - the log again; its folds now grow the table, and chunks scattered before that take the atomic replay
        for _ in range(8):
            new(agg)
            if agg.stats()["wide_log_mode"] == 1:
                break
        st = agg.stats()
        assert st["wide_log_mode"] == 1, st
        check(agg)
        replayed = st["wide_log_replayed"]
        while len(new_parts) < 20:
            new(agg)
            if agg.stats()["wide_log_replayed"] > replayed:
                break
        st = agg.stats()
        assert st["wide_log_replayed"] > replayed, st
        check(agg)
        assert st["records_ok"] == (launches[0] + len(new_parts)) * step and st["records_bad"] == 0
        # flows_5m never noticed any of it
        ref = po.Rollup(300)
        for i in range(3):
            for _ in range(times1[i]):
                ref.ingest(b1[int(o1[i * step]):int(o1[(i + 1) * step])], o1[i * step:(i + 1) * step + 1] - o1[i * step], 1)
        nn = len(new_parts) * step
        ref.ingest(b2[:int(o2[nn])], o2[:nn + 1], 1)
        assert agg.read_window().tobytes() == ref.rows().tobytes()


def test_wide_log_folds_early_when_segment_buffers_run_out(gpu_lib, fa, po, monkeypatch):
    """FA_WSEG_BUDGET stands in for a failing hipMalloc: with room for three pairs of segment buffers and eight chunks
    allowed, the fourth launch cannot get a pair - the oldest chunk is folded and its buffers are taken over; no error."""
    monkeypatch.setenv("FA_WIDE", "log")
    monkeypatch.setenv("FA_WIDE_LOG_CHUNKS", "8")
    monkeypatch.setenv("FA_WSEG_BUDGET", "3")
    n, step = 600_000, 100_000
    buf, off, rows, status = _stream(po, n, seed=411, universe_log2=16)
    with fa.FlowAgg(framed=True, key_sets=9, wide_capacity_log2=20, max_batch_records=step) as agg:
        _ingest(agg, buf, off, n, step, sync=False)
        st = agg.stats()
        assert st["wide_log_nomem_folds"] == 3 and st["wide_log_chunks"] == 3 and st["wide_log_recorded"] == 6, st  # (3 pairs of buffers: 3 chunks)
        assert agg.read_window_app().tobytes() == po.rollup_app(rows, status, 300).astype(fa.ROW_APP_DTYPE).tobytes()


def test_timeslot_closes_move_the_watermark_of_pending_chunks(gpu_lib, fa, po, monkeypatch):
    """A close of the OLDEST buckets must not fold the log: the chunks' watermark moves (and a chunk with nothing left is
    dropped whole).  Round 3 compared the window with the launch's time base - two buckets below its smallest sampled one -
    and every close of a real timeslot folded all chunks."""
    monkeypatch.setenv("FA_WIDE", "log")
    monkeypatch.setenv("FA_WIDE_LOG_CHUNKS", "8")
    n, step, sub = 400_000, 100_000, 60
    buf, off, rows, status = _stream(po, n, seed=421, universe_log2=14, span=600)
    t32 = rows["time_received"].astype(np.uint64).astype(np.uint32)
    with fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub, wide_capacity_log2=20, max_batch_records=step) as agg:
        _ingest(agg, buf, off, n, step, sync=False)
        t0 = int(agg.open_timeslots()[0])
        st0 = agg.stats()
        assert st0["wide_log_chunks"] == 4 and st0["wide_used"] == 0
        alive = np.ones(n, dtype=bool)
        for k in range(6):  # six closes, each removes the oldest sub-bucket
            ts = t0 + k * sub
            got = agg.close_window_app(ts)
            want = po.rollup_app(rows[alive], status[alive], sub, window=300, timeslot=ts).astype(fa.ROW_APP_DTYPE)
            assert got.tobytes() == want.tobytes(), k
            alive &= ~((t32 >= ts) & (t32 < ts + sub))
        st = agg.stats()
        assert st["wide_log_folded"] == 0 and st["wide_log_replayed"] == 0 and st["wide_used"] == 0, st  # nothing was folded
        assert st["wide_log_watermark_moves"] >= 6 and st["wide_log_dropped"] == 2, st                     # (600 s in 4 launches, 360 s closed: two chunks are gone whole)
        assert agg.read_window_app().tobytes() == po.rollup_app(rows[alive], status[alive], sub).astype(fa.ROW_APP_DTYPE).tobytes()
        # a drop that is not the oldest range still folds first (and stays exact)
        mid = t0 + 8 * sub
        got = agg.close_window_app(mid)
        assert got.tobytes() == po.rollup_app(rows[alive], status[alive], sub, window=300, timeslot=mid).astype(fa.ROW_APP_DTYPE).tobytes()
        assert agg.stats()["wide_log_chunks"] == 0


def test_late_records_are_counted_and_still_aggregated(gpu_lib, fa, po):
    n = 300_000
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=431, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    t32 = rows["time_received"].astype(np.uint64).astype(np.uint32)
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(buf, off)
        t0 = int(agg.open_timeslots()[0])
        assert agg.stats()["records_late"] == 0
        first = agg.close_window(t0)
        assert int(first["count"].sum()) == int((t32 < t0 + 300).sum())
        agg.ingest(buf, off)  # the same stream again: what falls into the closed window is late
        st = agg.stats()
        assert st["records_late"] == int((t32 < t0 + 300).sum()), st
        again = agg.read_window(t0)  # ... and visible to a later read of that timeslot, like a late INSERT into flows_5m
        assert again.tobytes() == first.tobytes()
        agg.close_window()  # close-all is not a time watermark
        agg.ingest(buf, off)
        assert agg.stats()["records_late"] == 2 * int((t32 < t0 + 300).sum())


@pytest.mark.parametrize("world", [2, 3, 8])
def test_device_partition_equals_the_numpy_restatement(gpu_lib, fa, po, world):
    n = 300_000
    buf, off, rows, status = _stream(po, n, seed=441, universe_log2=14)
    with fa.FlowAgg(framed=True, key_sets=63, cms_width_log2=14, topk_capacity_log2=16, subwindow_secs=60) as agg:
        agg.ingest(buf, off)
        t0 = int(agg.open_timeslots()[0])
        for kind, ts, k in ((fa.ROWS_APP, fa.ALL_TIMESLOTS, 0), (fa.ROWS_APP, t0 + 60, 0), (fa.ROWS_5M, fa.ALL_TIMESLOTS, 0),
                            (fa.ROWS_PORT_DST, 0, 0), (fa.ROWS_MINUTE, 0, 0), (fa.ROWS_TOPK_SRC, 0, 5000)):
            ptr, m = agg.rows_device(kind, ts, k)
            whole = agg.rows_fetch(kind, ptr, m)
            pptr, counts = agg.rows_partition_device(kind, ptr, m, world)
            assert sum(counts) == m and m > 0
            parts = agg.rows_fetch(kind, pptr, m)
            dest = fa.dist.partition_rows_host(parts, kind, world)
            assert np.array_equal(dest, np.repeat(np.arange(world), counts)), kind     # grouped by owner, owners as numpy says
            assert np.array_equal(np.bincount(fa.dist.partition_rows_host(whole, kind, world), minlength=world), counts)
            def canon(a):  # rows as u64 columns, sorted: the same multiset of rows
                mtx = np.ascontiguousarray(a).view(np.uint8).reshape(m, -1).view("<u8")
                return mtx[np.lexsort(mtx.T[::-1])]
            assert np.array_equal(canon(parts), canon(whole))
            if kind == fa.ROWS_APP and world > 2:
                assert min(counts) > 0.5 * m / world  # (a hash partition: balanced)


TWO_RANK_PARTITIONED = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import _pkg
fa = _pkg.load(); po = _pkg.load_oracle()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n, nparts = 300000, 8
gp = po.gen_params(mode=2, framed=1, seed=451, n_total=n, zipf_log2_universe=20, zipf_s_x100=80)
buf, off = po.gen_records(gp, 0, n)
raw = bytes(buf)
def shard(parts):
    idx = np.concatenate([np.arange(p, n, nparts) for p in parts]); idx.sort()
    recs = [raw[int(off[k]):int(off[k + 1])] for k in idx]
    o = np.zeros(len(recs) + 1, dtype=np.uint64); o[1:] = np.cumsum([len(r) for r in recs])
    return np.frombuffer(b"".join(recs), dtype=np.uint8), o
for mode in ("scatter", "log"):
    os.environ["FA_WIDE"] = mode
    kw = dict(framed=True, key_sets=9, subwindow_secs=60, max_batch_records=60000)
    with fa.FlowAgg(**kw) as agg, fa.FlowAgg(**kw) as whole:
        b, o = shard(fa.dist.partitions_of(rank, world, nparts))
        agg.ingest(b, o)
        whole.ingest(buf, off)
        ts = int(whole.open_timeslots()[0])
        for t in (ts, ts + 60, fa.ALL_TIMESLOTS):
            share = fa.dist.close_window_app_partitioned(agg, t)
            assert (fa.dist.partition_rows_host(share, fa.ROWS_APP, world) == rank).all() and len(share) > 0
            allp = fa.dist.allgather_struct(share, fa.dist.ROW_APP_DTYPE, device="cpu")
            assert sum(len(p) for p in allp) == len(fa.dist.merge_rows_app_host(allp))        # a key lives on ONE rank
            assert fa.dist.merge_rows_app_host(allp).tobytes() == whole.close_window_app(t).tobytes(), (mode, t)
        # the flows_5m rows through the same exchange
        s5 = fa.dist.rows_merged_partitioned(agg, fa.ROWS_5M)
        assert fa.dist.merge_rows_host(fa.dist.allgather_struct(s5, fa.dist.ROW5M_DTYPE, device="cpu")).tobytes() == whole.read_window().tobytes()
dist.destroy_process_group()
print("TWO_RANK_PARTITIONED_OK", rank)
'''


def test_two_ranks_partitioned_window_close(gpu_lib, tmp_path):
    import socket
    script = tmp_path / "two_rank_partitioned.py"
    script.write_text(TWO_RANK_PARTITIONED % {"root": ROOT})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FA_WIDE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.count("TWO_RANK_PARTITIONED_OK") == 2, (r.stdout[-1000:], r.stderr[-3000:])


def test_large_window_leaves_through_the_pinned_buffer_in_pieces(gpu_lib, fa, po, monkeypatch):
    """34 MB of (SrcAddr,DstPort,Proto) rows: more than half of the ctx's pinned buffer, so the copy-out runs in pieces
    (copy engine and host threads alternating on the two halves) - from the table and from the log, through the read call
    and through fa_rows_device + fa_rows_fetch, into a fresh and into a reused buffer."""
    n, step = 640_000, 160_000
    buf, off, rows, status = _stream(po, n, seed=461)
    want = po.rollup_app(rows, status, 300).astype(fa.ROW_APP_DTYPE)
    assert want.nbytes > 32 << 20
    for mode in ("scatter", "log"):
        monkeypatch.setenv("FA_WIDE", mode)
        with fa.FlowAgg(framed=True, key_sets=9, table_capacity_log2=16, wide_capacity_log2=21, max_batch_records=step) as agg:
            _ingest(agg, buf, off, n, step, sync=False)
            assert agg.read_window_app().tobytes() == want.tobytes(), mode
            reuse = np.empty(len(want) + 1000, dtype=fa.ROW_APP_DTYPE)
            got = agg.read_window_app(out=reuse)
            assert got.tobytes() == want.tobytes() and got.base is reuse, mode
            # a page-locked buffer of the caller's: one copy-engine transfer, no relay (also at an odd offset into it)
            pinned = fa.FlowAgg.pinned_rows(fa.ROWS_APP, len(want) + 7)
            got = agg.read_window_app(out=pinned)
            assert got.tobytes() == want.tobytes() and np.shares_memory(got, pinned), mode
            got = agg.read_window_app(out=pinned[3:])
            assert got.tobytes() == want.tobytes(), mode
            n_out = fa.C.c_size_t()
            small = np.empty(1000, dtype=fa.ROW_APP_DTYPE)  # the C contract: too small a buffer -> FA_ERR_CAPACITY and the size
            assert agg._L.fa_read_window_app(agg._h, fa.ALL_TIMESLOTS, small.ctypes.data, len(small), fa.C.byref(n_out)) == -6
            assert n_out.value == len(want)


@pytest.mark.parametrize("mode", ["scatter", "log"])
def test_drop_range_closes_a_tumbling_window_over_sub_buckets(gpu_lib, fa, po, monkeypatch, mode):
    """fa_drop_range: the five 60-s sub-buckets of a 5-minute window leave in one pass - both key sets, table and log."""
    monkeypatch.setenv("FA_WIDE", mode)
    n, step, sub = 300_000, 100_000, 60
    buf, off, rows, status = _stream(po, n, seed=471, universe_log2=14, span=900)
    t32 = rows["time_received"].astype(np.uint64).astype(np.uint32)
    with fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub, max_batch_records=step) as agg:
        _ingest(agg, buf, off, n, step, sync=False)
        t0 = int(agg.open_timeslots()[0])
        t0 -= t0 % 300
        alive = np.ones(n, dtype=bool)
        for w in (t0, t0 + 300):
            assert agg.read_window_app(w).tobytes() == po.rollup_app(rows[alive], status[alive], sub, window=300, timeslot=w).astype(fa.ROW_APP_DTYPE).tobytes()
            agg.drop_range(fa.ROWS_APP, w, w + 300)
            agg.drop_range(fa.ROWS_5M, w, w + 300)
            alive &= ~((t32 >= w) & (t32 < w + 300))
            assert agg.read_window_app().tobytes() == po.rollup_app(rows[alive], status[alive], sub).astype(fa.ROW_APP_DTYPE).tobytes()
            ref = po.Rollup(sub)
            idx = np.nonzero(alive)[0]
            if len(idx):
                a, b = int(idx[0]), int(idx[-1]) + 1  # (the stream is time-ordered: the live records are a suffix)
                assert alive[a:b].all()
                ref.ingest(buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a], 1)
            assert agg.read_window().tobytes() == ref.rows().tobytes()
        with pytest.raises(fa.FlowAggError):
            agg.drop_range(fa.ROWS_APP, t0 + 7, t0 + 300)
        assert agg.stats()["records_late"] == 0


def _varint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_device_side_framing_equals_the_offsets_path(gpu_lib, fa, po, mode):
    """offsets == NULL on the device path: the framed chain is cut into records on the GPU (framing.cuh) - the rows, every
    counter and every other key set's result equal the call with offsets; several launches per call (max_batch_records)."""
    import torch
    n = 700_000
    gp = po.gen_params(mode=mode, framed=1, seed=501, n_total=n, span_secs=900, zipf_log2_universe=14)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    d = torch.zeros(len(buf) + 64, dtype=torch.uint8, device="cuda")
    d[:len(buf)] = torch.from_numpy(np.ascontiguousarray(buf))
    torch.cuda.synchronize()
    with fa.FlowAgg(framed=True, key_sets=9, max_batch_records=250_000) as agg, fa.FlowAgg(framed=True, key_sets=9, max_batch_records=250_000) as want:
        agg.ingest_device(d.data_ptr(), len(buf), 0, 0)
        want.ingest(buf, off)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        assert agg.read_window_app().tobytes() == want.read_window_app().tobytes()
        st, sw = agg.stats(), want.stats()
        assert st["records_ok"] == n and st["records_bad"] == 0 and st["bytes_in"] == sw["bytes_in"] == len(buf) and st["batches"] == 3
    # the host entry point without offsets takes the same road (bytes uploaded as they are, cut on the device)
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(buf)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
    with fa.FlowAgg(framed=False) as agg, pytest.raises(fa.FlowAggError):
        agg.ingest_device(d.data_ptr(), len(buf), 0, 0)  # bare records need offsets


def test_device_side_framing_refuses_what_is_not_a_chain_and_survives_odd_ones(gpu_lib, fa, po):
    import torch
    n = 120_000
    gp = po.gen_params(mode=1, framed=1, seed=511, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)

    def run(stream, **kw):
        d = torch.zeros(len(stream) + 64, dtype=torch.uint8, device="cuda")
        if len(stream):
            d[:len(stream)] = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy())
        torch.cuda.synchronize()
        with fa.FlowAgg(framed=True, **kw) as agg:
            agg.ingest_device(d.data_ptr(), len(stream), 0, 0)
            return agg.read_window(), agg.stats()

    def host(stream):  # the host walk + the offsets path: what the device split has to equal
        o = [0]
        p = 0
        while p < len(stream):
            v = s = 0
            while True:
                b = stream[p]
                p += 1
                v |= (b & 0x7F) << s
                s += 7
                if not b & 0x80:
                    break
            p += v
            o.append(p)
        assert p == len(stream)
        with fa.FlowAgg(framed=True) as agg:
            agg.ingest(stream, np.array(o, dtype=np.uint64))
            return agg.read_window(), agg.stats()

    # 1. a stream whose last frame is cut short / whose first length runs past the end: FA_ERR_FRAMING, nothing ingested
    for broken in (raw[:-1], _varint(len(raw) + 5) + raw, raw + b"\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff\xff"):
        with pytest.raises(fa.FlowAggError) as ei:
            run(broken)
        assert ei.value.code == -7
    # 2. odd but valid chains equal the host walk: empty frames, a record of 40 KB (three blocks: settles by rounds), one of
    #    300 KB (19 blocks: more rounds than the device takes - the host walks), single-record and empty streams
    junk40 = _varint(40_000) + bytes(40_000)
    junk300 = _varint(300_000) + bytes([0x08, 0x01] * 150_000)
    for stream in (raw[:int(off[1000])] + b"\x00" * 5000 + raw[int(off[1000]):int(off[5000])],
                   raw[:int(off[70000])] + junk40 + raw[int(off[70000]):],
                   raw[:int(off[300])] + junk300 + raw[int(off[300]):int(off[90000])],
                   raw[:int(off[1])], b""):
        got, st = run(stream)
        want, sw = host(stream)
        assert got.tobytes() == want.tobytes()
        assert (st["records_ok"], st["records_bad"], st["bytes_in"]) == (sw["records_ok"], sw["records_bad"], sw["bytes_in"])


@pytest.mark.parametrize("mode,n", [(5, 2_000), (5, 60_000), (3, 150_000)])
def test_device_side_framing_repairs_wrong_guesses(gpu_lib, fa, po, mode, n, capfd, monkeypatch):
    """The guesses only decide the number of rounds.  A producer that marshals its fields in descending order (generator mode 5)
    has no plausible candidate anywhere: every block starts from its own first byte and is walked again from where its predecessor
    ends - about a block per round, on the device while the rounds last (9 blocks), on the host beyond (260 blocks); the GoFlow-shaped
    stream has 3-4 % of its guesses replaced, each walked again only up to the frame where the two walks meet.  Rows equal
    the offsets path every time."""
    import torch
    monkeypatch.setenv("FA_VERBOSE", "1")
    gp = po.gen_params(mode=mode, framed=1, seed=77, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    d = torch.zeros(len(buf) + 64, dtype=torch.uint8, device="cuda")
    d[:len(buf)] = torch.from_numpy(np.ascontiguousarray(buf))
    torch.cuda.synchronize()
    with fa.FlowAgg(framed=True) as agg:
        capfd.readouterr()
        agg.ingest_device(d.data_ptr(), len(buf), 0, 0)
        agg.sync()
        log = capfd.readouterr().err
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        st = agg.stats()
        assert st["records_ok"] == n and st["records_bad"] == 0
    line = [l for l in log.splitlines() if "[flowagg framing]" in l][-1]
    blocks = (len(buf) + 16383) // 16384
    if mode == 5 and blocks <= 12:
        assert "NOT" not in line and 3 < int(line.split("settled after ")[1].split()[0]) <= blocks + 1, line  # (a few blocks begin on a frame by chance)
    elif mode == 5:
        assert "NOT settled" in line, line
    else:
        assert "settled after" in line and "NOT" not in line, line


@pytest.mark.parametrize("seed", range(int(os.environ.get("FA_FUZZ_SEEDS", "12"))))
def test_device_side_framing_random_chains(gpu_lib, fa, po, seed, capfd, monkeypatch):
    """Chains drawn at random - runs of generated records of every producer, cut by junk frames (random bytes behind a length of
    0 .. 40 000: no candidate is plausible near them, the blocks behind them are walked again from where their predecessor ends),
    runs of empty frames, records in descending field order - cut on the device: rows and counters equal the offsets path with
    the offsets of a host walk.  What matters is the repair rounds: the walk from a replaced start meets the first walk somewhere
    (inside the same 256-byte sub-block, sub-blocks later, or never)."""
    import torch
    monkeypatch.setenv("FA_VERBOSE", "1")
    rng = np.random.default_rng(4400 + seed)
    parts = []
    total = 0
    want_bytes = int(rng.integers(1_200_000, 4_000_000))
    while total < want_bytes:
        kind = int(rng.integers(0, 10))
        if kind < 6:
            mode = int(rng.choice([1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 5]))
            n = int(rng.integers(50, 400 if mode == 5 else 6000))  # (descending field order: no candidates - a block per round, twelve rounds)
            gp = po.gen_params(mode=mode, framed=1, seed=int(rng.integers(1, 1 << 20)), n_total=n, span_secs=900, zipf_log2_universe=12)
            buf, _off = po.gen_records(gp, 0, n)
            piece = bytes(buf)
        elif kind < 8:
            ln = int(rng.choice([0, 1, int(rng.integers(2, 300)), int(rng.integers(300, 5000)), int(rng.integers(5000, 40000))]))
            piece = _varint(ln) + rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes()
        elif kind < 9:
            piece = b"\x00" * int(rng.integers(1, 700))
        else:  # a record-sized frame of zeros behind a few ordinary ones: plausible lengths, implausible payloads
            ln = int(rng.integers(40, 120))
            piece = (_varint(ln) + bytes(ln)) * int(rng.integers(1, 40))
        parts.append(piece)
        total += len(piece)
    stream = b"".join(parts)
    off = [0]
    p = 0
    while p < len(stream):
        v = sh = 0
        while True:
            b = stream[p]
            p += 1
            v |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        p += v
        off.append(p)
    assert p == len(stream)
    off = np.array(off, dtype=np.uint64)
    d = torch.zeros(len(stream) + 64, dtype=torch.uint8, device="cuda")
    d[:len(stream)] = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy())
    torch.cuda.synchronize()
    kw = dict(framed=True, key_sets=9, max_batch_records=len(off))
    with fa.FlowAgg(**kw) as agg, fa.FlowAgg(**kw) as want:
        capfd.readouterr()
        agg.ingest_device(d.data_ptr(), len(stream), 0, 0)
        agg.sync()
        log = capfd.readouterr().err
        want.ingest(np.frombuffer(stream, dtype=np.uint8), off)
        st, sw = agg.stats(), want.stats()
        assert (st["records_ok"], st["records_bad"], st["bytes_in"]) == (sw["records_ok"], sw["records_bad"], sw["bytes_in"]), (seed, log)
        assert st["records_ok"] + st["records_bad"] == len(off) - 1
        assert agg.read_window().tobytes() == want.read_window().tobytes()
        assert agg.read_window_app().tobytes() == want.read_window_app().tobytes()
    assert "[flowagg framing]" in log
    if os.environ.get("FA_FUZZ_PRINT"):  # (soak runs: how the rounds went)
        with capfd.disabled():
            print([l for l in log.splitlines() if "[flowagg framing]" in l][-1])


def test_first_big_launch_is_probed_before_the_rest_follows(gpu_lib, fa, po, monkeypatch):
    """A ctx whose FIRST launch is big (>= 2^22 records) and carries the (SrcAddr,DstPort,Proto) key set: its first 2^20 + 2^17
    records go ahead as a launch of their own, the counter feedback reads them, and a stream that opens a row per record has the
    rest of that launch recorded in the wide log instead of folded into the hash table.  Rows are the oracle's either way; a
    second ctx with the sink pinned (FA_WIDE=scatter: no probe) has the same rows and no log."""
    import torch
    monkeypatch.delenv("FA_WIDE", raising=False)
    n = 4_400_000
    probe = (1 << 20) + (1 << 17)
    kw = dict(mode=2, framed=1, seed=77, n_total=n, span_secs=600, zipf_log2_universe=22, zipf_s_x100=80)
    gp = po.gen_params(**kw)
    mp = fa.mock_params(**kw)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    want = po.rollup_app(rows, status, 300).astype(fa.ROW_APP_DTYPE)
    assert len(want) > 0.9 * n
    ref = po.Rollup(300)
    ref.ingest(buf, off, 1)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    dev = torch.device("cuda", 0)
    d_buf = torch.empty(n * 96 + 4096, dtype=torch.uint8, device=dev)
    d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
    for pinned in (False, True):
        if pinned:
            monkeypatch.setenv("FA_WIDE", "scatter")
        with fa.FlowAgg(framed=True, key_sets=ks, max_batch_records=n) as agg:
            wbytes = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
            assert wbytes == len(buf)
            agg.ingest_device(d_buf.data_ptr(), wbytes, d_off.data_ptr(), n)
            agg.sync()
            st = agg.stats()
            assert st["records_ok"] == n and st["bytes_in"] == wbytes
            first = st["wave_tile_launches"]
            if pinned:
                assert st["wide_log_recorded"] == 0 and st["wide_log_mode"] == 0
            else:
                assert first >= 2
                assert st["wide_log_mode"] == 1 and st["wide_log_recorded"] >= 1 and st["wide_log_records"] == n - probe
            got = agg.read_window_app()
            assert got.tobytes() == want.tobytes()
            assert agg.read_window().tobytes() == ref.rows().tobytes()
            # the next launch of the same ctx is not probed again
            agg.ingest_device(d_buf.data_ptr(), wbytes, d_off.data_ptr(), n)
            agg.sync()
            assert agg.stats()["wave_tile_launches"] == (2 * first if pinned else 2 * first - 1)
            assert agg.read_window_app().tobytes() == _scaled(want, 2).tobytes()
