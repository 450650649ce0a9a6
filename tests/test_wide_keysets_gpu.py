"""GPU parity for the wide key sets: (SrcAddr,DstPort,Proto) rollup (BASELINE config 5's second key set),
the dashboards' GROUP BY SrcPort/DstPort and per-minute series (viz-ch.json:74,358,604).  Through the
C-ABI, bit-exact against the oracle restatements (oracle/pyoracle.py) on the same seeded inputs."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def concat(records):
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in records])
    return np.frombuffer(b"".join(records), dtype=np.uint8), off


def _same(got, want, cols):
    assert len(got) == len(want), (len(got), len(want))
    for c in cols:
        assert np.array_equal(got[c], want[c]), (c, np.nonzero((got[c] != want[c]).reshape(len(got), -1).any(axis=1))[0][:10])


APP_COLS = ("date", "timeslot", "src_addr", "dst_port", "proto", "bytes", "packets", "count")


@pytest.mark.parametrize("mode,n,zs", [(0, 20000, 110), (1, 150000, 110), (2, 200000, 80), (2, 60000, 110)])
def test_app_rollup_concurrent_with_as_rollup(gpu_lib, fa, po, mode, n, zs):
    """config 5 shape: both key sets in one pass; each must equal its oracle."""
    gp = po.gen_params(mode=mode, framed=1, seed=50 + mode, n_total=n, span_secs=900, per_sec=60, zipf_s_x100=zs)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    ref = po.Rollup(300)
    ref.ingest(buf, off, 1)
    want_app = po.rollup_app(rows, status, 300)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    with fa.FlowAgg(framed=True, key_sets=ks, wide_capacity_log2=12) as agg:  # small table: growth is exercised
        h = n // 3
        agg.ingest(buf[:int(off[h])], off[:h + 1])
        agg.ingest(buf[int(off[h]):], off[h:] - off[h])
        got_app = agg.read_window_app()
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        st = agg.stats()
        assert st["wide_used"] == len(want_app) and st["wide_capacity"] >= 2 * len(want_app)
        _same(got_app, want_app, APP_COLS)
        # per-window close removes exactly that window
        slots = np.unique(want_app["timeslot"])
        first = agg.close_window_app(int(slots[0]))
        _same(first, want_app[want_app["timeslot"] == slots[0]], APP_COLS)
        rest = agg.read_window_app()
        _same(rest, want_app[want_app["timeslot"] != slots[0]], APP_COLS)
        assert agg.read_window().tobytes() == ref.rows().tobytes()  # the flows_5m table is untouched


def test_app_rollup_sliding_subwindows(gpu_lib, fa, po):
    """1-minute sub-buckets, 5-minute windows sliding by 60 s (config 5): every window == oracle."""
    n = 100000
    gp = po.gen_params(mode=2, framed=1, seed=55, n_total=n, span_secs=900, zipf_s_x100=80)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    with fa.FlowAgg(framed=True, key_sets=ks, window_secs=300, subwindow_secs=60) as agg:
        agg.ingest(buf, off)
        for start in (po.T0, po.T0 + 60, po.T0 + 420, po.T0 + 600):
            want = po.rollup_app(rows, status, 60, window=300, timeslot=start)
            _same(agg.read_window_app(start), want, APP_COLS)
        # closing a sliding window drops only its oldest sub-bucket
        agg.close_window_app(po.T0)
        want = po.rollup_app(rows, status, 60, window=300, timeslot=po.T0 + 60)
        _same(agg.read_window_app(po.T0 + 60), want, APP_COLS)
        assert len(agg.read_window_app(po.T0)) == len(po.rollup_app(rows[rows["time_received"] >= po.T0 + 60],
                                                                    status[rows["time_received"] >= po.T0 + 60], 60, window=300,
                                                                    timeslot=po.T0))


def test_dashboard_ports_and_minutes(gpu_lib, fa, po):
    n = 150000
    gp = po.gen_params(mode=2, framed=1, seed=57, n_total=n, span_secs=600)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    ks = fa.FA_KEYS_PORT_HIST | fa.FA_KEYS_MINUTE_SERIES
    with fa.FlowAgg(framed=True, key_sets=ks) as agg:
        agg.ingest(buf, off)
        for d in (0, 1):
            want = po.top_ports(rows, status, d)
            got = agg.top_ports(d)
            _same(got, want, ("port", "weight", "count"))
            _same(agg.top_ports(d, 10), want[:10], ("port", "weight", "count"))
        _same(agg.minute_series(), po.minute_series(rows, status), ("minute", "weight", "count"))
        assert len(agg.minute_series()) == 10
        agg.dashboard_reset()
        assert len(agg.top_ports(0)) == 0 and len(agg.minute_series()) == 0
        agg.ingest(buf, off)  # usable again after the reset
        _same(agg.minute_series(), po.minute_series(rows, status), ("minute", "weight", "count"))


def _enc(fa, fields):
    out = bytearray()
    for f, v in fields:
        if isinstance(v, (bytes, bytearray)):
            out += fa.schema.encode_varint((f << 3) | 2) + fa.schema.encode_varint(len(v)) + bytes(v)
        else:
            out += fa.schema.encode_varint(f << 3) + fa.schema.encode_varint(int(v))
    return bytes(out)


def test_wide_keysets_edge_values_and_bad_records(gpu_lib, fa, po):
    """UInt32 ports >= 65536, UInt64 products that wrap, all-zero weights, TimeFlowStart beyond 2^32,
    non-canonical order (deferred parsers), malformed records (dropped), every key set at once."""
    rng = np.random.default_rng(5)
    recs = []
    t0 = po.T0
    for i in range(6000):
        k = i % 12
        sport = [443, 65535, 65536, 70000, 0xFFFFFFFF, 0][i % 6]
        dport = [53, 65536 + (i % 7), 80, 0xFFFFFFFE][i % 4]
        by = [0, 1500, (1 << 63) + 3, (1 << 40)][i % 4]
        sr = [1, 0, 1000, (1 << 30)][(i // 4) % 4]
        tfs = t0 + (i % 400) if k != 5 else (1 << 32) + t0 + i  # narrowing to DateTime (create.sh:40)
        addr = bytes([10, i % 3, 0, i % 5]) if i % 2 else bytes([0x20, 1, 0xd, 0xb8] + [0] * 11 + [i % 9])
        fields = [(2, t0 + (i % 700)), (3, sr), (6, addr), (9, by), (10, i % 100), (14, 64512 + i % 4), (15, 64600),
                  (20, [6, 17, 0, 0xFFFFFFFF][i % 4]), (21, sport), (22, dport), (30, 0x0800 if i % 2 else 0x86dd), (38, tfs)]
        fields = [(f, v) for f, v in fields if isinstance(v, bytes) or v != 0]
        if k == 3:
            fields = fields[::-1]
        if k == 7:
            fields = fields + [(22, 1), (22, dport)]
            fields = [(f, v) for f, v in fields if isinstance(v, bytes) or v != 0]
        r = _enc(fa, fields)
        if k == 9:
            r = r[:-1] + b"\xff"  # truncated varint: malformed
        recs.append(r)
    buf, off = concat(recs)
    rows, status = po.decode_batch(buf, off, 0)
    assert status.sum() > 0
    ref = po.Rollup(300)
    bad = ref.ingest(buf, off, 0)
    ks = 63
    with fa.FlowAgg(framed=False, key_sets=ks) as agg:
        agg.ingest(buf, off)
        st = agg.stats()
        assert st["records_bad"] == bad == int(status.sum())
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        _same(agg.read_window_app(), po.rollup_app(rows, status, 300), APP_COLS)
        for d in (0, 1):
            _same(agg.top_ports(d), po.top_ports(rows, status, d), ("port", "weight", "count"))
        _same(agg.minute_series(), po.minute_series(rows, status), ("minute", "weight", "count"))
        # the sketches ride along in the same pass
        cms = agg.cms_read(fa.FA_KEYS_SRCADDR_CMS)
        ok = rows[status == 0]
        with np.errstate(over="ignore"):
            assert int(cms[0].sum(dtype=np.uint64)) == int((ok["bytes"] * ok["sampling_rate"]).sum(dtype=np.uint64))


def test_wide_shard_merge_equals_single_shard(gpu_lib, fa, po):
    """Two Kafka partitions on two contexts, merged at window close == one context (multi-GPU merge path)."""
    n = 80000
    gp = po.gen_params(mode=2, framed=1, seed=58, n_total=n, zipf_s_x100=80)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)
    rows, status = po.decode_batch(buf, off, 1)
    ks = fa.FA_KEYS_ADDR_PORT_PROTO | fa.FA_KEYS_PORT_HIST | fa.FA_KEYS_MINUTE_SERIES
    shards = []
    for p in range(2):
        b, o = concat([raw[int(off[k]):int(off[k + 1])] for k in range(p, n, 2)])
        agg = fa.FlowAgg(framed=True, key_sets=ks)
        agg.ingest(b, o)
        shards.append(agg)
    a, b = shards
    a.merge_rows_app(b.read_window_app())
    for d in (0, 1):
        a.merge_ports(d, b.top_ports(d))
    a.merge_minutes(b.minute_series())
    _same(a.read_window_app(), po.rollup_app(rows, status, 300), APP_COLS)
    for d in (0, 1):
        _same(a.top_ports(d), po.top_ports(rows, status, d), ("port", "weight", "count"))
    _same(a.minute_series(), po.minute_series(rows, status), ("minute", "weight", "count"))
    a.close()
    b.close()


def test_config5_shape_at_scale(gpu_lib, fa, po):
    """BASELINE config 5 shape at 3 M records generated in HBM (Zipf 0.8, every key set at once, 60 s sub-buckets):
    full grids, table growth from 2^20 slots, both exact rollups, ports and minutes against the oracle."""
    import torch
    n = 3_000_000
    kw = dict(mode=2, framed=1, seed=5, n_total=n, span_secs=900, zipf_log2_universe=22, zipf_s_x100=80)
    gp = po.gen_params(**kw)
    mp = fa.mock_params(**kw)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO | fa.FA_KEYS_PORT_HIST | fa.FA_KEYS_MINUTE_SERIES
    dev = torch.device("cuda", 0)
    with fa.FlowAgg(framed=True, key_sets=ks, window_secs=300, subwindow_secs=60, max_batch_records=n) as agg:
        d_buf = torch.empty(n * 96 + 4096, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        wbytes = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
        assert wbytes == len(buf)
        agg.ingest_device(d_buf.data_ptr(), wbytes, d_off.data_ptr(), n)
        want = po.rollup_app(rows, status, 60)
        got = agg.read_window_app()
        _same(got, want, APP_COLS)
        assert agg.stats()["wide_capacity"] >= 2 * len(want)
        start = po.T0 + 120  # a sliding 5-minute window
        _same(agg.read_window_app(start), po.rollup_app(rows, status, 60, window=300, timeslot=start), APP_COLS)
        ref = po.Rollup(60)
        ref.ingest(buf, off, 1)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        for d in (0, 1):
            _same(agg.top_ports(d), po.top_ports(rows, status, d), ("port", "weight", "count"))
        _same(agg.minute_series(), po.minute_series(rows, status), ("minute", "weight", "count"))


def _app_device_run(fa, po, n, chunk, **kw):
    """n records generated in HBM, ingested in chunks with key sets flows_5m + (SrcAddr,DstPort,Proto); returns
    (flows_5m rows, app rows, stats)."""
    import torch
    mp = fa.mock_params(**kw)
    dev = torch.device("cuda", 0)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_ADDR_PORT_PROTO
    with fa.FlowAgg(framed=True, key_sets=ks, wide_capacity_log2=16, max_batch_records=chunk) as agg:
        d_buf = torch.empty(chunk * 96 + 4096, dtype=torch.uint8, device=dev)
        d_off = torch.empty(chunk + 1, dtype=torch.int32, device=dev)
        for i0 in range(0, n, chunk):
            m = min(chunk, n - i0)
            w = agg.mock_generate_device(mp, i0, m, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), m)
        return agg.read_window(), agg.read_window_app(), agg.stats()


@pytest.mark.parametrize("variant", ["default", "scatter", "atomic", "tiny_segments"])
@pytest.mark.parametrize("zs,lu", [(80, 22), (140, 10)])
def test_app_scatter_sink_equals_oracle(gpu_lib, fa, po, monkeypatch, variant, zs, lu):
    """The (SrcAddr,DstPort,Proto) scatter sink (wagg.cuh: tuples per table region, one workgroup per region, plain
    loads and stores) against the oracle on 2 M records in four launches, with table growth from 2^16 slots between
    them: almost-all-distinct keys (Zipf 0.8 over 2^22 addresses) and heavy duplicates (Zipf 1.4 over 2^10: the LDS
    deduplication and the segment-overflow fallback of a heavy key's region).  default = the library chooses per launch
    (scatter sink while many records open new rows, atomics otherwise); the same with the sink forced on
    (FA_WIDE=scatter), off (FA_WIDE=atomic) and with segments far too small (FA_SEG_CAP: most tuples overflow into
    the atomic path)."""
    if variant in ("atomic", "scatter"):
        monkeypatch.setenv("FA_WIDE", variant)
    if variant == "tiny_segments":
        monkeypatch.setenv("FA_WIDE", "scatter")
        monkeypatch.setenv("FA_SEG_CAP", "40")
    n, chunk = 2_000_000, 500_000
    kw = dict(mode=2, framed=1, seed=77, n_total=n, span_secs=900, zipf_log2_universe=lu, zipf_s_x100=zs)
    gp = po.gen_params(**kw)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    ref = po.Rollup(300)
    ref.ingest(buf, off, 1)
    want_app = po.rollup_app(rows, status, 300)
    got, got_app, st = _app_device_run(fa, po, n, chunk, **kw)
    assert got.tobytes() == ref.rows().tobytes()
    _same(got_app, want_app, APP_COLS)
    # (wide_used = rows of the hash table: the library's own choice may keep a stream that opens a row per record in its log)
    assert (st["wide_used"] <= len(want_app) if variant == "default" else st["wide_used"] == len(want_app)) and st["records_ok"] == n


def test_one_windows_rows_as_48_byte_rows(gpu_lib, fa, po):
    """fa_read_window_app48 / fa_close_window_app48: the rows of fa_read_window_app without the date / timeslot they share -
    tumbling and sliding windows, table and log, the close removes what fa_close_window_app removes."""
    n = 200_000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=4801, n_total=n, zipf_log2_universe=14, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    for sub in (0, 60):
        with fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub) as agg, fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub) as twin:
            agg.ingest(buf, off)
            twin.ingest(buf, off)
            t0 = int(agg.open_timeslots()[0])
            for ts in [t0, t0 + 300] + ([t0 + 120] if sub else []):
                want = agg.read_window_app(ts)
                got, date = agg.read_window_app48(ts)
                assert len(got) == len(want) > 0 and date == ts // 86400
                for f in ("src_addr", "dst_port", "proto", "bytes", "packets", "count"):
                    assert np.array_equal(got[f], want[f]), f
                assert (want["timeslot"] == ts).all() and (want["date"] == date).all()
            with pytest.raises(fa.FlowAggError):
                agg.read_window_app48(fa.ALL_TIMESLOTS)
            got, _ = agg.read_window_app48(t0, close=True)
            assert np.array_equal(got["count"], twin.close_window_app(t0)["count"])
            assert agg.read_window_app().tobytes() == twin.read_window_app().tobytes()
            small = np.empty(3, dtype=fa.ROW_APP48_DTYPE)  # too small a buffer: the binding asks again with room
            assert len(agg.read_window_app48(t0 + 300, out=small)[0]) == len(twin.read_window_app(t0 + 300))


@pytest.mark.parametrize("sub,mode", [(0, "zipf"), (60, "zipf"), (0, "one_block")])
def test_48_byte_rows_in_two_halves_into_page_locked_memory(gpu_lib, fa, po, monkeypatch, sub, mode):
    """A large window read into PAGE-LOCKED memory leaves in two halves cut by key (the first half's copy overlaps the second half's
    sort; FA_APP48_SPLIT sets the threshold - 2^22 rows in production): byte-identical to the one-piece read, for tumbling and
    sliding windows, with sixteen addresses in all (a lopsided cut), when the buffer is too small (the
    caller learns how many rows to make room for), and for the close."""
    monkeypatch.setenv("FA_APP48_SPLIT", "8192")
    n = 300_000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=4901, n_total=n, zipf_log2_universe=16 if mode == "zipf" else 4, span_secs=600)
    buf, off = po.gen_records(gp, 0, n)

    def pinned48(rows):
        return fa.FlowAgg.pinned_rows(fa.ROWS_APP, rows).view(np.uint8)[: rows * fa.ROW_APP48_DTYPE.itemsize].view(fa.ROW_APP48_DTYPE)

    with fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub, wide_capacity_log2=20) as agg, \
            fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub, wide_capacity_log2=20) as twin:
        agg.ingest(buf, off)
        twin.ingest(buf, off)
        t0 = int(agg.open_timeslots()[0])
        for ts in [t0, t0 + 300] + ([t0 + 120] if sub else []):
            want, date_w = twin.read_window_app48(ts)  # (pageable buffer: in one piece)
            assert len(want) > 8192 or mode == "one_block"
            out = pinned48(len(want) + 10)
            got, date = agg.read_window_app48(ts, out=out)
            assert date == date_w and got.tobytes() == want.tobytes(), (ts, len(got), len(want))
            if len(want) > 8192:
                with pytest.raises(fa.FlowAggError) as e:  # straight through the C-ABI: the binding would ask again by itself
                    agg._chk(agg._L.fa_read_window_app48(agg._h, ts, pinned48(len(want) // 3).ctypes.data, len(want) // 3, C.byref(C.c_size_t()), C.byref(C.c_uint32())))
                assert e.value.code == -6
                assert agg.read_window_app48(ts, out=pinned48(len(want) // 3))[0].tobytes() == want.tobytes()  # ... and gets every row
        want, _ = twin.read_window_app48(t0, close=True)
        got, _ = agg.read_window_app48(t0, out=pinned48(len(want)), close=True)
        assert got.tobytes() == want.tobytes()
        assert agg.read_window_app().tobytes() == twin.read_window_app().tobytes()
