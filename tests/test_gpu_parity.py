"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the
same seeded inputs.  Bit-exact everywhere (integer / byte work)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

COLS = ["time_received", "time_flow_start", "sampling_rate", "bytes", "packets", "sequence_num",
        "src_as", "dst_as", "etype", "proto", "src_port", "dst_port", "sampler_address",
        "src_addr", "dst_addr"]


def oracle_rows(po, buf, off, framed):
    """Decode every record with the oracle -> (structured array, status array)."""
    return po.decode_batch(buf, off, framed)


def assert_decode_equal(got, want, wstatus):
    assert np.array_equal(got["status"], wstatus), np.nonzero(got["status"] != wstatus)[0][:10]
    for c in COLS:
        assert np.array_equal(got[c], want[c]), (c, np.nonzero(got[c] != want[c])[0][:10])


def concat(records):
    off = np.zeros(len(records) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in records])
    return np.frombuffer(b"".join(records), dtype=np.uint8), off


@pytest.mark.parametrize("mode,framed", [(0, 1), (0, 0), (1, 1), (1, 0), (2, 1)])
def test_decode_matches_oracle(gpu_lib, fa, po, mode, framed):
    n = 30000
    gp = po.gen_params(mode=mode, framed=framed, seed=5 + mode, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    want, wstatus = oracle_rows(po, buf, off, framed)
    assert wstatus.sum() == 0
    # the generator's own truth table agrees too
    truth = po.gen_rows(gp, 0, 64)
    for c in COLS:
        assert np.array_equal(truth[c], want[c][:64]), c
    with fa.FlowAgg(framed=bool(framed)) as agg:
        got = agg.decode(buf, off)
    assert_decode_equal(got, want, wstatus)


def test_edge_cases_golden(gpu_lib, fa, po):
    """SURVEY Appendix A.2 + upb-pinned edge vectors: every record, bare and framed."""
    cases = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))["cases"]
    recs = [bytes.fromhex(c["hex"]) for c in cases]
    for framed in (0, 1):
        rr = [fa.schema.frame(r) for r in recs] if framed else recs
        buf, off = concat(rr)
        with fa.FlowAgg(framed=bool(framed)) as agg:
            got = agg.decode(buf, off)
        for k, c in enumerate(cases):
            if c["expect"] is None:
                assert got["status"][k] == 1, (c["name"], "should be a bad record")
                continue
            assert got["status"][k] == 0, (c["name"], "should decode")
            e = c["expect"]
            assert int(got["time_received"][k]) == e["TimeReceived"], c["name"]
            assert int(got["src_as"][k]) == e["SrcAS"], c["name"]
            assert int(got["dst_as"][k]) == e["DstAS"], c["name"]
            assert int(got["bytes"][k]) == e["Bytes"], c["name"]
            assert int(got["packets"][k]) == e["Packets"], c["name"]
            assert int(got["etype"][k]) == e["EType"], c["name"]
            assert bytes(got["src_addr"][k]).hex() == e["SrcAddr"], c["name"]
            assert bytes(got["dst_addr"][k]).hex() == e["DstAddr"], c["name"]
            assert bytes(got["sampler_address"][k]).hex() == e["SamplerAddress"], c["name"]
            for col, key in (("time_flow_start", "TimeFlowStart"), ("sequence_num", "SequenceNum"),
                             ("sampling_rate", "SamplingRate"), ("proto", "Proto"),
                             ("src_port", "SrcPort"), ("dst_port", "DstPort")):
                assert int(got[col][k]) == e[key], (c["name"], key)


def test_fuzzed_records_match_oracle(gpu_lib, fa, po):
    """Mutation-fuzzed records (committed fixture): decode + status identical to the oracle."""
    blob = np.load(os.path.join(GOLDEN, "fuzz_records.npz"))
    buf, off = blob["buf"], blob["off"]
    want, wstatus = oracle_rows(po, buf, off, 0)
    assert 0 < wstatus.sum() < len(wstatus)
    with fa.FlowAgg(framed=False) as agg:
        got = agg.decode(buf, off)
    assert_decode_equal(got, want, wstatus)
    # and the expectations recorded from upb when the fixture was made
    assert np.array_equal(wstatus, blob["upb_status"])


@pytest.mark.parametrize("mode,n,per_sec", [(0, 10000, 4), (0, 200000, 300), (1, 300000, 0), (2, 100000, 0)])
def test_rollup_matches_oracle(gpu_lib, fa, po, mode, n, per_sec):
    gp = po.gen_params(mode=mode, framed=1, seed=21 + mode, n_total=n, per_sec=per_sec or 4)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    want = ref.rows()
    with fa.FlowAgg(framed=True, table_capacity_log2=16) as agg:  # small table: exercises growth
        agg.ingest(buf, off)
        got = agg.read_window()
        st = agg.stats()
        assert st["records_ok"] == n and st["records_bad"] == 0
        assert got.tobytes() == want.tobytes()
        # per-window close returns the same rows and empties the table
        slots = agg.open_timeslots()
        assert list(slots) == sorted(set(want["timeslot"]))
        parts = [agg.close_window(int(ts)) for ts in slots]
        assert np.concatenate(parts).tobytes() == want.tobytes()
        assert len(agg.read_window()) == 0


def test_rollup_config1_bare_9_groups(gpu_lib, fa, po):
    """BASELINE config 1 shape: 10k bare mocker records -> 9 (SrcAS,DstAS) groups."""
    n = 10000
    gp = po.gen_params(mode=0, framed=0, seed=1, n_total=n, per_sec=4)
    buf, off = po.gen_records(gp, 0, n)
    with fa.FlowAgg(framed=False) as agg:
        agg.ingest(buf, off)
        rows = agg.read_window()
    pairs = {}
    for r in rows:
        k = (int(r["src_as"]), int(r["dst_as"]))
        b, p, c = pairs.get(k, (0, 0, 0))
        pairs[k] = (b + int(r["bytes"]), p + int(r["packets"]), c + int(r["count"]))
    assert len(pairs) == 9
    assert sum(v[2] for v in pairs.values()) == n
    truth = po.gen_rows(gp, 0, n)
    assert sum(v[0] for v in pairs.values()) == int(truth["bytes"].sum())
    assert sum(v[1] for v in pairs.values()) == int(truth["packets"].sum())


def test_rollup_with_bad_and_exotic_records(gpu_lib, fa, po):
    """Malformed records are counted and dropped (inserter.go:125-126); exotic but valid
    encodings (groups, 10-byte varints, long tags) take the generic device parser."""
    blob = np.load(os.path.join(GOLDEN, "fuzz_records.npz"))
    buf, off = blob["buf"], blob["off"]
    ref = po.Rollup(300)
    bad = ref.ingest(buf, off, 0)
    with fa.FlowAgg(framed=False) as agg:
        agg.ingest(buf, off)
        got = agg.read_window()
        st = agg.stats()
    assert st["records_bad"] == bad
    assert st["records_ok"] == len(off) - 1 - bad
    assert st["records_slow"] >= bad
    assert got.tobytes() == ref.rows().tobytes()


def test_stream_without_offsets(gpu_lib, fa, po):
    """offsets=NULL: the framed stream is split on the host by the varint prefixes."""
    n = 5000
    gp = po.gen_params(mode=1, framed=1, seed=9, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    ref = po.Rollup(300)
    ref.ingest(buf, off, 1)
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(buf, None)
        assert agg.read_window().tobytes() == ref.rows().tobytes()


def test_large_records_beyond_lds_tile(gpu_lib, fa, po):
    """Records bigger than the 32 KiB LDS tile (huge unknown LEN field) and tiles whose
    256 records do not fit one pass."""
    big_unknown = fa.schema.encode_varint((1000 << 3) | 2) + fa.schema.encode_varint(40000) + b"\xab" * 40000
    mid_unknown = fa.schema.encode_varint((1001 << 3) | 2) + fa.schema.encode_varint(300) + b"\xcd" * 300
    gp = po.gen_params(mode=1, framed=0, seed=33, n_total=2000)
    buf, off = po.gen_records(gp, 0, 2000)
    raw = bytes(buf)
    recs = []
    for k in range(2000):
        r = raw[int(off[k]):int(off[k + 1])]
        if k % 500 == 7:
            r = big_unknown + r
        elif k % 3 == 0:
            r = r + mid_unknown
        recs.append(fa.schema.frame(r))
    b2, o2 = concat(recs)
    ref = po.Rollup(300)
    assert ref.ingest(b2, o2, 1) == 0
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(b2, o2)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        assert agg.stats()["records_ok"] == 2000


def test_device_generator_matches_oracle(gpu_lib, fa, po):
    import torch
    for mode, framed in [(0, 1), (1, 1), (2, 1), (0, 0)]:
        n = 50000
        gp = po.gen_params(mode=mode, framed=framed, seed=77, n_total=n, per_sec=7)
        want_buf, want_off = po.gen_records(gp, 1234, n)
        mp = fa.mock_params(mode=mode, framed=framed, seed=77, n_total=n, per_sec=7)
        d_buf = torch.empty(n * 96 + 256, dtype=torch.uint8, device="cuda")
        d_off = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        with fa.FlowAgg(framed=bool(framed)) as agg:
            w = agg.mock_generate_device(mp, 1234, n, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
        assert w == len(want_buf)
        assert np.array_equal(d_off.cpu().numpy().astype(np.uint64), want_off)
        assert np.array_equal(d_buf[:w].cpu().numpy(), want_buf)
        hb, ho = fa.mock_generate_host(mp, 1234, n)
        assert np.array_equal(hb, want_buf) and np.array_equal(ho, want_off)


def test_cms_matches_oracle(gpu_lib, fa, po):
    n = 60000
    gp = po.gen_params(mode=2, framed=1, seed=3, n_total=n, zipf_log2_universe=16)
    buf, off = po.gen_records(gp, 0, n)
    truth = po.gen_rows(gp, 0, n)
    depth, wl2, seed = 4, 12, 0xC0FFEE
    want_src = np.zeros(depth << wl2, dtype=np.uint64)
    want_dst = np.zeros(depth << wl2, dtype=np.uint64)
    for k in range(n):
        w = (int(truth["bytes"][k]) * int(truth["sampling_rate"][k])) & (2**64 - 1)
        po.cms_update(want_src, depth, wl2, seed, bytes(truth["src_addr"][k]), w)
        po.cms_update(want_dst, depth, wl2, seed, bytes(truth["dst_addr"][k]), w)
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_SRCADDR_CMS | fa.FA_KEYS_DSTADDR_CMS
    with fa.FlowAgg(framed=True, key_sets=ks, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed) as agg:
        agg.ingest(buf, off)
        got_src = agg.cms_read(fa.FA_KEYS_SRCADDR_CMS).reshape(-1)
        got_dst = agg.cms_read(fa.FA_KEYS_DSTADDR_CMS).reshape(-1)
        key = bytes(truth["src_addr"][0])
        assert agg.cms_query(fa.FA_KEYS_SRCADDR_CMS, key) == po.cms_query(want_src, depth, wl2, seed, key)
        ref = po.Rollup(300)
        ref.ingest(buf, off, 1)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
    assert np.array_equal(got_src, want_src)
    assert np.array_equal(got_dst, want_dst)


def test_merge_rows_equals_single_shard(gpu_lib, fa, po):
    """Partition i mod P -> P contexts; merging their rows == the single-shard result."""
    n, P = 80000, 4
    gp = po.gen_params(mode=1, framed=1, seed=8, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)
    recs = [raw[int(off[k]):int(off[k + 1])] for k in range(n)]
    ref = po.Rollup(300)
    ref.ingest(buf, off, 1)
    with fa.FlowAgg(framed=True) as total:
        for p in range(P):
            b, o = concat(recs[p::P])
            with fa.FlowAgg(framed=True) as shard:
                shard.ingest(b, o)
                total.merge_rows(shard.read_window())
        assert total.read_window().tobytes() == ref.rows().tobytes()


def test_sliding_subwindows(gpu_lib, fa, po):
    """60 s sub-buckets: any 5-minute window starting on a minute boundary equals the
    oracle's rollup of exactly those records (tumbling windows are the reference's case)."""
    n = 120000
    gp = po.gen_params(mode=1, framed=1, seed=12, n_total=n, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    truth = po.gen_rows(gp, 0, n)
    raw = bytes(buf)
    with fa.FlowAgg(framed=True, window_secs=300, subwindow_secs=60) as agg:
        agg.ingest(buf, off)
        for start in (po.T0, po.T0 + 60, po.T0 + 420):
            sel = np.nonzero((truth["time_received"] >= start) & (truth["time_received"] < start + 300))[0]
            ref = po.Rollup(86400)  # one bucket: everything selected falls in one "window"
            b, o = concat([raw[int(off[k]):int(off[k + 1])] for k in sel])
            ref.ingest(b, o, 1)
            want = ref.rows()
            got = agg.read_window(start)
            assert len(got) == len(want)
            for col in ("src_as", "dst_as", "etype", "bytes", "packets", "count"):
                assert np.array_equal(got[col], want[col]), col
            assert (got["timeslot"] == start).all()


# ---- scatter sink / parser tiers -------------------------------------------------------------
def _enc_record(fa, fields):
    """fields: list of (field_number, int | bytes) in the order they are to be serialized."""
    out = bytearray()
    for f, v in fields:
        if isinstance(v, (bytes, bytearray)):
            out += fa.schema.encode_varint((f << 3) | 2) + fa.schema.encode_varint(len(v)) + bytes(v)
        else:
            out += fa.schema.encode_varint(f << 3) + fa.schema.encode_varint(int(v))
    return bytes(out)


def _custom_records(fa, po, n, seed):
    """Generator truth rows re-encoded with values and layouts that leave the fast paths:
    tuple-path escapes (big Bytes/Packets/Etype, far-away timestamps), non-canonical field
    order, duplicates (last wins), unknown fields."""
    gp = po.gen_params(mode=1, framed=0, seed=seed, n_total=n)
    rows = po.gen_rows(gp, 0, n)
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 20, n)
    recs = []
    for i in range(n):
        r = rows[i]
        t = int(r["time_received"])
        by, pk, et = int(r["bytes"]), int(r["packets"]), int(r["etype"])
        k = int(kind[i])
        if k == 0:
            by = [1 << 28, (1 << 28) + i, (1 << 40) + i, (1 << 63) + 5, (1 << 64) - 1][i % 5]
        elif k == 1:
            pk = [1 << 15, (1 << 15) + i, (1 << 33) + 7][i % 3]
        elif k == 2:
            et = [65536, 0x12345678, 0xFFFFFFFF][i % 3]
        elif k == 3:
            t += 300 * int(rng.integers(20, 400))  # far outside the batch's 16-bucket span
        alen = 16 if et == 0x86dd else 4
        fields = [(2, t), (3, int(r["sampling_rate"])), (4, int(r["sequence_num"])),
                  (6, bytes(r["src_addr"][:alen])), (7, bytes(r["dst_addr"][:alen])), (9, by), (10, pk),
                  (14, int(r["src_as"])), (15, int(r["dst_as"])), (21, int(r["src_port"])),
                  (22, int(r["dst_port"])), (30, et), (38, int(r["time_flow_start"]))]
        fields = [(f, v) for f, v in fields if isinstance(v, bytes) or v != 0]  # proto3 zero omission
        if k in (4, 5, 6):
            fields = fields[::-1]                                    # any order is legal protobuf
        elif k == 7:
            fields = fields + [(14, 4242), (14, int(r["src_as"]))]  # duplicates: last one wins
        elif k == 8:
            fields = [(1000, 77)] + fields + [(1001, b"xyz")]       # unknown fields are skipped
        elif k == 9:
            fields = [(1, 3), (5, t + 9), (11, b"\x0a\x00\x00\x01"), (18, 17), (19, 18), (23, 1), (26, 0x12)] + fields
            fields.sort(key=lambda fv: fv[0])                        # full GoFlow-style record, canonical
        recs.append(_enc_record(fa, fields))
    return recs


@pytest.mark.parametrize("sink", ["auto", "scatter", "direct"])
def test_rollup_escapes_and_noncanonical_records(gpu_lib, fa, po, monkeypatch, sink):
    n = 48000 if sink == "auto" else 9000
    monkeypatch.setenv("FA_SINK", sink)
    recs = [fa.schema.frame(r) for r in _custom_records(fa, po, n, seed=77)]
    buf, off = concat(recs)
    ref = po.Rollup(300)
    assert ref.ingest(buf, off, 1) == 0
    want = ref.rows()
    with fa.FlowAgg(framed=True) as agg:
        agg.ingest(buf, off)
        got = agg.read_window()
        st = agg.stats()
    assert st["records_ok"] == n and st["records_bad"] == 0
    assert got.tobytes() == want.tobytes()
    assert st["records_retried"] > 0.15 * n          # reversed / duplicated / unknown-field records
    if sink != "direct":
        assert 0 < st["records_direct"] < 0.5 * n    # tuple escapes took the device-wide-table path


def test_decode_noncanonical_records(gpu_lib, fa, po):
    recs = _custom_records(fa, po, 20000, seed=78)
    buf, off = concat(recs)
    want, wstatus = oracle_rows(po, buf, off, 0)
    assert wstatus.sum() == 0
    with fa.FlowAgg(framed=False) as agg:
        got = agg.decode(buf, off)
    assert_decode_equal(got, want, wstatus)


@pytest.mark.parametrize("tile", ["wave", "wg", ""])
def test_tile_kernel_variants_agree_with_oracle(gpu_lib, fa, po, monkeypatch, tile):
    """Both ingest kernels (wave-private tiles + LDS tuple bins: the default / 256-thread workgroup tiles) over
    several batches incl. escapes, hot keys and a record size that shrinks the tiles."""
    if tile:
        monkeypatch.setenv("FA_TILE", tile)
    else:
        monkeypatch.delenv("FA_TILE", raising=False)
    n = 400000
    gp = po.gen_params(mode=1, framed=1, seed=31, n_total=n, span_secs=1500)
    buf, off = po.gen_records(gp, 0, n)
    recs = _custom_records(fa, po, 40000, 32)
    big = [_enc_record(fa, [(2, po.T0 + 5), (14, 70000 + (i % 50)), (15, 9), (9, 10), (1000, b"z" * 300)]) for i in range(40000)]
    ref = po.Rollup(300)
    with fa.FlowAgg(framed=True) as agg:
        for lo, hi in ((0, n // 2), (n // 2, n)):
            b, o = buf[int(off[lo]):int(off[hi])], off[lo:hi + 1] - off[lo]
            agg.ingest(b, o)
            ref.ingest(b, o, 1)
            agg.sync()
        st = agg.stats()
        assert st["wave_tile_launches"] == (0 if tile == "wg" else 2)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
    ref = po.Rollup(300)
    with fa.FlowAgg(framed=False) as agg:
        for rr in (recs, big, recs):
            b, o = concat(rr)
            agg.ingest(b, o)
            ref.ingest(b, o, 0)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        assert agg.stats()["records_ok"] == 120000


def test_scatter_sink_hot_keys_and_many_batches(gpu_lib, fa, po, monkeypatch):
    """Same context, several batches, skewed keys (mocker: 9 groups) and uniform keys mixed:
    segment overflow, the hot-key table and repeated aggregation passes all stay exact."""
    monkeypatch.setenv("FA_SINK", "scatter")
    ref = po.Rollup(300)
    with fa.FlowAgg(framed=True, table_capacity_log2=14) as agg:
        for step, (mode, n) in enumerate([(0, 70000), (1, 90000), (2, 40000), (1, 50000)]):
            gp = po.gen_params(mode=mode, framed=1, seed=100 + step, n_total=n, per_sec=300)
            buf, off = po.gen_records(gp, 0, n)
            assert ref.ingest(buf, off, 1) == 0
            agg.ingest(buf, off)
        got = agg.read_window()
        st = agg.stats()
    assert st["records_ok"] == 250000
    assert got.tobytes() == ref.rows().tobytes()


def test_topk_matches_oracle(gpu_lib, fa, po):
    """fa_topk = every distinct address ranked by its Count-Min estimate of sum(Bytes*SamplingRate)
    (viz-ch.json:233,479): identical to the CPU sketch + exhaustive ranking, and an over-estimate of
    the exact GROUP BY the dashboard runs."""
    n = 60000
    gp = po.gen_params(mode=2, framed=1, seed=31, n_total=n, zipf_log2_universe=11)
    buf, off = po.gen_records(gp, 0, n)
    truth = po.gen_rows(gp, 0, n)
    depth, wl2, seed = 4, 10, 0xBEEF
    for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
        cms = np.zeros(depth << wl2, dtype=np.uint64)
        exact = {}
        for k in range(n):
            w = (int(truth["bytes"][k]) * int(truth["sampling_rate"][k])) & (2**64 - 1)
            key = bytes(truth[col][k])
            po.cms_update(cms, depth, wl2, seed, key, w)
            exact[key] = (exact.get(key, 0) + w) & (2**64 - 1)
        want = sorted(((po.cms_query(cms, depth, wl2, seed, key), key) for key in exact), key=lambda t: (-t[0], t[1]))
        with fa.FlowAgg(framed=True, key_sets=fa.FA_KEYS_AS_PAIR | ks, cms_depth=depth, cms_width_log2=wl2,
                        cms_seed=seed, topk_capacity_log2=14) as agg:
            agg.ingest(buf, off)
            got = agg.topk(ks, 100)
            everything = agg.topk(ks, 1 << 20)
            # candidates from "another partition": unseen keys join the set and are ranked by the sketch
            extra = np.frombuffer(bytes(range(32)), dtype=np.uint8).reshape(2, 16)
            agg.topk_merge_keys(ks, extra)
            merged = agg.topk(ks, 1 << 20)
        assert len(got) == 100 and len(everything) == len(exact)
        for row, (w, key) in zip(got, want[:100]):
            assert bytes(row["key"]) == key and int(row["weight"]) == w
        assert all(int(r["weight"]) >= exact[bytes(r["key"])] for r in everything)  # CMS never under-estimates
        assert len(merged) == len(exact) + 2
        for r in merged:
            if bytes(r["key"]) in (bytes(extra[0]), bytes(extra[1])):
                assert int(r["weight"]) == po.cms_query(cms, depth, wl2, seed, bytes(r["key"]))


def test_topk_reports_overflow(gpu_lib, fa, po):
    n = 40000
    gp = po.gen_params(mode=2, framed=1, seed=32, n_total=n, zipf_log2_universe=16)
    buf, off = po.gen_records(gp, 0, n)
    with fa.FlowAgg(framed=True, key_sets=fa.FA_KEYS_SRCADDR_CMS, topk_capacity_log2=8) as agg:
        agg.ingest(buf, off)
        with pytest.raises(fa.FlowAggError) as ei:
            agg.topk(fa.FA_KEYS_SRCADDR_CMS, 10)
        assert ei.value.code == -5


def _large_batch_checksum_run(fa, po, mode, n=4_000_000):
    """4 M records generated in HBM, ingested three times; rows checked against the multi-threaded oracle's
    order-independent checksum.  Returns the ctx statistics."""
    import torch

    def mix64(z):
        z = z.astype(np.uint64)
        with np.errstate(over="ignore"):
            z ^= z >> np.uint64(30)
            z *= np.uint64(0xbf58476d1ce4e5b9)
            z ^= z >> np.uint64(27)
            z *= np.uint64(0x94d049bb133111eb)
            z ^= z >> np.uint64(31)
        return z

    def checksum(rows):
        with np.errstate(over="ignore"):
            a = (rows["timeslot"].astype(np.uint64) << np.uint64(32)) | rows["etype"].astype(np.uint64)
            b = (rows["src_as"].astype(np.uint64) << np.uint64(32)) | rows["dst_as"].astype(np.uint64)
            h = mix64(a ^ mix64(b))
            v = rows["bytes"] * np.uint64(3) + rows["packets"] * np.uint64(5) + rows["count"] * np.uint64(7) + np.uint64(1)
            return int((h * v).sum(dtype=np.uint64))

    gp = po.gen_params(mode=mode, framed=1, seed=40 + mode, n_total=n, span_secs=900, per_sec=50_000)
    want = po.bench_rollup(gp, 0, n, 8)
    assert want["bad"] == 0
    mp = fa.mock_params(mode=mode, framed=1, seed=40 + mode, n_total=n, span_secs=900, per_sec=50_000)
    dev = torch.device("cuda", 0)
    with fa.FlowAgg(framed=True, max_batch_records=n) as agg:
        cap = n * 96 + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        w = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), cap, d_off.data_ptr())
        assert w == want["wire_bytes"]
        for _ in range(3):  # the same batch three times: sums triple, groups stay
            agg.ingest_device(d_buf.data_ptr(), w, d_off.data_ptr(), n)
        rows = agg.read_window()
        assert len(rows) == want["groups"]
        for col in ("bytes", "packets", "count"):
            assert (rows[col] % np.uint64(3) == 0).all()
            rows[col] //= np.uint64(3)
        assert checksum(rows) == want["checksum"]
        assert int(rows["count"].sum()) == n
        return agg.stats()


@pytest.mark.parametrize("mode,tile", [(1, ""), (2, ""), (0, ""), (1, "wg")])
def test_large_device_batches_checksum_equals_oracle(gpu_lib, fa, po, monkeypatch, mode, tile):
    """Bench-scale batches (full grids, every LDS bin cycling hundreds of times).  Small batches do not exercise
    the cross-wave bin hand-over enough to catch ordering mistakes there."""
    if tile:
        monkeypatch.setenv("FA_TILE", tile)
    st = _large_batch_checksum_run(fa, po, mode)
    assert st["wave_tile_launches"] == (0 if tile == "wg" else 3)


@pytest.mark.parametrize("tile,cap", [("", 40), ("", 48), ("wg", 40)])
def test_segment_overflow_fallbacks_stay_exact(gpu_lib, fa, po, monkeypatch, tile, cap):
    """Segments far too small for the batch (FA_SEG_CAP, a test knob): full bins that find the front part full,
    single tuples that find the back part full and the workgroup kernel's plain overflow all fall back to the
    device-wide table - slower, but the rows must not change."""
    monkeypatch.setenv("FA_SEG_CAP", str(cap))
    if tile:
        monkeypatch.setenv("FA_TILE", tile)
    st = _large_batch_checksum_run(fa, po, 1)
    if not tile:  # (the workgroup kernel's 1536 segments per partition hold ~10 tuples each here: rarely above 40)
        assert st["records_direct"] > (500_000 if cap == 40 else 10_000), st  # the fallbacks really ran (of 12 M records)


def test_config3_shape_sketches_and_topk_at_scale(gpu_lib, fa, po):
    """BASELINE config 3 shape at 3 M records (Zipf 1.1 addresses, 2^20 universe, weight Bytes*SamplingRate):
    both sketches bit-exact against the CPU sketch (wave-level folding of equal addresses, replicated
    counters folded before the read), top-100 identical to ranking every distinct address by its estimate,
    estimates never below the exact GROUP BY."""
    import torch
    n = 3_000_000
    gp = po.gen_params(mode=2, framed=1, seed=3, n_total=n, zipf_log2_universe=20, zipf_s_x100=110)
    mp = fa.mock_params(mode=2, framed=1, seed=3, n_total=n, zipf_log2_universe=20, zipf_s_x100=110)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    depth, wl2, seed = 4, 16, 0x5EED
    ks = fa.FA_KEYS_AS_PAIR | fa.FA_KEYS_SRCADDR_CMS | fa.FA_KEYS_DSTADDR_CMS
    dev = torch.device("cuda", 0)
    with fa.FlowAgg(framed=True, key_sets=ks, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=21,
                    max_batch_records=n) as agg:
        d_buf = torch.empty(n * 96 + 4096, dtype=torch.uint8, device=dev)
        d_off = torch.empty(n + 1, dtype=torch.int32, device=dev)
        wbytes = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
        assert wbytes == len(buf)
        agg.ingest_device(d_buf.data_ptr(), wbytes, d_off.data_ptr(), n)
        for col, key_set in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            want = po.cms_sketch_numpy(rows[col], w, depth, wl2, seed)
            got = agg.cms_read(key_set).reshape(-1)
            assert np.array_equal(got, want), col
            # exact GROUP BY address (viz-ch.json:233,479) and the ranking by estimate
            keys = np.ascontiguousarray(rows[col]).view([("k", "u1", 16)]).reshape(-1)
            uniq, inv = np.unique(keys, return_inverse=True)
            exact = np.zeros(len(uniq), dtype=np.uint64)
            with np.errstate(over="ignore"):
                np.add.at(exact, inv, w)
            ukeys = uniq.view(np.uint8).reshape(-1, 16)
            top = agg.topk(key_set, 100)
            everything = agg.topk(key_set, 1 << 21)
            assert len(everything) == len(uniq)
            got_w = {bytes(r["key"]): int(r["weight"]) for r in everything}
            for k, e in zip(ukeys, exact):
                assert got_w[bytes(k)] >= int(e)  # never an under-estimate
            order = sorted(got_w.items(), key=lambda kv: (-kv[1], kv[0]))[:100]
            assert [(bytes(r["key"]), int(r["weight"])) for r in top] == order
            for r in top[:10]:
                assert int(r["weight"]) == po.cms_query(want, depth, wl2, seed, bytes(r["key"]))




def test_empty_and_tiny_batches(gpu_lib, fa, po):
    """n = 0 (no records, empty buffer), n = 1, an empty record (valid protobuf: every column 0), and a batch
    made of malformed records only."""
    with fa.FlowAgg(framed=False) as agg:
        agg.ingest(b"", np.zeros(1, dtype=np.uint64))  # n = 0
        assert len(agg.read_window()) == 0 and agg.stats()["records_ok"] == 0
        one = bytes.fromhex("10a0ceb9b706" "70eafb03" "78e9fb03" "48db0b" "5063")  # not canonical order: Bytes after the ASNs
        agg.ingest(one, np.array([0, len(one)], dtype=np.uint64))
        rows = agg.read_window()
        assert len(rows) == 1 and int(rows["bytes"][0]) == 1499 and int(rows["packets"][0]) == 99 and int(rows["count"][0]) == 1
        agg.ingest(b"", np.array([0, 0], dtype=np.uint64))  # one empty record: TimeReceived = 0 -> timeslot 0
        rows = agg.read_window()
        assert len(rows) == 2 and int(rows["timeslot"][0]) == 0 and int(rows["count"][0]) == 1
        bad = [bytes.fromhex("70ffffffff"), bytes.fromhex("00"), bytes.fromhex("3210" + "11" * 8)]
        blob = b"".join(bad)
        offs = np.array([0, 5, 6, 6 + 10], dtype=np.uint64)
        agg.ingest(blob, offs)
        st = agg.stats()
        assert st["records_bad"] == 3 and st["records_ok"] == 2
        assert len(agg.read_window()) == 2
    ref = po.Rollup(300)
    assert ref.ingest(np.frombuffer(blob, dtype=np.uint8), offs, 0) == 3


def test_readme_samples_on_device(gpu_lib, fa, po):
    """README.md:155-161,180-183 of the reference through the C-ABI: fa_decode gives the printed flows_raw rows,
    fa_ingest + fa_close_window the printed flows_5m rows (tests/golden/readme_samples.json)."""
    import json
    from test_oracle_golden import GOLDEN, readme_batch, readme_render
    fx = json.load(open(os.path.join(GOLDEN, "readme_samples.json")))
    buf, off = readme_batch(fa, fx)
    with fa.FlowAgg(framed=True) as agg:
        decoded = agg.decode(buf, off)
        assert not decoded["status"].any()
        agg.ingest(buf, off)
        rows = agg.close_window()
    readme_render(fa, fx, decoded, rows)


def test_mock_generator_refuses_a_call_of_4_gib(gpu_lib, fa):
    """Device offsets are 32-bit: a generator call whose records add up to 4 GiB or more is refused with the size it would have
    written (found by `bench.py --mode goflow` at the default chunk: 33 M records of 165 bytes wrapped the offsets silently)."""
    import torch
    n = 30_000_000  # x ~156 bytes = 4.7 GB
    mp = fa.mock_params(mode=fa.MOCK_GOFLOW, framed=1, seed=3, n_total=n, span_secs=900, per_sec=400_000)
    d_buf = torch.empty(4096, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    with fa.FlowAgg(framed=True) as agg:
        with pytest.raises(fa.FlowAggError) as e:
            agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())
        assert e.value.code == -1 and "4 GiB" in str(e.value)  # FA_ERR_ARG
        small = agg.mock_generate_device(mp, 0, 16, d_buf.data_ptr(), d_buf.numel(), d_off.data_ptr())  # (the ctx is fine)
        assert 16 * 100 < small < 16 * 200
