"""CPU tests: the oracle against the committed golden fixtures (upb-pinned), against
live upb when the protobuf wheel is importable, and its own invariants."""
import json
import os
import random

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_edge_cases(po):
    cases = json.load(open(os.path.join(GOLDEN, "edge_cases.json")))["cases"]
    assert len(cases) >= 60
    for c in cases:
        got = po.decode(bytes.fromhex(c["hex"]))
        want = c["expect"]
        if want is None:
            assert got is None, c["name"]
            continue
        assert got is not None, c["name"]
        for k, v in want.items():
            g = got[k].hex() if isinstance(got[k], bytes) else got[k]
            assert g == v, (c["name"], k, g, v)


def test_kat1_flows_5m_contribution(po):
    """key and contribution to flows_5m."""
    cases = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "edge_cases.json")))["cases"]}
    rec = bytes.fromhex(cases["KAT-1 all mocker fields"]["hex"])
    framed = bytes([len(rec)]) + rec
    assert framed[0] == 0x52
    r = po.Rollup(300)
    assert r.ingest(np.frombuffer(framed, dtype=np.uint8), np.array([0, len(framed)], dtype=np.uint64), 1) == 0
    rows = r.rows()
    assert len(rows) == 1
    row = rows[0]
    assert (row["date"], row["timeslot"], row["src_as"], row["dst_as"], row["etype"]) == (
        19987, 1726899900, 65002, 65001, 34525)
    assert (row["bytes"], row["packets"], row["count"]) == (1499, 99, 1)


def test_fuzz_fixture_matches_upb(po):
    z = np.load(os.path.join(GOLDEN, "fuzz_records.npz"))
    rows, status = po.decode_batch(z["buf"], z["off"], 0)
    assert np.array_equal(status, z["upb_status"])
    assert status.sum() > 1000
    colmap = {"time_received": "upb_TimeReceived", "time_flow_start": "upb_TimeFlowStart",
              "sequence_num": "upb_SequenceNum", "sampling_rate": "upb_SamplingRate",
              "src_as": "upb_SrcAS", "dst_as": "upb_DstAS", "etype": "upb_EType", "proto": "upb_Proto",
              "src_port": "upb_SrcPort", "dst_port": "upb_DstPort", "bytes": "upb_Bytes",
              "packets": "upb_Packets", "sampler_address": "upb_SamplerAddress",
              "src_addr": "upb_SrcAddr", "dst_addr": "upb_DstAddr"}
    for c, u in colmap.items():
        assert np.array_equal(rows[c].astype(np.uint64) if rows[c].ndim == 1 else rows[c], z[u]), c
    # the single-record entry point agrees with the batch one
    raw = bytes(z["buf"])
    for k in range(0, len(status), 97):
        d = po.decode(raw[int(z["off"][k]):int(z["off"][k + 1])])
        assert (d is None) == bool(status[k])


def test_rollup_fixture(po):
    fx = json.load(open(os.path.join(GOLDEN, "rollup_2000.json")))
    recs = [bytes.fromhex(h) for h in fx["records_hex"]]
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in recs])
    r = po.Rollup(300)
    assert r.ingest(np.frombuffer(b"".join(recs), dtype=np.uint8), off, 0) == 0
    rows = r.rows()
    got = [[int(x[f]) for f in ("date", "timeslot", "src_as", "dst_as", "etype", "bytes", "packets", "count")]
           for x in rows]
    assert got == fx["rows"]
    assert sum(x[7] for x in got) == len(recs)


def test_live_upb_cross_check(po, fa):
    pytest.importorskip("google.protobuf")
    L = fa.schema.message_class("full")
    rng = random.Random(99)
    for i in range(500):
        m = L()
        m.TimeReceived = rng.randrange(2**34)
        m.Bytes = rng.randrange(2**64)
        m.Packets = rng.randrange(2**20)
        m.SrcAS = rng.randrange(2**32)
        m.DstAS = rng.randrange(2**32)
        m.Etype = rng.choice([0x0800, 0x86DD, 0])
        m.SrcAddr = bytes(rng.randrange(256) for _ in range(rng.choice([0, 4, 16])))
        m.MPLS1Label = rng.randrange(2**20)
        m.DstCountry = "FR"
        d = po.decode(m.SerializeToString())
        assert d["TimeReceived"] == m.TimeReceived and d["Bytes"] == m.Bytes
        assert d["SrcAS"] == m.SrcAS and d["DstAS"] == m.DstAS and d["EType"] == m.Etype
        assert d["SrcAddr"] == m.SrcAddr + b"\0" * (16 - len(m.SrcAddr))
        assert d["Packets"] == m.Packets


def test_generator_roundtrip_and_mocker_distribution(po):
    """mocker.go:57-91: value ranges, 9 AS pairs, field set; decode(generate(i)) == truth(i)."""
    n = 20000
    for mode in (po.GEN_MOCKER, po.GEN_ASPAIRS, po.GEN_ZIPF):
        for framed in (0, 1):
            gp = po.gen_params(mode=mode, framed=framed, seed=4, n_total=n)
            buf, off = po.gen_records(gp, 0, 2000)
            truth = po.gen_rows(gp, 0, 2000)
            raw = bytes(buf)
            for k in range(2000):
                d = po.decode(raw[int(off[k]):int(off[k + 1])], framed=bool(framed))
                assert d is not None
                assert d["TimeReceived"] == truth["time_received"][k]
                assert d["Bytes"] == truth["bytes"][k] and d["Packets"] == truth["packets"][k]
                assert d["SrcAddr"] == bytes(truth["src_addr"][k])
                assert d["DstPort"] == truth["dst_port"][k]
    gp = po.gen_params(mode=po.GEN_MOCKER, framed=1, seed=1, n_total=n, per_sec=4)
    t = po.gen_rows(gp, 0, n)
    assert t["bytes"].max() < 1500 and t["packets"].max() < 100
    assert set(t["src_as"]) == {65000, 65001, 65002} and set(t["dst_as"]) == {65000, 65001, 65002}
    assert (t["etype"] == 0x86DD).all() and (t["sampling_rate"] == 1).all()
    assert (t["src_addr"][:, :15] == np.array([0x20, 1, 0x0d, 0xb8, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0])).all()
    assert t["time_received"][0] == po.T0 and t["time_received"][n - 1] == po.T0 + (n - 1) // 4
    lens = np.diff(po.gen_records(gp, 0, n)[1].astype(np.int64))
    assert 64 <= lens.min() and lens.max() <= 85 and 82.5 < lens.mean() < 84.5


def test_aspairs_config2_shape(po):
    n = 400000
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=2, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    r = po.Rollup(300)
    assert r.ingest(buf, off, 1) == 0
    rows = r.rows()
    assert set(rows["timeslot"]) == {po.T0, po.T0 + 300, po.T0 + 600}
    assert set(rows["etype"]) == {0x0800, 0x86DD}
    assert rows["count"].sum() == n
    assert 65000 < len({(a, b) for a, b in zip(rows["src_as"], rows["dst_as"])}) <= 65536


def test_frame_split(po):
    gp = po.gen_params(mode=po.GEN_MOCKER, framed=1, seed=6, n_total=100)
    buf, off = po.gen_records(gp, 0, 100)
    out = np.zeros(102, dtype=np.uint64)
    n = po.lib().fo_frame_split(buf.ctypes.data, buf.size, out.ctypes.data, out.size)
    assert n == 100 and np.array_equal(out[:101], off)
    assert po.lib().fo_frame_split(buf.ctypes.data, buf.size - 1, out.ctypes.data, out.size) == 2**64 - 1


def test_rollup_merge_is_shard_invariant(po):
    """Kafka partitions are the shard unit; the merged rollup equals the single-shard one."""
    n = 50000
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=10, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    whole = po.Rollup(300)
    whole.ingest(buf, off, 1)
    raw = bytes(buf)
    total = po.Rollup(300)
    for p in range(8):
        recs = [raw[int(off[k]):int(off[k + 1])] for k in range(p, n, 8)]
        o = np.zeros(len(recs) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(r) for r in recs])
        part = po.Rollup(300)
        part.ingest(np.frombuffer(b"".join(recs), dtype=np.uint8), o, 1)
        total.merge(part)
    assert total.rows().tobytes() == whole.rows().tobytes()


def test_cms_overestimates_only(po):
    depth, wl2, seed = 4, 10, 1
    cms = np.zeros(depth << wl2, dtype=np.uint64)
    rng = random.Random(3)
    exact = {}
    for _ in range(5000):
        key = bytes([rng.randrange(40)] + [0] * 15)
        w = rng.randrange(1000)
        exact[key] = exact.get(key, 0) + w
        po.cms_update(cms, depth, wl2, seed, key, w)
    for k, v in exact.items():
        assert po.cms_query(cms, depth, wl2, seed, k) >= v


def test_wide_keyset_restatements_against_plain_dicts(po):
    """The numpy restatements of the wide key sets (rollup_app / top_ports / minute_series) against an
    independent row-at-a-time Python restatement of the same SQL on a small sample."""
    n = 3000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=77, n_total=n, span_secs=700, zipf_log2_universe=6)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    M = (1 << 64) - 1
    app, ports, minutes = {}, ({}, {}), {}
    for r in rows[status == 0]:
        t = int(r["time_received"]) & 0xFFFFFFFF
        ts = t - t % 300
        k = (ts // 86400, ts, bytes(r["src_addr"]), int(r["dst_port"]), int(r["proto"]))
        b, p, c = app.get(k, (0, 0, 0))
        app[k] = ((b + int(r["bytes"])) & M, (p + int(r["packets"])) & M, c + 1)
        w = (int(r["bytes"]) * int(r["sampling_rate"])) & M
        for d, col in enumerate(("src_port", "dst_port")):
            ww, cc = ports[d].get(int(r[col]), (0, 0))
            ports[d][int(r[col])] = ((ww + w) & M, cc + 1)
        tf = int(r["time_flow_start"]) & 0xFFFFFFFF
        ww, cc = minutes.get(tf - tf % 60, (0, 0))
        minutes[tf - tf % 60] = ((ww + w) & M, cc + 1)
    got = po.rollup_app(rows, status, 300)
    want = sorted(app.items())
    assert len(got) == len(want)
    for g, (k, v) in zip(got, want):
        assert (int(g["date"]), int(g["timeslot"]), bytes(g["src_addr"]), int(g["dst_port"]), int(g["proto"])) == k
        assert (int(g["bytes"]), int(g["packets"]), int(g["count"])) == v
    for d in (0, 1):
        got = po.top_ports(rows, status, d)
        want = sorted(ports[d].items(), key=lambda kv: (-kv[1][0], kv[0]))
        assert [(int(g["port"]), int(g["weight"]), int(g["count"])) for g in got] == [(k, v[0], v[1]) for k, v in want]
    got = po.minute_series(rows, status)
    assert [(int(g["minute"]), int(g["weight"]), int(g["count"])) for g in got] == [(k, v[0], v[1]) for k, v in sorted(minutes.items())]
    # sliding window fold: 5 one-minute sub-buckets == direct 300 s window starting on a minute
    start = po.T0 + 120
    sel = (rows["time_received"] >= start) & (rows["time_received"] < start + 300)
    a = po.rollup_app(rows, status, 60, window=300, timeslot=start)
    b = po.rollup_app(rows[sel], status[sel], 86400)
    for col in ("src_addr", "dst_port", "proto", "bytes", "packets", "count"):
        assert np.array_equal(a[col], b[col]), col


def test_numpy_sketch_restatement_matches_c_oracle(po):
    rng = np.random.default_rng(3)
    keys = rng.integers(0, 256, size=(3000, 16), dtype=np.uint8)
    keys[:500] = keys[0]  # a heavy hitter
    w = rng.integers(0, 2**63, size=3000, dtype=np.uint64) * np.uint64(3)  # wraps
    depth, wl2, seed = 4, 9, 0xABCDEF
    want = np.zeros(depth << wl2, dtype=np.uint64)
    for k in range(len(keys)):
        po.cms_update(want, depth, wl2, seed, bytes(keys[k]), int(w[k]))
    assert np.array_equal(po.cms_sketch_numpy(keys, w, depth, wl2, seed), want)


def test_sketch_definition_is_pinned(po):
    """The sketch definition (DESIGN.md "Sketch": prefix-partitioned Count-Min) against the committed columns of a few
    fixed keys (tests/golden/sketch_columns.json, made by tests/golden/make_sketch_columns.py): the C oracle, the numpy
    restatement and the tools' restatement (tools/config3_run.py) all give exactly these.  A change of the definition has
    to change the fixture - deliberately."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(root, "tests", "golden", "sketch_columns.json")))["vectors"]
    sys.path.insert(0, os.path.join(root, "tools"))
    import config3_run
    L = po.lib()
    phi = 0x9E3779B97F4A7C15
    for v in fx:
        key, seed, wl2, want = bytes.fromhex(v["key"]), v["seed"], v["width_log2"], v["columns"]
        assert [int(L.fo_cms_column(key, seed, wl2, r)) for r in range(len(want))] == want, v
        lo = np.frombuffer(key[:8], dtype="<u8").copy()
        hi = np.frombuffer(key[8:], dtype="<u8").copy()
        with np.errstate(over="ignore"):
            a = config3_run.mix64(lo ^ config3_run.mix64(np.array([(seed + phi) & (2**64 - 1)], dtype=np.uint64))[0])
            h1 = config3_run.mix64(a ^ hi)
        for r, c in enumerate(want):
            assert int(po.cms_columns(a, h1, wl2, r)[0]) == c and int(config3_run.columns(a, h1, wl2, r)[0]) == c, (v, r)
            assert c >> wl2 == 0
        # all rows of a key share the partition prefix
        pbits = min(8, wl2 - 4)
        assert len({c >> (wl2 - pbits) for c in want}) == 1


def readme_render(fa, fx, decoded_rows, rows5m):
    """Renders decoded rows / flows_5m rows the way clickhouse-client printed them in the reference's README and
    compares cell by cell with the README's own table rows (tests/golden/readme_samples.json)."""
    import datetime
    utc = lambda t: datetime.datetime.fromtimestamp(int(t), datetime.timezone.utc)
    cell = lambda l: [c.strip() for c in l.strip("│").split("│")]
    for line, idx in zip(fx["readme_flows_raw"], fx["raw_row_record_index"]):
        d = decoded_rows[idx]
        t32 = int(d["time_received"]) & 0xFFFFFFFF  # flows_raw: TimeReceived UInt64 -> DateTime (create.sh:41,64-68)
        got = [utc(t32).strftime("%Y-%m-%d"), utc(t32).strftime("%Y-%m-%d %H:%M:%S"),
               fa.format_addr(bytes(d["src_addr"]), int(d["etype"])), fa.format_addr(bytes(d["dst_addr"]), int(d["etype"])),
               str(int(d["bytes"])), str(int(d["packets"]))]
        assert got == cell(line), (got, line)
    got5 = []
    for r in rows5m:
        e, b, p, c = int(r["etype"]), int(r["bytes"]), int(r["packets"]), int(r["count"])
        got5.append([(datetime.date(1970, 1, 1) + datetime.timedelta(days=int(r["date"]))).isoformat(),
                     utc(r["timeslot"]).strftime("%Y-%m-%d %H:%M:%S"), str(int(r["src_as"])), str(int(r["dst_as"])),
                     "[%d]" % e, "[%d]" % b, "[%d]" % p, "[%d]" % c, str(b), str(p), str(c)])
    assert got5 == [cell(l) for l in fx["readme_flows_5m"]], got5


def readme_batch(fa, fx):
    recs = [fa.schema.frame(bytes.fromhex(h)) for h in fx["records_hex"]]  # ClickHouse path: -proto.fixedlen=true
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in recs])
    return np.frombuffer(b"".join(recs), dtype=np.uint8), off


def test_readme_samples_oracle(po, fa):
    """The reference's only result-bearing text (README.md:155-161,180-183): the oracle reproduces the printed
    flows_raw and flows_5m rows (Date, toStartOfFiveMinute, IPv6NumToString, [EType] key, sums, counts)."""
    fx = json.load(open(os.path.join(GOLDEN, "readme_samples.json")))
    buf, off = readme_batch(fa, fx)
    rows, status = po.decode_batch(buf, off, framed=1)
    assert not status.any()
    r = po.Rollup(300)
    assert r.ingest(buf, off, 1) == 0
    readme_render(fa, fx, rows, r.rows())


def test_candidates_contract_restatement_finds_the_heavy_hitters(po):
    """oracle/pyoracle.py topk_candidates (fa_config.topk_mode = FA_TOPK_CANDIDATES): on a skewed stream cut into batches the
    candidates hold far fewer keys than the stream has addresses, nothing joins during the first batch, and the first 50 of their
    ranking are the first 50 of the ranking of every address; thresholds never fall below the total-weight floor."""
    n, nb, depth, wl2, seed = 120_000, 6, 4, 14, 7
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=97, n_total=n, zipf_log2_universe=14)
    rows = po.gen_rows(gp, 0, n)
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    step = n // nb
    batches = [(rows["src_addr"][i * step:(i + 1) * step], w[i * step:(i + 1) * step]) for i in range(nb)]
    sk, cand, est, thetas = po.topk_candidates(batches, depth, wl2, seed, track=256, capacity_log2=12)
    assert np.array_equal(sk, po.cms_sketch_numpy(rows["src_addr"], w, depth, wl2, seed))
    keys = np.unique(np.ascontiguousarray(rows["src_addr"]), axis=0)
    full = sorted(zip((-po.cms_estimates_numpy(sk, keys, depth, wl2, seed).astype(object)).tolist(), [bytes(k) for k in keys]))
    mine = sorted(zip((-est.astype(object)).tolist(), [bytes(k) for k in cand]))
    assert 100 <= len(cand) < len(keys) // 4 and mine[:50] == full[:50]
    _, c1, _, t1 = po.topk_candidates(batches[:1], depth, wl2, seed, track=256, capacity_log2=12)
    assert len(c1) == 0 and t1[0] >= 1
    total = 0
    for (_, bw), th in zip(batches, thetas):
        total = (total + int(bw.sum(dtype=np.uint64))) & (2**64 - 1)
        assert th >= max(total >> 10, 1)


def test_topk_contract_fixture(po):
    """tests/golden/topk_contract.json (make_topk_contract.py): the estimate-bin map at fixed values and the candidates contract on a
    seeded stream - thresholds per boundary, candidates held, the ranking's digest.  The restatement may not drift from it; the
    library is compared with the restatement on the GPU (tests/test_topk_gpu.py)."""
    import hashlib
    import importlib.util
    fx = json.load(open(os.path.join(GOLDEN, "topk_contract.json")))
    for v in fx["bins"]:
        e, b = int(v["estimate"]), v["bin"]
        assert int(po.topk_bin(np.array([e], dtype=np.uint64))[0]) == b and po.topk_bin_floor(b) == int(v["floor"])
        assert po.topk_bin_floor(b) <= e and (b == 1919 or po.topk_bin_floor(b + 1) > e)  # (the map is monotone: floor(b) <= e < floor(b + 1))
    spec = importlib.util.spec_from_file_location("make_topk_contract", os.path.join(GOLDEN, "make_topk_contract.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    c = fx["candidates"]
    again = mk.case(n=c["records"], nb=c["batches"], depth=c["depth"], wl2=c["width_log2"], seed=c["seed"], track=c["track"], cap=c["capacity_log2"],
                    gen_seed=c["generator_seed"])
    assert again == c
    assert hashlib.sha256(b"").hexdigest() != c["sets"]["src_addr"]["ranking_sha256"] and c["sets"]["src_addr"]["candidates"] >= c["track"]


def test_linear_app_checksum_equals_the_grouped_rollups(po):
    """fo_app_checksum_stream (bench.py's config-5 / group checks at 100 M records, where a CPU group-by of the (SrcAddr,DstPort,
    Proto) rows takes minutes): the checksum over RECORDS == the same checksum over the rows of pyoracle.rollup_app, window by
    window; it moves when a sum moves between keys or a row is split; strict order is what rules the split out."""
    n = 120_000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=5, n_total=n, span_secs=900, zipf_log2_universe=12, zipf_s_x100=80)
    buf, off = po.gen_records(gp, 0, n)
    rows, st = po.decode_batch(buf, off, framed=1)
    app = po.rollup_app(rows, st)
    sums, cnts, outside = po.app_checksum_stream(gp, 0, n, 3, 300, po.T0, 3)
    assert outside == 0 and int(cnts.sum()) == n
    for k in range(3):
        w = app[app["timeslot"] == po.T0 + 300 * k]
        assert len(w) and po.app_rows_checksum(w) == int(sums[k]) and int(w["count"].sum()) == int(cnts[k])
        assert po.app_rows_strictly_ascending(w)
        bad = w.copy()
        bad["bytes"][0] += 1
        bad["bytes"][1] -= 1  # the same total, another key
        assert po.app_rows_checksum(bad) != int(sums[k])
        split = np.concatenate([w[:1], w])  # a key twice: the linear checksum cannot see a row split in two halves ...
        split["bytes"][0] = split["bytes"][1] // 2
        split["bytes"][1] -= split["bytes"][0]
        split["packets"][0] = split["count"][0] = 0
        assert po.app_rows_checksum(split) == int(sums[k]) and not po.app_rows_strictly_ascending(split)  # ... the order check does
    # slots that do not cover the stream: the rest is counted outside
    _, c2, out2 = po.app_checksum_stream(gp, 0, n, 2, 300, po.T0 + 300, 1)
    assert out2 == n - int(cnts[1]) and int(c2[0]) == int(cnts[1])


def test_bench_restatements_agree_with_the_oracle(po):
    """bench.py's secondary blocks rank the whole address universe with numpy (16 M addresses x 2 forms: the C oracle's per-key
    query would take minutes): its restatements of the generator's addresses and of the sketch's columns == the oracle's."""
    import bench
    L, depth, wl2, seed = 10, 4, 12, 0x5EED
    for dst in (0, 1):
        lo, hi = bench._universe_keys(L, dst)
        for rank in (0, 1, 7, 513, (1 << L) - 1):
            for v6 in (0, 1):
                key = po.zipf_key(rank, dst, v6)
                i = rank + (v6 << L)
                assert lo[i].tobytes() + hi[i].tobytes() == key, (dst, rank, v6)
    n = 50_000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=3, n_total=n, zipf_log2_universe=L, zipf_s_x100=110)
    c_src = np.zeros(depth << wl2, dtype=np.uint64)
    c_dst = np.zeros(depth << wl2, dtype=np.uint64)
    po.cms_stream(gp, 0, n, 2, depth, wl2, seed, c_src, c_dst)
    for dst, cms in enumerate((c_src, c_dst)):
        lo, hi = bench._universe_keys(L, dst)
        est = bench._estimates(cms, lo, hi, depth, wl2, seed)
        keys = np.stack([lo, hi], axis=1).view(np.uint8).reshape(-1, 16)
        assert np.array_equal(est, po.cms_estimates_numpy(cms.reshape(depth, -1), keys, depth, wl2, seed))
        for i in (0, 5, 300, (1 << L) + 17):
            assert int(est[i]) == po.cms_query(cms, depth, wl2, seed, keys[i].tobytes())
        want = bench._want_top100(cms, L, dst, depth, wl2, seed)
        uniq = {}
        for k, e in zip(keys, est):
            uniq[k.tobytes()] = int(e)
        assert want == sorted(uniq.items(), key=lambda kv: (-kv[1], kv[0]))[:100]
