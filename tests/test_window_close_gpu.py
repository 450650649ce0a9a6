"""The device-resident window close (ABI 5: fa_rows_device / fa_rows_merge_device / fa_drop_window).

Two ctxs on the one GPU stand for two ranks: each ingests half of the Kafka partitions, their device row buffers are
concatenated in HBM (what the RCCL all-gather leaves on every rank) and merged by the library's kernels - every row
kind, tumbling and sliding windows - against a ctx that ingested everything and against the oracle.  The transport
(nccl world 1, gloo with two processes) is covered in test_dist_nccl_gpu.py / test_ingest_sinks_gpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _shards(po, n, seed, nparts=8, zipf_log2=14):
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=seed, n_total=n, zipf_log2_universe=zipf_log2, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)

    def shard(parts):
        idx = np.concatenate([np.arange(p, n, nparts) for p in parts])
        idx.sort()
        recs = [raw[int(off[k]):int(off[k + 1])] for k in idx]
        o = np.zeros(len(recs) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(r) for r in recs])
        return np.frombuffer(b"".join(recs), dtype=np.uint8), o
    return buf, off, shard(range(0, nparts, 2)), shard(range(1, nparts, 2))


def _cat_device(fa, torch, ctxs, kind, timeslot, k=0):
    """every ctx's fa_rows_device result, back to back in one device buffer (the all-gather's destination)"""
    rb = fa.ROW_DTYPES[kind].itemsize
    parts = []
    for c in ctxs:
        ptr, n = c.rows_device(kind, timeslot, k)
        if n:
            parts.append(torch.as_tensor(fa.dist._DevArray(ptr, n * rb, "|u1"), device="cuda").clone())
    if not parts:
        return torch.empty(0, dtype=torch.uint8, device="cuda"), 0
    buf = torch.cat(parts)
    torch.cuda.synchronize()  # the ctx reads the buffer on ITS stream: torch's copies must have landed (as dist.allgather_device_rows does)
    return buf, buf.numel() // rb


@pytest.mark.parametrize("sub", [0, 60])
def test_device_merge_of_two_ctxs_equals_single_ctx(gpu_lib, fa, po, sub):
    import torch
    n = 300_000
    buf, off, (b0, o0), (b1, o1) = _shards(po, n, seed=91)
    kw = dict(framed=True, key_sets=63, cms_width_log2=14, topk_capacity_log2=16, subwindow_secs=sub)
    with fa.FlowAgg(**kw) as a0, fa.FlowAgg(**kw) as a1, fa.FlowAgg(**kw) as whole:
        a0.ingest(b0, o0)
        a1.ingest(b1, o1)
        whole.ingest(buf, off)
        slots = whole.open_timeslots()
        windows = [fa.ALL_TIMESLOTS, int(slots[0])] + ([int(slots[0]) + 60, int(slots[0]) + 240] if sub else [int(slots[0]) + 300])
        for ts in windows:
            for kind, read in ((fa.ROWS_5M, whole.read_window), (fa.ROWS_APP, whole.read_window_app)):
                dbuf, tot = _cat_device(fa, torch, (a0, a1), kind, ts)
                ptr, m = a0.rows_merge_device(kind, dbuf.data_ptr(), tot)
                got = a0.rows_fetch(kind, ptr, m)
                want = read(ts)
                assert len(want) > 0 and got.tobytes() == want.tobytes(), (kind, ts, len(got), len(want))
        for kind, want in ((fa.ROWS_PORT_SRC, whole.top_ports(0)), (fa.ROWS_PORT_DST, whole.top_ports(1)), (fa.ROWS_MINUTE, whole.minute_series())):
            dbuf, tot = _cat_device(fa, torch, (a0, a1), kind, 0)
            ptr, m = a1.rows_merge_device(kind, dbuf.data_ptr(), tot)
            assert a1.rows_fetch(kind, ptr, m).tobytes() == want.tobytes(), kind
        # the oracle's word on the merged flows_5m rows and the (SrcAddr,DstPort,Proto) rows of the whole stream
        ref = po.Rollup(sub or 300)
        ref.ingest(buf, off, 1)
        dbuf, tot = _cat_device(fa, torch, (a0, a1), fa.ROWS_5M, fa.ALL_TIMESLOTS)
        ptr, m = a0.rows_merge_device(fa.ROWS_5M, dbuf.data_ptr(), tot)
        assert a0.rows_fetch(fa.ROWS_5M, ptr, m).tobytes() == ref.rows().tobytes()
        rows, status = po.decode_batch(buf, off, 1)
        dbuf, tot = _cat_device(fa, torch, (a0, a1), fa.ROWS_APP, fa.ALL_TIMESLOTS)
        ptr, m = a0.rows_merge_device(fa.ROWS_APP, dbuf.data_ptr(), tot)
        assert a0.rows_fetch(fa.ROWS_APP, ptr, m).tobytes() == po.rollup_app(rows, status, sub or 300).astype(fa.ROW_APP_DTYPE).tobytes()
        # heavy hitters: merged sketch view on both (the all-reduce's result), each ctx's first k rows, merged
        for key_set, kind in ((fa.FA_KEYS_SRCADDR_CMS, fa.ROWS_TOPK_SRC), (fa.FA_KEYS_DSTADDR_CMS, fa.ROWS_TOPK_DST)):
            for c in (a0, a1):
                st = c.device_state()
                for own, merged, wh in ((st.cms_src, st.cms_src_merged, whole.device_state().cms_src), (st.cms_dst, st.cms_dst_merged, whole.device_state().cms_dst)):
                    t = torch.as_tensor(fa.dist._DevArray(merged, st.cms_words), device="cuda")
                    t.copy_(torch.as_tensor(fa.dist._DevArray(wh, st.cms_words), device="cuda"))  # = sum over the two shards (checked below)
                c.merged_view_set(True)
            torch.cuda.synchronize()  # (torch wrote the merged views on its stream; the ctxs read them on theirs)
            assert np.array_equal(a0.cms_read(key_set), whole.cms_read(key_set))
            k = 60
            dbuf, tot = _cat_device(fa, torch, (a0, a1), kind, 0, k)
            assert tot == 2 * k
            ptr, m = a0.rows_merge_device(kind, dbuf.data_ptr(), tot, k)
            assert a0.rows_fetch(kind, ptr, m).tobytes() == whole.topk(key_set, k).tobytes()
            for c in (a0, a1):
                c.merged_view_set(False)
        # the shards' own sketches do add up to the whole stream's (what the RCCL all-reduce computes)
        for key_set in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS):
            with np.errstate(over="ignore"):
                assert np.array_equal(a0.cms_read(key_set) + a1.cms_read(key_set), whole.cms_read(key_set))
        # drop_window removes what a close removes: afterwards the merged remainder == the single ctx after its close
        ts = int(slots[0])
        whole.close_window(ts)
        whole.close_window_app(ts)
        for c in (a0, a1):
            c.drop_window(fa.ROWS_5M, ts)
            c.drop_window(fa.ROWS_APP, ts)
        for kind, read in ((fa.ROWS_5M, whole.read_window), (fa.ROWS_APP, whole.read_window_app)):
            dbuf, tot = _cat_device(fa, torch, (a0, a1), kind, fa.ALL_TIMESLOTS)
            ptr, m = a1.rows_merge_device(kind, dbuf.data_ptr(), tot)
            assert a1.rows_fetch(kind, ptr, m).tobytes() == read(fa.ALL_TIMESLOTS).tobytes(), kind


def test_merge_rows_device_between_two_ctxs(gpu_lib, fa, po):
    """fa_window_rows_device of shard A folded into ctx B by fa_merge_rows_device (the ABI-4 exchange), tumbling
    and sub-bucketed, == the oracle's rollup of the whole stream; rows off the bucket grid are refused."""
    import torch
    n = 200_000
    buf, off, (b0, o0), (b1, o1) = _shards(po, n, seed=92)
    for sub in (0, 60):
        with fa.FlowAgg(framed=True, subwindow_secs=sub) as a0, fa.FlowAgg(framed=True, subwindow_secs=sub) as a1:
            a0.ingest(b0, o0)
            a1.ingest(b1, o1)
            ptr, m = a0.window_rows_device(fa.ALL_TIMESLOTS)
            mine = torch.as_tensor(fa.dist._DevArray(ptr, m * 48, "|u1"), device="cuda").clone()
            torch.cuda.synchronize()  # (a1 reads the copy on its own stream)
            a1.merge_rows_device(mine.data_ptr(), m)
            ref = po.Rollup(sub or 300)
            ref.ingest(buf, off, 1)
            assert a1.close_window().tobytes() == ref.rows().tobytes()
    with fa.FlowAgg(framed=True) as a:
        bad = np.zeros(4, dtype=fa.ROW5M_DTYPE)
        bad["timeslot"] = [300, 600, 601, 900]
        bad["count"] = 1
        t = torch.from_numpy(bad.view(np.uint8)).cuda()
        with pytest.raises(fa.FlowAggError):
            a.merge_rows_device(t.data_ptr(), 4)
        with pytest.raises(fa.FlowAggError):
            a.rows_merge_device(fa.ROWS_5M, t.data_ptr(), 4)
        assert len(a.read_window()) == 0


def test_rows_merge_device_random_rows_against_dict_groupby(gpu_lib, fa):
    """fa_rows_merge_device on random row sets of every kind (duplicate keys, u64 wrap-around, address byte order, weight
    ties) against a row-at-a-time dict group-by."""
    import torch
    rng = np.random.default_rng(7)
    M = 2**64

    def u64(n, big):
        return rng.integers(0, 2**64 if big else 1000, n, dtype=np.uint64)
    with fa.FlowAgg(framed=True, key_sets=63, cms_width_log2=10, topk_capacity_log2=10) as agg:
        def merged(kind, rows, k=0):
            t = torch.from_numpy(np.ascontiguousarray(rows).view(np.uint8).reshape(-1).copy()).cuda()
            ptr, m = agg.rows_merge_device(kind, t.data_ptr(), len(rows), k)
            return agg.rows_fetch(kind, ptr, m)
        for trial in range(6):
            big = trial % 2 == 0
            n = int(rng.integers(1, 5000))
            r = np.zeros(n, dtype=fa.ROW5M_DTYPE)
            r["timeslot"] = rng.integers(0, 3, n) * 300 + 86400 * rng.integers(0, 2, n)
            r["date"] = r["timeslot"] // 86400
            r["src_as"], r["dst_as"] = rng.integers(0, 5, n), rng.choice([0, 7, 2**31, 2**32 - 1], n)
            r["etype"] = rng.choice([0x800, 0x86dd], n)
            r["bytes"], r["packets"], r["count"] = u64(n, big), u64(n, big), u64(n, False)
            ref = {}
            for x in r:
                key = (int(x["date"]), int(x["timeslot"]), int(x["src_as"]), int(x["dst_as"]), int(x["etype"]))
                b, p, c = ref.get(key, (0, 0, 0))
                ref[key] = ((b + int(x["bytes"])) % M, (p + int(x["packets"])) % M, (c + int(x["count"])) % M)
            got = merged(fa.ROWS_5M, r)
            assert [(int(x["date"]), int(x["timeslot"]), int(x["src_as"]), int(x["dst_as"]), int(x["etype"])) for x in got] == sorted(ref)
            assert [(int(x["bytes"]), int(x["packets"]), int(x["count"])) for x in got] == [ref[key] for key in sorted(ref)]
            r = np.zeros(n, dtype=fa.ROW_APP_DTYPE)
            r["timeslot"] = rng.integers(0, 2, n) * 300
            r["src_addr"] = rng.integers(0, 2, (n, 16)) * rng.integers(1, 256, (n, 16))
            r["src_addr"][:, 2:7] = 0
            r["src_addr"][:, 9:15] = 0
            r["dst_port"], r["proto"] = rng.integers(0, 3, n), rng.integers(0, 2, n)
            r["bytes"], r["packets"], r["count"] = u64(n, big), u64(n, big), u64(n, False)
            ref = {}
            for x in r:
                key = (int(x["date"]), int(x["timeslot"]), bytes(x["src_addr"]), int(x["dst_port"]), int(x["proto"]))
                b, p, c = ref.get(key, (0, 0, 0))
                ref[key] = ((b + int(x["bytes"])) % M, (p + int(x["packets"])) % M, (c + int(x["count"])) % M)
            got = merged(fa.ROWS_APP, r)
            assert [(int(x["date"]), int(x["timeslot"]), bytes(x["src_addr"]), int(x["dst_port"]), int(x["proto"])) for x in got] == sorted(ref)
            assert [(int(x["bytes"]), int(x["packets"]), int(x["count"])) for x in got] == [ref[key] for key in sorted(ref)]
            for kind, dtype, key in ((fa.ROWS_PORT_DST, fa.PORT_ROW_DTYPE, "port"), (fa.ROWS_MINUTE, fa.MINUTE_ROW_DTYPE, "minute")):
                r = np.zeros(n, dtype=dtype)
                r[key] = rng.choice([0, 1, 5, 65535, 65536, 2**32 - 1], n) if key == "port" else rng.integers(0, 12, n) * 60
                r["weight"], r["count"] = u64(n, big) if trial else np.uint64(3), u64(n, False)
                ref = {}
                for x in r:
                    w, c = ref.get(int(x[key]), (0, 0))
                    ref[int(x[key])] = ((w + int(x["weight"])) % M, (c + int(x["count"])) % M)
                order = sorted(ref, key=(lambda q: (-ref[q][0], q)) if key == "port" else (lambda q: q))
                got = merged(kind, r)
                assert [int(x[key]) for x in got] == order
                assert [(int(x["weight"]), int(x["count"])) for x in got] == [ref[q] for q in order]
                if key == "port":
                    assert merged(kind, r, 3).tobytes() == got[:3].tobytes()
            r = np.zeros(n, dtype=fa.TOPK_DTYPE)
            keys = rng.integers(0, 256, (40, 16)).astype(np.uint8)
            keys[:, 1:15] = 0
            pick = rng.integers(0, 40, n)
            r["key"] = keys[pick]
            wts = rng.integers(0, 4, 40).astype(np.uint64)  # the same key carries the same estimate everywhere; ties between keys
            r["weight"] = wts[pick]
            ref = {bytes(keys[i]): int(wts[i]) for i in set(pick.tolist())}
            order = sorted(ref, key=lambda q: (-ref[q], q))
            got = merged(fa.ROWS_TOPK_SRC, r, 25)
            assert [bytes(x["key"]) for x in got] == order[:25] and [int(x["weight"]) for x in got] == [ref[q] for q in order[:25]]


def test_count_min_scatter_sink_segment_overflow_is_bit_exact(gpu_lib, fa, po, monkeypatch):
    """the Count-Min scatter sink with segments far too small (FA_SEG_CAP now also caps the sketch
    segments): full front parts send chunks to the atomic fallback, full back parts single tuples, the end-of-kernel
    drain too - and the position counters stay at their caps (taken back when nothing was stored).  Counter for counter
    the CPU sketch."""
    import torch
    monkeypatch.setenv("FA_SEG_CAP", "40")
    n = 1_000_000
    gp = po.gen_params(mode=2, framed=1, seed=35, n_total=n, zipf_log2_universe=18, zipf_s_x100=110)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    seed = 0xBEEF
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=4, cms_width_log2=16, cms_seed=seed, topk_capacity_log2=20, max_batch_records=n) as agg:
        mp = fa.mock_params(mode=fa.MOCK_ZIPF, framed=1, seed=35, n_total=n, zipf_log2_universe=18, zipf_s_x100=110)
        cap = n * fa.mock_record_cap(fa.MOCK_ZIPF) + 4096
        d_buf = torch.empty(cap, dtype=torch.uint8, device="cuda")
        d_off = torch.empty(n + 1, dtype=torch.int32, device="cuda")
        wb = agg.mock_generate_device(mp, 0, n, d_buf.data_ptr(), cap, d_off.data_ptr())
        assert wb == len(buf)
        agg.ingest_device(d_buf.data_ptr(), wb, d_off.data_ptr(), n)
        for col, key_set in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            assert np.array_equal(agg.cms_read(key_set).reshape(-1), po.cms_sketch_numpy(rows[col], w, 4, 16, seed)), col
        ref = po.Rollup(300)
        ref.ingest(buf, off, 1)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        assert agg.stats()["wave_tile_launches"] == 1


@pytest.mark.parametrize("n", [3000, 200_000])
def test_broken_device_offsets_are_bad_records_not_faults(gpu_lib, fa, po, n):
    """fa_ingest_device trusts nobody's offsets: bounds beyond the buffer or running backwards make the records that own
    them bad (counted, dropped) - nothing outside the buffer is read - and every other record of the batch, including the
    rest of a tile whose first or last bound is broken, is aggregated exactly (wave-tile kernel and workgroup-tile kernel)."""
    import torch
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=44, n_total=n, span_secs=600)
    buf, off = po.gen_records(gp, 0, n)
    off32 = off.astype(np.uint32)
    rng = np.random.default_rng(3)
    picks = np.sort(rng.choice(np.arange(2, n - 2, 3), size=min(200, n // 20), replace=False))
    picks = np.concatenate([picks, [64, 128, 191]]) if n > 1000 else picks   # tile boundaries of the wave-tile kernel among them
    picks = np.unique(picks)
    broken = off32.copy()
    kinds = rng.integers(0, 3, len(picks))
    for k, kind in zip(picks, kinds):
        broken[k] = [np.uint32(len(buf) + 5000), np.uint32(0xfffffff0), np.uint32(max(int(off32[k]) - 5000, 0) if off32[k] > 5000 else len(buf) + 77)][kind]
    bad_recs = set()
    for k in picks:
        bad_recs.add(int(k) - 1)
        bad_recs.add(int(k))
    good = np.array(sorted(set(range(n)) - bad_recs))
    ref = po.Rollup(300)
    raw = bytes(buf)
    recs = [raw[int(off[i]):int(off[i + 1])] for i in good]
    o = np.zeros(len(recs) + 1, dtype=np.uint64)
    o[1:] = np.cumsum([len(r) for r in recs])
    assert ref.ingest(np.frombuffer(b"".join(recs), dtype=np.uint8), o, 1) == 0
    with fa.FlowAgg(framed=True, max_batch_records=n) as agg:
        d_buf = torch.zeros(len(buf) + 64, dtype=torch.uint8, device="cuda")
        d_buf[:len(buf)] = torch.from_numpy(np.ascontiguousarray(buf)).cuda()
        d_off = torch.from_numpy(broken.view(np.int32)).cuda()
        torch.cuda.synchronize()  # (torch's fills and copies run on its stream, the ingest on the ctx's)
        agg.ingest_device(d_buf.data_ptr(), len(buf), d_off.data_ptr(), n)
        st = agg.stats()
        assert st["records_bad"] == len(bad_recs) and st["records_ok"] == n - len(bad_recs), (st["records_bad"], len(bad_recs), st["records_ok"])
        assert agg.read_window().tobytes() == ref.rows().tobytes()


def test_sketch_error_bound_at_100M_records(gpu_lib):
    """the epsilon bound of the prefix-partitioned Count-Min sketch checked inside the suite at >= 100 M records
    (tools/config3_run.py, the runner of the 1 B-record evidence, on a 117 M-record Zipf-1.1 stream): both sketches
    bit-exact against the CPU sketch on the 100 M prefix and on the whole stream, top-100 == the ranking of the whole
    2^25-address universe, estimates never below the exact GROUP BY weight and within eps * total weight (eps = e / width)
    for at least 1 - e^-depth of the addresses."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "config3_run.py"), "--records", "116666669", "--prefix", "100000000"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["prefix_records"] >= 100_000_000 and out["records"] == 116666669
    assert out["sketch_bit_exact_prefix"] and out["sketch_bit_exact_full_stream"] and out["top100_equals_ranking_of_the_whole_universe"]
    for tag in ("src", "dst"):
        assert out["prefix_%s_never_underestimates" % tag]
        assert out["prefix_%s_share_within_eps" % tag] >= out["prefix_%s_required_share" % tag]


def test_rows_merge_device_long_runs(gpu_lib, fa):
    """Degenerate input of fa_rows_merge_device: thousands of rows with ONE key (and a second key with a medium run) - the
    heads sum the first rows of a run, the rest is added by the rows themselves with atomics; sums wrap mod 2^64."""
    import torch
    M = 2**64
    rng = np.random.default_rng(11)
    with fa.FlowAgg(framed=True, key_sets=63, cms_width_log2=10, topk_capacity_log2=10) as agg:
        def merged(kind, rows):
            t = torch.from_numpy(np.ascontiguousarray(rows).view(np.uint8).reshape(-1).copy()).cuda()
            ptr, m = agg.rows_merge_device(kind, t.data_ptr(), len(rows))
            return agg.rows_fetch(kind, ptr, m)
        n = 20000
        which = np.where(np.arange(n) % 400 == 7, 1, 0)           # key 1: a run of 50 rows; key 0: the rest
        r = np.zeros(n, dtype=fa.ROW5M_DTYPE)
        r["timeslot"] = 600
        r["date"] = 0
        r["src_as"] = which
        r["bytes"] = rng.integers(0, 2**64, n, dtype=np.uint64)
        r["packets"] = rng.integers(0, 1000, n, dtype=np.uint64)
        r["count"] = 1
        got = merged(fa.ROWS_5M, r)
        assert len(got) == 2 and list(got["src_as"]) == [0, 1]
        for k in (0, 1):
            sel = r[which == k]
            assert int(got["bytes"][k]) == int(sum(int(x) for x in sel["bytes"]) % M)
            assert int(got["packets"][k]) == int(sel["packets"].sum()) and int(got["count"][k]) == len(sel)
        a = np.zeros(n, dtype=fa.ROW_APP_DTYPE)
        a["timeslot"] = 300
        a["src_addr"][:, 0] = 9
        a["dst_port"] = which
        a["bytes"] = 3
        a["count"] = 2
        got = merged(fa.ROWS_APP, a)
        assert len(got) == 2 and list(got["count"]) == [2 * int((which == 0).sum()), 2 * int((which == 1).sum())]
        p = np.zeros(n, dtype=fa.PORT_ROW_DTYPE)
        p["port"] = which * 70000
        p["weight"] = rng.integers(0, 2**40, n, dtype=np.uint64)
        p["count"] = 1
        got = merged(fa.ROWS_PORT_SRC, p)
        want = {k: int(p["weight"][which == k].sum()) for k in (0, 1)}
        assert {int(x["port"]) // 70000: int(x["weight"]) for x in got} == want and int(got["weight"][0]) >= int(got["weight"][1])
        t = np.zeros(n, dtype=fa.TOPK_DTYPE)
        t["key"][:, 3] = which
        t["weight"] = np.where(which == 1, 5, 9).astype(np.uint64)
        got = merged(fa.ROWS_TOPK_DST, t)
        assert len(got) == 2 and list(got["weight"]) == [9, 5]


def test_hot_address_cache_survives_launches_and_a_reset(gpu_lib, fa, po):
    """The per-workgroup hot-address caches of the sketch variants keep their entries from launch to launch (sinks.cuh,
    HotAddrs / KArgs::hot_seed).  Whatever they hold: sketches and distinct sets stay exact - over several launches of one
    stream (the later ones start with seeded caches) and across fa_cms_reset (seeded addresses that the next stream never
    carries must not reappear in its set, the ones it does carry must)."""
    n = 600_000
    depth, wl2, seed = 4, 16, 0xBEEF
    KS = (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS))

    def stream(gseed, log2):
        gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=gseed, n_total=n, zipf_log2_universe=log2, zipf_s_x100=110)
        buf, off = po.gen_records(gp, 0, n)
        rows, status = po.decode_batch(buf, off, 1)
        assert status.sum() == 0
        return buf, off, rows

    def check(agg, rows, times, what):
        with np.errstate(over="ignore"):
            w = rows["bytes"] * rows["sampling_rate"]
        for col, ks in KS:
            want = po.cms_sketch_numpy(rows[col], w, depth, wl2, seed) * np.uint64(times)
            assert np.array_equal(agg.cms_read(ks).reshape(-1), want), (what, col)
            distinct = {bytes(k) for k in np.unique(rows[col], axis=0)}
            got = agg.topk(ks, 1 << 16)
            assert {bytes(r["key"]) for r in got} == distinct and len(got) == len(distinct), (what, col)

    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=16,
                    max_batch_records=n) as agg:
        buf, off, rows = stream(5, 11)
        for rep in range(4):   # launches 2..4 run on seeded caches
            agg.ingest(buf, off)
        check(agg, rows, 4, "four launches")
        for _, ks in KS:
            agg.cms_reset(ks)
        buf2, off2, rows2 = stream(6, 9)   # 512 addresses: the head is shared with the first stream, its tail is gone
        agg.ingest(buf2, off2)
        check(agg, rows2, 1, "after the reset")
        agg.ingest(buf2, off2)
        check(agg, rows2, 2, "after the reset, seeded")
        assert agg.stats()["records_bad"] == 0


@pytest.mark.parametrize("key_sets", [3, 5, 6])
def test_one_sketch_alone_through_the_scheduled_fold(gpu_lib, fa, po, key_sets):
    """cms_agg_kernel's schedule (persistent workgroups, heaviest partition first, heavy partitions in slices - sizes of the
    previous launch) with ONE sketch enabled (256 logical partitions instead of 512) and without the flows_5m key set: three
    launches of a skewed stream (the second and third run on a real schedule), counters against the CPU sketch."""
    n = 1_500_000
    depth, wl2, seed = 4, 20, 0x51CE
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=44, n_total=n, zipf_log2_universe=16, zipf_s_x100=110)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    with fa.FlowAgg(framed=True, key_sets=key_sets, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=18,
                    max_batch_records=n) as agg:
        for rep in range(3):
            agg.ingest(buf, off)
        for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            if key_sets & ks:
                want = po.cms_sketch_numpy(rows[col], w, depth, wl2, seed) * np.uint64(3)
                assert np.array_equal(agg.cms_read(ks).reshape(-1), want), col
                distinct = {bytes(k) for k in np.unique(rows[col], axis=0)}
                assert {bytes(r["key"]) for r in agg.topk(ks, 1 << 18)} == distinct, col
        st = agg.stats()
        assert st["records_ok"] == 3 * n and st["records_bad"] == 0
        if key_sets & fa.FA_KEYS_AS_PAIR:
            ref = po.Rollup(300)
            ref.ingest(buf, off, 1)
            want_rows = ref.rows()
            for c in ("bytes", "packets", "count"):
                want_rows[c] *= np.uint64(3)
            assert agg.read_window().tobytes() == want_rows.tobytes()


@pytest.mark.parametrize("chunks,cap_log2,key_sets", [(8, 20, 9), (1, 20, 9), (0, 20, 9), (2, 10, 9), (8, 18, 63)])
def test_wide_log_mode_equals_the_oracle(gpu_lib, fa, po, monkeypatch, chunks, cap_log2, key_sets):
    """FA_WIDE=log: a launch's (SrcAddr,DstPort,Proto) tuples stay in their scatter segments (chunks) instead of being folded into
    the hash table; reads take table rows and chunk tuples through the same device merge; a close of the oldest buckets is the
    chunks' watermark, any other drop folds them first; more than FA_WIDE_LOG_CHUNKS pending chunks: the oldest is folded
    (wagg_kernel, or the atomic replay once the table has grown - cap_log2 10).  Whatever the mix: rows == the oracle's."""
    monkeypatch.setenv("FA_WIDE", "log")
    monkeypatch.setenv("FA_WIDE_LOG_CHUNKS", str(chunks))
    monkeypatch.setenv("FA_WLOG_RANGE_CHECK", "1")  # the chunks' bucket ranges as the ingest kernel kept them == a scan of the chunks
    n, parts = 240_000, 4
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=77, n_total=n, zipf_log2_universe=12, span_secs=900)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    sub = 60

    def want(ts, sel=None):
        r, s = (rows, status) if sel is None else (rows[sel], status[sel])
        if ts == fa.ALL_TIMESLOTS:
            return po.rollup_app(r, s, sub).astype(fa.ROW_APP_DTYPE)
        return po.rollup_app(r, s, sub, window=300, timeslot=ts).astype(fa.ROW_APP_DTYPE)

    with fa.FlowAgg(framed=True, key_sets=key_sets, subwindow_secs=sub, wide_capacity_log2=cap_log2, cms_width_log2=12, topk_capacity_log2=14,
                    max_batch_records=n // parts) as agg:
        step = n // parts
        for i in range(parts):
            a, b = i * step, (i + 1) * step
            agg.ingest(buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a])
        slots = agg.open_timeslots()
        t0 = int(slots[0])
        for ts in (fa.ALL_TIMESLOTS, t0, t0 + 60, t0 + 300):
            assert agg.read_window_app(ts).tobytes() == want(ts).tobytes(), ("read", ts)
        # flows_5m is untouched by the mode
        ref = po.Rollup(sub)
        ref.ingest(buf, off, 1)
        assert agg.read_window().tobytes() == ref.rows().tobytes()
        # a sliding close removes the oldest sub-bucket only ...
        t32 = rows["time_received"].astype(np.uint64).astype(np.uint32)
        assert agg.close_window_app(t0).tobytes() == want(t0).tobytes()
        alive = t32 >= t0 + 60
        assert agg.read_window_app(t0 + 60).tobytes() == want(t0 + 60, alive).tobytes()
        assert agg.read_window_app(fa.ALL_TIMESLOTS).tobytes() == want(fa.ALL_TIMESLOTS, alive).tobytes()
        # ... more records (a launch behind the drop: its tuples of old buckets are late rows, not dropped ones) ...
        agg.ingest(buf[: int(off[step])], off[: step + 1])
        sel2 = np.concatenate([np.nonzero(alive)[0], np.arange(step)])
        assert agg.read_window_app(fa.ALL_TIMESLOTS).tobytes() == want(fa.ALL_TIMESLOTS, sel2).tobytes()
        # ... and a drop that is NOT the oldest buckets (the chunks are folded into the table first)
        mid = t0 + 300
        got = agg.close_window_app(mid)
        assert got.tobytes() == want(mid, sel2).tobytes()
        keep = sel2[~((t32[sel2] >= mid) & (t32[sel2] < mid + 60))]
        assert agg.read_window_app(fa.ALL_TIMESLOTS).tobytes() == want(fa.ALL_TIMESLOTS, keep).tobytes()
        st = agg.stats()
        assert st["records_bad"] == 0 and st["records_ok"] == n + step
        assert st["wave_tile_launches"] == parts + 1  # (every launch went through the scatter sink: its tuples were a chunk)


@pytest.mark.parametrize("sub", [0, 60])
def test_wide_log_bucket_ranges_of_a_stream_out_of_time_order(gpu_lib, fa, po, monkeypatch, sub):
    """Kafka partitions are close to time-ordered, the ingest kernel's bookkeeping of a chunk's bucket range has a short way for a
    tile whose records share a bucket - here the records arrive in random order (every tile spans all windows): the ranges still
    equal a scan of the chunks (FA_WLOG_RANGE_CHECK), the closes in window order still return the oracle's rows and leave nothing."""
    monkeypatch.setenv("FA_WIDE", "log")
    monkeypatch.setenv("FA_WLOG_RANGE_CHECK", "1")
    n, parts = 200_000, 4
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=177, n_total=n, zipf_log2_universe=12, span_secs=1200)
    buf0, off0 = po.gen_records(gp, 0, n)
    raw = bytes(buf0)
    order = np.random.default_rng(5).permutation(n)
    recs = [raw[int(off0[k]):int(off0[k + 1])] for k in order]
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in recs])
    buf = np.frombuffer(b"".join(recs), dtype=np.uint8)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    gran = sub or 300
    with fa.FlowAgg(framed=True, key_sets=9, subwindow_secs=sub, wide_capacity_log2=18, max_batch_records=n // parts) as agg:
        step = n // parts
        for i in range(parts):
            a, b = i * step, (i + 1) * step
            agg.ingest(buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a])
        assert agg.stats()["wide_log_chunks"] >= 1
        if sub:  # sliding form: one read of everything (the ranges are fetched - and checked - by the first read)
            assert agg.read_window_app(fa.ALL_TIMESLOTS).tobytes() == po.rollup_app(rows, status, gran).astype(fa.ROW_APP_DTYPE).tobytes()
        else:
            for ts in [int(t) for t in agg.open_timeslots()]:
                want = po.rollup_app(rows, status, 300, window=300, timeslot=ts).astype(fa.ROW_APP_DTYPE)
                assert agg.close_window_app(ts).tobytes() == want.tobytes(), ts
            assert len(agg.read_window_app(fa.ALL_TIMESLOTS)) == 0
