"""fa_topk: the pre-selection of the rows that can rank among the first k (maintenance.cuh, topk_scan_kernel) and the
candidates mode (fa_config.topk_mode = FA_TOPK_CANDIDATES) against its restatement in oracle/pyoracle.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ranked(keys, est):
    return sorted(zip((-est.astype(object)).tolist(), [bytes(k) for k in keys]))


@pytest.mark.parametrize("zipf_log2,wl2", [(12, 12), (16, 14)])
def test_topk_is_a_prefix_of_the_full_ranking_for_every_k(gpu_lib, fa, po, zipf_log2, wl2):
    """k rows out of many: whatever the histogram's threshold bin selects, the first k rows are the first k of the ranking of
    EVERY distinct address (ties by key) - also when most estimates fall into a few bins (small sketch, heavy ties)."""
    n = 150_000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=1201, n_total=n, zipf_log2_universe=zipf_log2)
    buf, off = po.gen_records(gp, 0, n)
    rows = po.gen_rows(gp, 0, n)
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    depth, seed = 4, 0xABCD
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=zipf_log2 + 2) as agg:
        agg.ingest(buf, off)
        for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            sk = po.cms_sketch_numpy(rows[col], w, depth, wl2, seed)
            keys = np.unique(np.ascontiguousarray(rows[col]), axis=0)
            want = _ranked(keys, po.cms_estimates_numpy(sk, keys, depth, wl2, seed))
            full = agg.topk(ks, 1 << 22)
            assert [(bytes(r["key"]), int(r["weight"])) for r in full] == [(k, -e) for e, k in want]
            ks_up = (1, 2, 7, 64, 100, 1000, len(want) - 1, len(want), len(want) + 5)
            for k in ks_up + ks_up[::-1] + (100, 100):  # (descending: the reads that know a bin the k-th estimate reaches - one pass)
                got = agg.topk(ks, k)
                assert got.tobytes() == full[:k].tobytes(), k
        # more records: estimates and sets have grown, the remembered bins are still lower bounds
        agg.ingest(buf[:int(off[n // 2])], off[:n // 2 + 1])
        for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            sk = po.cms_sketch_numpy(np.concatenate([rows[col], rows[col][:n // 2]]), np.concatenate([w, w[:n // 2]]), depth, wl2, seed)
            keys = np.unique(np.ascontiguousarray(rows[col]), axis=0)
            want = _ranked(keys, po.cms_estimates_numpy(sk, keys, depth, wl2, seed))
            for k in (100, 7, 1000):
                assert [(bytes(r["key"]), int(r["weight"])) for r in agg.topk(ks, k)] == [(kk, -e) for e, kk in want[:k]], k
        agg.cms_reset(fa.FA_KEYS_SRCADDR_CMS)  # everything starts over: a remembered bin would be far too high
        agg.ingest(buf[:int(off[2000])], off[:2001])
        sk = po.cms_sketch_numpy(rows["src_addr"][:2000], w[:2000], depth, wl2, seed)
        keys = np.unique(np.ascontiguousarray(rows["src_addr"][:2000]), axis=0)
        want = _ranked(keys, po.cms_estimates_numpy(sk, keys, depth, wl2, seed))
        assert [(bytes(r["key"]), int(r["weight"])) for r in agg.topk(fa.FA_KEYS_SRCADDR_CMS, 50)] == [(kk, -e) for e, kk in want[:50]]


def _batches(po, n, nb, seed, zipf_log2, junk=False):
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=seed, n_total=n, zipf_log2_universe=zipf_log2)
    step = n // nb
    out = []
    for b in range(nb):
        buf, off = po.gen_records(gp, b * step, step)
        rows = po.gen_rows(gp, b * step, step)
        if junk and b == 1:  # records only the complete parser takes (3-byte tag in front) and a malformed one
            raw = bytes(buf)
            extra = bytes.fromhex("0480800101")  # field 2048 (a 3-byte tag: only the complete parser decides it): valid, nothing projected - an all-zero record, address ::
            raw += extra + bytes.fromhex("0570ffffffff")
            off = np.concatenate([off, [int(off[-1]) + len(extra), int(off[-1]) + len(extra) + 6]]).astype(np.uint64)  # (+ a malformed one: a 5-byte frame whose varint never ends)
            buf = np.frombuffer(raw, dtype=np.uint8)
            z = np.zeros(1, dtype=rows.dtype)
            rows = np.concatenate([rows, z])
        out.append((buf, off, rows))
    return out


@pytest.mark.parametrize("n,nb,zipf_log2,wl2,track,cap", [(480_000, 6, 14, 14, 64, 12), (60_000, 12, 12, 12, 32, 10), (480_000, 4, 16, 16, 1024, 14)])
def test_candidates_mode_equals_its_restatement(gpu_lib, fa, po, n, nb, zipf_log2, wl2, track, cap):
    """Every launch is a batch of the contract: the candidates the library holds after the stream, their estimates and the
    ranking == oracle/pyoracle.py topk_candidates - through the scatter sink (80 k-record launches), the direct sink (5 k) and
    the deferred parsers; sketches bit-exact as in the exact mode."""
    depth, seed = 4, 0x5EED
    bs = _batches(po, n, nb, 1301 + nb, zipf_log2, junk=True)
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed, topk_capacity_log2=cap,
                    topk_mode=fa.TOPK_CANDIDATES, topk_track=track) as agg:
        for buf, off, _ in bs:
            agg.ingest(buf, off)
        st = agg.stats()
        assert st["kernel_launches"] == nb and st["records_bad"] == 1
        for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            batches = []
            for _, _, rows in bs:
                with np.errstate(over="ignore"):
                    batches.append((rows[col], rows["bytes"] * rows["sampling_rate"]))
            sk, cand, est, thetas = po.topk_candidates(batches, depth, wl2, seed, track=track, capacity_log2=cap)
            assert np.array_equal(agg.cms_read(ks).reshape(-1), sk)
            want = _ranked(cand, est)
            got = agg.topk(ks, 1 << 20)
            assert len(cand) >= 16 and thetas[-1] > 1
            assert [(bytes(r["key"]), int(r["weight"])) for r in got] == [(k, -e) for e, k in want], col
            assert agg.topk(ks, 10).tobytes() == got[:10].tobytes()
        # a reset starts the contract over: nothing is admitted during the next first batch
        agg.cms_reset(fa.FA_KEYS_SRCADDR_CMS)
        agg.ingest(bs[0][0], bs[0][1])
        assert len(agg.topk(fa.FA_KEYS_SRCADDR_CMS, 10)) == 0
        agg.ingest(bs[1][0], bs[1][1])
        assert len(agg.topk(fa.FA_KEYS_SRCADDR_CMS, 10)) == 10


def test_candidates_mode_finds_the_exact_modes_top_k_on_a_skewed_stream(gpu_lib, fa, po):
    """The two contracts differ in what they keep, not - on a stream whose heavy hitters recur in every batch - in what they
    report: top 100 of the candidates == top 100 of every address."""
    n, nb = 1_200_000, 8
    bs = _batches(po, n, nb, 1401, 18)
    kw = dict(framed=True, key_sets=7, cms_width_log2=16)
    with fa.FlowAgg(topk_capacity_log2=20, **kw) as exact, fa.FlowAgg(topk_capacity_log2=14, topk_mode=fa.TOPK_CANDIDATES, **kw) as cand:
        for buf, off, _ in bs:
            exact.ingest(buf, off)
            cand.ingest(buf, off)
        for ks in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS):
            assert np.array_equal(exact.cms_read(ks), cand.cms_read(ks))
            assert cand.topk(ks, 100).tobytes() == exact.topk(ks, 100).tobytes()
            assert len(cand.topk(ks, 1 << 20)) < len(exact.topk(ks, 1 << 20)) // 8
        assert cand.read_window().tobytes() == exact.read_window().tobytes()


@pytest.mark.parametrize("depth,wl2,track,cap", [(1, 4, 8, 14), (3, 9, 16, 12), (5, 20, 64, 12), (16, 8, 32, 14), (2, 5, 4, 14)])
def test_odd_sketch_geometries_in_both_topk_modes(gpu_lib, fa, po, depth, wl2, track, cap):
    """Sketches that do not go through the scatter sink (too narrow, too deep: every update is an atomic), rows beyond the fourth in
    the estimates, fewer than 64 counters per row in the candidate bits: the exact mode's reads are prefixes of the full ranking,
    the candidates mode equals its restatement."""
    n, nb, seed = 200_000, 5, 0xC0FFEE
    bs = _batches(po, n, nb, 1500 + depth, 12)
    rows_all = np.concatenate([r for _, _, r in bs])
    with np.errstate(over="ignore"):
        w_all = rows_all["bytes"] * rows_all["sampling_rate"]
    kw = dict(framed=True, key_sets=7, cms_depth=depth, cms_width_log2=wl2, cms_seed=seed)
    with fa.FlowAgg(topk_capacity_log2=15, **kw) as exact, fa.FlowAgg(topk_capacity_log2=cap, topk_mode=fa.TOPK_CANDIDATES, topk_track=track, **kw) as cand:
        for buf, off, _ in bs:
            exact.ingest(buf, off)
            cand.ingest(buf, off)
        for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
            sk = po.cms_sketch_numpy(rows_all[col], w_all, depth, wl2, seed)
            assert np.array_equal(exact.cms_read(ks).reshape(-1), sk) and np.array_equal(cand.cms_read(ks).reshape(-1), sk)
            keys = np.unique(np.ascontiguousarray(rows_all[col]), axis=0)
            want = _ranked(keys, po.cms_estimates_numpy(sk, keys, depth, wl2, seed))
            full = exact.topk(ks, 1 << 20)
            assert [(bytes(r["key"]), int(r["weight"])) for r in full] == [(k, -e) for e, k in want]
            for k in (5, 100, 1000, 100, 5):
                assert exact.topk(ks, k).tobytes() == full[:k].tobytes(), k
            batches = []
            for _, _, rows in bs:
                with np.errstate(over="ignore"):
                    batches.append((rows[col], rows["bytes"] * rows["sampling_rate"]))
            _, cset, cest, thetas = po.topk_candidates(batches, depth, wl2, seed, track=track, capacity_log2=cap)
            got = cand.topk(ks, 1 << 20)
            assert [(bytes(r["key"]), int(r["weight"])) for r in got] == [(k, -e) for e, k in _ranked(cset, cest)], (col, len(got), len(cset))
            st = cand.stats()
            assert st["topk_theta_" + ("src" if col == "src_addr" else "dst")] == thetas[-1]


def test_candidates_mode_equals_the_golden_fixture(gpu_lib, fa, po):
    """tests/golden/topk_contract.json pins the candidates contract on a seeded stream (thresholds, candidates held, the ranking's
    digest; the restatement is pinned to it on the CPU): the library, fed the same stream in the same launches, lands on it."""
    import hashlib
    import json
    import os
    c = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "topk_contract.json")))["candidates"]
    n, nb = c["records"], c["batches"]
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=c["generator_seed"], n_total=n, zipf_log2_universe=c["zipf_log2_universe"])
    buf, off = po.gen_records(gp, 0, n)
    step = n // nb
    with fa.FlowAgg(framed=True, key_sets=7, cms_depth=c["depth"], cms_width_log2=c["width_log2"], cms_seed=c["seed"], topk_capacity_log2=c["capacity_log2"],
                    topk_mode=fa.TOPK_CANDIDATES, topk_track=c["track"], max_batch_records=step) as agg:
        for i in range(nb):
            a, b = i * step, (i + 1) * step
            agg.ingest(buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a])
        st = agg.stats()
        assert st["kernel_launches"] == nb
        for col, ks, tag in (("src_addr", fa.FA_KEYS_SRCADDR_CMS, "src"), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS, "dst")):
            want = c["sets"][col]
            got = agg.topk(ks, 1 << 16)
            assert st["topk_theta_" + tag] == want["thetas"][-1] and len(got) == want["candidates"]
            h = hashlib.sha256()
            for r in got:
                h.update(bytes(r["key"]) + int(r["weight"]).to_bytes(8, "little"))
            assert h.hexdigest() == want["ranking_sha256"]
            assert [[bytes(r["key"]).hex(), int(r["weight"])] for r in got[:5]] == want["first"]


def _zero_weight_records(fa, n_heavy, n_zero, scatter_pad=0):
    """Hand-made framed records: n_heavy addresses with Bytes * SamplingRate > 0 and n_zero addresses whose records all carry
    SamplingRate = 0 (the dashboards' weight, viz-ch.json:233, is then 0: still a row of GROUP BY SrcAddr)."""
    ev = fa.schema.encode_varint
    recs, keys, weights = [], [], []
    for i in range(n_heavy + n_zero):
        addr = bytes([10, i >> 16 & 255, i >> 8 & 255, i & 255]) + bytes(12)
        rate = 0 if i >= n_heavy else 1 + i % 7
        nbytes = 100 + i
        p = b"\x10" + ev(fa.T0 + 1)                       # TimeReceived (2)
        if rate:
            p += b"\x18" + ev(rate)                        # SamplingRate (3): proto3 leaves a zero out
        p += b"\x32\x10" + addr + b"\x3a\x10" + addr      # SrcAddr (6), DstAddr (7)
        p += b"\x48" + ev(nbytes) + b"\x50" + ev(1)       # Bytes (9), Packets (10)
        p += b"\x70" + ev(65000) + b"\x78" + ev(65001)    # SrcAS (14), DstAS (15)
        p += b"\xf0\x01" + ev(0x800)                       # Etype (30)
        recs.append(fa.schema.frame(p))
        keys.append(addr)
        weights.append(rate * nbytes)
    recs = recs * (1 + scatter_pad)
    off = np.concatenate([[0], np.cumsum([len(r) for r in recs])]).astype(np.uint64)
    return np.frombuffer(b"".join(recs), dtype=np.uint8), off, keys, [w * (1 + scatter_pad) for w in weights]


@pytest.mark.parametrize("n_heavy,n_zero,pad", [(40, 300, 0), (0, 500, 0), (3000, 5000, 9)])
def test_topk_keeps_zero_weight_addresses(gpu_lib, fa, n_heavy, n_zero, pad):
    """an address whose estimate is 0 is still a row of the ranking.  With the threshold in bin 0 (fewer than k
    keys, or a k-th estimate of 0) the histogram path must return those rows too: fa_topk(k) == the first k rows of fa_topk(all)
    == the ranking of every distinct address, through the direct sink (small batch) and the scatter sink (80 k records)."""
    buf, off, keys, weights = _zero_weight_records(fa, n_heavy, n_zero, scatter_pad=pad)
    with fa.FlowAgg(framed=True, key_sets=7, cms_width_log2=16, topk_capacity_log2=15) as agg:  # (2^16 columns: no collisions among <= 8000 keys would be luck -
        agg.ingest(buf, off)                                                                   #  the expectation below is computed from the sketch itself)
        assert agg.stats()["records_bad"] == 0
        for ks in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS):
            est = {k: agg.cms_query(ks, k) for k in keys}
            assert all(est[k] >= w for k, w in zip(keys, weights))
            want = sorted(((-e, k) for k, e in est.items()))
            full = agg.topk(ks, 1 << 15)
            assert [(bytes(r["key"]), int(r["weight"])) for r in full] == [(k, -e) for e, k in want]
            nz = sum(1 for e in est.values() if e)
            for k in (1, max(nz - 1, 1), nz, nz + 1, nz + 50, len(keys) - 1, len(keys), len(keys) + 10, nz + 1, 1):
                got = agg.topk(ks, k)
                assert got.tobytes() == full[:k].tobytes(), (k, len(got), nz)
