"""The window close of a GROUP of contexts inside one process (ABI 7, fa_group_*; csrc/group_host.inc).

The reference's consumer is one process with a goroutine per claimed partition (inserter/inserter.go:167-196).  Here every
partition has its own ctx - all of them on the one GPU of the test box, which exercises everything but the xGMI hop: the
peer-copy transport degrades to device copies - and the group's results are compared, byte for byte, with ONE ctx that
ingested every partition, with the oracle's rollup of the whole stream, and with the host-side merges dist.py's tests use."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _partitions(po, n, seed, nparts, zipf_log2=14, span_secs=900):
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=seed, n_total=n, zipf_log2_universe=zipf_log2, span_secs=span_secs)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)
    parts = []
    for p in range(nparts):  # record i belongs to Kafka partition i % nparts
        recs = [raw[int(off[k]):int(off[k + 1])] for k in range(p, n, nparts)]
        o = np.zeros(len(recs) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(r) for r in recs])
        parts.append((np.frombuffer(b"".join(recs), dtype=np.uint8), o))
    return buf, off, parts


def _sorted_app(rows):
    addr = np.ascontiguousarray(rows["src_addr"])
    hi = addr[:, :8].copy().view(">u8").reshape(-1)
    lo = addr[:, 8:].copy().view(">u8").reshape(-1)
    return rows[np.lexsort((rows["proto"], rows["dst_port"], lo, hi, rows["timeslot"], rows["date"]))]


@pytest.mark.parametrize("nparts,sub", [(2, 0), (2, 60), (8, 0), (8, 60), (3, 0)])
def test_group_close_equals_one_ctx_that_ingested_everything(gpu_lib, fa, po, nparts, sub):
    n = 240_000
    buf, off, parts = _partitions(po, n, seed=700 + nparts, nparts=nparts)
    kw = dict(framed=True, key_sets=63, cms_width_log2=14, topk_capacity_log2=16, subwindow_secs=sub)
    ref = po.Rollup(sub or 300)
    assert ref.ingest(buf, off, 1) == 0
    members = [fa.FlowAgg(**kw) for _ in range(nparts)]
    whole = fa.FlowAgg(**kw)
    try:
        for m, (b, o) in zip(members, parts):
            m.ingest(b, o)
        whole.ingest(buf, off)
        with fa.FlowGroup(members) as g:
            assert g.transport == fa.GROUP_PEER
            slots = whole.open_timeslots()
            assert g.open_timeslots().tobytes() == slots.tobytes()
            st = g.stats()
            assert st["records_ok"] == n and st["records_bad"] == 0 and st["bytes_in"] == len(buf) and st["batches"] == nparts
            # flows_5m: the whole table == the oracle's rollup of ALL partitions; windows (tumbling / sliding) == the single ctx
            assert g.read_window(fa.ROWS_5M).tobytes() == ref.rows().tobytes()
            t0 = int(slots[0])
            windows = [fa.ALL_TIMESLOTS, t0] + ([t0 + 60, t0 + 240] if sub else [t0 + 300])
            for ts in windows:
                want = whole.read_window(ts)
                got = g.read_window(fa.ROWS_5M, ts)
                assert got.tobytes() == want.tobytes() and (len(want) or ts != t0)
                # ... and == the host-side merge of the members' own windows (what dist.py's CPU tests use)
                assert fa.dist.merge_rows_host([m.read_window(ts) for m in members]).tobytes() == want.tobytes()
                want = whole.read_window_app(ts)
                assert g.read_window(fa.ROWS_APP, ts).tobytes() == want.tobytes()
                # hash-partitioned: the shares back to back, every key in exactly one share - the share of its owner
                rows, shares = g.read_window_partitioned(fa.ROWS_APP, ts)
                assert sum(shares) == len(rows) == len(want) and len(shares) == nparts
                owner = fa.dist.partition_rows_host(rows, fa.ROWS_APP, nparts)
                assert owner.tolist() == np.repeat(np.arange(nparts), shares).tolist()
                at = 0
                for s in shares:  # inside a share: emit order
                    assert rows[at:at + s].tobytes() == _sorted_app(rows[at:at + s]).tobytes()
                    at += s
                assert _sorted_app(rows).tobytes() == want.tobytes()
                rows5, shares5 = g.read_window_partitioned(fa.ROWS_5M, ts)
                assert fa.dist.merge_rows_host([rows5]).tobytes() == whole.read_window(ts).tobytes() and sum(shares5) == len(rows5)
            # dashboards: ports (cut at k AFTER the merge), minutes
            for dst, kind in ((0, fa.ROWS_PORT_SRC), (1, fa.ROWS_PORT_DST)):
                assert g.read_window(kind, k=50).tobytes() == whole.top_ports(dst, 50).tobytes()
                assert g.read_window(kind).tobytes() == whole.top_ports(dst).tobytes()
            assert g.read_window(fa.ROWS_MINUTE).tobytes() == whole.minute_series().tobytes()
            # sketches + top-k: every member's merged view == the single ctx's sketch; the group's top-k == its top-k
            for key_set in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS):
                assert g.topk(key_set, 100).tobytes() == whole.topk(key_set, 100).tobytes()
                sk = whole.cms_read(key_set)
                for m in members:
                    assert m.cms_read(key_set).tobytes() == sk.tobytes()
            g.allreduce_sketches()  # again: out of place, nothing is counted twice
            assert members[-1].cms_read(fa.FA_KEYS_SRCADDR_CMS).tobytes() == whole.cms_read(fa.FA_KEYS_SRCADDR_CMS).tobytes()
            # closes remove from every member what the single ctx's close removes
            assert g.close_window(fa.ROWS_5M, t0).tobytes() == whole.close_window(t0).tobytes()
            assert g.read_window(fa.ROWS_5M).tobytes() == whole.read_window().tobytes()
            rows, shares = g.close_window_partitioned(fa.ROWS_APP, t0)
            assert _sorted_app(rows).tobytes() == whole.close_window_app(t0).tobytes()
            assert g.read_window(fa.ROWS_APP).tobytes() == whole.read_window_app().tobytes()
            assert g.close_window(fa.ROWS_APP).tobytes() == whole.close_window_app().tobytes()
            assert len(g.read_window(fa.ROWS_APP)) == 0 and len(g.read_window_partitioned(fa.ROWS_APP)[0]) == 0
            # ingest goes on behind a close; the merged view is stale then and the next top-k recomputes it
            members[0].ingest(*parts[0])
            whole.ingest(*parts[0])
            assert g.topk(fa.FA_KEYS_SRCADDR_CMS, 20).tobytes() == whole.topk(fa.FA_KEYS_SRCADDR_CMS, 20).tobytes()
            assert g.read_window(fa.ROWS_5M).tobytes() == whole.read_window().tobytes()
    finally:
        for m in members + [whole]:
            m.close()


def test_group_of_one_and_small_buffers(gpu_lib, fa, po):
    """n = 1 is the single ctx; FA_ERR_CAPACITY reports the rows needed and removes nothing."""
    import ctypes as C
    n = 60_000
    buf, off, parts = _partitions(po, n, seed=811, nparts=2)
    kw = dict(framed=True, key_sets=9)
    with fa.FlowAgg(**kw) as a, fa.FlowAgg(**kw) as b, fa.FlowAgg(**kw) as whole:
        a.ingest(*parts[0])
        b.ingest(*parts[1])
        whole.ingest(buf, off)
        with fa.FlowGroup([a]) as g1:
            assert g1.read_window(fa.ROWS_5M).tobytes() == a.read_window().tobytes()
            assert g1.read_window_partitioned(fa.ROWS_APP)[0].tobytes() == a.read_window_app().tobytes()
        with fa.FlowGroup([a, b]) as g:
            want = whole.read_window()
            out = np.empty(4, dtype=fa.ROW5M_DTYPE)
            need = C.c_size_t()
            L = fa.lib()
            for fn in (lambda: L.fa_group_close_window(g._h, fa.ROWS_5M, fa.ALL_TIMESLOTS, out.ctypes.data, 4, C.byref(need)),
                       lambda: L.fa_group_close_window_partitioned(g._h, fa.ROWS_5M, fa.ALL_TIMESLOTS, out.ctypes.data, 4, None, C.byref(need))):
                assert fn() == -6 and need.value == len(want)
            assert g.read_window(fa.ROWS_5M, cap=4).tobytes() == want.tobytes()  # (the binding asks again with room)
            assert g.close_window(fa.ROWS_5M).tobytes() == want.tobytes()
            assert len(g.read_window(fa.ROWS_5M)) == 0


def test_a_failing_member_fails_the_call_for_everyone_and_drops_nothing(gpu_lib, fa, po):
    n = 120_000
    buf, off, parts = _partitions(po, n, seed=821, nparts=2, zipf_log2=20)
    kw = dict(framed=True, key_sets=7, cms_width_log2=14)
    with fa.FlowAgg(topk_capacity_log2=18, **kw) as a, fa.FlowAgg(topk_capacity_log2=8, **kw) as b, fa.FlowAgg(topk_capacity_log2=18, **kw) as whole:
        a.ingest(*parts[0])
        b.ingest(*parts[1])  # far more distinct addresses than 2^8 slots: its top-k cannot be answered
        whole.ingest(buf, off)
        with fa.FlowGroup([a, b]) as g:
            with pytest.raises(fa.FlowAggError) as ei:
                g.topk(fa.FA_KEYS_SRCADDR_CMS, 10)
            assert ei.value.code == -5 and "member 1" in str(ei.value)
            assert b"member 1" in fa.lib().fa_last_error(a._h)  # the other member reports it too
            # the group is still usable for what does not need the failed piece, and nothing was dropped
            assert g.read_window(fa.ROWS_5M).tobytes() == whole.read_window().tobytes()
            sk = whole.cms_read(fa.FA_KEYS_SRCADDR_CMS)
            g.allreduce_sketches()
            assert a.cms_read(fa.FA_KEYS_SRCADDR_CMS).tobytes() == sk.tobytes()


def test_group_create_checks_its_members(gpu_lib, fa):
    with fa.FlowAgg(key_sets=1) as a, fa.FlowAgg(key_sets=9) as b, fa.FlowAgg(key_sets=1, subwindow_secs=60) as c:
        for bad in ([a, b], [a, c], [a, a], []):
            with pytest.raises(fa.FlowAggError) as ei:
                fa.FlowGroup(bad)
            assert ei.value.code == -1
        with pytest.raises(fa.FlowAggError) as ei:  # RCCL wants a GPU per member
            fa.FlowGroup([a, fa.FlowAgg(key_sets=1)], transport=fa.GROUP_RCCL)
        assert ei.value.code == -8
        with fa.FlowGroup([a]) as g:
            with pytest.raises(fa.FlowAggError) as ei:  # a ctx is a member of one group at a time
                fa.FlowGroup([a, fa.FlowAgg(key_sets=1)])
            assert ei.value.code == -1 and "already a member" in str(ei.value)
            with pytest.raises(fa.FlowAggError):
                g.allreduce_sketches()  # no sketch key set
            with pytest.raises(fa.FlowAggError):
                g.close_window(fa.ROWS_PORT_SRC)  # not a windowed kind


def test_group_create_wants_one_topk_contract(gpu_lib, fa):
    """a group ranks by ONE top-k contract - mixed exact / candidates members are refused, and so are candidates members
    whose thresholds would follow different ranks or set sizes (the capacity is part of theta); in the exact mode the capacity
    is only a size.  A ctx leaves its group when the group is destroyed."""
    kw = dict(key_sets=7, cms_width_log2=14)
    with fa.FlowAgg(topk_capacity_log2=12, **kw) as ex, fa.FlowAgg(topk_capacity_log2=14, **kw) as ex2, \
            fa.FlowAgg(topk_mode=fa.TOPK_CANDIDATES, topk_capacity_log2=12, **kw) as ca, \
            fa.FlowAgg(topk_mode=fa.TOPK_CANDIDATES, topk_capacity_log2=12, topk_track=64, **kw) as ca_track, \
            fa.FlowAgg(topk_mode=fa.TOPK_CANDIDATES, topk_capacity_log2=13, **kw) as ca_cap, \
            fa.FlowAgg(topk_mode=fa.TOPK_CANDIDATES, topk_capacity_log2=12, **kw) as ca2:
        for bad in ([ex, ca], [ca, ca_track], [ca, ca_cap]):
            with pytest.raises(fa.FlowAggError) as ei:
                fa.FlowGroup(bad)
            assert ei.value.code == -1 and "top-k" in str(ei.value)
        with fa.FlowGroup([ex, ex2]) as g:
            assert g.transport == fa.GROUP_PEER
        with fa.FlowGroup([ca, ca2]), fa.FlowGroup([ex, ex2]):  # (released above: members again)
            pass


def test_group_over_rccl_world_1(gpu_lib, fa, po):
    """FA_GROUP_RCCL through a real ncclCommInitAll - one member, the box has one GPU: communicator creation, the grouped
    ncclAllReduce into the merged view and the teardown run; world > 1 needs the 8-GPU node."""
    n = 80_000
    buf, off, _ = _partitions(po, n, seed=831, nparts=1)
    kw = dict(framed=True, key_sets=7, cms_width_log2=14, topk_capacity_log2=16)
    with fa.FlowAgg(**kw) as a, fa.FlowAgg(**kw) as whole:
        a.ingest(buf, off)
        whole.ingest(buf, off)
        with fa.FlowGroup([a], transport=fa.GROUP_RCCL) as g:
            assert g.transport == fa.GROUP_RCCL
            g.allreduce_sketches()
            st = a.device_state()
            assert st.cms_src_merged and a.cms_read(fa.FA_KEYS_SRCADDR_CMS).tobytes() == whole.cms_read(fa.FA_KEYS_SRCADDR_CMS).tobytes()
            assert g.topk(fa.FA_KEYS_DSTADDR_CMS, 50).tobytes() == whole.topk(fa.FA_KEYS_DSTADDR_CMS, 50).tobytes()
            assert g.read_window(fa.ROWS_5M).tobytes() == whole.read_window().tobytes()


def test_group_topk_in_candidates_mode(gpu_lib, fa, po):
    """Members that keep candidates only: every member admits by ITS sketch and threshold, the group ranks the union by the MERGED
    estimate - on a skewed stream whose heavy hitters recur in every batch of every partition the rows of a single ctx that keeps
    every address; the merged views equal its sketch; fa_group_stats reports the highest threshold and the candidates held."""
    n, nparts, nb = 960_000, 4, 6
    buf, off, parts = _partitions(po, n, seed=905, nparts=nparts, zipf_log2=18)
    kw = dict(framed=True, key_sets=7, cms_width_log2=16)
    members = [fa.FlowAgg(topk_capacity_log2=14, topk_mode=fa.TOPK_CANDIDATES, topk_track=128, **kw) for _ in range(nparts)]
    whole = fa.FlowAgg(topk_capacity_log2=20, **kw)
    try:
        for m, (b, o) in zip(members, parts):
            step = (len(o) - 1) // nb
            for i in range(nb):  # six launches per member: candidates join from the second on
                lo, hi = i * step, (len(o) - 1 if i == nb - 1 else (i + 1) * step)
                m.ingest(b[int(o[lo]):int(o[hi])], o[lo:hi + 1] - o[lo])
        whole.ingest(buf, off)
        with fa.FlowGroup(members) as g:
            for ks in (fa.FA_KEYS_SRCADDR_CMS, fa.FA_KEYS_DSTADDR_CMS):
                assert g.topk(ks, 50).tobytes() == whole.topk(ks, 50).tobytes()
                assert members[1].cms_read(ks).tobytes() == whole.cms_read(ks).tobytes()
            st = g.stats()
            assert st["records_ok"] == n and st["topk_theta_src"] > 1 and 50 <= st["topk_candidates_src"] < 4 * (1 << 14)
    finally:
        for m in members + [whole]:
            m.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("FA_FUZZ_SEEDS", "10"))))  # (soak runs: FA_FUZZ_SEEDS=100)
def test_random_sessions_of_a_group_against_a_model(gpu_lib, fa, po, seed):
    """What a consumer does to a group over a day, drawn at random: batches into the members (with offsets, or as a bare framed
    chain the library cuts itself), closes of the oldest window of the whole topic in both forms, reads, top-k of either contract
    in between, a sketch reset - against a model kept in numpy (rows of the batches merged the way SummingMergeTree would, rows
    of a closed window taken out; late records open their window again)."""
    rng = np.random.default_rng(7000 + seed)
    nm = int(rng.integers(1, 5))
    nb = int(rng.integers(3, 9))
    n = int(rng.integers(60_000, 400_000))
    depth, wl2, cseed = int(rng.integers(2, 5)), int(rng.integers(15, 18)), int(rng.integers(1, 1 << 30))
    candidates = bool(rng.integers(0, 2))
    gp = po.gen_params(mode=int(rng.choice([po.GEN_ZIPF, po.GEN_ASPAIRS, po.GEN_GOFLOW])), framed=1, seed=7100 + seed, n_total=n,
                       span_secs=int(rng.choice([600, 1500])), zipf_log2_universe=int(rng.integers(8, 15)), zipf_s_x100=110)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    assert status.sum() == 0
    with np.errstate(over="ignore"):
        wgt = rows["bytes"] * rows["sampling_rate"]
    bounds = [0] + sorted(int(c) for c in rng.choice(np.arange(1, n), size=nb - 1, replace=False)) + [n]
    kw = dict(framed=True, key_sets=15, cms_depth=depth, cms_width_log2=wl2, cms_seed=cseed, topk_capacity_log2=16 if candidates else 20,
              topk_mode=fa.TOPK_CANDIDATES if candidates else fa.TOPK_EXACT, topk_track=64, wide_capacity_log2=int(rng.integers(10, 20)),
              table_capacity_log2=int(rng.integers(10, 18)), max_batch_records=max(b - a for a, b in zip(bounds, bounds[1:])))
    members = [fa.FlowAgg(**kw) for _ in range(nm)]
    live5 = np.zeros(0, dtype=fa.ROW5M_DTYPE)
    live_app = np.zeros(0, dtype=fa.ROW_APP_DTYPE)
    seen = np.zeros(n, dtype=bool)  # records the sketches hold
    per_member = [[] for _ in range(nm)]  # candidates mode: every member follows the contract over ITS launches
    try:
        with fa.FlowGroup(members) as g:
            for i, (a, b) in enumerate(zip(bounds, bounds[1:])):
                m = int(rng.integers(0, nm))
                piece, o = buf[int(off[a]):int(off[b])], off[a:b + 1] - off[a]
                if rng.integers(0, 3) == 0:
                    members[m].ingest(piece, None)  # a framed chain: cut by the library (on the device from 1 MiB)
                else:
                    members[m].ingest(piece, o)
                ref = po.Rollup(300)
                ref.ingest(piece, o, 1)
                live5 = fa.dist.merge_rows_host([live5, ref.rows()])
                live_app = fa.dist.merge_rows_app_host([live_app, po.rollup_app(rows[a:b], status[a:b], 300).astype(fa.ROW_APP_DTYPE)])
                seen[a:b] = True
                per_member[m].append((a, b))
                op = int(rng.integers(0, 5))
                if op == 0 and len(live5):  # the oldest window of the topic leaves, flows_5m merged, (SrcAddr,DstPort,Proto) by owner
                    ts = int(g.open_timeslots()[0])
                    assert ts == int(live5["timeslot"].min())
                    got5 = g.close_window(fa.ROWS_5M, ts)
                    assert got5.tobytes() == live5[live5["timeslot"] == ts].tobytes(), (seed, i)
                    live5 = live5[live5["timeslot"] != ts]
                    gota, shares = g.close_window_partitioned(fa.ROWS_APP, ts)
                    assert sum(shares) == len(gota)
                    assert fa.dist.merge_rows_app_host([gota]).tobytes() == live_app[live_app["timeslot"] == ts].tobytes(), (seed, i)
                    assert len(fa.dist.merge_rows_app_host([gota])) == len(gota)  # a key in exactly one share
                    live_app = live_app[live_app["timeslot"] != ts]
                elif op == 1:
                    assert g.read_window(fa.ROWS_5M).tobytes() == live5.tobytes(), (seed, i)
                elif op == 2 and not candidates:  # the whole topic's heavy hitters: exact with respect to the merged sketch
                    for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
                        sk = po.cms_sketch_numpy(rows[col][seen], wgt[seen], depth, wl2, cseed)
                        keys = np.unique(np.ascontiguousarray(rows[col][seen]), axis=0)
                        est = po.cms_estimates_numpy(sk, keys, depth, wl2, cseed)
                        want = sorted(zip((-est.astype(object)).tolist(), [bytes(k) for k in keys]))[:50]
                        assert [(bytes(r["key"]), int(r["weight"])) for r in g.topk(ks, 50)] == [(k, -e) for e, k in want], (seed, i, col)
                elif op == 3 and not candidates and i + 1 < nb:  # every member's sketches and distinct-address sets start over
                    for mem in members:
                        mem.cms_reset(fa.FA_KEYS_SRCADDR_CMS)
                        mem.cms_reset(fa.FA_KEYS_DSTADDR_CMS)
                    seen[:] = False
            st = g.stats()
            assert st["records_ok"] == n and st["records_bad"] == 0
            assert g.read_window(fa.ROWS_5M).tobytes() == live5.tobytes()
            gota, _ = g.read_window_partitioned(fa.ROWS_APP)
            assert fa.dist.merge_rows_app_host([gota]).tobytes() == live_app.tobytes()
            g.allreduce_sketches()
            for col, ks in (("src_addr", fa.FA_KEYS_SRCADDR_CMS), ("dst_addr", fa.FA_KEYS_DSTADDR_CMS)):
                sk = po.cms_sketch_numpy(rows[col][seen], wgt[seen], depth, wl2, cseed)
                assert np.array_equal(members[int(rng.integers(0, nm))].cms_read(ks).reshape(-1), sk), (seed, col)
                if candidates:  # the union of the members' candidate sets, ranked by the merged estimate
                    cand = [po.topk_candidates([(rows[col][a:b], wgt[a:b]) for a, b in per_member[m]], depth, wl2, cseed, track=64, capacity_log2=16)[1]
                            for m in range(nm) if per_member[m]]
                    keys = np.unique(np.concatenate(cand), axis=0) if cand else np.zeros((0, 16), dtype=np.uint8)
                    est = po.cms_estimates_numpy(sk, keys, depth, wl2, cseed) if len(keys) else np.zeros(0, dtype=np.uint64)
                    want = sorted(zip((-est.astype(object)).tolist(), [bytes(k) for k in keys]))
                    got = g.topk(ks, 1 << 16)
                    assert [(bytes(r["key"]), int(r["weight"])) for r in got] == [(k, -e) for e, k in want], (seed, col, len(got), len(want))
    finally:
        for mem in members:
            mem.close()
