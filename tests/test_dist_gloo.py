"""N>1 host path on CPU: world_size-2 gloo.  Kafka partitions are sharded over ranks,
each rank aggregates its shard (here with the oracle standing in for the GPU - the
product kernels cannot run without a HIP device), and the window-close exchange
(all-gather of rows + re-aggregation; sketch all-reduce) is the real dist.py code."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import _pkg
    fa = _pkg.load()
    po = _pkg.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, nparts = 40000, 8
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=10, n_total=n)
    buf, off = po.gen_records(gp, 0, n)
    raw = bytes(buf)
    mine = fa.dist.partitions_of(rank, world, nparts)
    shard = po.Rollup(300)
    for p in mine:  # record i lives in partition i mod nparts (mocker sets no key, mocker.go:103-106)
        recs = [raw[int(off[k]):int(off[k + 1])] for k in range(p, n, nparts)]
        o = np.zeros(len(recs) + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(r) for r in recs])
        shard.ingest(np.frombuffer(b"".join(recs), dtype=np.uint8), o, 1)
    parts = fa.dist.allgather_rows(shard.rows().astype(fa.dist.ROW5M_DTYPE), device="cpu")
    merged = fa.dist.merge_rows_host(parts)
    whole = po.Rollup(300)
    whole.ingest(buf, off, 1)
    ok = merged.tobytes() == whole.rows().tobytes()
    # dense sketch merge: u64 wrap-around sum == int64 all-reduce bit for bit
    sk = np.full(1024, np.uint64(2**63 + 12345 + rank), dtype=np.uint64)
    t = torch.from_numpy(sk.view(np.int64))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    want = (sum(2**63 + 12345 + r for r in range(world))) % 2**64
    ok = ok and int(sk[0]) == want
    # wide key sets: per-shard partial rows (oracle restatements standing in for the GPU) -> all-gather -> merge
    rows, status = po.decode_batch(buf, off, 1)
    sel = np.zeros(n, dtype=bool)
    for p in mine:
        sel[p::nparts] = True
    d = fa.dist
    app = d.merge_rows_app_host(d.allgather_struct(po.rollup_app(rows[sel], status[sel], 300), d.ROW_APP_DTYPE, device="cpu"))
    ok = ok and app.tobytes() == po.rollup_app(rows, status, 300).astype(d.ROW_APP_DTYPE).tobytes()
    # ... and the hash-partitioned exchange of the same rows (large sets): every rank ends up with the keys it owns, complete;
    # the union of the shares is the all-gathered result
    from types import SimpleNamespace
    kinds = SimpleNamespace(APP=1, R5M=0)
    my_app = po.rollup_app(rows[sel], status[sel], 300).astype(d.ROW_APP_DTYPE)
    share = d.merge_rows_app_host([d.alltoall_struct(my_app, d.partition_rows_host(my_app, kinds.APP, world), d.ROW_APP_DTYPE)])
    ok = ok and bool((d.partition_rows_host(share, kinds.APP, world) == rank).all()) and 0 < len(share) < len(app)
    ok = ok and d.merge_rows_app_host(d.allgather_struct(share, d.ROW_APP_DTYPE, device="cpu")).tobytes() == app.tobytes()
    my5 = shard.rows().astype(d.ROW5M_DTYPE)
    share5 = d.merge_rows_host([d.alltoall_struct(my5, d.partition_rows_host(my5, kinds.R5M, world), d.ROW5M_DTYPE)])
    ok = ok and d.merge_rows_host(d.allgather_struct(share5, d.ROW5M_DTYPE, device="cpu")).tobytes() == merged.tobytes()
    # a rank that fails before / behind a collective of the close: every rank raises, nobody waits (dist._all_ok)
    try:
        d._all_ok(rank != 1, "a test", own_error=ValueError("rank 1's own error") if rank == 1 else None)
        ok = False
    except ValueError:
        ok = ok and rank == 1
    except d.RankFailed:
        ok = ok and rank != 1
    for dst in (0, 1):
        ports = d.merge_ports_host(d.allgather_struct(po.top_ports(rows[sel], status[sel], dst), d.PORT_ROW_DTYPE, device="cpu"))
        ok = ok and ports.tobytes() == po.top_ports(rows, status, dst).tobytes()
    mins = d.merge_minutes_host(d.allgather_struct(po.minute_series(rows[sel], status[sel]), d.MINUTE_ROW_DTYPE, device="cpu"))
    ok = ok and mins.tobytes() == po.minute_series(rows, status).tobytes()
    # BASELINE config 4 shape on the CPU: per-shard Count-Min sketch -> all-reduce == the sketch of the whole stream bit
    # for bit; the ranks' candidate keys -> all-gather -> union == every distinct key (dist.topk_merged's exchange)
    good = status == 0
    with np.errstate(over="ignore"):
        w = rows["bytes"] * rows["sampling_rate"]
    mysk = po.cms_sketch_numpy(rows["src_addr"][sel & good], w[sel & good], 4, 12, 7)
    ts = torch.from_numpy(mysk.view(np.int64))
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    ok = ok and mysk.tobytes() == po.cms_sketch_numpy(rows["src_addr"][good], w[good], 4, 12, 7).tobytes()
    # dist.topk_merged's exchange: every rank ranks ITS distinct keys by the MERGED sketch and sends its first k rows;
    # the top k of the union == the top k over every distinct key of the whole stream (ties: key bytes ascending)
    def estimates(keys16):
        cols = [po.cms_sketch_numpy(keys16[i:i + 1], np.ones(1, dtype=np.uint64), 4, 12, 7).reshape(4, -1).argmax(axis=1) for i in range(len(keys16))]
        sk = mysk.reshape(4, -1)
        return np.array([min(int(sk[r, c[r]]) for r in range(4)) for c in cols], dtype=np.uint64)

    def topk(keys16, k):
        est = estimates(keys16)
        order = sorted(range(len(keys16)), key=lambda i: (-int(est[i]), bytes(keys16[i])))[:k]
        return [(bytes(keys16[i]), int(est[i])) for i in order]

    def uniq(a):
        return np.unique(np.ascontiguousarray(a).view([("k", "u1", 16)]).reshape(-1)).view(np.uint8).reshape(-1, 16)
    k = 25
    mykeys = uniq(rows["src_addr"][sel & good])
    mine_top = topk(mykeys, k)
    sent = np.array([list(x[0]) for x in mine_top], dtype=np.uint8).reshape(-1, 16)
    union = uniq(np.concatenate(d.allgather_bytes(sent, device="cpu")).reshape(-1, 16))
    allkeys = uniq(rows["src_addr"][good])
    ok = ok and topk(union, k) == topk(allkeys, k) and len(allkeys) > len(mykeys) > k
    q.put((rank, ok, len(merged), len(mine)))
    dist.destroy_process_group()


def test_two_rank_window_close_merge():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert sorted(r[3] for r in res) == [4, 4]


def test_merge_rows_host_sums_duplicates():
    sys.path.insert(0, ROOT)
    import _pkg
    fa = _pkg.load()
    a = np.zeros(2, dtype=fa.dist.ROW5M_DTYPE)
    a["timeslot"] = [300, 600]
    a["src_as"] = 1
    a["bytes"] = [2**64 - 1, 5]
    a["count"] = 1
    b = a.copy()
    b["bytes"] = [2, 7]
    m = fa.dist.merge_rows_host([a, b])
    assert len(m) == 2 and list(m["timeslot"]) == [300, 600]
    assert list(m["bytes"]) == [1, 12] and list(m["count"]) == [2, 2]  # wraps mod 2^64
    assert fa.dist.partitions_of(1, 4, 8) == [1, 5]


def test_host_merges_match_plain_dict_groupby():
    """merge_rows_host / merge_rows_app_host / merge_ports_host / merge_minutes_host against a row-at-a-time dict
    group-by on random partial row sets (duplicate keys inside and across parts, u64 wrap-around, empty parts)."""
    sys.path.insert(0, ROOT)
    import _pkg
    d = _pkg.load().dist
    rng = np.random.default_rng(5)
    M = 2**64

    def u64(n, big):
        return rng.integers(0, 2**64 if big else 1000, n, dtype=np.uint64)

    for trial in range(20):
        nparts = int(rng.integers(1, 5))
        big = trial % 2 == 0
        # flows_5m rows
        parts, ref = [], {}
        for _ in range(nparts):
            n = int(rng.integers(0, 60))
            r = np.zeros(n, dtype=d.ROW5M_DTYPE)
            r["timeslot"] = rng.integers(0, 3, n) * 300 + 86400 * rng.integers(0, 2, n)
            r["date"] = r["timeslot"] // 86400
            r["src_as"], r["dst_as"] = rng.integers(0, 3, n), rng.integers(0, 3, n)
            r["etype"] = rng.choice([0x800, 0x86dd], n)
            r["bytes"], r["packets"], r["count"] = u64(n, big), u64(n, big), u64(n, False)
            parts.append(r)
            for x in r:
                k = (int(x["date"]), int(x["timeslot"]), int(x["src_as"]), int(x["dst_as"]), int(x["etype"]))
                b, p, c = ref.get(k, (0, 0, 0))
                ref[k] = ((b + int(x["bytes"])) % M, (p + int(x["packets"])) % M, (c + int(x["count"])) % M)
        got = d.merge_rows_host(parts)
        assert [(int(x["date"]), int(x["timeslot"]), int(x["src_as"]), int(x["dst_as"]), int(x["etype"])) for x in got] == sorted(ref)
        assert [(int(x["bytes"]), int(x["packets"]), int(x["count"])) for x in got] == [ref[k] for k in sorted(ref)]
        # (SrcAddr, DstPort, Proto) rows: sorted by the address BYTES
        parts, ref = [], {}
        for _ in range(nparts):
            n = int(rng.integers(0, 60))
            r = np.zeros(n, dtype=d.ROW_APP_DTYPE)
            r["timeslot"] = rng.integers(0, 2, n) * 300
            r["src_addr"] = rng.integers(0, 2, (n, 16)) * rng.integers(1, 256, (n, 16))
            r["src_addr"][:, 2:15] = 0
            r["dst_port"], r["proto"] = rng.integers(0, 3, n), rng.integers(0, 2, n)
            r["bytes"], r["packets"], r["count"] = u64(n, big), u64(n, big), u64(n, False)
            parts.append(r)
            for x in r:
                k = (int(x["date"]), int(x["timeslot"]), bytes(x["src_addr"]), int(x["dst_port"]), int(x["proto"]))
                b, p, c = ref.get(k, (0, 0, 0))
                ref[k] = ((b + int(x["bytes"])) % M, (p + int(x["packets"])) % M, (c + int(x["count"])) % M)
        got = d.merge_rows_app_host(parts)
        assert [(int(x["date"]), int(x["timeslot"]), bytes(x["src_addr"]), int(x["dst_port"]), int(x["proto"])) for x in got] == sorted(ref)
        assert [(int(x["bytes"]), int(x["packets"]), int(x["count"])) for x in got] == [ref[k] for k in sorted(ref)]
        # port group-by: ORDER BY weight DESC, port; minute series: ORDER BY minute
        for dtype, key, merge in ((d.PORT_ROW_DTYPE, "port", d.merge_ports_host), (d.MINUTE_ROW_DTYPE, "minute", d.merge_minutes_host)):
            parts, ref = [], {}
            for _ in range(nparts):
                n = int(rng.integers(0, 40))
                r = np.zeros(n, dtype=dtype)
                r[key] = rng.integers(0, 12, n)
                r["weight"], r["count"] = u64(n, big), u64(n, False)
                parts.append(r)
                for x in r:
                    w, c = ref.get(int(x[key]), (0, 0))
                    ref[int(x[key])] = ((w + int(x["weight"])) % M, (c + int(x["count"])) % M)
            got = merge(parts)
            order = sorted(ref, key=(lambda k: (-ref[k][0], k)) if key == "port" else (lambda k: k))
            assert [int(x[key]) for x in got] == order
            assert [(int(x["weight"]), int(x["count"])) for x in got] == [ref[k] for k in order]
