"""Multi-GPU layer: Kafka partitions sharded one process per GPU, merged at window close.

The reference's only parallelism is Kafka-partition data parallelism (topic
created with ``--partitions 2``, ``compose/docker-compose-clickhouse-mock.yml:18``;
sarama runs one ``ConsumeClaim`` per claimed partition, ``inserter/inserter.go:176``).
Here partition p is owned by rank ``p % world``; each rank keeps private group-by
state in HBM for the whole window (no data-path collective), and the only
exchange is at window close:

* dense Count-Min sketches -> ``all_reduce(SUM)`` of the library's device buffers into
  the ctx's merged view (out of place, idempotent; RCCL over xGMI when the backend is
  ``nccl``; u64 wrap-around sum == int64 sum bit for bit);
* sparse rows, small sets (flows_5m: 393 k rows x 48 B for BASELINE config 2; ports, minutes, top-k candidates) ->
  ``all_gather`` of each rank's compacted rows, then a local re-aggregation (sum is a commutative monoid, so the
  merged table equals the single-shard table exactly) - every rank ends up with the whole result;
* sparse rows, LARGE sets ((SrcAddr,DstPort,Proto): a window of BASELINE config 5 is 16.6 M rows x 56 B = 930 MB per
  rank) -> hash-partitioned exchange (``rows_merged_partitioned``): every rank cuts its rows by ``hash(key) * world >> 64``
  on the device, ONE ``all_to_all_single`` moves each group to its owner (xGMI is point to point: every link carries
  1 / world of a rank's rows instead of every rank receiving everything), and each rank sorts and sums 1 / world of the
  keys - and emits that share (SURVEY.md 8(e) option (ii)).

Works with ``gloo`` on CPU tensors (host logic tests) and ``nccl`` on the GPUs.
"""
from __future__ import annotations

import numpy as np

ROW5M_DTYPE = np.dtype([
    ("date", "<u4"), ("timeslot", "<u4"), ("src_as", "<u4"), ("dst_as", "<u4"),
    ("etype", "<u4"), ("_pad", "<u4"), ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8"),
])


ROW_APP_DTYPE = np.dtype([
    ("date", "<u4"), ("timeslot", "<u4"), ("src_addr", "u1", 16), ("dst_port", "<u4"), ("proto", "<u4"),
    ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8"),
])
PORT_ROW_DTYPE = np.dtype([("port", "<u4"), ("_pad", "<u4"), ("weight", "<u8"), ("count", "<u8")])
MINUTE_ROW_DTYPE = np.dtype([("minute", "<u4"), ("_pad", "<u4"), ("weight", "<u8"), ("count", "<u8")])


def partitions_of(rank: int, world: int, n_partitions: int):
    """Kafka partitions owned by `rank` (round-robin, like a balanced consumer group)."""
    return [p for p in range(n_partitions) if p % world == rank]


def merge_rows_host(parts) -> np.ndarray:
    """SummingMergeTree collapse (create.sh:70-90) of partial row sets: rows with equal
    (date,timeslot,src_as,dst_as,etype) are summed (mod 2^64); output sorted by key."""
    parts = [np.ascontiguousarray(p, dtype=ROW5M_DTYPE) for p in parts if len(p)]
    if not parts:
        return np.zeros(0, dtype=ROW5M_DTYPE)
    rows = np.concatenate(parts)
    order = np.lexsort((rows["etype"], rows["dst_as"], rows["src_as"], rows["timeslot"], rows["date"]))
    rows = rows[order]
    key = np.stack([rows[f] for f in ("date", "timeslot", "src_as", "dst_as", "etype")], axis=1)
    first = np.ones(len(rows), dtype=bool)
    first[1:] = (key[1:] != key[:-1]).any(axis=1)
    starts = np.nonzero(first)[0]
    out = rows[starts].copy()
    with np.errstate(over="ignore"):
        for f in ("bytes", "packets", "count"):
            out[f] = np.add.reduceat(rows[f], starts)
    return out


def _merge_sorted_groups(rows, key_cols, sum_cols):
    """rows sorted so that equal keys are adjacent -> one row per key, sum_cols added mod 2^64."""
    if len(rows) == 0:
        return rows
    first = np.ones(len(rows), dtype=bool)
    diff = np.zeros(len(rows) - 1, dtype=bool)
    for c in key_cols:
        a = rows[c]
        d = a[1:] != a[:-1]
        diff |= d.reshape(len(d), -1).any(axis=1)
    first[1:] = diff
    starts = np.nonzero(first)[0]
    out = rows[starts].copy()
    with np.errstate(over="ignore"):
        for f in sum_cols:
            out[f] = np.add.reduceat(rows[f], starts)
    return out


def merge_rows_app_host(parts) -> np.ndarray:
    """Partial (SrcAddr,DstPort,Proto) row sets -> one row per key, sorted like fa_read_window_app."""
    parts = [np.ascontiguousarray(p, dtype=ROW_APP_DTYPE) for p in parts if len(p)]
    if not parts:
        return np.zeros(0, dtype=ROW_APP_DTYPE)
    rows = np.concatenate(parts)
    addr = np.ascontiguousarray(rows["src_addr"])
    hi = addr[:, :8].copy().view(">u8").reshape(-1)
    lo = addr[:, 8:].copy().view(">u8").reshape(-1)
    order = np.lexsort((rows["proto"], rows["dst_port"], lo, hi, rows["timeslot"], rows["date"]))
    return _merge_sorted_groups(rows[order], ("date", "timeslot", "src_addr", "dst_port", "proto"), ("bytes", "packets", "count"))


def merge_ports_host(parts) -> np.ndarray:
    """Partial GROUP BY port row sets -> merged, ORDER BY weight DESC, port (viz-ch.json:358,604)."""
    parts = [np.ascontiguousarray(p, dtype=PORT_ROW_DTYPE) for p in parts if len(p)]
    if not parts:
        return np.zeros(0, dtype=PORT_ROW_DTYPE)
    rows = np.concatenate(parts)
    rows = _merge_sorted_groups(rows[np.argsort(rows["port"], kind="stable")], ("port",), ("weight", "count"))
    return rows[np.lexsort((rows["port"], np.uint64(0xFFFFFFFFFFFFFFFF) - rows["weight"]))]


def merge_minutes_host(parts) -> np.ndarray:
    """Partial per-minute series -> merged, ORDER BY minute (viz-ch.json:74)."""
    parts = [np.ascontiguousarray(p, dtype=MINUTE_ROW_DTYPE) for p in parts if len(p)]
    if not parts:
        return np.zeros(0, dtype=MINUTE_ROW_DTYPE)
    rows = np.concatenate(parts)
    return _merge_sorted_groups(rows[np.argsort(rows["minute"], kind="stable")], ("minute",), ("weight", "count"))


def allgather_struct(rows: np.ndarray, dtype, group=None, device=None):
    """All ranks receive every rank's rows of `dtype` (list indexed by rank)."""
    rows = np.ascontiguousarray(rows, dtype=dtype)
    return [b.view(dtype).copy() for b in allgather_bytes(rows.view(np.uint8).reshape(-1), group=group, device=device)]


def alltoall_struct(rows: np.ndarray, dest: np.ndarray, dtype, group=None):
    """Host rows to their owners: rank r receives every rank's rows with dest == r (one array, senders in rank order).
    The host twin of fa_rows_partition_device + alltoall_device_rows (CPU tests; oracle rows in the tools)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rows = np.ascontiguousarray(rows, dtype=dtype)
    order = np.argsort(dest, kind="stable")
    counts = np.bincount(dest, minlength=world).astype(np.int64)
    sc = torch.from_numpy(counts.copy())
    rc = torch.zeros(world, dtype=torch.int64)
    dist.all_to_all_single(rc, sc, group=group)
    payload = torch.from_numpy(np.ascontiguousarray(rows[order]).view(np.uint8).reshape(-1).copy())
    out = torch.empty(int(rc.sum()) * dtype.itemsize, dtype=torch.uint8)
    dist.all_to_all_single(out, payload, output_split_sizes=[int(v) * dtype.itemsize for v in rc.tolist()],
                           input_split_sizes=[int(v) * dtype.itemsize for v in counts.tolist()], group=group)
    return out.numpy().view(dtype).copy()


class _DevArray:
    """Minimal __cuda_array_interface__ view of library-owned HBM."""

    def __init__(self, ptr: int, n: int, typestr: str = "<i8"):
        self.__cuda_array_interface__ = {
            "shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def _key_words(rows: np.ndarray, kind: int):
    """The merge-order key words of a row kind, least significant first (merge.cuh, RowOps<KIND>::key)."""
    from . import ROWS_5M, ROWS_APP, ROWS_MINUTE, ROWS_PORT_DST, ROWS_PORT_SRC
    u64 = np.uint64
    if kind == ROWS_5M:
        return [(rows["dst_as"].astype(u64) << u64(32)) | rows["etype"].astype(u64),
                (rows["timeslot"].astype(u64) << u64(32)) | rows["src_as"].astype(u64)]
    if kind == ROWS_APP:
        addr = np.ascontiguousarray(rows["src_addr"])
        return [(rows["dst_port"].astype(u64) << u64(32)) | rows["proto"].astype(u64),
                addr[:, 8:].copy().view(">u8").reshape(-1).astype(u64), addr[:, :8].copy().view(">u8").reshape(-1).astype(u64),
                rows["timeslot"].astype(u64)]
    if kind in (ROWS_PORT_SRC, ROWS_PORT_DST):
        return [rows["port"].astype(u64)]
    if kind == ROWS_MINUTE:
        return [rows["minute"].astype(u64)]
    key = np.ascontiguousarray(rows["key"])  # top-k rows
    return [key[:, 8:].copy().view(">u8").reshape(-1).astype(u64), key[:, :8].copy().view(">u8").reshape(-1).astype(u64)]


def partition_rows_host(rows: np.ndarray, kind: int, world: int) -> np.ndarray:
    """Owner rank of every row: numpy restatement of merge.cuh row_dest (fa_rows_partition_device)."""
    h = np.full(len(rows), 0x9E3779B97F4A7C15, dtype=np.uint64)
    for w in _key_words(rows, kind):
        h = _mix64(h ^ w)
    return (((h >> np.uint64(32)) * np.uint64(world)) >> np.uint64(32)).astype(np.int64)


class RankFailed(RuntimeError):
    """Another rank could not produce its part of a window close: every rank raises instead of waiting in a collective."""


def _all_ok(ok: bool, what: str, group=None, own_error=None):
    """One small all-reduce: either every rank goes on or every rank raises (a rank that failed BEFORE a collective would
    leave the others waiting in it forever; one that failed behind it would let them drop a window it still holds)."""
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    if own_error is not None:
        raise own_error
    if int(t.item()) == 0:
        raise RankFailed("window close: a rank failed in %s" % what)


def allgather_device_rows(ptr: int, n: int, row_bytes: int, group=None):
    """Every rank's n rows (row_bytes each, sitting in HBM at ptr) back to back in ONE device buffer on every rank.
    -> (uint8 cuda tensor, total rows).  Counts travel first; the payload is gathered into slices of the destination
    buffer - no padding to the largest rank, no host copy of the rows.  `nccl`: RCCL moves HBM to HBM (uneven
    all_gather).  Any other backend (gloo: two ranks sharing the one GPU of a test box) only replaces the transport:
    rows are staged through host memory for the collective and land in the same device buffer."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    on_device = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    xdev = dev if on_device else torch.device("cpu")
    cnt = torch.tensor([n], dtype=torch.int64, device=xdev)
    counts = [torch.zeros(1, dtype=torch.int64, device=xdev) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    if min(counts) < 0:  # (n = -1: that rank's fa_rows_device failed - rows_merged)
        raise RankFailed("window close: rank(s) %s could not produce their rows" % [r for r, v in enumerate(counts) if v < 0])
    total = sum(counts)
    out = torch.empty(max(total, 1) * row_bytes, dtype=torch.uint8, device=dev)
    mine = (torch.as_tensor(_DevArray(ptr, n * row_bytes, "|u1"), device=dev) if n
            else torch.empty(0, dtype=torch.uint8, device=dev))
    if total == 0:
        return out[:0], 0
    starts = np.concatenate([[0], np.cumsum(counts)]) * row_bytes
    if on_device:
        # all-gather with uneven counts = one broadcast per rank into that rank's slice of the destination (queued
        # together on the RCCL stream; what ProcessGroupNCCL itself does for uneven all_gather outputs)
        rank = dist.get_rank(group)
        work = []
        for r in range(world):
            if not counts[r]:
                continue
            sl = out[int(starts[r]):int(starts[r + 1])]
            if r == rank:
                sl.copy_(mine)
            src = r if group is None else dist.get_global_rank(group, r)
            work.append(dist.broadcast(sl, src=src, group=group, async_op=True))
        for w in work:
            w.wait()
    else:
        h = mine.cpu()
        cap = max(counts) * row_bytes
        pad = torch.zeros(cap, dtype=torch.uint8)
        pad[:h.numel()] = h
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        for r in range(world):
            if counts[r]:
                out[int(starts[r]):int(starts[r + 1])].copy_(bufs[r][:counts[r] * row_bytes])
    torch.cuda.synchronize()
    return out[:total * row_bytes], total


def rows_merged(agg, kind, timeslot=0xFFFFFFFF, k_local=0, k=0, group=None) -> np.ndarray:
    """One kind of rows merged across ranks at window close, identical on every rank: fa_rows_device (this rank's
    result in HBM: extracted, sub-buckets folded, sorted) -> all-gather of the device buffers -> fa_rows_merge_device
    (radix sort + segmented sums in HBM, emit order) -> one copy of the result to the host."""
    from . import ROW_DTYPES, FlowAggError
    err = None
    try:
        ptr, n = agg.rows_device(kind, timeslot, k_local)
    except FlowAggError as e:  # (the others must not wait for this rank in the collective: it takes part with count -1)
        err, ptr, n = e, 0, -1
    try:
        buf, total = allgather_device_rows(ptr, n, ROW_DTYPES[kind].itemsize, group=group)
    except RankFailed:
        if err is not None:
            raise err
        raise
    rows = None
    try:
        mptr, m = agg.rows_merge_device(kind, buf.data_ptr(), total, k)
        rows = agg.rows_fetch(kind, mptr, m)
    except FlowAggError as e:
        err = e
    _all_ok(err is None, "the merge of the gathered rows", group=group, own_error=err)
    return rows


def alltoall_device_rows(ptr: int, counts, row_bytes: int, group=None):
    """The partitioned rows of this rank (HBM at ptr, group r = counts[r] rows, back to back) -> the groups every rank
    holds for THIS rank, back to back in one device buffer.  -> (uint8 cuda tensor, rows received).  `nccl`: one
    all_to_all_single over RCCL, HBM to HBM; other backends (gloo on a shared test GPU) stage through host memory."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    on_device = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device())
    xdev = dev if on_device else torch.device("cpu")
    sc = torch.tensor([int(v) for v in counts], dtype=torch.int64, device=xdev)
    rc = torch.zeros(world, dtype=torch.int64, device=xdev)
    dist.all_to_all_single(rc, sc, group=group)
    rcounts = [int(v) for v in rc.tolist()]
    if min(rcounts) < 0 or min(int(v) for v in counts) < 0:  # (counts of -1: this rank's own failure - it must not go on either)
        raise RankFailed("window close: rank(s) %s could not produce their rows" % ([r for r, v in enumerate(rcounts) if v < 0] or "this one"))
    n = sum(int(v) for v in counts if v > 0)
    total = sum(rcounts)
    mine = (torch.as_tensor(_DevArray(ptr, n * row_bytes, "|u1"), device=dev) if n else torch.empty(0, dtype=torch.uint8, device=dev))
    insp = [max(int(v), 0) * row_bytes for v in counts]
    outsp = [v * row_bytes for v in rcounts]
    if on_device:
        out = torch.empty(max(total, 1) * row_bytes, dtype=torch.uint8, device=dev)
        dist.all_to_all_single(out[:total * row_bytes], mine, output_split_sizes=outsp, input_split_sizes=insp, group=group)
    else:
        h_out = torch.empty(total * row_bytes, dtype=torch.uint8)
        dist.all_to_all_single(h_out, mine.cpu(), output_split_sizes=outsp, input_split_sizes=insp, group=group)
        out = torch.empty(max(total, 1) * row_bytes, dtype=torch.uint8, device=dev)
        out[:total * row_bytes].copy_(h_out)
    torch.cuda.synchronize()
    return out[:total * row_bytes], total


def rows_merged_partitioned(agg, kind, timeslot=0xFFFFFFFF, group=None) -> np.ndarray:
    """THIS rank's share of one kind of rows merged across ranks: the keys with partition_rows_host(...) == rank, complete
    (every rank's rows of those keys, summed) and in the kind's merge order.  fa_rows_device -> fa_rows_partition_device
    (groups by owner, in HBM) -> one all-to-all -> fa_rows_merge_device on 1 / world of the keys -> copy out.  The union
    of the ranks' shares is what rows_merged returns on every rank - without any rank receiving or sorting all of it."""
    import torch.distributed as dist
    from . import ROW_DTYPES, FlowAggError
    world = dist.get_world_size(group)
    err = None
    try:
        ptr, n = agg.rows_device(kind, timeslot)
        pptr, counts = agg.rows_partition_device(kind, ptr, n, world)
    except FlowAggError as e:
        err, pptr, counts = e, 0, [-1] * world
    try:
        buf, total = alltoall_device_rows(pptr, counts, ROW_DTYPES[kind].itemsize, group=group)
    except RankFailed:
        if err is not None:
            raise err
        raise
    rows = None
    try:
        mptr, m = agg.rows_merge_device(kind, buf.data_ptr(), total, 0)
        rows = agg.rows_fetch(kind, mptr, m)
    except FlowAggError as e:
        err = e
    _all_ok(err is None, "the merge of its share", group=group, own_error=err)
    return rows


def close_window_app_partitioned(agg, timeslot, group=None) -> np.ndarray:
    """Window close of the (SrcAddr,DstPort,Proto) key set across ranks, hash-partitioned: every rank returns ITS share of
    the merged rows (sorted; a key appears on exactly one rank) and drops the window.  What a sharded sink wants - each
    rank inserts its rows - and what keeps a 16.6 M-row window from being received and sorted `world` times."""
    from . import ROWS_APP
    rows = rows_merged_partitioned(agg, ROWS_APP, timeslot, group=group)
    agg.drop_window(ROWS_APP, timeslot)
    return rows


def close_window_merged(agg, timeslot, group=None, device=None) -> np.ndarray:
    """Window close across ranks; every rank returns the same merged flows_5m rows and drops the window from its
    table (sliding windows: the oldest sub-bucket, like fa_close_window).  The ranks' own tables are never mixed, so
    sliding windows take the same path as tumbling ones."""
    from . import ROWS_5M
    rows = rows_merged(agg, ROWS_5M, timeslot, group=group)
    agg.drop_window(ROWS_5M, timeslot)
    return rows


def close_window_app_merged(agg, timeslot, group=None, device=None) -> np.ndarray:
    """Window close of the (SrcAddr,DstPort,Proto) key set across ranks (identical result on every rank)."""
    from . import ROWS_APP
    rows = rows_merged(agg, ROWS_APP, timeslot, group=group)
    agg.drop_window(ROWS_APP, timeslot)
    return rows


def top_ports_merged(agg, dst, k=None, group=None, device=None) -> np.ndarray:
    """GROUP BY port across ranks (every rank's port rows travel: a port just below the cut everywhere can lead overall)."""
    from . import ROWS_PORT_DST, ROWS_PORT_SRC
    return rows_merged(agg, ROWS_PORT_DST if dst else ROWS_PORT_SRC, k_local=0, k=0 if k is None else max(int(k), 1), group=group)[:k]


def minute_series_merged(agg, group=None, device=None) -> np.ndarray:
    from . import ROWS_MINUTE
    return rows_merged(agg, ROWS_MINUTE, group=group)


def allgather_rows(rows: np.ndarray, group=None, device=None):
    """All ranks receive every rank's rows (list indexed by rank) - host rows (tests, oracle rows in bench.py)."""
    return allgather_struct(rows, ROW5M_DTYPE, group=group, device=device)


def allreduce_sketches(agg, group=None):
    """Window-close merge of the dense state across ranks: RCCL all-reduce (sum, u64 == i64 bit for bit) of every
    rank's Count-Min sketches INTO the ctx's merged view - out of place: the rank's own sketches stay as they are,
    so the call is idempotent (calling it twice, or again after more ingest, recomputes the view; nothing is ever
    counted twice) and fa_topk / fa_cms_read answer from the merged view until the ctx ingests again.  The dense
    port histograms are NOT reduced here: ports travel as rows (top_ports_merged), like every other sparse result."""
    import torch
    import torch.distributed as dist
    st = agg.device_state()  # (settles the stream: the sketch copies are folded)
    on_device = dist.get_backend(group) == "nccl"  # gloo (harness tests on a 1-GPU box): staged through host memory
    for own, merged in ((st.cms_src, st.cms_src_merged), (st.cms_dst, st.cms_dst_merged)):
        if own:
            t = torch.as_tensor(_DevArray(merged, st.cms_words), device="cuda")
            mine = torch.as_tensor(_DevArray(own, st.cms_words), device="cuda")
            if on_device:
                t.copy_(mine)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            else:
                h = mine.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                t.copy_(h)
    torch.cuda.synchronize()
    agg.merged_view_set(True)


def allgather_bytes(arr: np.ndarray, group=None, device=None):
    """All ranks receive every rank's uint8 payload (list indexed by rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    flat = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
    n = torch.tensor([flat.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    mine = torch.zeros(max(max(counts), 1), dtype=torch.uint8)
    mine[:flat.size] = torch.from_numpy(flat.copy())
    mine = mine.to(dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    return [b.cpu().numpy()[:c].copy() for c, b in zip(counts, bufs)]


def topk_merged(agg, key_set, k, candidates_per_rank=None, group=None, device=None):
    """Heavy hitters across ranks at window close, exact with respect to the merged sketch: all-reduce the sketches
    (dense, RCCL) into the merged view, then every rank ranks ITS distinct addresses by the merged estimate and sends
    its first k rows; the merged top k is the top k of their union.  (A key of the global top k ranks at least as high
    among the keys of any rank that saw it - and the merged estimate of a key is the same on every rank - so it is in
    that rank's k rows.  Round 2 exchanged every distinct key: 16 M per rank, 95 s over host memory.)
    candidates_per_rank is kept for callers of the old interface; values below k would make the result approximate
    and are raised to k."""
    from . import FA_KEYS_SRCADDR_CMS, ROWS_TOPK_DST, ROWS_TOPK_SRC
    allreduce_sketches(agg, group=group)
    kind = ROWS_TOPK_SRC if key_set == FA_KEYS_SRCADDR_CMS else ROWS_TOPK_DST
    return rows_merged(agg, kind, k_local=max(int(k), 1), k=max(int(k), 1), group=group)[:k]
