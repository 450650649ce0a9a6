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
* sparse flows_5m rows -> ``all_gather`` of each rank's compacted rows, then a
  local re-aggregation (sum is a commutative monoid, so the merged table equals
  the single-shard table exactly).  Row sets are tens of MB at most (393 k rows x
  48 B for BASELINE config 2), far below the point where a hash-partitioned
  all-to-all would pay off on 153 GB/s xGMI links.

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


def close_window_app_merged(agg, timeslot, group=None, device=None) -> np.ndarray:
    """Window close of the (SrcAddr,DstPort,Proto) key set across ranks (identical result on every rank)."""
    local = agg.close_window_app(timeslot)
    return merge_rows_app_host(allgather_struct(local, ROW_APP_DTYPE, group=group, device=device))


def top_ports_merged(agg, dst, k=None, group=None, device=None) -> np.ndarray:
    rows = merge_ports_host(allgather_struct(agg.top_ports(dst), PORT_ROW_DTYPE, group=group, device=device))
    return rows if k is None else rows[:k]


def minute_series_merged(agg, group=None, device=None) -> np.ndarray:
    return merge_minutes_host(allgather_struct(agg.minute_series(), MINUTE_ROW_DTYPE, group=group, device=device))


def allgather_rows(rows: np.ndarray, group=None, device=None):
    """All ranks receive every rank's rows (list indexed by rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    rows = np.ascontiguousarray(rows, dtype=ROW5M_DTYPE)
    n = torch.tensor([len(rows)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    mine = torch.zeros(cap * ROW5M_DTYPE.itemsize, dtype=torch.uint8)
    if len(rows):
        mine[:rows.nbytes] = torch.from_numpy(rows.view(np.uint8).reshape(-1))
    mine = mine.to(dev)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    out = []
    for c, b in zip(counts, bufs):
        a = b.cpu().numpy()[:c * ROW5M_DTYPE.itemsize]
        out.append(a.view(ROW5M_DTYPE).copy())
    return out


def close_window_merged(agg, timeslot, group=None, device=None) -> np.ndarray:
    """Window close across ranks; every rank returns the same merged flows_5m rows.

    `nccl`: device side - every rank's rows of the window are compacted and sorted in HBM
    (fa_window_rows_device), all-gathered over RCCL straight out of / into device memory, the other ranks' rows
    are folded into the rank's own table by a kernel (fa_merge_rows_device) and the merged window leaves through
    the ordinary device-sorted close.  No host sort, no Python re-aggregation.
    Other backends (gloo on CPU tensors: the harness tests): rows travel through host memory, merged with numpy."""
    import torch
    import torch.distributed as dist
    # (sliding windows keep the newer sub-buckets in the table after a close: folding other ranks' rows into it would
    # count them again at the next close - those closes take the host path as well)
    sliding = agg.cfg.subwindow_secs not in (0, agg.cfg.window_secs) and timeslot != 0xFFFFFFFF
    if dist.get_backend(group) != "nccl" or sliding:
        local = agg.close_window(timeslot)
        return merge_rows_host(allgather_rows(local, group=group, device=device))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ptr, n = agg.window_rows_device(timeslot)
    dev = torch.device("cuda", torch.cuda.current_device())
    cnt = torch.tensor([n], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    words = max(max(counts), 1) * (ROW5M_DTYPE.itemsize // 8)
    mine = torch.zeros(words, dtype=torch.int64, device=dev)
    if n:
        mine[:n * (ROW5M_DTYPE.itemsize // 8)].copy_(torch.as_tensor(_DevArray(ptr, n * (ROW5M_DTYPE.itemsize // 8)), device="cuda"))
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    torch.cuda.synchronize()
    for r in range(world):
        if r != rank and counts[r]:
            agg.merge_rows_device(bufs[r].data_ptr(), counts[r])
    return agg.close_window(timeslot)


class _DevArray:
    """Minimal __cuda_array_interface__ view of library-owned HBM (int64 words)."""

    def __init__(self, ptr: int, words: int):
        self.__cuda_array_interface__ = {
            "shape": (words,), "typestr": "<i8", "data": (ptr, False), "version": 2, "strides": None}


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


def merge_topk_candidates(parts, k_keep=None) -> np.ndarray:
    """Union of per-rank candidate keys (uint8[n,16] each), duplicates removed, sorted."""
    keys = [np.ascontiguousarray(p, dtype=np.uint8).reshape(-1, 16) for p in parts if len(p)]
    if not keys:
        return np.zeros((0, 16), dtype=np.uint8)
    allk = np.unique(np.concatenate(keys).view([("k", "u1", 16)]).reshape(-1))
    return allk.view(np.uint8).reshape(-1, 16)


def topk_merged(agg, key_set, k, candidates_per_rank=None, group=None, device=None):
    """Heavy hitters across ranks at window close: all-reduce the sketches (dense, exact), exchange
    every rank's local candidates (its top `candidates_per_rank` keys; all distinct keys when None),
    add the union to the local set and rank by the merged estimate.  With candidates_per_rank=None
    the result equals the single-GPU result bit for bit."""
    ncand = candidates_per_rank if candidates_per_rank is not None else (1 << 30)
    local = agg.topk(key_set, ncand)
    allreduce_sketches(agg, group=group)
    parts = allgather_bytes(local["key"], group=group, device=device)
    agg.topk_merge_keys(key_set, merge_topk_candidates(parts))
    return agg.topk(key_set, k)
