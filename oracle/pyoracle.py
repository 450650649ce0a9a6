"""ctypes binding for the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  See oracle/flow_oracle.h for scope and parity status.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class FoRow(C.Structure):
    _fields_ = [
        ("time_received", C.c_uint64), ("time_flow_start", C.c_uint64),
        ("sampling_rate", C.c_uint64), ("bytes", C.c_uint64), ("packets", C.c_uint64),
        ("sequence_num", C.c_uint32), ("src_as", C.c_uint32), ("dst_as", C.c_uint32),
        ("etype", C.c_uint32), ("proto", C.c_uint32), ("src_port", C.c_uint32),
        ("dst_port", C.c_uint32), ("_pad", C.c_uint32),
        ("sampler_address", C.c_uint8 * 16), ("src_addr", C.c_uint8 * 16),
        ("dst_addr", C.c_uint8 * 16),
    ]

    def as_dict(self):
        return {
            "TimeReceived": self.time_received, "TimeFlowStart": self.time_flow_start,
            "SequenceNum": self.sequence_num, "SamplingRate": self.sampling_rate,
            "SamplerAddress": bytes(self.sampler_address), "SrcAddr": bytes(self.src_addr),
            "DstAddr": bytes(self.dst_addr), "SrcAS": self.src_as, "DstAS": self.dst_as,
            "EType": self.etype, "Proto": self.proto, "SrcPort": self.src_port,
            "DstPort": self.dst_port, "Bytes": self.bytes, "Packets": self.packets,
        }


ROW5M_DTYPE = np.dtype([
    ("date", "<u4"), ("timeslot", "<u4"), ("src_as", "<u4"), ("dst_as", "<u4"),
    ("etype", "<u4"), ("_pad", "<u4"), ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8"),
])
assert ROW5M_DTYPE.itemsize == 48

ROW_DTYPE = np.dtype([
    ("time_received", "<u8"), ("time_flow_start", "<u8"), ("sampling_rate", "<u8"),
    ("bytes", "<u8"), ("packets", "<u8"), ("sequence_num", "<u4"), ("src_as", "<u4"),
    ("dst_as", "<u4"), ("etype", "<u4"), ("proto", "<u4"), ("src_port", "<u4"),
    ("dst_port", "<u4"), ("_pad", "<u4"), ("sampler_address", "u1", 16),
    ("src_addr", "u1", 16), ("dst_addr", "u1", 16),
])
assert ROW_DTYPE.itemsize == C.sizeof(FoRow) == 120


class GenParams(C.Structure):
    _fields_ = [
        ("mode", C.c_uint32), ("framed", C.c_uint32), ("seed", C.c_uint64),
        ("n_total", C.c_uint64), ("t0", C.c_uint64), ("span_secs", C.c_uint32),
        ("per_sec", C.c_uint32), ("zipf_log2_universe", C.c_uint32),
        ("zipf_s_x100", C.c_uint32),
    ]


GEN_MOCKER, GEN_ASPAIRS, GEN_ZIPF, GEN_GOFLOW, GEN_DISTINCT, GEN_REVERSED = 0, 1, 2, 3, 4, 5
GEN_MAX_RECORD = 256
T0 = 1_600_000_200  # multiple of 300


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("flow_oracle.c", "flow_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    L.fo_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FoRow)]
    L.fo_decode.restype = C.c_int
    L.fo_decode_framed.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(FoRow)]
    L.fo_decode_framed.restype = C.c_int
    L.fo_decode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.fo_decode_batch.restype = C.c_uint64
    L.fo_frame_split.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.fo_frame_split.restype = C.c_size_t
    L.fo_rollup_new.argtypes = [C.c_uint32]
    L.fo_rollup_new.restype = C.c_void_p
    L.fo_rollup_free.argtypes = [C.c_void_p]
    L.fo_rollup_add.argtypes = [C.c_void_p, C.POINTER(FoRow)]
    L.fo_rollup_ingest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.fo_rollup_ingest.restype = C.c_uint64
    L.fo_rollup_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.fo_rollup_size.argtypes = [C.c_void_p]
    L.fo_rollup_size.restype = C.c_size_t
    L.fo_rollup_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]
    L.fo_rollup_rows.restype = C.c_size_t
    L.fo_cms_column.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32]
    L.fo_cms_column.restype = C.c_uint32
    L.fo_cms_update.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_char_p, C.c_uint64]
    L.fo_cms_query.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_char_p]
    L.fo_cms_query.restype = C.c_uint64
    L.fo_gen_record_len.argtypes = [C.POINTER(GenParams), C.c_uint64]
    L.fo_gen_record_len.restype = C.c_uint32
    L.fo_gen_records.argtypes = [C.POINTER(GenParams), C.c_uint64, C.c_uint64, C.c_void_p,
                                 C.c_size_t, C.c_void_p]
    L.fo_gen_records.restype = C.c_size_t
    L.fo_gen_row.argtypes = [C.POINTER(GenParams), C.c_uint64, C.POINTER(FoRow)]
    L.fo_bench_rollup.argtypes = [C.POINTER(GenParams), C.c_uint64, C.c_uint64, C.c_int, u64p,
                                  u64p, u64p, u64p]
    L.fo_bench_rollup.restype = C.c_double
    _LIB = L
    return L


def decode(payload: bytes, framed=False):
    """-> dict of the 15 projected columns, or None for a bad record."""
    row = FoRow()
    f = lib().fo_decode_framed if framed else lib().fo_decode
    rc = f(payload, len(payload), C.byref(row))
    return row.as_dict() if rc == 0 else None


def decode_batch(buf: np.ndarray, off: np.ndarray, framed=0):
    """-> (rows ROW_DTYPE[n], status uint32[n]); bad records are zeroed with status 1."""
    buf = np.ascontiguousarray(buf, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    n = len(off) - 1
    rows = np.zeros(n, dtype=ROW_DTYPE)
    status = np.zeros(n, dtype=np.uint32)
    lib().fo_decode_batch(buf.ctypes.data, off.ctypes.data, n, framed, rows.ctypes.data, status.ctypes.data)
    return rows, status


def gen_params(mode=GEN_MOCKER, framed=1, seed=1, n_total=0, t0=T0, span_secs=900, per_sec=4,
               zipf_log2_universe=24, zipf_s_x100=110):
    return GenParams(mode, framed, seed, n_total, t0, span_secs, per_sec, zipf_log2_universe,
                     zipf_s_x100)


def gen_records(gp: GenParams, i0: int, n: int):
    """-> (buf uint8[nbytes], offsets uint64[n+1])"""
    buf = np.empty(n * (200 if gp.mode == GEN_GOFLOW else 96) + 256, dtype=np.uint8)
    off = np.empty(n + 1, dtype=np.uint64)
    w = lib().fo_gen_records(C.byref(gp), i0, n, buf.ctypes.data, buf.size, off.ctypes.data)
    assert w != 2**64 - 1
    return buf[:w].copy(), off


def gen_rows(gp: GenParams, i0: int, n: int):
    out = np.zeros(n, dtype=ROW_DTYPE)
    row = FoRow()
    for k in range(n):
        lib().fo_gen_row(C.byref(gp), i0 + k, C.byref(row))
        C.memmove(out.ctypes.data + k * ROW_DTYPE.itemsize, C.byref(row), ROW_DTYPE.itemsize)
    return out


class Rollup:
    def __init__(self, granule=300):
        self._h = lib().fo_rollup_new(granule)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().fo_rollup_free(self._h)
            self._h = None

    def ingest(self, buf: np.ndarray, off: np.ndarray, framed=1) -> int:
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        return lib().fo_rollup_ingest(self._h, buf.ctypes.data, off.ctypes.data, len(off) - 1, framed)

    def merge(self, other: "Rollup"):
        lib().fo_rollup_merge(self._h, other._h)

    def rows(self, timeslot=0xFFFFFFFF) -> np.ndarray:
        n = lib().fo_rollup_size(self._h)
        out = np.zeros(max(n, 1), dtype=ROW5M_DTYPE)
        k = lib().fo_rollup_rows(self._h, timeslot, out.ctypes.data, n)
        return out[:k]


def cms_update(cms: np.ndarray, depth, width_log2, seed, key: bytes, weight: int):
    lib().fo_cms_update(cms.ctypes.data, depth, width_log2, seed, key, weight)


def cms_query(cms: np.ndarray, depth, width_log2, seed, key: bytes) -> int:
    return lib().fo_cms_query(cms.ctypes.data, depth, width_log2, seed, key)


def cms_stream(gp: GenParams, i0: int, n: int, threads: int, depth: int, width_log2: int, seed: int, cms_src, cms_dst,
               exact_src=None, exact_dst=None):
    """Adds the sketches (and optionally the exact per-(rank, v6) weights) of generator records [i0, i0+n) - see
    fo_cms_stream in flow_oracle.h.  Arrays are uint64 numpy arrays, modified in place."""
    L = lib()
    L.fo_cms_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fo_cms_stream.restype = None
    L.fo_cms_stream(C.byref(gp), i0, n, threads, depth, width_log2, seed, cms_src.ctypes.data, cms_dst.ctypes.data,
                    None if exact_src is None else exact_src.ctypes.data, None if exact_dst is None else exact_dst.ctypes.data)


def app_checksum_stream(gp: GenParams, i0: int, n: int, threads: int, gran: int, slot0: int, nslots: int):
    """-> (checksum per timeslot, records per timeslot, records outside): fo_app_checksum_stream - the linear checksum of the
    (SrcAddr,DstPort,Proto) rollup over generator records [i0, i0+n), no group-by (flow_oracle.h)."""
    L = lib()
    L.fo_app_checksum_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.fo_app_checksum_stream.restype = C.c_uint64
    sums = np.zeros(nslots, dtype=np.uint64)
    cnts = np.zeros(nslots, dtype=np.uint64)
    outside = L.fo_app_checksum_stream(C.byref(gp), i0, n, threads, gran, slot0, nslots, sums.ctypes.data, cnts.ctypes.data)
    return sums, cnts, int(outside)


def _mix64(z):
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xbf58476d1ce4e5b9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94d049bb133111eb)
        z ^= z >> np.uint64(31)
    return z


def app_rows_checksum(rows, timeslot=None):
    """The rows' side of app_checksum_stream: sum h(key) * (3*bytes + 5*packets + 7*count) mod 2^64 over (SrcAddr,DstPort,Proto)
    rows (ROW_APP_DTYPE, or 48-byte rows without date / timeslot when `timeslot` names the window they belong to)."""
    if len(rows) == 0:
        return 0
    addr = np.ascontiguousarray(rows["src_addr"]).view("<u8").reshape(-1, 2)
    ts = rows["timeslot"].astype(np.uint64) if timeslot is None else np.full(len(rows), timeslot, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _mix64(((ts << np.uint64(32)) | rows["dst_port"].astype(np.uint64)) ^ np.uint64(0x9E3779B97F4A7C15))
        h = _mix64(h ^ addr[:, 0])
        h = _mix64(h ^ addr[:, 1])
        h = _mix64(h ^ rows["proto"].astype(np.uint64))
        v = rows["bytes"] * np.uint64(3) + rows["packets"] * np.uint64(5) + rows["count"] * np.uint64(7)
        return int((h * v).sum(dtype=np.uint64))


def app_rows_strictly_ascending(rows) -> bool:
    """Every key once, in the emit order of fa_read_window_app: (timeslot,) SrcAddr bytes, DstPort, Proto strictly ascending."""
    if len(rows) < 2:
        return True
    addr = np.ascontiguousarray(rows["src_addr"])
    cols = [addr[:, :8].copy().view(">u8").reshape(-1).astype(np.uint64), addr[:, 8:].copy().view(">u8").reshape(-1).astype(np.uint64),
            rows["dst_port"].astype(np.uint64), rows["proto"].astype(np.uint64)]
    if "timeslot" in rows.dtype.names:
        cols.insert(0, rows["timeslot"].astype(np.uint64))
    lt = np.zeros(len(rows) - 1, dtype=bool)   # decided "ascending" by an earlier column
    eq = np.ones(len(rows) - 1, dtype=bool)    # equal in every earlier column
    for c in cols:
        lt |= eq & (c[:-1] < c[1:])
        eq &= c[:-1] == c[1:]
    return bool(lt.all())


def zipf_key(rank: int, dst: int, v6: int) -> bytes:
    out = C.create_string_buffer(16)
    L = lib()
    L.fo_zipf_key.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_char_p]
    L.fo_zipf_key.restype = None
    L.fo_zipf_key(rank, dst, v6, out)
    return out.raw


def bench_rollup(gp: GenParams, i0: int, n: int, threads: int):
    wire, groups, bad, cs = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    dt = lib().fo_bench_rollup(C.byref(gp), i0, n, threads, C.byref(wire), C.byref(groups),
                               C.byref(bad), C.byref(cs))
    return {"seconds": dt, "wire_bytes": wire.value, "groups": groups.value, "bad": bad.value,
            "checksum": cs.value}


class BenchResult(C.Structure):
    _fields_ = [("seconds", C.c_double), ("decode_min", C.c_double), ("decode_max", C.c_double), ("decode_mean", C.c_double),
                ("merge_seconds", C.c_double), ("wire_bytes", C.c_uint64), ("groups", C.c_uint64), ("bad", C.c_uint64),
                ("checksum", C.c_uint64), ("rows", C.c_uint64), ("threads", C.c_uint32), ("_pad", C.c_uint32)]


def bench_rollup_ex(gp: GenParams, i0: int, n: int, threads: int, groups_hint: int = 0, want_rows: bool = False):
    """fo_bench_rollup_ex: tables sized from `groups_hint` and first-touched before the start barrier; per-thread
    decode times in the result; want_rows: also the merged rows (ROW5M_DTYPE, sorted) under "rows"."""
    L = lib()
    L.fo_bench_rollup_ex.argtypes = [C.POINTER(GenParams), C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(BenchResult),
                                     C.c_void_p, C.c_size_t]
    L.fo_bench_rollup_ex.restype = C.c_int
    res = BenchResult()
    rows = None
    cap = 0
    if want_rows:
        cap = max(int(groups_hint) * 2, 1 << 20)
        rows = np.zeros(cap, dtype=ROW5M_DTYPE)
    rc = L.fo_bench_rollup_ex(C.byref(gp), i0, n, threads, groups_hint, C.byref(res), None if rows is None else rows.ctypes.data, cap)
    if rc != 0 and want_rows:  # more groups than guessed: again with room for all of them
        cap = int(res.rows)
        rows = np.zeros(cap, dtype=ROW5M_DTYPE)
        rc = L.fo_bench_rollup_ex(C.byref(gp), i0, n, threads, groups_hint, C.byref(res), rows.ctypes.data, cap)
    assert rc == 0
    out = {"seconds": res.seconds, "wire_bytes": res.wire_bytes, "groups": res.groups, "bad": res.bad, "checksum": res.checksum,
           "threads": res.threads, "decode_seconds_min": res.decode_min, "decode_seconds_max": res.decode_max,
           "decode_seconds_mean": res.decode_mean, "merge_seconds": res.merge_seconds}
    if want_rows:
        out["rows"] = rows[:int(res.rows)].copy()
    return out


# ---- wide key sets and dashboard read side (numpy restatements over decoded rows) -----------------
# Same status as the flows_5m rollup: "parity unpinned" (restated from the SQL; ClickHouse cannot run
# here).  Input = (rows ROW_DTYPE, status) of decode_batch(); bad records are dropped
# (inserter/inserter.go:125-126).  All sums are UInt64 and wrap mod 2^64.
ROW_APP_DTYPE = np.dtype([
    ("date", "<u4"), ("timeslot", "<u4"), ("src_addr", "u1", 16), ("dst_port", "<u4"), ("proto", "<u4"),
    ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8"),
])
PORT_ROW_DTYPE = np.dtype([("port", "<u4"), ("_pad", "<u4"), ("weight", "<u8"), ("count", "<u8")])
MINUTE_ROW_DTYPE = np.dtype([("minute", "<u4"), ("_pad", "<u4"), ("weight", "<u8"), ("count", "<u8")])


def _group_sum(keys, values):
    """keys: list of uint64 arrays, most significant first; values: list of uint64 arrays.
    -> (index of the first row of every group in sorted order, [summed values])."""
    order = np.lexsort(tuple(reversed(keys)))
    ks = [k[order] for k in keys]
    first = np.ones(len(order), dtype=bool)
    if len(order) > 1:
        diff = np.zeros(len(order) - 1, dtype=bool)
        for k in ks:
            diff |= k[1:] != k[:-1]
        first[1:] = diff
    starts = np.nonzero(first)[0]
    with np.errstate(over="ignore"):
        sums = [np.add.reduceat(v[order].astype(np.uint64), starts) if len(order) else v[:0] for v in values]
    return order[starts] if len(order) else order, sums


def rollup_app(rows, status, granule=300, window=None, timeslot=None):
    """GROUP BY Date, Timeslot, SrcAddr, DstPort, Proto -> sum(Bytes), sum(Packets), count(): the second
    key set of BASELINE config 5, with the Date/Timeslot rule of flows_5m_view
    (compose/clickhouse/create.sh:92-110: toDate / toStartOfFiveMinute of the DateTime-narrowed
    TimeReceived, create.sh:39,66).  window/timeslot: fold the sub-buckets [timeslot, timeslot+window)
    into one row per key (sliding windows over `granule`-second sub-buckets)."""
    r = rows[status == 0]
    t32 = (r["time_received"] & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    ts = t32 - t32 % np.uint64(granule)
    if timeslot is not None:
        keep = (ts >= np.uint64(timeslot)) & (ts < np.uint64(timeslot + (window or granule)))
        r, ts = r[keep], ts[keep]
        ts = np.full(len(r), timeslot, dtype=np.uint64)
    addr = np.ascontiguousarray(r["src_addr"])
    a_hi = addr[:, :8].copy().view(">u8").reshape(-1).astype(np.uint64)   # byte-lexicographic order
    a_lo = addr[:, 8:].copy().view(">u8").reshape(-1).astype(np.uint64)
    keys = [ts, a_hi, a_lo, r["dst_port"].astype(np.uint64), r["proto"].astype(np.uint64)]
    idx, (b, p, c) = _group_sum(keys, [r["bytes"], r["packets"], np.ones(len(r), dtype=np.uint64)])
    out = np.zeros(len(idx), dtype=ROW_APP_DTYPE)
    out["timeslot"] = ts[idx]
    out["date"] = ts[idx] // np.uint64(86400)
    out["src_addr"] = addr[idx]
    out["dst_port"] = r["dst_port"][idx]
    out["proto"] = r["proto"][idx]
    out["bytes"], out["packets"], out["count"] = b, p, c
    return out


def top_ports(rows, status, dst=0):
    """SELECT SrcPort|DstPort AS port, sum(Bytes*SamplingRate) AS sumbytes ... GROUP BY port ORDER BY
    sumbytes DESC (compose/grafana/dashboards/viz-ch.json:358,604); ties ordered by port."""
    r = rows[status == 0]
    port = r["dst_port" if dst else "src_port"].astype(np.uint64)
    with np.errstate(over="ignore"):
        w = r["bytes"] * r["sampling_rate"]
    idx, (ws, cs) = _group_sum([port], [w, np.ones(len(r), dtype=np.uint64)])
    out = np.zeros(len(idx), dtype=PORT_ROW_DTYPE)
    out["port"] = port[idx]
    out["weight"], out["count"] = ws, cs
    order = np.lexsort((out["port"], np.uint64(0xFFFFFFFFFFFFFFFF) - out["weight"]))
    return out[order]


def minute_series(rows, status):
    """SELECT toStartOfMinute(TimeFlowStart) AS t, sum(Bytes*SamplingRate) ... GROUP BY t ORDER BY t
    (viz-ch.json:74); TimeFlowStart narrowed UInt64 -> DateTime as in flows_raw (create.sh:40)."""
    r = rows[status == 0]
    t32 = (r["time_flow_start"] & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    minute = t32 - t32 % np.uint64(60)
    with np.errstate(over="ignore"):
        w = r["bytes"] * r["sampling_rate"]
    idx, (ws, cs) = _group_sum([minute], [w, np.ones(len(r), dtype=np.uint64)])
    out = np.zeros(len(idx), dtype=MINUTE_ROW_DTYPE)
    out["minute"] = minute[idx]
    out["weight"], out["count"] = ws, cs
    return out


def cms_sketch_numpy(keys16: np.ndarray, weights: np.ndarray, depth: int, width_log2: int, seed: int) -> np.ndarray:
    """Vectorised Count-Min sketch over uint8[n,16] keys (the same columns as fo_cms_column / fo_cms_update in
    flow_oracle.c, restated in numpy so that bench-scale inputs can be checked; pinned against the C functions by
    tests/test_oracle_golden.py).  -> uint64[depth << width_log2]."""
    def mix64(z):
        z = z.astype(np.uint64)
        with np.errstate(over="ignore"):
            z ^= z >> np.uint64(30)
            z *= np.uint64(0xbf58476d1ce4e5b9)
            z ^= z >> np.uint64(27)
            z *= np.uint64(0x94d049bb133111eb)
            z ^= z >> np.uint64(31)
        return z
    k = np.ascontiguousarray(keys16, dtype=np.uint8).reshape(-1, 16)
    lo = k[:, :8].copy().view("<u8").reshape(-1)
    hi = k[:, 8:].copy().view("<u8").reshape(-1)
    w = np.ascontiguousarray(weights, dtype=np.uint64)
    out = np.zeros(depth << width_log2, dtype=np.uint64)
    with np.errstate(over="ignore"):
        s0 = mix64(np.array([(seed + 0x9E3779B97F4A7C15) & (2**64 - 1)], dtype=np.uint64))[0]
        a = mix64(lo ^ s0)
        for r in range(depth):
            np.add.at(out, cms_columns(a, mix64(a ^ hi), width_log2, r) + (r << width_log2), w)
    return out


def cms_estimates_numpy(sketch: np.ndarray, keys16: np.ndarray, depth: int, width_log2: int, seed: int) -> np.ndarray:
    """Count-Min estimates (minimum over the rows) of uint8[n,16] keys in a sketch laid out like cms_sketch_numpy's - the ranking
    weight of fa_topk (viz-ch.json:233,479 rank by sum(Bytes*SamplingRate); the sketch over-estimates it)."""
    def mix64(z):
        z = z.astype(np.uint64)
        with np.errstate(over="ignore"):
            z ^= z >> np.uint64(30)
            z *= np.uint64(0xbf58476d1ce4e5b9)
            z ^= z >> np.uint64(27)
            z *= np.uint64(0x94d049bb133111eb)
            z ^= z >> np.uint64(31)
        return z
    k = np.ascontiguousarray(keys16, dtype=np.uint8).reshape(-1, 16)
    lo = k[:, :8].copy().view("<u8").reshape(-1)
    hi = k[:, 8:].copy().view("<u8").reshape(-1)
    sk = np.ascontiguousarray(sketch, dtype=np.uint64).reshape(-1)
    with np.errstate(over="ignore"):
        s0 = mix64(np.array([(seed + 0x9E3779B97F4A7C15) & (2**64 - 1)], dtype=np.uint64))[0]
        a = mix64(lo ^ s0)
        h1 = mix64(a ^ hi)
    est = np.full(len(k), np.uint64(0xFFFFFFFFFFFFFFFF), dtype=np.uint64)
    for r in range(depth):
        est = np.minimum(est, sk[cms_columns(a, h1, width_log2, r) + (r << width_log2)])
    return est


def topk_bin(est):
    """fa_topk's monotone estimate bins (csrc/sinks.cuh topk_bin): values below 64 exactly, above that 32 steps per octave."""
    est = np.asarray(est, dtype=np.uint64)
    out = est.astype(np.int64).copy()
    big = est >= 64
    if big.any():
        v = est[big]
        e = np.array([int(x).bit_length() - 1 for x in v.tolist()], dtype=np.uint64)
        out[big] = (((e - np.uint64(4)) << np.uint64(5)) | ((v >> (e - np.uint64(5))) & np.uint64(31))).astype(np.int64)
    return out


def topk_bin_floor(b: int) -> int:
    """The smallest estimate that maps to bin b (csrc/maintenance.cuh topk_bin_floor)."""
    if b < 64:
        return b
    e = (b >> 5) + 4
    return (1 << e) | ((b & 31) << (e - 5))


def topk_candidates(batches, depth, width_log2, seed, track=1024, capacity_log2=20):
    """The heavy-hitter contract of fa_config.topk_mode = FA_TOPK_CANDIDATES (include/flowagg.h), restated: batches = the
    ingest launches in order, each (keys uint8[n,16], weights uint64[n]) of its good records.
      R_t = R_(t-1) u { x in batch t : estimate_(t-1)(x) >= theta_(t-1) }          (nothing is admitted in the first batch)
      theta_t = max(lower edge of the topk_bin that holds rank `track` among estimates_t(R_t) [0 while |R_t| < track],
                    N_t >> (max(capacity_log2, 8) - 2), 1),   N_t = total weight so far = sum of sketch row 0 (mod 2^64)
    -> (sketch, candidates uint8[m,16] sorted, their final estimates, thetas per boundary)."""
    sketch = np.zeros(depth << width_log2, dtype=np.uint64)
    cand = np.zeros((0, 16), dtype=np.uint8)
    theta = None
    thetas = []
    for keys, w in batches:
        keys = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1, 16)
        if theta is not None and len(keys):
            uniq = np.unique(keys, axis=0)
            est_prev = cms_estimates_numpy(sketch, uniq, depth, width_log2, seed)
            cand = np.unique(np.concatenate([cand, uniq[est_prev >= np.uint64(theta)]]), axis=0)
        with np.errstate(over="ignore"):
            sketch = sketch + cms_sketch_numpy(keys, w, depth, width_log2, seed)
        theta_k = 0
        if len(cand) >= track:
            bins = np.sort(topk_bin(cms_estimates_numpy(sketch, cand, depth, width_log2, seed)))[::-1]
            theta_k = topk_bin_floor(int(bins[track - 1]))
        with np.errstate(over="ignore"):
            total = int(sketch[:1 << width_log2].sum(dtype=np.uint64))
        theta = max(theta_k, total >> (max(capacity_log2, 8) - 2), 1)
        thetas.append(theta)
    est = cms_estimates_numpy(sketch, cand, depth, width_log2, seed) if len(cand) else np.zeros(0, dtype=np.uint64)
    return sketch, cand, est, thetas


def cms_columns(a, h1, width_log2: int, row: int):
    """Columns of row `row` for keys given by their two hashes (a = mix64(lo ^ mix64(seed + phi)), h1 = mix64(a ^ hi)):
    the prefix-partitioned sketch of flow_oracle.c (fo_cms_column), vectorised.  -> int64 array."""
    pbits = min(8, width_log2 - 4)
    sub = width_log2 - pbits
    with np.errstate(over="ignore"):
        prefix = (h1 & np.uint64((1 << pbits) - 1)).astype(np.int64)
        l1 = (h1 >> np.uint64(32)).astype(np.uint32)
        l2 = ((a | np.uint64(1)) >> np.uint64(32)).astype(np.uint32) | np.uint32(1)
        low = ((l1 + np.uint32(row) * l2) >> np.uint32(32 - sub)).astype(np.int64)
    return (prefix << sub) | low
