"""flow-pipeline_amd - MI355X-native flow-aggregation stage (host binding).

Thin ctypes binding over the C-ABI of ``libflowagg.so`` (``include/flowagg.h``).
The library replaces one hot path of cloudflare/flow-pipeline: Kafka message
bytes -> proto3 FlowMessage decode -> 15-column projection -> 5-minute
(SrcAS, DstAS) sum(Bytes)/sum(Packets)/count() rollup, i.e. the ClickHouse chain
``flows -> flows_raw -> flows_5m`` of ``compose/clickhouse/create.sh:5-110``.

There is no CPU fallback: if the HIP library is missing or no GPU is present the
calls raise.  (The directory name contains a hyphen; load it with
``_pkg.load()`` from the repository root or ``importlib`` - see ``_pkg.py``.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import dist, schema, spill  # noqa: F401  (re-export)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflowagg.so")
if os.environ.get("FA_LIB_VARIANT"):  # A/B experiments only (tools/): libflowagg_<variant>.so built with make OUT=... EXTRA=...
    LIB_PATH = os.path.join(_HERE, "libflowagg_%s.so" % os.environ["FA_LIB_VARIANT"])

FA_KEYS_AS_PAIR = 1
FA_KEYS_SRCADDR_CMS = 2
FA_KEYS_DSTADDR_CMS = 4
FA_KEYS_ADDR_PORT_PROTO = 8
FA_KEYS_PORT_HIST = 16
FA_KEYS_MINUTE_SERIES = 32
ALL_TIMESLOTS = 0xFFFFFFFF

MOCK_MOCKER, MOCK_ASPAIRS, MOCK_ZIPF, MOCK_GOFLOW, MOCK_DISTINCT, MOCK_REVERSED = 0, 1, 2, 3, 4, 5
T0 = 1_600_000_200  # multiple of 300

ERRORS = {
    -1: "FA_ERR_ARG", -2: "FA_ERR_NO_DEVICE", -3: "FA_ERR_HIP", -4: "FA_ERR_NOMEM",
    -5: "FA_ERR_TABLE_FULL", -6: "FA_ERR_CAPACITY", -7: "FA_ERR_FRAMING", -8: "FA_ERR_UNSUPPORTED",
}


class FlowAggError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "FA_ERR"), code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("window_secs", C.c_uint32), ("subwindow_secs", C.c_uint32),
        ("table_capacity_log2", C.c_uint32), ("cms_depth", C.c_uint32),
        ("cms_width_log2", C.c_uint32), ("cms_seed", C.c_uint64), ("key_sets", C.c_uint32),
        ("framed", C.c_int32), ("max_batch_records", C.c_uint32), ("topk_capacity_log2", C.c_uint32),
        ("wide_capacity_log2", C.c_uint32), ("topk_mode", C.c_uint32), ("topk_track", C.c_uint32), ("reserved", C.c_uint32 * 1),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "records_ok", "records_bad", "records_slow", "bytes_in", "batches", "table_used",
        "table_capacity", "kernel_ns", "kernel_ns_total", "kernel_launches", "batch_ns_total",
        "records_direct", "records_retried", "wide_used", "wide_capacity", "wave_tile_launches",
        "compact_tuple_launches", "records_misfit_compact", "decode_ns_total", "decode_launches",
        "records_late", "wide_log_chunks", "wide_log_bytes", "wide_log_records", "wide_log_recorded", "wide_log_folded",
        "wide_log_replayed", "wide_log_dropped", "wide_log_watermark_moves", "wide_log_nomem_folds", "wide_log_mode",
        "topk_theta_src", "topk_theta_dst", "topk_candidates_src", "topk_candidates_dst", "learnt_order_launches",
        "host_ingest_ns", "host_stage_wait_ns", "host_stage_copy_ns")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class MockParams(C.Structure):
    _fields_ = [
        ("mode", C.c_uint32), ("framed", C.c_uint32), ("seed", C.c_uint64),
        ("n_total", C.c_uint64), ("t0", C.c_uint64), ("span_secs", C.c_uint32),
        ("per_sec", C.c_uint32), ("zipf_log2_universe", C.c_uint32), ("zipf_s_x100", C.c_uint32),
    ]


class Columns(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "time_received", "time_flow_start", "sampling_rate", "bytes", "packets",
        "sequence_num", "src_as", "dst_as", "etype", "proto", "src_port", "dst_port",
        "sampler_address", "src_addr", "dst_addr", "status")]


class DeviceState(C.Structure):
    _fields_ = [("cms_src", C.c_void_p), ("cms_dst", C.c_void_p), ("cms_words", C.c_size_t),
                ("port_hist", C.c_void_p), ("port_hist_words", C.c_size_t),
                ("cms_src_merged", C.c_void_p), ("cms_dst_merged", C.c_void_p)]


ROW5M_DTYPE = np.dtype([
    ("date", "<u4"), ("timeslot", "<u4"), ("src_as", "<u4"), ("dst_as", "<u4"),
    ("etype", "<u4"), ("_pad", "<u4"), ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8"),
])
FLOW_ROW_DTYPE = np.dtype([
    ("time_received", "<u8"), ("time_flow_start", "<u8"), ("sampling_rate", "<u8"),
    ("bytes", "<u8"), ("packets", "<u8"), ("sequence_num", "<u4"), ("src_as", "<u4"),
    ("dst_as", "<u4"), ("etype", "<u4"), ("proto", "<u4"), ("src_port", "<u4"),
    ("dst_port", "<u4"), ("status", "<u4"), ("sampler_address", "u1", 16),
    ("src_addr", "u1", 16), ("dst_addr", "u1", 16),
])
TOPK_DTYPE = np.dtype([("key", "u1", 16), ("weight", "<u8")])
ROW_APP_DTYPE = np.dtype([
    ("date", "<u4"), ("timeslot", "<u4"), ("src_addr", "u1", 16), ("dst_port", "<u4"), ("proto", "<u4"),
    ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8"),
])
ROW_APP48_DTYPE = np.dtype([("src_addr", "u1", 16), ("dst_port", "<u4"), ("proto", "<u4"), ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8")])
PORT_ROW_DTYPE = np.dtype([("port", "<u4"), ("_pad", "<u4"), ("weight", "<u8"), ("count", "<u8")])
MINUTE_ROW_DTYPE = np.dtype([("minute", "<u4"), ("_pad", "<u4"), ("weight", "<u8"), ("count", "<u8")])
assert ROW5M_DTYPE.itemsize == 48 and FLOW_ROW_DTYPE.itemsize == 120
assert ROW_APP_DTYPE.itemsize == 56 and PORT_ROW_DTYPE.itemsize == 24 and MINUTE_ROW_DTYPE.itemsize == 24
ROW_DTYPES = [ROW5M_DTYPE, ROW_APP_DTYPE, PORT_ROW_DTYPE, PORT_ROW_DTYPE, MINUTE_ROW_DTYPE, TOPK_DTYPE, TOPK_DTYPE]  # by row kind

# every symbol include/flowagg.h declares (checked by the CPU test-suite)
EXPORTS = [
    "fa_abi_version", "fa_create", "fa_destroy", "fa_last_error", "fa_ingest", "fa_ingest_device",
    "fa_sync", "fa_decode", "fa_decode_device", "fa_open_timeslots", "fa_close_window",
    "fa_read_window", "fa_window_rows_device", "fa_merge_rows_device", "fa_topk", "fa_topk_merge_keys", "fa_cms_query", "fa_cms_read", "fa_cms_reset",
    "fa_device_state_get", "fa_merged_view_set", "fa_merge_rows", "fa_merge_allreduce", "fa_stats",
    "fa_mock_generate_device", "fa_mock_generate_host",
    "fa_read_window_app", "fa_close_window_app", "fa_merge_rows_app", "fa_top_ports", "fa_merge_ports",
    "fa_minute_series", "fa_merge_minutes", "fa_dashboard_reset", "fa_rows_to_rowbinary", "fa_format_addr",
    "fa_row_bytes", "fa_rows_device", "fa_rows_merge_device", "fa_rows_fetch", "fa_drop_window", "fa_rows_partition_device", "fa_drop_range",
    "fa_group_create", "fa_group_destroy", "fa_group_last_error", "fa_group_size", "fa_group_transport", "fa_group_open_timeslots",
    "fa_group_read_window", "fa_group_close_window", "fa_group_read_window_partitioned", "fa_group_close_window_partitioned",
    "fa_group_allreduce_sketches", "fa_group_topk", "fa_group_stats", "fa_read_window_app48", "fa_close_window_app48",
    "fa_reserve_ingest",
]
GROUP_PEER, GROUP_RCCL = 0, 1
TOPK_EXACT, TOPK_CANDIDATES = 0, 1
# row kinds of the device-resident window close (include/flowagg.h, ABI 5)
ROWS_5M, ROWS_APP, ROWS_PORT_SRC, ROWS_PORT_DST, ROWS_MINUTE, ROWS_TOPK_SRC, ROWS_TOPK_DST = range(7)

_LIB = None


def build(force=False):
    """Compile libflowagg.so for gfx950 with hipcc (cross-compiles without a GPU).

    Safe under `torch.distributed.run` (every rank calls it): the staleness check and the compile run under an
    exclusive file lock, the compiler writes to a private name and the result is renamed into place, so a rank
    never loads a half-written library and at most one rank compiles."""
    import fcntl
    srcdir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(srcdir, f) for f in os.listdir(srcdir) if not f.endswith(".so")] + [
        os.path.join(_HERE, "..", "include", "flowagg.h")]

    stamp = LIB_PATH + ".srchash"  # the sources the library was built from (a copy of the tree may reset every mtime)

    def stamped():
        try:
            with open(stamp) as f:
                return f.read().strip()
        except OSError:
            return None

    def stale():
        if not os.path.exists(LIB_PATH):
            return True
        if stamped() == source_hash():
            return False  # (built from exactly these sources, whatever the mtimes say)
        return stamped() is not None or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)

    def write_stamp():
        try:
            with open(stamp + ".tmp.%d" % os.getpid(), "w") as f:
                f.write(source_hash() + "\n")
            os.replace(stamp + ".tmp.%d" % os.getpid(), stamp)
        except OSError:
            pass

    if os.environ.get("FA_LIB_VARIANT"):  # A/B libraries are built by hand with their own EXTRA flags: never rebuilt here
        if not os.path.exists(LIB_PATH):
            raise FlowAggError(-2, "%s is not built" % LIB_PATH)
        return LIB_PATH
    if not (force or stale()):
        if stamped() is None:
            write_stamp()  # (a library built by hand with make, newer than its sources: remember what it was built from)
        return LIB_PATH
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or stale():
                tmp = LIB_PATH + ".tmp.%d" % os.getpid()
                try:
                    subprocess.check_call(["make", "-C", srcdir, "-B", "OUT=" + tmp], stdout=subprocess.DEVNULL)
                    os.replace(tmp, LIB_PATH)
                    write_stamp()
                finally:
                    if os.path.exists(tmp):
                        os.unlink(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the library's sources: stamps profiles (tools/prof_summary.py) so that
    bench.py only quotes PMC traffic measured on the code it is running."""
    import hashlib
    srcdir = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(srcdir)) + [os.path.join("..", "..", "include", "flowagg.h")]:
        if f.endswith((".cuh", ".hip", ".h", ".inc", "Makefile")):
            with open(os.path.join(srcdir, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def lib():
    """Load libflowagg.so.  Raises (never falls back) when it is not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise FlowAggError(-2, "libflowagg.so is not built (run `python -c 'import __graft_entry__ as g; "
                               "g.build()'` or `make -C flow-pipeline_amd/csrc`); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, u32, u64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64
    szp = C.POINTER(C.c_size_t)
    L.fa_abi_version.restype = u32
    L.fa_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.fa_destroy.argtypes = [vp]
    L.fa_destroy.restype = None
    L.fa_last_error.argtypes = [vp]
    L.fa_last_error.restype = C.c_char_p
    L.fa_ingest.argtypes = [vp, vp, sz, vp, sz]
    L.fa_ingest_device.argtypes = [vp, vp, sz, vp, sz]
    L.fa_sync.argtypes = [vp]
    L.fa_reserve_ingest.argtypes = [vp, sz, sz]
    L.fa_decode.argtypes = [vp, vp, sz, vp, sz, vp]
    L.fa_decode_device.argtypes = [vp, vp, sz, vp, sz, C.POINTER(Columns)]
    L.fa_open_timeslots.argtypes = [vp, vp, sz, szp]
    L.fa_close_window.argtypes = [vp, u32, vp, sz, szp]
    L.fa_read_window.argtypes = [vp, u32, vp, sz, szp]
    L.fa_window_rows_device.argtypes = [vp, u32, C.POINTER(vp), szp]
    L.fa_merge_rows_device.argtypes = [vp, vp, sz]
    L.fa_topk.argtypes = [vp, u32, sz, vp, sz, szp]
    L.fa_topk_merge_keys.argtypes = [vp, u32, vp, sz]
    L.fa_cms_query.argtypes = [vp, u32, C.c_char_p, C.POINTER(u64)]
    L.fa_cms_read.argtypes = [vp, u32, vp, sz]
    L.fa_cms_reset.argtypes = [vp, u32]
    L.fa_device_state_get.argtypes = [vp, C.POINTER(DeviceState)]
    L.fa_merged_view_set.argtypes = [vp, C.c_int]
    L.fa_merge_rows.argtypes = [vp, vp, sz]
    L.fa_merge_allreduce.argtypes = [vp, vp]
    L.fa_stats.argtypes = [vp, C.POINTER(Stats)]
    L.fa_mock_generate_device.argtypes = [vp, C.POINTER(MockParams), u64, u64, vp, sz, vp, C.POINTER(u64)]
    L.fa_mock_generate_host.argtypes = [C.POINTER(MockParams), u64, u64, vp, sz, vp, C.POINTER(u64)]
    L.fa_read_window_app.argtypes = [vp, u32, vp, sz, szp]
    L.fa_close_window_app.argtypes = [vp, u32, vp, sz, szp]
    L.fa_merge_rows_app.argtypes = [vp, vp, sz]
    L.fa_top_ports.argtypes = [vp, C.c_int, sz, vp, sz, szp]
    L.fa_merge_ports.argtypes = [vp, C.c_int, vp, sz]
    L.fa_minute_series.argtypes = [vp, vp, sz, szp]
    L.fa_merge_minutes.argtypes = [vp, vp, sz]
    L.fa_dashboard_reset.argtypes = [vp]
    L.fa_rows_to_rowbinary.argtypes = [vp, sz, vp, sz, szp]
    L.fa_format_addr.argtypes = [C.c_char_p, u32, C.c_char_p, sz]
    L.fa_row_bytes.argtypes = [C.c_int]
    L.fa_row_bytes.restype = sz
    L.fa_rows_device.argtypes = [vp, C.c_int, u32, sz, C.POINTER(vp), szp]
    L.fa_rows_merge_device.argtypes = [vp, C.c_int, vp, sz, sz, C.POINTER(vp), szp]
    L.fa_rows_fetch.argtypes = [vp, C.c_int, vp, sz, vp, sz]
    L.fa_rows_partition_device.argtypes = [vp, C.c_int, vp, sz, u32, C.POINTER(vp), szp]
    L.fa_drop_window.argtypes = [vp, C.c_int, u32]
    L.fa_drop_range.argtypes = [vp, C.c_int, u32, u32]
    L.fa_read_window_app48.argtypes = [vp, u32, vp, sz, szp, C.POINTER(u32)]
    L.fa_close_window_app48.argtypes = [vp, u32, vp, sz, szp, C.POINTER(u32)]
    L.fa_group_create.argtypes = [C.POINTER(vp), sz, u32, C.POINTER(vp)]
    L.fa_group_destroy.argtypes = [vp]
    L.fa_group_destroy.restype = None
    L.fa_group_last_error.argtypes = [vp]
    L.fa_group_last_error.restype = C.c_char_p
    L.fa_group_size.argtypes = [vp]
    L.fa_group_size.restype = sz
    L.fa_group_transport.argtypes = [vp]
    L.fa_group_open_timeslots.argtypes = [vp, vp, sz, szp]
    L.fa_group_read_window.argtypes = [vp, C.c_int, u32, sz, vp, sz, szp]
    L.fa_group_close_window.argtypes = [vp, C.c_int, u32, vp, sz, szp]
    L.fa_group_read_window_partitioned.argtypes = [vp, C.c_int, u32, vp, sz, szp, szp]
    L.fa_group_close_window_partitioned.argtypes = [vp, C.c_int, u32, vp, sz, szp, szp]
    L.fa_group_allreduce_sketches.argtypes = [vp]
    L.fa_group_topk.argtypes = [vp, u32, sz, vp, sz, szp]
    L.fa_group_stats.argtypes = [vp, C.POINTER(Stats)]
    _LIB = L
    return L


def mock_params(mode=MOCK_MOCKER, framed=1, seed=1, n_total=0, t0=T0, span_secs=900, per_sec=4,
                zipf_log2_universe=24, zipf_s_x100=110):
    return MockParams(mode, framed, seed, n_total, t0, span_secs, per_sec, zipf_log2_universe,
                      zipf_s_x100)


def mock_record_cap(mode: int) -> int:
    """Upper bound of the mean framed record size of a generator mode (buffer sizing)."""
    return 200 if mode == MOCK_GOFLOW else 96


def mock_generate_host(mp: MockParams, i0: int, n: int):
    """Host twin of the device producer -> (bytes uint8[], offsets uint64[n+1])."""
    buf = np.empty(n * mock_record_cap(mp.mode) + 256, dtype=np.uint8)
    off = np.empty(n + 1, dtype=np.uint64)
    w = C.c_uint64()
    rc = lib().fa_mock_generate_host(C.byref(mp), i0, n, buf.ctypes.data, buf.size,
                                     off.ctypes.data, C.byref(w))
    if rc:
        raise FlowAggError(rc, "fa_mock_generate_host")
    return buf[:w.value].copy(), off


def rows_to_rowbinary(rows: np.ndarray) -> bytes:
    """flows_5m rows -> `INSERT INTO flows_5m FORMAT RowBinary` payload (create.sh:70-90 column list)."""
    r = np.ascontiguousarray(rows, dtype=ROW5M_DTYPE)
    out = np.empty(len(r) * 70, dtype=np.uint8)
    n = C.c_size_t()
    rc = lib().fa_rows_to_rowbinary(r.ctypes.data, len(r), out.ctypes.data, out.size, C.byref(n))
    if rc:
        raise FlowAggError(rc, "fa_rows_to_rowbinary")
    return out[:n.value].tobytes()


def format_addr(addr, etype: int) -> str:
    """The dashboards' address string (viz-ch.json:233,479): dotted IPv4 of the first four bytes when
    EType = 0x800, ClickHouse IPv6NumToString of the FixedString(16) otherwise."""
    a = bytes(addr)
    if len(a) != 16:
        raise ValueError("FixedString(16) expected")
    out = C.create_string_buffer(46)
    rc = lib().fa_format_addr(a, etype, out, 46)
    if rc:
        raise FlowAggError(rc, "fa_format_addr")
    return out.value.decode("ascii")


def rowbinary_to_rows(blob: bytes) -> np.ndarray:
    """Inverse of rows_to_rowbinary (plain Python decoder, for tests and offline checks)."""
    import struct
    out = []
    p = 0
    while p < len(blob):
        date, ts, sa, da = struct.unpack_from("<HIII", blob, p)
        p += 14
        arr = []
        for fmt, size in (("<I", 4), ("<Q", 8), ("<Q", 8), ("<Q", 8)):
            if blob[p] != 1:
                raise ValueError("ETypeMap arrays hold exactly one element")
            arr.append(struct.unpack_from(fmt, blob, p + 1)[0])
            p += 1 + size
        b, pk, c = struct.unpack_from("<QQQ", blob, p)
        p += 24
        if (b, pk, c) != tuple(arr[1:]):
            raise ValueError("ETypeMap values differ from the scalar columns")
        out.append((date, ts, sa, da, arr[0], 0, b, pk, c))
    return np.array(out, dtype=ROW5M_DTYPE)


_ROWBINARY_DTYPE = np.dtype([  # one flows_5m row of fa_rows_to_rowbinary (create.sh:70-90 column list), packed: 70 bytes
    ("date", "<u2"), ("timeslot", "<u4"), ("src_as", "<u4"), ("dst_as", "<u4"),
    ("n_etype", "u1"), ("m_etype", "<u4"), ("n_bytes", "u1"), ("m_bytes", "<u8"), ("n_packets", "u1"), ("m_packets", "<u8"), ("n_count", "u1"), ("m_count", "<u8"),
    ("bytes", "<u8"), ("packets", "<u8"), ("count", "<u8")])
assert _ROWBINARY_DTYPE.itemsize == 70


def rowbinary_to_rows_fast(blob) -> np.ndarray:
    """rowbinary_to_rows for large outputs (numpy, no Python loop): the rows of fa_rows_to_rowbinary all have the same shape
    (one element per ETypeMap array) - anything else raises."""
    b = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray)) else np.ascontiguousarray(blob, dtype=np.uint8)
    if b.size % 70:
        raise ValueError("not a whole number of 70-byte flows_5m RowBinary rows")
    r = b.view(_ROWBINARY_DTYPE)
    if not ((r["n_etype"] == 1) & (r["n_bytes"] == 1) & (r["n_packets"] == 1) & (r["n_count"] == 1)).all():
        raise ValueError("ETypeMap arrays hold exactly one element")
    if not ((r["m_bytes"] == r["bytes"]) & (r["m_packets"] == r["packets"]) & (r["m_count"] == r["count"])).all():
        raise ValueError("ETypeMap values differ from the scalar columns")
    out = np.zeros(len(r), dtype=ROW5M_DTYPE)
    for f in ("date", "timeslot", "src_as", "dst_as", "bytes", "packets", "count"):
        out[f] = r[f]
    out["etype"] = r["m_etype"]
    return out


class FlowAgg:
    """One aggregation context = one (Kafka partition, GPU) pair."""

    def __init__(self, device=0, window_secs=300, subwindow_secs=0, table_capacity_log2=20,
                 key_sets=FA_KEYS_AS_PAIR, framed=True, cms_depth=4, cms_width_log2=20,
                 cms_seed=0x5EED, max_batch_records=0, topk_capacity_log2=0, wide_capacity_log2=0, topk_mode=0, topk_track=0):
        self._L = lib()
        self.cfg = Config(device, window_secs, subwindow_secs, table_capacity_log2, cms_depth,
                          cms_width_log2, cms_seed, key_sets, 1 if framed else 0, max_batch_records,
                          topk_capacity_log2, wide_capacity_log2, topk_mode, topk_track)
        h = C.c_void_p()
        rc = self._L.fa_create(C.byref(self.cfg), C.byref(h))
        if rc:
            raise FlowAggError(rc, (self._L.fa_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.fa_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise FlowAggError(rc, (self._L.fa_last_error(self._h) or b"").decode())

    # -- ingest ---------------------------------------------------------------
    def ingest(self, buf, offsets=None):
        """Host bytes (bytes / uint8 array) + uint64 offsets (n+1) or None (framed stream)."""
        b = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else \
            np.ascontiguousarray(buf, dtype=np.uint8)
        if offsets is None:
            self._chk(self._L.fa_ingest(self._h, b.ctypes.data, b.size, None, 0))
        else:
            o = np.ascontiguousarray(offsets, dtype=np.uint64)
            self._chk(self._L.fa_ingest(self._h, b.ctypes.data, b.size, o.ctypes.data, len(o) - 1))

    def ingest_device(self, d_buf_ptr: int, nbytes: int, d_off_ptr: int, n: int):
        """Device-resident batch: raw device pointers (e.g. torch tensor .data_ptr())."""
        self._chk(self._L.fa_ingest_device(self._h, d_buf_ptr, nbytes, d_off_ptr, n))

    def sync(self):
        self._chk(self._L.fa_sync(self._h))

    def reserve_ingest(self, nbytes: int, records: int):
        """Allocates the pinned staging slots (and their device twins) for host batches of up to nbytes / records now."""
        self._chk(self._L.fa_reserve_ingest(self._h, nbytes, records))

    # -- decode / projection -----------------------------------------------------
    def decode(self, buf, offsets) -> np.ndarray:
        b = np.frombuffer(buf, dtype=np.uint8) if isinstance(buf, (bytes, bytearray)) else \
            np.ascontiguousarray(buf, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(o) - 1
        out = np.zeros(n, dtype=FLOW_ROW_DTYPE)
        self._chk(self._L.fa_decode(self._h, b.ctypes.data, b.size, o.ctypes.data, n, out.ctypes.data))
        return out

    def decode_device(self, d_buf_ptr: int, nbytes: int, d_off_ptr: int, n: int) -> Columns:
        cols = Columns()
        self._chk(self._L.fa_decode_device(self._h, d_buf_ptr, nbytes, d_off_ptr, n, C.byref(cols)))
        return cols

    # -- window close ---------------------------------------------------------------
    def _rows_call(self, fn, timeslot, dtype=ROW5M_DTYPE):
        n = C.c_size_t()
        # one call in the common case: the table's group count bounds the rows of any window (a too small buffer
        # would make the library extract and sort the window a second time)
        st = self.stats()
        cap = max(1 << 12, int(st["table_used"] if dtype is ROW5M_DTYPE else st["wide_used"]))
        while True:
            out = np.empty(cap, dtype=dtype)
            rc = fn(self._h, timeslot, out.ctypes.data, cap, C.byref(n))
            if rc == -6:  # FA_ERR_CAPACITY: n holds the required size
                cap = n.value
                continue
            self._chk(rc)
            return out[:n.value]

    def read_window(self, timeslot=ALL_TIMESLOTS) -> np.ndarray:
        return self._rows_call(self._L.fa_read_window, timeslot)

    def close_window(self, timeslot=ALL_TIMESLOTS) -> np.ndarray:
        return self._rows_call(self._L.fa_close_window, timeslot)

    def window_rows_device(self, timeslot=ALL_TIMESLOTS):
        """-> (device pointer, n): the window's rows sorted in HBM (valid until the ctx's next window / ingest call)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.fa_window_rows_device(self._h, timeslot, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def merge_rows_device(self, d_rows_ptr: int, n: int):
        self._chk(self._L.fa_merge_rows_device(self._h, d_rows_ptr, n))

    # -- device-resident window close (ABI 5) -----------------------------------------------
    def rows_device(self, kind: int, timeslot=ALL_TIMESLOTS, k: int = 0):
        """-> (device pointer, n): this ctx's result for (kind, timeslot) as the matching read call returns it, left
        in HBM (valid until the ctx's next call)."""
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.fa_rows_device(self._h, kind, timeslot, k, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def rows_merge_device(self, kind: int, d_rows_ptr: int, n: int, k: int = 0):
        """Merges n rows in HBM (several ranks' rows_device results back to back) -> (device pointer, n) in emit order."""
        p, m = C.c_void_p(), C.c_size_t()
        self._chk(self._L.fa_rows_merge_device(self._h, kind, d_rows_ptr, n, k, C.byref(p), C.byref(m)))
        return p.value or 0, m.value

    def rows_fetch(self, kind: int, d_rows_ptr: int, n: int, out: np.ndarray | None = None) -> np.ndarray:
        """n rows of `kind` from HBM to the host (large results leave through the ctx's pinned buffer in pieces).  out: a
        buffer of the kind's dtype to reuse (a consumer keeps one per row kind: fresh pages cost more than the copy)."""
        if out is None or len(out) < n or out.dtype != ROW_DTYPES[kind]:
            out = np.empty(n, dtype=ROW_DTYPES[kind])
        if n:
            self._chk(self._L.fa_rows_fetch(self._h, kind, d_rows_ptr, n, out.ctypes.data, len(out)))
        return out[:n]

    @staticmethod
    def pinned_rows(kind: int, n: int) -> np.ndarray:
        """A row buffer of `kind` in page-locked host memory (torch's pinned allocator): fa_rows_fetch / the window reads then
        write into it with ONE copy-engine transfer instead of relaying through the ctx's pinned slots with host threads (57 against
        54 GB/s, and no host core busy)."""
        import torch
        dt = ROW_DTYPES[kind]
        t = torch.empty(max(n, 1) * dt.itemsize, dtype=torch.uint8, pin_memory=True)
        return t.numpy().view(dt)  # (the array keeps the tensor alive through .base)

    def rows_partition_device(self, kind: int, d_rows_ptr: int, n: int, world: int):
        """n rows in HBM regrouped by owning rank (hash of the key) -> (device pointer, counts per rank)."""
        p = C.c_void_p()
        counts = (C.c_size_t * world)()
        self._chk(self._L.fa_rows_partition_device(self._h, kind, d_rows_ptr, n, world, C.byref(p), counts))
        return p.value or 0, [int(x) for x in counts]

    def drop_window(self, kind: int, timeslot=ALL_TIMESLOTS):
        """Removes what close_window / close_window_app would remove after emitting `timeslot`."""
        self._chk(self._L.fa_drop_window(self._h, kind, timeslot))

    def drop_range(self, kind: int, timeslot_lo: int, timeslot_hi: int):
        """Removes every (sub-)bucket whose start lies in [timeslot_lo, timeslot_hi) in one pass (tumbling closes over sub-buckets)."""
        self._chk(self._L.fa_drop_range(self._h, kind, timeslot_lo, timeslot_hi))

    def open_timeslots(self) -> np.ndarray:
        n = C.c_size_t()
        cap = 1 << 10
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            rc = self._L.fa_open_timeslots(self._h, out.ctypes.data, cap, C.byref(n))
            if rc == -6:
                cap = n.value
                continue
            self._chk(rc)
            return out[:n.value].copy()

    def merge_rows(self, rows: np.ndarray):
        r = np.ascontiguousarray(rows, dtype=ROW5M_DTYPE)
        self._chk(self._L.fa_merge_rows(self._h, r.ctypes.data, len(r)))

    # -- second exact key set: (SrcAddr, DstPort, Proto) -------------------------------
    def read_window_app(self, timeslot=ALL_TIMESLOTS, out: np.ndarray | None = None) -> np.ndarray:
        # With pending log chunks the table's row count says nothing about the size of a window (and a too small buffer
        # makes the library collect and sort the window twice): the result is left in HBM first, then fetched.
        if out is None and self.stats()["wide_log_chunks"] == 0:
            return self._rows_call(self._L.fa_read_window_app, timeslot, ROW_APP_DTYPE)
        ptr, n = self.rows_device(ROWS_APP, timeslot)
        return self.rows_fetch(ROWS_APP, ptr, n, out=out)

    def close_window_app(self, timeslot=ALL_TIMESLOTS, out: np.ndarray | None = None) -> np.ndarray:
        if out is None and self.stats()["wide_log_chunks"] == 0:
            return self._rows_call(self._L.fa_close_window_app, timeslot, ROW_APP_DTYPE)
        rows = self.read_window_app(timeslot, out=out)
        self.drop_window(ROWS_APP, timeslot)
        return rows

    def read_window_app48(self, timeslot: int, out: np.ndarray | None = None, close=False):
        """ONE window's (SrcAddr,DstPort,Proto) rows without the date / timeslot they share: 48-byte rows -> (rows, date)."""
        fn = self._L.fa_close_window_app48 if close else self._L.fa_read_window_app48
        n, date = C.c_size_t(), C.c_uint32()
        cap = len(out) if out is not None else 1 << 12
        while True:
            if out is None or len(out) < cap or out.dtype != ROW_APP48_DTYPE:
                out = np.empty(cap, dtype=ROW_APP48_DTYPE)
            rc = fn(self._h, timeslot, out.ctypes.data, len(out), C.byref(n), C.byref(date))
            if rc == -6:
                cap = n.value
                out = None
                continue
            self._chk(rc)
            return out[:n.value], date.value

    def merge_rows_app(self, rows: np.ndarray):
        r = np.ascontiguousarray(rows, dtype=ROW_APP_DTYPE)
        self._chk(self._L.fa_merge_rows_app(self._h, r.ctypes.data, len(r)))

    # -- dashboard read side -------------------------------------------------------------
    def top_ports(self, dst: int, k=1 << 32) -> np.ndarray:
        """GROUP BY SrcPort (dst=0) / DstPort (dst=1) ORDER BY sum(Bytes*SamplingRate) DESC, first k rows."""
        n = C.c_size_t()
        cap = 1 << 12
        while True:
            out = np.zeros(cap, dtype=PORT_ROW_DTYPE)
            rc = self._L.fa_top_ports(self._h, dst, min(int(k), 1 << 40), out.ctypes.data, cap, C.byref(n))
            if rc == -6:
                cap = n.value
                continue
            self._chk(rc)
            return out[:n.value].copy()

    def merge_ports(self, dst: int, rows: np.ndarray):
        r = np.ascontiguousarray(rows, dtype=PORT_ROW_DTYPE)
        self._chk(self._L.fa_merge_ports(self._h, dst, r.ctypes.data, len(r)))

    def minute_series(self) -> np.ndarray:
        n = C.c_size_t()
        cap = 1 << 10
        while True:
            out = np.zeros(cap, dtype=MINUTE_ROW_DTYPE)
            rc = self._L.fa_minute_series(self._h, out.ctypes.data, cap, C.byref(n))
            if rc == -6:
                cap = n.value
                continue
            self._chk(rc)
            return out[:n.value].copy()

    def merge_minutes(self, rows: np.ndarray):
        r = np.ascontiguousarray(rows, dtype=MINUTE_ROW_DTYPE)
        self._chk(self._L.fa_merge_minutes(self._h, r.ctypes.data, len(r)))

    def dashboard_reset(self):
        self._chk(self._L.fa_dashboard_reset(self._h))

    # -- sketches -------------------------------------------------------------------
    def cms_read(self, key_set) -> np.ndarray:
        words = self.cfg.cms_depth << self.cfg.cms_width_log2
        out = np.zeros(words, dtype=np.uint64)
        self._chk(self._L.fa_cms_read(self._h, key_set, out.ctypes.data, words))
        return out.reshape(self.cfg.cms_depth, -1)

    def cms_query(self, key_set, key: bytes) -> int:
        w = C.c_uint64()
        self._chk(self._L.fa_cms_query(self._h, key_set, key, C.byref(w)))
        return w.value

    def cms_reset(self, key_set):
        self._chk(self._L.fa_cms_reset(self._h, key_set))

    def topk(self, key_set, k) -> np.ndarray:
        """k heaviest addresses by Count-Min estimate of sum(Bytes*SamplingRate): (key, weight) rows."""
        k = min(int(k), 1 << (self.cfg.topk_capacity_log2 or 20))
        out = np.zeros(k, dtype=TOPK_DTYPE)
        n = C.c_size_t()
        self._chk(self._L.fa_topk(self._h, key_set, k, out.ctypes.data, k, C.byref(n)))
        return out[:n.value].copy()

    def topk_merge_keys(self, key_set, keys: np.ndarray):
        """Adds candidate keys (uint8[n,16]) found by other contexts / ranks."""
        kk = np.ascontiguousarray(keys, dtype=np.uint8).reshape(-1, 16)
        self._chk(self._L.fa_topk_merge_keys(self._h, key_set, kk.ctypes.data, len(kk)))

    def device_state(self) -> DeviceState:
        st = DeviceState()
        self._chk(self._L.fa_device_state_get(self._h, C.byref(st)))
        return st

    def merged_view_set(self, valid: bool):
        """Declare the merged sketch view (device_state().cms_*_merged) valid / stale."""
        self._chk(self._L.fa_merged_view_set(self._h, 1 if valid else 0))

    def merge_allreduce(self, rccl_comm_ptr: int):
        """In-library RCCL path: ncclAllReduce of the sketches into the merged view (out of place, idempotent)."""
        self._chk(self._L.fa_merge_allreduce(self._h, rccl_comm_ptr))

    def stats(self) -> dict:
        s = Stats()
        self._chk(self._L.fa_stats(self._h, C.byref(s)))
        return s.as_dict()

    # -- synthetic producer ------------------------------------------------------------
    def mock_generate_device(self, mp: MockParams, i0: int, n: int, d_buf_ptr: int, cap: int,
                             d_off_ptr: int) -> int:
        w = C.c_uint64()
        self._chk(self._L.fa_mock_generate_device(self._h, C.byref(mp), i0, n, d_buf_ptr, cap,
                                                  d_off_ptr, C.byref(w)))
        return w.value


class FlowGroup:
    """The window close of several contexts inside ONE process (ABI 7+, fa_group_*): one FlowAgg per (Kafka partition, GPU),
    results merged in HBM.  The multi-process twin is flow-pipeline_amd.dist (one rank per GPU under torchrun)."""

    def __init__(self, members, transport=GROUP_PEER):
        self._L = lib()
        self.members = list(members)
        arr = (C.c_void_p * len(self.members))(*[m._h for m in self.members])
        h = C.c_void_p()
        rc = self._L.fa_group_create(arr, len(self.members), transport, C.byref(h))
        if rc:
            raise FlowAggError(rc, (self._L.fa_group_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.fa_group_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc:
            raise FlowAggError(rc, (self._L.fa_group_last_error(self._h) or b"").decode())

    @property
    def transport(self) -> int:
        return self._L.fa_group_transport(self._h)

    def _rows(self, call, kind, cap=1 << 12):
        n = C.c_size_t()
        while True:
            out = np.empty(cap, dtype=ROW_DTYPES[kind])
            rc = call(out, cap, n)
            if rc == -6:
                cap = n.value
                continue
            self._chk(rc)
            return out[:n.value]

    def open_timeslots(self) -> np.ndarray:
        n = C.c_size_t()
        cap = 1 << 10
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            rc = self._L.fa_group_open_timeslots(self._h, out.ctypes.data, cap, C.byref(n))
            if rc == -6:
                cap = n.value
                continue
            self._chk(rc)
            return out[:n.value].copy()

    def read_window(self, kind=ROWS_5M, timeslot=ALL_TIMESLOTS, k=0, cap=1 << 12) -> np.ndarray:
        return self._rows(lambda out, c, n: self._L.fa_group_read_window(self._h, kind, timeslot, k, out.ctypes.data, c, C.byref(n)), kind, cap)

    def close_window(self, kind=ROWS_5M, timeslot=ALL_TIMESLOTS, cap=1 << 12) -> np.ndarray:
        return self._rows(lambda out, c, n: self._L.fa_group_close_window(self._h, kind, timeslot, out.ctypes.data, c, C.byref(n)), kind, cap)

    def _part(self, fn, kind, timeslot, cap):
        shares = (C.c_size_t * len(self.members))()
        rows = self._rows(lambda out, c, n: fn(self._h, kind, timeslot, out.ctypes.data, c, shares, C.byref(n)), kind, cap)
        return rows, [int(v) for v in shares]

    def read_window_partitioned(self, kind=ROWS_APP, timeslot=ALL_TIMESLOTS, cap=1 << 12):
        """-> (rows, share sizes): the members' shares back to back, share 0 first; a key lives in exactly one share."""
        return self._part(self._L.fa_group_read_window_partitioned, kind, timeslot, cap)

    def close_window_partitioned(self, kind=ROWS_APP, timeslot=ALL_TIMESLOTS, cap=1 << 12):
        return self._part(self._L.fa_group_close_window_partitioned, kind, timeslot, cap)

    def allreduce_sketches(self):
        self._chk(self._L.fa_group_allreduce_sketches(self._h))

    def topk(self, key_set, k) -> np.ndarray:
        out = np.zeros(max(int(k), 1), dtype=TOPK_DTYPE)
        n = C.c_size_t()
        self._chk(self._L.fa_group_topk(self._h, key_set, int(k), out.ctypes.data, len(out), C.byref(n)))
        return out[:n.value].copy()

    def stats(self) -> dict:
        s = Stats()
        self._chk(self._L.fa_group_stats(self._h, C.byref(s)))
        return s.as_dict()
