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


GEN_MOCKER, GEN_ASPAIRS, GEN_ZIPF = 0, 1, 2
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
    L.fo_hash_key16.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32]
    L.fo_hash_key16.restype = C.c_uint64
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
    buf = np.empty(n * 96 + 256, dtype=np.uint8)
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


def bench_rollup(gp: GenParams, i0: int, n: int, threads: int):
    wire, groups, bad, cs = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    dt = lib().fo_bench_rollup(C.byref(gp), i0, n, threads, C.byref(wire), C.byref(groups),
                               C.byref(bad), C.byref(cs))
    return {"seconds": dt, "wire_bytes": wire.value, "groups": groups.value, "bad": bad.value,
            "checksum": cs.value}
