"""CPU tests of the drop-in boundary: libflowagg.so loads, exports every symbol
include/flowagg.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "flowagg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(fa):
    if not os.path.exists(fa.LIB_PATH):
        fa.build()
    L = C.CDLL(fa.LIB_PATH)
    names = header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libflowagg.so does not export %s" % n
    assert sorted(fa.EXPORTS) == names, "python binding and header disagree"
    assert fa.lib().fa_abi_version() == 8


def test_struct_layouts_match_header(fa):
    assert C.sizeof(fa.Config) == 64
    assert C.sizeof(fa.Stats) == 312  # ABI 8: + host_ingest_ns, host_stage_wait_ns, host_stage_copy_ns
    assert C.sizeof(fa.MockParams) == 48
    assert fa.ROW5M_DTYPE.itemsize == 48 and fa.FLOW_ROW_DTYPE.itemsize == 120


def test_no_cpu_fallback(fa):
    """Without a HIP device the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(fa.FlowAggError) as ei:
        fa.FlowAgg()
    assert ei.value.code == -2


def test_product_code_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may reference it."""
    pkg = os.path.join(ROOT, "flow-pipeline_amd")
    for dp, _dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp", ".go", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "flow_oracle" not in txt and "pyoracle" not in txt, f


def test_host_generator_matches_oracle(fa, po):
    """fa_mock_generate_host (no GPU needed) emits the same bytes as the oracle generator."""
    import numpy as np
    if not os.path.exists(fa.LIB_PATH):
        fa.build()
    for mode in (0, 1, 2):
        for framed in (0, 1):
            gp = po.gen_params(mode=mode, framed=framed, seed=42, n_total=5000, per_sec=3)
            mp = fa.mock_params(mode=mode, framed=framed, seed=42, n_total=5000, per_sec=3)
            wb, wo = po.gen_records(gp, 100, 5000)
            hb, ho = fa.mock_generate_host(mp, 100, 5000)
            assert np.array_equal(hb, wb) and np.array_equal(ho, wo)


def test_rowbinary_sink_known_answer(fa):
    """flows_5m row of SURVEY Appendix A.1 KAT-1 -> RowBinary bytes (create.sh:70-90 column order), written
    out by hand from the ClickHouse RowBinary rules: LE fixed-width ints, arrays = LEB128 length + elements."""
    import numpy as np
    rows = np.zeros(2, dtype=fa.ROW5M_DTYPE)
    rows[0] = (19987, 1726899900, 65002, 65001, 34525, 0, 1499, 99, 1)
    rows[1] = (19987, 1726899900, 1, 2, 0x800, 0, 2**64 - 1, 2**63, 7)
    blob = fa.rows_to_rowbinary(rows)
    want0 = ("134e" "bc66ee66" "eafd0000" "e9fd0000"
             "01" "dd860000" "01" "db05000000000000" "01" "6300000000000000" "01" "0100000000000000"
             "db05000000000000" "6300000000000000" "0100000000000000")
    assert len(blob) == 140 and blob[:70].hex() == want0
    assert fa.rowbinary_to_rows(blob).tobytes() == rows.tobytes()
    bad = rows[:1].copy()
    bad["date"] = 70000  # Date is UInt16 days
    with pytest.raises(fa.FlowAggError):
        fa.rows_to_rowbinary(bad)


def test_format_addr_matches_glibc_and_readme(fa):
    """fa_format_addr = the dashboards' address expression (viz-ch.json:233,479).  IPv6NumToString follows BIND's
    inet_ntop6, so glibc's inet_ntop is the second opinion; README.md:155-161,186-221 hold the known answers."""
    import random
    import socket
    import struct
    if not os.path.exists(fa.LIB_PATH):
        fa.build()
    v6 = lambda s: socket.inet_pton(socket.AF_INET6, s)
    # README.md:155-161 (mocker addresses as ClickHouse prints them) and README.md:191 (192.168.1.1 stored as the
    # FixedString of its little-endian UInt32 renders as "101:a8c0::")
    assert fa.format_addr(v6("2001:db8:0:1::80"), 0x86dd) == "2001:db8:0:1::80"
    assert fa.format_addr(v6("2001:db8:0:1::"), 0x86dd) == "2001:db8:0:1::"
    assert fa.format_addr(struct.pack("<I", 3232235777) + bytes(12), 0x86dd) == "101:a8c0::"
    # what GoFlow stores for IPv4 (4 bytes in network order + 12 NULs) under the EType = 0x800 branch
    assert fa.format_addr(bytes([192, 168, 1, 1]) + bytes(12), 0x800) == "192.168.1.1"
    assert fa.format_addr(bytes([0, 0, 0, 0]) + bytes(12), 0x800) == "0.0.0.0"
    assert fa.format_addr(bytes([255, 255, 255, 255]) + b"\x01" * 12, 0x800) == "255.255.255.255"  # bytes 5..16 ignored
    assert fa.format_addr(bytes(16), 0) == "::"
    rng = random.Random(7)
    cases = [bytes(16), bytes(15) + b"\x01", bytes(10) + b"\xff\xff" + bytes([1, 2, 3, 4]), bytes(12) + bytes([1, 2, 3, 4]),
             bytes(10) + b"\xff\xfe" + bytes([1, 2, 3, 4]), b"\x00\x01" + bytes(14), b"\xff" * 16]
    for _ in range(20000):
        words = [rng.choice((0, 0, 0, 1, 0xff, 0x100, 0xffff, rng.getrandbits(16))) for _ in range(8)]
        cases.append(struct.pack(">8H", *words))
    for a in cases:
        assert fa.format_addr(a, 0x86dd) == socket.inet_ntop(socket.AF_INET6, a), a.hex()
    out = C.create_string_buffer(4)
    assert fa.lib().fa_format_addr(bytes(15) + b"\x01", 0x86dd, out, 3) == -6  # "::1" needs 4 bytes
    assert fa.lib().fa_format_addr(bytes(15) + b"\x01", 0x86dd, out, 4) == 0 and out.value == b"::1"
