#!/usr/bin/env python3
"""Generates the committed golden fixtures in this directory.

Run in the build container only (needs /root/reference and the protobuf wheel):

    python tests/golden/make_golden.py

What pins what
--------------
* The field tables in flow-pipeline_amd/schema.py are checked against the
  reference's own schema sources: the text of pb-ext/flow.proto and the gzipped
  FileDescriptorProto embedded at pb-ext/flow.pb.go:650-714.
* Every expectation below is produced by upb-protobuf (google.protobuf 7.35.1)
  parsing the record with that schema - an implementation independent of both
  the C oracle and the HIP kernels - plus a pure-Python dict group-by restating
  compose/clickhouse/create.sh:64-110 for the rollup fixture.
* The one policy on top of upb: an address field (SrcAddr/DstAddr/SamplerAddress)
  occurrence longer than 16 bytes makes the record bad (FixedString(16) overflow,
  SURVEY.md 8(a)-4).
The reference ships no tests or vectors of its own, and neither Go
nor ClickHouse exist in this image, so these fixtures are the strongest pin
available; the ClickHouse-side semantics remain "parity unpinned".
"""
import gzip
import json
import os
import random
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402

fa = _pkg.load()
schema = fa.schema
REF = "/root/reference"


def check_schema_against_reference():
    txt = open(os.path.join(REF, "pb-ext/flow.proto")).read()
    ref = sorted((int(n), nm, t) for t, nm, n in re.findall(r"^\s*(\w+)\s+(\w+)\s*=\s*(\d+);", txt, re.M)
                 if nm not in schema.FLOW_TYPES)
    mine = sorted((n, nm, ("FlowType" if t == "enum" else t)) for nm, n, _k, t in schema.LIGHT)
    assert ref == mine, "schema.LIGHT != pb-ext/flow.proto"
    go = open(os.path.join(REF, "pb-ext/flow.pb.go")).read()
    blk = go[go.index("var fileDescriptor_3864ec3df39bbe77"):]
    bs = bytes(int(x, 16) for x in re.findall(r"0x([0-9a-f]{2})", blk))
    from google.protobuf import descriptor_pb2
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.ParseFromString(gzip.decompress(bs))
    ref2 = sorted((f.number, f.name, f.type) for f in fdp.message_type[0].field)
    mine2 = sorted((n, nm, schema._PB_TYPE[t]) for nm, n, _k, t in schema.FULL)
    assert ref2 == mine2, "schema.FULL != descriptor embedded in pb-ext/flow.pb.go"


L = schema.message_class("light")


def overlong(b):
    """top-level occurrence of an address field (6, 7, 11; LEN) longer than 16 bytes"""
    p, depth = 0, 0

    def vi(p, maxb):
        v = 0
        for i in range(maxb):
            if p >= len(b):
                return None, p
            c = b[p]
            p += 1
            v |= (c & 0x7F) << (7 * i)
            if not c & 0x80:
                return v, p
        return None, p
    while p < len(b):
        t, p = vi(p, 5)
        if t is None:
            return False
        f, w = t >> 3, t & 7
        if w == 0:
            v, p = vi(p, 10)
            if v is None:
                return False
        elif w == 1:
            p += 8
        elif w == 5:
            p += 4
        elif w == 2:
            ln, p = vi(p, 5)
            if ln is None:
                return False
            if depth == 0 and f in (6, 7, 11) and ln > 16:
                return True
            p += ln
        elif w == 3:
            depth += 1
        elif w == 4:
            depth -= 1
        else:
            return False
    return False


def upb_decode(b):
    """-> dict of the 15 projected columns (create.sh:7-27) or None (bad record)"""
    if overlong(b):
        return None
    m = L()
    try:
        m.ParseFromString(b)
    except Exception:
        return None
    d = {}
    for col, (_num, kind, _typ) in schema.PROJECTED.items():
        v = getattr(m, "Etype" if col == "EType" else col)
        if kind == "b":
            if len(v) > 16:
                return None
            v = (v + b"\0" * (16 - len(v))).hex()
        d[col] = v
    return d


EDGE = {
    # SURVEY.md Appendix A.1
    "KAT-1 all mocker fields": "10a0ceb9b706180120c0c407321020010db80000000100000000000000803a1020010db800000001000000000000002048db0b506370eafb0378e9fb03a801ffff03b001bb03f001dd8d02b002a0ceb9b706",
    "KAT-2 zero omission": "1080f7c4d5061801321020010db80000000100000000000000003a1020010db800000001000000000000000070e8fb0378e8fb03f001dd8d02b00280f7c4d506",
    # Appendix A.2
    "SrcAS twice, last wins": "70e8fb037001",
    "uint32 field 7-byte varint": "70ffffffffff1f",
    "uint32 field 10-byte varint": "70ffffffffffffffffff01",
    "Bytes = 2^63": "4880808080808080808001",
    "Bytes 10th byte excess bits": "48ffffffffffffffffff7f",
    "11-byte varint": "48ffffffffffffffffffff01",
    "unknown varint then SrcAS": "c03e057001",
    "unknown LEN then SrcAS": "c23e036162637001",
    "unknown fixed32 then SrcAS": "c53e010203047001",
    "unknown fixed64 then SrcAS": "c13e01020304050607087001",
    "SrcAddr 4 bytes": "3204c0a80101",
    "SrcAddr empty": "3200",
    "SrcAddr 20 bytes": "3214" + "11" * 20,
    "SrcAddr 20 bytes then 4 bytes": "3214" + "11" * 20 + "3204c0a80101",
    "SrcAddr LEN past end": "3210" + "11" * 8,
    "field 14 as LEN": "720161",
    "field 14 as LEN truncated": "7205",
    "enum out of range": "0805",
    "field number 0": "0001",
    "empty payload": "",
    # further upb-pinned rules
    "tag 3 bytes (field 1000)": "c03e057001",
    "tag non-minimal 2 bytes": "f00007",
    "tag padded to 5 bytes": "f08080800007",
    "tag 6 bytes": "f0808080800007",
    "tag 5th byte 0x0f": "f0ffffff0f07",
    "tag 5th byte 0x10": "f0ffffff1007",
    "wire type 6": "7601",
    "wire type 7": "7701",
    "group empty": "c33ec43e7001",
    "group with content": "c33e0805c43e7001",
    "group mismatched end": "c33ecc3e7001",
    "group unterminated": "c33e0805",
    "stray end group": "c43e7001",
    "nested groups": "c33ecb3ecc3ec43e7001",
    "group on known field": "73747002",
    "known field inside group not applied": "c33e3204c0a80101c43e7001",
    "LEN overrun inside group": "c33e3210c0a80101c43e7001",
    "field 0 inside group tolerated": "7b000000007c7003",
    "group depth 100": "c33e" * 100 + "c43e" * 100 + "7001",
    "group depth 101": "c33e" * 101 + "c43e" * 101 + "7001",
    "fixed64 on known field": "7101020304050607087003",
    "fixed32 on known field": "75010203047004",
    "fixed32 truncated": "75010203",
    "fixed64 truncated": "7101020304050607",
    "LEN size 2-byte encoding": "328100aa7005",
    "LEN size padded 5 bytes": "32858080800011223344557001",
    "LEN size padded 6 bytes": "3285808080800011223344557001",
    "LEN size huge": "32ffffffff0f",
    "varint truncated at end": "7080",
    "tag truncated at end": "7001f0",
    "unknown varint 10 bytes": "c03effffffffffffffffff017007",
    "unknown varint 11 bytes": "c03effffffffffffffffffff017007",
    "enum negative 10 bytes": "08ffffffffffffffffff017008",
    "string field 100 invalid utf8 skipped": "a20602fffe7009",
    "TimeReceived 7-byte varint (2^42)": "1080808080808001",
    "all three addresses 16 bytes": "3210" + "aa" * 16 + "3a10" + "bb" * 16 + "5a10" + "cc" * 16,
    "address 16 then 17 bytes": "3210" + "aa" * 16 + "3211" + "bb" * 17,
    "SamplerAddress 1 byte": "5a01ff",
    "ports and proto": "a00106a801bb03b001d0860370017801",
    "max values": "10ffffffffffffffffff01" "18ffffffffffffffffff01" "20ffffffff0f" "48ffffffffffffffffff01"
                  "50ffffffffffffffffff01" "70ffffffff0f" "78ffffffff0f" "a001ffffffff0f" "a801ffffffff0f"
                  "b001ffffffff0f" "f001ffffffff0f" "b002ffffffffffffffffff01",
}


def make_edge_cases():
    cases = []
    for name, hx in EDGE.items():
        b = bytes.fromhex(hx)
        cases.append({"name": name, "hex": hx, "expect": upb_decode(b)})
    json.dump({"source": "upb-protobuf 7.35.1 + pb-ext/flow.proto schema; see make_golden.py",
               "cases": cases}, open(os.path.join(HERE, "edge_cases.json"), "w"), indent=1)
    return cases


def rand_message(rng, i):
    """A FlowMessage with a wide mix of fields (encoded by upb with the FULL schema so
    non-projected fields, 2-byte tags and strings appear on the wire)."""
    F = schema.message_class("full")
    m = F()
    t = 1_600_000_200 + rng.randrange(0, 1500)
    m.TimeReceived = t
    m.TimeFlowStart = t - rng.randrange(0, 3)
    m.SamplingRate = rng.choice([0, 1, 1, 1000, 2**40])
    m.SequenceNum = i
    v6 = rng.random() < 0.5
    m.Etype = 0x86DD if v6 else 0x0800
    m.SrcAddr = bytes(rng.randrange(256) for _ in range(16 if v6 else 4))
    m.DstAddr = bytes(rng.randrange(256) for _ in range(16 if v6 else 4))
    m.Bytes = rng.choice([0, rng.randrange(1500), rng.randrange(2**40), 2**64 - 1 - rng.randrange(1000)])
    m.Packets = rng.choice([0, rng.randrange(100), rng.randrange(2**33)])
    m.SrcAS = rng.choice([0, 65000 + rng.randrange(4), rng.randrange(2**32)])
    m.DstAS = rng.choice([65000 + rng.randrange(4), 4294967295])
    m.Proto = rng.choice([0, 6, 17])
    m.SrcPort = rng.randrange(65536)
    m.DstPort = rng.choice([0, 53, 443, rng.randrange(65536)])
    if rng.random() < 0.3:
        m.SamplerAddress = bytes(rng.randrange(256) for _ in range(rng.choice([4, 16])))
    if rng.random() < 0.5:
        m.Type = rng.randrange(5)
        m.TimeFlowEnd = t
        m.InIf = rng.randrange(100)
        m.TCPFlags = rng.randrange(64)
        m.SrcMac = rng.randrange(2**48)
        m.SrcCountry = rng.choice(["", "US", "DE"])
        m.DstASDB = rng.randrange(2**32)
        m.NextHop = bytes(rng.randrange(256) for _ in range(4))
        m.HasMPLS = rng.random() < 0.5
    return m.SerializeToString()


def make_rollup_fixture():
    """2000 upb-encoded records -> expected flows_5m rows by a Python dict group-by
    (create.sh:64-110: Date=toDate(t), Timeslot=t-t%300, key + EType, sum/sum/count)."""
    rng = random.Random(2024)
    recs = [rand_message(rng, i) for i in range(2000)]
    agg = {}
    for r in recs:
        d = upb_decode(r)
        assert d is not None
        t = d["TimeReceived"] & 0xFFFFFFFF
        ts = t - t % 300
        k = (ts // 86400, ts, d["SrcAS"], d["DstAS"], d["EType"])
        b, p, c = agg.get(k, (0, 0, 0))
        agg[k] = ((b + d["Bytes"]) % 2**64, (p + d["Packets"]) % 2**64, c + 1)
    rows = [list(k) + list(v) for k, v in sorted(agg.items())]
    json.dump({"source": "upb-encoded (FULL schema) records; rows = python dict restatement of create.sh:64-110",
               "records_hex": [r.hex() for r in recs], "rows": rows},
              open(os.path.join(HERE, "rollup_2000.json"), "w"))


def make_readme_fixture():
    """The only result-bearing text of the reference: the sample outputs of README.md:155-161 (flows_raw) and
    README.md:180-183 (flows_5m after OPTIMIZE).  The table rows are copied VERBATIM from the README at
    generation time; the records are upb-encoded mocker-shaped FlowMessages (mocker.go:57-91) chosen so that the five
    printed flows_raw rows are among them and every (SrcAS,DstAS) group adds up to the printed flows_5m row.  (The
    README samples come from two different runs of an unseeded mocker, so the mapping of the five raw rows to groups
    is ours; what the fixture pins is the rendering: Date, Timeslot = toStartOfFiveMinute, DateTime narrowing,
    IPv6NumToString, the [EType] key, sums and counts.)"""
    import calendar
    lines = open(os.path.join(REF, "README.md")).read().split("\n")
    raw_rows = [l for l in lines[154:161] if l.startswith("\u2502")]
    agg_rows = [l for l in lines[179:183] if l.startswith("\u2502")]
    assert len(raw_rows) == 5 and len(agg_rows) == 3, (raw_rows, agg_rows)
    cell = lambda l: [c.strip() for c in l.strip("\u2502").split("\u2502")]
    F = schema.message_class("light")
    t38 = calendar.timegm((2020, 3, 22, 21, 26, 38))
    ip = lambda s: __import__("socket").inet_pton(__import__("socket").AF_INET6, s)
    raws = [cell(l) for l in raw_rows]  # Date, TimeReceived, Src, Dst, Bytes, Packets
    def rec(t, src, dst, b, pk, sa, da, seq):
        m = F()
        m.TimeReceived = t; m.TimeFlowStart = t; m.SamplingRate = 1; m.SequenceNum = seq
        m.SrcAddr = ip(src); m.DstAddr = ip(dst); m.Bytes = b; m.Packets = pk
        m.SrcAS = sa; m.DstAS = da; m.Etype = 0x86dd; m.SrcPort = 1000 + seq; m.DstPort = 2000 + seq
        return m.SerializeToString()
    R = {i: (t38 + (r[1].endswith(":39")), r[2], r[3], int(r[4]), int(r[5])) for i, r in enumerate(raws)}
    plan = [  # (group DstAS, [(raw row index | None, bytes, packets)])
        (65000, [(3, 0, 0), (2, 0, 0), (0, 0, 0), (None, 757, 6)]),
        (65001, [(1, 0, 0), (None, 800, 75), (None, 749, 72)]),
        (65002, [(4, 0, 0), (None, 1000, 50), (None, 1000, 50), (None, 1000, 55), (None, 1000, 60), (None, 697, 50)]),
    ]
    recs, seq, where = [], 0, {}
    for da, items in plan:
        for ri, b, pk in items:
            if ri is None:
                recs.append(rec(t38 + seq % 2, "2001:db8:0:1::%x" % (seq + 1), "2001:db8:0:1::%x" % (seq + 7), b, pk, 65001, da, seq))
            else:
                t, s_, d_, b_, p_ = R[ri]
                where[ri] = seq
                recs.append(rec(t, s_, d_, b_, p_, 65001, da, seq))
            seq += 1
    json.dump({"source": "README.md:155-161 and README.md:180-183 of the reference, table rows verbatim",
               "raw_row_record_index": [where[i] for i in range(5)],  # README raw rows 0..4 -> position in records_hex
               "readme_flows_raw": raw_rows, "readme_flows_5m": agg_rows,
               "records_hex": [r.hex() for r in recs]},
              open(os.path.join(HERE, "readme_samples.json"), "w"), ensure_ascii=False, indent=1)


SNIPS = [bytes.fromhex(x) for x in [
    "c33e", "c43e", "cb3e", "cc3e", "73", "74", "7001", "7205", "71", "75", "c03e05",
    "ffffffffffffffffff01", "8080808000", "f0ffffff0f", "3200", "3214" + "11" * 20, "a206", "00", "80",
    "ff", "0a", "12", "1a", "c23e03616263", "7b", "7c", "48ffffffffffffffffff7f", "1080808080808001"]]


def make_fuzz_fixture(n=20000):
    rng = random.Random(7)
    seeds = [rand_message(rng, i) for i in range(400)]
    seeds += [bytes.fromhex(h) for h in EDGE.values()]
    out = []
    for _ in range(n):
        r = bytearray(rng.choice(seeds))
        for _ in range(rng.randrange(0, 4)):
            op = rng.randrange(6)
            if op == 0 and r:
                r[rng.randrange(len(r))] = rng.randrange(256)
            elif op == 1 and r:
                r[rng.randrange(len(r))] ^= 1 << rng.randrange(8)
            elif op == 2:
                p = rng.randrange(len(r) + 1)
                r[p:p] = rng.choice(SNIPS)
            elif op == 3 and r:
                p = rng.randrange(len(r))
                del r[p:p + rng.randrange(1, 5)]
            elif op == 4:
                r = r[:rng.randrange(len(r) + 1)]
            else:
                p = rng.randrange(len(r) + 1)
                r[p:p] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 12)))
        out.append(bytes(r))
    status = np.array([0 if upb_decode(r) is not None else 1 for r in out], dtype=np.uint32)
    off = np.zeros(len(out) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in out])
    buf = np.frombuffer(b"".join(out), dtype=np.uint8)
    # expected columns for the OK records, as decoded by upb
    exp = {c: [] for c in schema.PROJECTED}
    for r, s in zip(out, status):
        d = upb_decode(r) if s == 0 else None
        for c in schema.PROJECTED:
            exp[c].append(d[c] if d else (("00" * 16) if schema.PROJECTED[c][1] == "b" else 0))
    cols = {}
    for c, (_n, kind, typ) in schema.PROJECTED.items():
        if kind == "b":
            cols["upb_" + c] = np.frombuffer(bytes.fromhex("".join(exp[c])), dtype=np.uint8).reshape(-1, 16)
        else:
            cols["upb_" + c] = np.array(exp[c], dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "fuzz_records.npz"), buf=buf, off=off, upb_status=status, **cols)
    print("fuzz fixture: %d records, %d bad, %d bytes" % (len(out), int(status.sum()), len(buf)))


if __name__ == "__main__":
    check_schema_against_reference()
    cases = make_edge_cases()
    print("edge cases:", len(cases), "bad:", sum(1 for c in cases if c["expect"] is None))
    make_rollup_fixture()
    make_readme_fixture()
    make_fuzz_fixture()
