"""The rollup's SQL reading, engine-executed (CPU test).

ClickHouse cannot run in this image, so `oracle/flow_oracle.c` restates compose/clickhouse/create.sh:64-68
(flows_raw_view) and :92-110 (flows_5m_view) by hand.  This test takes the two SELECTs themselves, translated
MECHANICALLY into the SQL dialect that is here (the bundled sqlite3), runs them over the decoded records of the
committed fixture tests/golden/rollup_2000.json, and asserts that the engine's result equals the oracle's rollup
(`fo_rollup`) and the fixture's rows.  It does not make parity green - nothing in this image can - but the GROUP BY /
sum / count() semantics are executed by an SQL engine instead of being re-read by us.

Translation table (everything else is the reference's text, identifier for identifier):
    toDate(TimeReceived)               -> TimeReceived / 86400      (UTC days; parity domain 65536 <= t < 2^32)
    toStartOfFiveMinute(TimeReceived)  -> TimeReceived - TimeReceived % 300
    UInt64 -> DateTime narrowing       -> TimeReceived & 4294967295 (flows_raw.TimeReceived is DateTime, create.sh:39)
    [EType] AS `ETypeMap.EType`        -> EType AS ETypeMap_EType   (a one-element array groups like its element)
    sum(UInt64)                        -> sqlite integers are signed 64-bit and sum() refuses to wrap, so every UInt64
                                          column travels as two 32-bit halves that are summed apart and recombined
                                          mod 2^64 - the wrap-around ClickHouse's UInt64 sum has.
"""
import json
import os
import sqlite3

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REF_CREATE = "/root/reference/compose/clickhouse/create.sh"

# compose/clickhouse/create.sh:64-68, as the reference has it
CH_FLOWS_RAW_VIEW = """CREATE MATERIALIZED VIEW IF NOT EXISTS flows_raw_view TO flows_raw
    AS SELECT
        toDate(TimeReceived) AS Date,
        *
       FROM flows;"""
# compose/clickhouse/create.sh:92-110
CH_FLOWS_5M_VIEW = """CREATE MATERIALIZED VIEW IF NOT EXISTS flows_5m_view TO flows_5m
    AS
        SELECT
            Date,
            toStartOfFiveMinute(TimeReceived) AS Timeslot,
            SrcAS,
            DstAS,

            [EType] AS \\`ETypeMap.EType\\`,
            [Bytes] AS \\`ETypeMap.Bytes\\`,
            [Packets] AS \\`ETypeMap.Packets\\`,
            [Count] AS \\`ETypeMap.Count\\`,

            sum(Bytes) AS Bytes,
            sum(Packets) AS Packets,
            count() AS Count

        FROM flows_raw
        GROUP BY Date, Timeslot, SrcAS, DstAS, \\`ETypeMap.EType\\`;"""

# the same two statements in sqlite's dialect (translation table in the module docstring)
SQLITE_FLOWS_RAW_VIEW = """CREATE VIEW flows_raw AS SELECT
        (TimeReceived & 4294967295) / 86400 AS Date,
        (TimeReceived & 4294967295) AS TimeReceived,
        SrcAS, DstAS, EType, BytesLo, BytesHi, PacketsLo, PacketsHi
       FROM flows;"""
SQLITE_FLOWS_5M_VIEW = """CREATE VIEW flows_5m AS
        SELECT
            Date,
            TimeReceived - TimeReceived % 300 AS Timeslot,
            SrcAS,
            DstAS,

            EType AS ETypeMap_EType,

            sum(BytesLo) AS BytesLo, sum(BytesHi) AS BytesHi,
            sum(PacketsLo) AS PacketsLo, sum(PacketsHi) AS PacketsHi,
            count() AS Count

        FROM flows_raw
        GROUP BY Date, Timeslot, SrcAS, DstAS, ETypeMap_EType;"""


def _squash(s):
    return " ".join(s.split())


def test_embedded_clickhouse_text_is_the_references():
    """The ClickHouse statements quoted above are create.sh's (checked wherever the reference is present)."""
    if not os.path.exists(REF_CREATE):
        import pytest
        pytest.skip("reference not on this box")
    text = _squash(open(REF_CREATE).read())
    assert _squash(CH_FLOWS_RAW_VIEW) in text
    assert _squash(CH_FLOWS_5M_VIEW) in text


def _fixture_columns(po):
    fx = json.load(open(os.path.join(GOLDEN, "rollup_2000.json")))
    recs = [bytes.fromhex(h) for h in fx["records_hex"]]
    off = np.zeros(len(recs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in recs])
    buf = np.frombuffer(b"".join(recs), dtype=np.uint8)
    rows, status = po.decode_batch(buf, off, 0)
    assert not status.any()
    return fx, buf, off, rows


def test_flows_5m_select_in_sqlite_equals_oracle_rollup(po):
    fx, buf, off, cols = _fixture_columns(po)
    db = sqlite3.connect(":memory:")
    # the `flows` table (create.sh:5-27), the columns the two views read; UInt64 sums as two halves
    db.execute("CREATE TABLE flows (TimeReceived INTEGER, SrcAS INTEGER, DstAS INTEGER, EType INTEGER, "
               "BytesLo INTEGER, BytesHi INTEGER, PacketsLo INTEGER, PacketsHi INTEGER)")
    m32 = 0xFFFFFFFF
    db.executemany("INSERT INTO flows VALUES (?,?,?,?,?,?,?,?)", [
        (int(t), int(sa), int(da), int(et), int(b) & m32, int(b) >> 32, int(p) & m32, int(p) >> 32)
        for t, sa, da, et, b, p in zip(cols["time_received"], cols["src_as"], cols["dst_as"], cols["etype"], cols["bytes"], cols["packets"])])
    db.execute(SQLITE_FLOWS_RAW_VIEW)
    db.execute(SQLITE_FLOWS_5M_VIEW)
    # (SummingMergeTree ORDER BY (Date, Timeslot, SrcAS, DstAS, ETypeMap.EType), create.sh:90: the order a final read shows)
    got = []
    for d, ts, sa, da, et, bl, bh, pl, ph, c in db.execute(
            "SELECT Date, Timeslot, SrcAS, DstAS, ETypeMap_EType, BytesLo, BytesHi, PacketsLo, PacketsHi, Count FROM flows_5m "
            "ORDER BY Date, Timeslot, SrcAS, DstAS, ETypeMap_EType"):
        got.append([d, ts, sa, da, et, ((bh << 32) + bl) % 2**64, ((ph << 32) + pl) % 2**64, c])
    # 1. the engine's rows are the fixture's rows (a Python dict group-by wrote those)
    assert got == fx["rows"]
    # 2. ... and the C oracle's rollup of the same wire bytes
    r = po.Rollup(300)
    assert r.ingest(buf, off, 0) == 0
    want = [[int(x[f]) for f in ("date", "timeslot", "src_as", "dst_as", "etype", "bytes", "packets", "count")] for x in r.rows()]
    assert got == want
    # the fixture exercises what the translation table claims: wrapping sums and times beyond one window
    exact = {}
    for t, sa, da, et, b in zip(cols["time_received"], cols["src_as"], cols["dst_as"], cols["etype"], cols["bytes"]):
        k = (int(t) - int(t) % 300, int(sa), int(da), int(et))
        exact[k] = exact.get(k, 0) + int(b)
    assert any(v >= 2**64 for v in exact.values()) and len({x[1] for x in got}) >= 5
