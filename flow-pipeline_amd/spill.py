"""Columnar spill of the decoded projection: `flows_raw`-equivalent Parquet files.

The dashboards of the reference scan ``flows_raw`` (``viz-ch.json:74-604``); with the
ClickHouse Kafka-engine table and its materialized views gone, the raw rows
have no home unless the stage writes them somewhere.  ``spill_parquet`` turns the rows
``fa_decode`` returns (the 15 projected columns, ``compose/clickhouse/create.sh:7-27``) into
one Parquet file with the column list and types of ``flows_raw`` (``create.sh:36-62``):

    Date Date, TimeReceived DateTime, TimeFlowStart DateTime, SequenceNum UInt32,
    SamplingRate UInt64, SamplerAddress/SrcAddr/DstAddr FixedString(16), SrcAS, DstAS, EType,
    Proto, SrcPort, DstPort UInt32, Bytes, Packets UInt64

``Date = toDate(TimeReceived)`` (``create.sh:66``) and the UInt64 -> DateTime narrowing of the two
time columns are applied here as ClickHouse applies them in ``flows_raw_view``; malformed records
(status != 0) are dropped (``inserter/inserter.go:125-126``).  ClickHouse loads such a file with
``INSERT INTO flows_raw FORMAT Parquet``; offline checks read it with pyarrow / pandas.
Host-side Python only (pyarrow): nothing here is on the hot path.
"""
from __future__ import annotations

import numpy as np

COLUMNS = ["Date", "TimeReceived", "TimeFlowStart", "SequenceNum", "SamplingRate", "SamplerAddress", "SrcAddr",
           "DstAddr", "SrcAS", "DstAS", "EType", "Proto", "SrcPort", "DstPort", "Bytes", "Packets"]


def to_arrow(rows: np.ndarray):
    """rows: structured array with the fields of FLOW_ROW_DTYPE (``status`` optional) -> pyarrow.Table."""
    import pyarrow as pa
    if "status" in rows.dtype.names:
        rows = rows[rows["status"] == 0]
    t_recv = (rows["time_received"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)   # DateTime (create.sh:39)
    t_flow = (rows["time_flow_start"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)  # DateTime (create.sh:40)

    def fixed16(col):
        flat = np.ascontiguousarray(rows[col]).reshape(-1)
        return pa.FixedSizeBinaryArray.from_buffers(pa.binary(16), len(rows), [None, pa.py_buffer(flat.tobytes())])

    cols = {
        "Date": pa.array((t_recv // np.uint32(86400)).astype(np.int32), type=pa.date32()),  # toDate, UTC (create.sh:66)
        # DateTime = UInt32 seconds; Parquet has no second-resolution timestamp (pyarrow would widen to ms), and
        # ClickHouse reads a UInt32 Parquet column into a DateTime column as is
        "TimeReceived": pa.array(t_recv, type=pa.uint32()),
        "TimeFlowStart": pa.array(t_flow, type=pa.uint32()),
        "SequenceNum": pa.array(rows["sequence_num"], type=pa.uint32()),
        "SamplingRate": pa.array(rows["sampling_rate"], type=pa.uint64()),
        "SamplerAddress": fixed16("sampler_address"),
        "SrcAddr": fixed16("src_addr"),
        "DstAddr": fixed16("dst_addr"),
        "SrcAS": pa.array(rows["src_as"], type=pa.uint32()),
        "DstAS": pa.array(rows["dst_as"], type=pa.uint32()),
        "EType": pa.array(rows["etype"], type=pa.uint32()),
        "Proto": pa.array(rows["proto"], type=pa.uint32()),
        "SrcPort": pa.array(rows["src_port"], type=pa.uint32()),
        "DstPort": pa.array(rows["dst_port"], type=pa.uint32()),
        "Bytes": pa.array(rows["bytes"], type=pa.uint64()),
        "Packets": pa.array(rows["packets"], type=pa.uint64()),
    }
    return pa.table([cols[c] for c in COLUMNS], names=COLUMNS)


def spill_parquet(rows: np.ndarray, path: str, compression="zstd") -> int:
    """Writes one flows_raw-shaped Parquet file; returns the number of rows written."""
    import pyarrow.parquet as pq
    table = to_arrow(rows)
    pq.write_table(table, path, compression=compression)
    return table.num_rows
