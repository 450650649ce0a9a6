"""flows_raw-equivalent Parquet spill: schema of compose/clickhouse/create.sh:36-62, and the
dashboard's raw-table queries (viz-ch.json:74,358) answered from the file."""
import numpy as np
import pytest


def _check_file(fa, po, rows, status, path):
    import pyarrow.parquet as pq
    t = pq.read_table(path)
    assert t.column_names == fa.spill.COLUMNS
    ok = rows[status == 0]
    assert t.num_rows == len(ok)
    assert str(t.schema.field("Date").type) == "date32[day]" and str(t.schema.field("TimeReceived").type) == "uint32"
    assert str(t.schema.field("SrcAddr").type) == "fixed_size_binary[16]" and str(t.schema.field("Bytes").type) == "uint64"
    d = t.to_pydict()
    assert d["SrcAddr"][:50] == [bytes(x) for x in ok["src_addr"][:50]]
    assert np.array_equal(np.array(d["Bytes"], dtype=np.uint64), ok["bytes"])
    assert np.array_equal(np.array(d["DstPort"], dtype=np.uint32), ok["dst_port"])
    tr = t.column("TimeReceived").to_numpy().astype(np.uint64)
    assert np.array_equal(tr, ok["time_received"] & np.uint64(0xFFFFFFFF))
    days = t.column("Date").to_numpy().astype("datetime64[D]").astype(np.int64)
    assert np.array_equal(days, (tr // np.uint64(86400)).astype(np.int64))
    # viz-ch.json:74 on the file: GROUP BY toStartOfMinute(TimeFlowStart) -> sum(Bytes*SamplingRate)
    tf = t.column("TimeFlowStart").to_numpy().astype(np.uint64)
    with np.errstate(over="ignore"):
        w = np.array(d["Bytes"], dtype=np.uint64) * np.array(d["SamplingRate"], dtype=np.uint64)
    minute = tf - tf % np.uint64(60)
    want = po.minute_series(rows, status)
    for m, ww in zip(want["minute"], want["weight"]):
        assert int(w[minute == np.uint64(m)].sum(dtype=np.uint64)) == int(ww)


def test_spill_schema_and_queries_from_oracle_rows(fa, po, tmp_path):
    n = 5000
    gp = po.gen_params(mode=po.GEN_ZIPF, framed=1, seed=8, n_total=n, span_secs=300)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    status[::97] = 1  # pretend some records were malformed: they must not reach the file
    x = np.zeros(n, dtype=fa.FLOW_ROW_DTYPE)
    for c in rows.dtype.names:
        if c in x.dtype.names:
            x[c] = rows[c]
    x["status"] = status
    path = str(tmp_path / "flows_raw.parquet")
    assert fa.spill.spill_parquet(x, path) == int((status == 0).sum())
    _check_file(fa, po, rows, status, path)


@pytest.mark.gpu
def test_spill_of_gpu_decoded_rows(gpu_lib, fa, po, tmp_path):
    n = 20000
    gp = po.gen_params(mode=po.GEN_ASPAIRS, framed=1, seed=9, n_total=n, span_secs=600)
    buf, off = po.gen_records(gp, 0, n)
    rows, status = po.decode_batch(buf, off, 1)
    with fa.FlowAgg(framed=True) as agg:
        got = agg.decode(buf, off)
    path = str(tmp_path / "flows_raw.parquet")
    assert fa.spill.spill_parquet(got, path) == n
    _check_file(fa, po, rows, status, path)
