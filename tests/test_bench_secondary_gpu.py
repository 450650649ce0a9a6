"""bench.py's `secondary` blocks at small sizes: the blocks the driver's one bench line carries beside the headline (BASELINE configs 3 and
5, the 8-context group close, the C++ consumer) each return `ok` with every parity boolean true - the same code paths as the 200 M /
100 M / 64 M-record runs, in seconds."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def env(gpu_lib, fa, po):
    import torch
    import bench
    return bench, torch, torch.device("cuda", 0)


@pytest.fixture(scope="module")
def zipf_ref(env, po):
    bench = env[0]
    return bench._Zipf3(po, 6_000_000, L=18, wl2=16)


@pytest.mark.parametrize("cand", [False, True])
def test_secondary_config3(env, fa, po, zipf_ref, cand):
    bench, torch, dev = env
    out = bench.sec_config3(fa, po, torch, dev, zipf_ref, cand, chunk=1_000_000)
    assert out["ok"], out
    assert out["sketches_bit_exact"] and out["top100_equals_ranking_of_the_whole_universe"] and out["rollup_count_equals_records"]
    assert out["launches"] == 6 and 0 < out["frac"] < 1 and len(out["topk100_ms_per_call"]) == 6


def test_secondary_config5(env, fa, po):
    bench, torch, dev = env
    out = bench.sec_config5(fa, po, torch, dev, n=6_000_000, chunk=1_000_000, span=1800)
    assert out["ok"], out
    assert out["flows_5m_aligned_windows_bit_exact"] and out["sliding_window_bit_exact"] and out["app_windows_bit_exact_by_linear_checksum_and_strict_order"]
    assert len(out["close_app_window_ms"]) == 6 and out["app_rows_left_after_all_closes"] == 0 and out["app_rows"] > 5_000_000


def test_secondary_group8(env, fa, po, zipf_ref):
    bench, torch, dev = env
    out = bench.sec_group8(fa, po, torch, dev, zipf_ref, members=8, chunk=250_000, cand=True)
    assert out["ok"], out
    for k in ("flows_5m_merged_equals_oracle_rollup_of_all_partitions", "closed_window_equals_its_read", "app_windows_bit_exact_by_linear_checksum_and_strict_order",
              "merged_sketches_bit_exact", "top100_equals_ranking_of_the_whole_universe"):
        assert out[k], k
    assert out["members"] == 8 and len(out["read_app_window_partitioned_ms"]) == 6


def test_secondary_host_consume(env, fa, po):
    bench, torch, dev = env
    out = bench.sec_host_consume(fa, po, torch, dev, n=2_000_000, nparts=4, flush_count=1000)
    assert out["ok"], out
    assert out["insert_count_equals_records"] and out["rowbinary_equals_oracle_rollup_of_all_partitions"]
    assert out["copied_bytes"] == 0 and out["batches"] == 4 and len(out["phases_by_partition"]) == 4  # (one 36 MB batch per claim, handed over in place)


def test_run_secondary_isolates_a_failing_block(env, fa, po, monkeypatch):
    """A block that raises is reported as {"ok": false, "error": ...}; the others still run, the headline is never at stake."""
    import types
    bench, torch, dev = env

    def boom(*a, **kw):
        raise RuntimeError("simulated")
    monkeypatch.setattr(bench, "sec_config5", boom)
    monkeypatch.setattr(bench, "sec_host_consume", lambda *a, **kw: {"ok": True})
    monkeypatch.setattr(bench, "sec_config3", lambda *a, **kw: {"ok": True})
    monkeypatch.setattr(bench, "sec_group8", lambda *a, **kw: {"ok": True})
    monkeypatch.setattr(bench, "_Zipf3", lambda po_, n: types.SimpleNamespace(seconds=0.0))
    args = types.SimpleNamespace(secondary_budget=100, secondary_host_records=1, secondary_config5_records=1, secondary_config3_records=1)
    sec = bench.run_secondary(fa, po, torch, dev, args)
    assert sec["ok"] is False and sec["config5"]["ok"] is False and "simulated" in sec["config5"]["error"]
    assert sec["config3_exact"]["ok"] and sec["group8"]["ok"] and sec["host_consume"]["ok"]
