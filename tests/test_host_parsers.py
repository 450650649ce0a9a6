"""CPU tests of the device parsers' host instantiations (wire.cuh compiles for host and device)
and of the host-side arithmetic the kernels rely on.  No GPU needed."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_fuzz_against_oracle(po):
    """parse_canon / parse_fast may only answer "sure" with exactly the oracle's columns;
    parse_generic must agree with the oracle on every input (tests/host_parsers.hip)."""
    po.build()
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_parsers")
    src = os.path.join(ROOT, "tests", "host_parsers.hip")
    deps = [src, os.path.join(ROOT, "flow-pipeline_amd", "csrc", "wire.cuh")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call([
            "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-value", "-o", exe, src,
            "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    res = subprocess.run([exe, "100000"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = {l.split("cases=")[0].strip(): l for l in res.stdout.splitlines() if "cases=" in l}
    # the fast tier must accept plain generator output and canonical records with small values
    for name in ("generator (4 modes)", "random schema, canonical small"):
        f = dict(kv.split("=") for kv in lines[name].split() if "=" in kv)
        assert f["canon_sure"] == f["cases"] and f["full_sure"] == f["cases"] and f["FAIL"] == "0", lines[name]
    # the template walks take every record of the producer they are compiled for (mocker.go's field list: the mocker,
    # ASPAIRS, ZIPF and DISTINCT generators; GoFlow's 33 fields: both)
    f = dict(kv.split("=") for kv in lines["generator (4 modes)"].split() if "=" in kv)
    assert f["tmpl_mocker_sure"] == f["cases"] and f["tmpl_goflow_sure"] == f["cases"], lines["generator (4 modes)"]
    f = dict(kv.split("=") for kv in lines["generator (goflow)"].split() if "=" in kv)
    assert f["tmpl_goflow_sure"] == f["cases"] and f["tmpl_mocker_sure"] == "0", lines["generator (goflow)"]
    # GoFlow's field list with short values of every width: the template walk takes every record whose values fit (about a quarter)
    f = dict(kv.split("=") for kv in lines["goflow fields, every width"].split() if "=" in kv)
    assert int(f["tmpl_goflow_sure"]) * 5 >= int(f["cases"]) and f["FAIL"] == "0", lines["goflow fields, every width"]
    # descending field order: no ordered walk takes a record, the learnt order takes every one (the first is what it learns from)
    f = dict(kv.split("=") for kv in lines["generator (reversed)"].split() if "=" in kv)
    assert f["canon_sure"] == "0" and f["full_sure"] == "0" and f["fast_sure"] == f["cases"] and f["FAIL"] == "0", lines["generator (reversed)"]
    assert int(f["seq_sure"]) >= int(f["cases"]) - 20 and 1 <= int(f["seq_learnt"]) <= 10, lines["generator (reversed)"]
    # the 67-field producer (pb-ext/flow.pb.go:57-147): the FULL canonical walk must take every record
    for name in ("generator (goflow)", "67-field, canonical small"):
        f = dict(kv.split("=") for kv in lines[name].split() if "=" in kv)
        assert f["full_sure"] == f["cases"] and f["canon_sure"] == "0" and f["FAIL"] == "0", lines[name]


def test_tuple_formats_round_trip_and_balance():
    """table.cuh scatter-sink tuples compiled for the host: wide / compact round trips, fits() is exactly the
    documented value range, the compact format's partition bijection recovers SrcAS[7:0], partitions stay balanced."""
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_tuples")
    src = os.path.join(ROOT, "tests", "host_tuples.hip")
    deps = [src, os.path.join(ROOT, "flow-pipeline_amd", "csrc", "table.cuh")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", exe, src])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_packed_sort_keys_keep_the_order():
    """rowplan.cuh (window close: the radix sort runs over the bits that differ inside the row set only): ordering by the
    packed words == ordering by the full key, for random masks, widths and duplicates (tests/host_rowplan.hip)."""
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_rowplan")
    src = os.path.join(ROOT, "tests", "host_rowplan.hip")
    deps = [src, os.path.join(ROOT, "flow-pipeline_amd", "csrc", "rowplan.cuh")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", exe, src])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_sketch_definition_kernel_headers_equal_oracle(po):
    """sinks.cuh's sketch definition (cms_hash2 / cms_key / cms_column / cms_low, host-compiled) == oracle/flow_oracle.c
    fo_cms_column for every width, and the partition invariants of the scatter sink (tests/host_sketch.hip)."""
    po.build()
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "host_sketch")
    src = os.path.join(ROOT, "tests", "host_sketch.hip")
    csrc = os.path.join(ROOT, "flow-pipeline_amd", "csrc")
    deps = [src, os.path.join(ROOT, "oracle", "liboracle.so")] + [os.path.join(csrc, f) for f in ("sinks.cuh", "table.cuh", "wide.cuh", "wire.cuh")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        subprocess.check_call([
            "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-Wno-unused-value", "-o", exe, src,
            "-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    res = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("OK"), res.stdout + res.stderr


def test_time_bucket_reciprocal_is_exact():
    """sinks.cuh time_bucket(): floor(double(t) * (1/g)(1+2^-40)) == t // g for every u32 t.
    Checked at every multiple of g (+-1) near the ends of the range and on random t."""
    rng = np.random.default_rng(1)
    grans = [g for g in range(60, 86401) if 86400 % g == 0]
    assert 300 in grans and 60 in grans
    for g in grans:
        c = (1.0 / g) * (1.0 + 1.0 / 1099511627776.0)
        q = np.concatenate([np.arange(0, 2000, dtype=np.uint64), (2**32 - 1) // g - np.arange(0, 2000, dtype=np.uint64),
                            rng.integers(0, (2**32 - 1) // g, 4000, dtype=np.uint64)])
        for d in (-1, 0, 1, g - 1):
            t = q * np.uint64(g) + np.uint64(d % g if d >= 0 else 0)
            if d == -1:
                t = np.where(q > 0, q * np.uint64(g) - np.uint64(1), np.uint64(0))
            t = t[t < 2**32]
            got = np.floor(t.astype(np.float64) * c).astype(np.uint64)
            assert np.array_equal(got, t // np.uint64(g)), g
        t = rng.integers(0, 2**32, 20000, dtype=np.uint64)
        assert np.array_equal(np.floor(t.astype(np.float64) * c).astype(np.uint64), t // np.uint64(g)), g
