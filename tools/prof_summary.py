#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (rocprofv3 rocpd SQLite databases) into a short
text summary: per-kernel average duration (kernel trace) and every collected counter per launch of
each kernel.

FETCH_SIZE / WRITE_SIZE on gfx950, calibrated on known byte counts in this library's access patterns
(tools/micro/fetch_calib.hip, profiles/r05_fetch_calibration.json):
  * FETCH_SIZE counts 64 B per fabric read request.  A coalesced streaming read (4, 8 or 16 B per lane alike, and the
    LDS DMA) moves 128-byte lines and is counted at HALF its bytes (x0.500 measured): double it.  The offsets pattern
    (8-byte loads at a 4-byte stride) reads its array at x0.563: the array's bytes x 1.125 after doubling.
  * RANDOM requests are counted 1 : 1 at 64 B each, whatever the lane asked for: 16 + 8 B of a 32-byte slot = 64 B
    (x2.0 of the 32 B), a whole 64-byte line = 64 B (x1.0), an 8-byte counter = 64 B (x8) - also when the table sits in
    the Infinity Cache (32 MiB sketch: x7).  Doubling THESE over-states them 2x (round 4 did, for every kernel).
  * WRITE_SIZE is the traffic: coalesced stores and 64-byte store units 1 : 1, a single 8-byte store costs a 32-byte
    sector (x4).
So per kernel: streaming kernels (aggregation walks, generators, table scans) FETCH x 2; the ingest kernels mix a
stream of known size (wire bytes + offsets, from the profiled command's own JSON line) with random requests:
calibrated read = 2 x (the stream's share of the counter) + 1 x (the rest)."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]


def db(sub):
    f = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


t = db("trace")
if t:
    print("== kernel trace: rocprofv3 --kernel-trace --stats (view top_kernels; durations in us) ==")
    print("%-72s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-72s %6d %12.1f %12.2f %6.2f%%" % (name[:72], calls, total, avg, pct))

for sub in sorted(os.listdir(out)):
    if not sub.startswith("pmc_") or not os.path.isdir(os.path.join(out, sub)):
        continue
    d = db(sub)
    if not d:
        print("== %s: no database (see %s.log) ==" % (sub, sub))
        continue
    try:
        counters = [r[0] for r in d.execute("select distinct counter_name from counters_collection")]
    except sqlite3.Error as e:
        print("== %s: %s ==" % (sub, e))
        continue
    for counter in counters:
        unit = " (KiB as reported)" if counter in ("FETCH_SIZE", "WRITE_SIZE") else ""
        print("== rocprofv3 --pmc %s, per launch%s ==" % (counter, unit))
        q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
             "where counter_name=? group by kernel_name order by sum(value) desc limit 6")
        for name, n, avg, dur in d.execute(q, (counter,)):
            extra = ""
            if counter in ("FETCH_SIZE", "WRITE_SIZE"):
                extra = " = %9.1f MB" % (avg * 1024 / 1e6)
            if counter == "FETCH_SIZE":
                extra += "  | x2 if all of it streams = %.1f MB" % (avg * 2 * 1024 / 1e6)
            print("%-56s launches=%-4d avg=%14.1f%s  avg_dur=%.1f us" % (name[:56], n, avg, extra, dur / 1e3))

# machine-readable HBM traffic of the hot-path kernels (bench.py's roofline.traffic reads the copy committed under
# profiles/), by the calibration in this file's header.
import json
import re


def profiled_stream():
    """(wire bytes, records) per ingest launch of the profiled command, from the JSON line it printed (trace.log)."""
    for log in ("trace.log", "pmc_fetch.log"):
        path = os.path.join(out, log)
        if not os.path.exists(path):
            continue
        lines = [l for l in open(path, errors="replace") if l.startswith("{")]
        if not lines:
            continue
        try:
            d = json.loads(lines[-1])
        except ValueError:
            continue
        if "wire_bytes" in d and d.get("launches"):  # tools/config3_run.py
            return d["wire_bytes"] / d["launches"], d["records"] / d["launches"]
        cfg = d.get("config") or {}
        if cfg.get("wire_bytes_per_gpu_per_step") and cfg.get("launches_per_step"):  # bench.py
            return cfg["wire_bytes_per_gpu_per_step"] / cfg["launches_per_step"], cfg["records_per_gpu_per_step"] / cfg["launches_per_step"]
    return None, None


STREAMING = ("agg8_kernel", "agg_kernel", "cms_agg_kernel", "wagg_kernel", "cand_bits_kernel", "cand_scan_kernel")
wire, recs = profiled_stream()
traffic = {}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    d = db(sub)
    if not d:
        continue
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
         "where counter_name=? group by kernel_name")
    for name, n, avg, dur in d.execute(q, (counter,)):
        key = ("wtile_kernel" if "wtile_kernel" in name else "tile_kernel" if "tile_kernel<0" in name
               else "cms_agg_kernel" if "cms_agg_kernel" in name else "wagg_kernel" if "wagg_kernel" in name else "agg8_kernel" if "agg8_kernel" in name
               else "agg_kernel" if "fa::agg_kernel" in name else "deferred_kernel" if "deferred_kernel<0" in name
               else "cand_bits_kernel" if "cand_bits_kernel" in name else "cand_scan_kernel" if "cand_scan_kernel" in name
               else "probe_kernel" if "probe_kernel" in name else None)
        if key:
            traffic.setdefault(key, {})[counter] = avg * 1024.0
            traffic[key]["launches"] = n
            traffic[key]["avg_dur_us_under_pmc"] = dur / 1e3
for k, v in traffic.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    raw = v["FETCH_SIZE"]
    v["read_bytes_if_all_streaming"] = 2.0 * raw
    if k in STREAMING:
        v["read_bytes_calibrated"] = 2.0 * raw
        v["read_model"] = "streaming walk: FETCH_SIZE x 2"
    elif k in ("wtile_kernel", "tile_kernel") and wire:
        stream_raw = min(raw, (wire + 1.125 * 4.0 * recs) / 2.0)  # what the wire bytes and the offsets show up as in the counter
        v["read_bytes_calibrated"] = 2.0 * stream_raw + (raw - stream_raw)
        v["read_model"] = ("stream (%.1f MB wire + %.1f MB offsets x 1.125) at x2, the remaining %.1f MB of the counter = random 64-byte requests at x1"
                           % (wire / 1e6, 4.0 * recs / 1e6, (raw - stream_raw) / 1e6))
    else:
        v["read_bytes_calibrated"] = raw
        v["read_model"] = "uncalibrated pattern: FETCH_SIZE as reported (random requests are counted 1 : 1)"
    v["read_bytes_corrected"] = v["read_bytes_calibrated"]
    v["traffic_bytes"] = v["read_bytes_calibrated"] + v["WRITE_SIZE"]
    v["traffic_bytes_if_all_streaming"] = 2.0 * raw + v["WRITE_SIZE"]
if traffic:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _pkg
    tot = sum(v.get("traffic_bytes", 0.0) for v in traffic.values())
    print("== calibrated HBM traffic per launch (header of tools/prof_summary.py) ==")
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1].get("traffic_bytes", 0.0)):
        if "traffic_bytes" in v:
            print("%-18s read %8.1f MB + written %8.1f MB = %8.1f MB   (%s)" % (k, v["read_bytes_calibrated"] / 1e6, v["WRITE_SIZE"] / 1e6, v["traffic_bytes"] / 1e6, v["read_model"]))
    if wire:
        print("path %.1f MB per launch against %.1f MB of wire bytes: %.3fx" % (tot / 1e6, wire / 1e6, tot / wire))
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/profile.sh; calibration: tools/micro/fetch_calib.hip",
                   "source_hash": _pkg.load().source_hash(),  # bench.py quotes these numbers only when it runs the same sources
                   "bench_args": os.environ.get("PROF_BENCH_ARGS", ""), "wire_bytes_per_launch": wire, "records_per_launch": recs,
                   "path_traffic_bytes": tot, "kernels": traffic}, f, indent=1)
