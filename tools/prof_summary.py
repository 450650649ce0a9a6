#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (rocprofv3 rocpd SQLite databases) into a short
text summary: per-kernel average duration (kernel trace) and every collected counter per launch of
each kernel.  gfx950 note: FETCH_SIZE under-reports wide coalesced reads 2x (MI355X_MICROARCH.md
"HBM"), so the corrected figure is printed beside the raw one."""
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
                extra += "  | x2 gfx950 wide-read correction = %.1f MB" % (avg * 2 * 1024 / 1e6)
            print("%-56s launches=%-4d avg=%14.1f%s  avg_dur=%.1f us" % (name[:56], n, avg, extra, dur / 1e3))

# machine-readable HBM traffic of the two hot-path kernels (bench.py's roofline.traffic reads the copy
# committed under profiles/): reads = FETCH_SIZE x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md
# "HBM"), writes = WRITE_SIZE as reported (1:1 on coalesced dword stores: gen_len_kernel writes 4 B per
# record and reports exactly that).
import json

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
               else "probe_kernel" if "probe_kernel" in name else None)
        if key:
            traffic.setdefault(key, {})[counter] = avg * 1024.0
            traffic[key]["launches"] = n
            traffic[key]["avg_dur_us_under_pmc"] = dur / 1e3
for k, v in traffic.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["read_bytes_corrected"] = 2.0 * v["FETCH_SIZE"]
        v["traffic_bytes"] = 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
if traffic:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _pkg
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), tools/profile.sh",
                   "source_hash": _pkg.load().source_hash(),  # bench.py quotes these numbers only when it runs the same sources
                   "bench_args": os.environ.get("PROF_BENCH_ARGS", ""), "kernels": traffic}, f, indent=1)
