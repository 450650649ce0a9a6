#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory (rocprofv3 rocpd SQLite databases)
into a short text summary: per-kernel average duration (kernel trace) and
FETCH_SIZE / WRITE_SIZE per launch of each kernel.  gfx950 note: FETCH_SIZE
under-reports wide coalesced reads 2x (MI355X_MICROARCH.md "HBM"), so the corrected
figure is printed beside the raw one."""
import glob
import os
import sqlite3
import sys

out = sys.argv[1]


def db(sub):
    f = glob.glob(os.path.join(out, sub, "*.db"))
    return sqlite3.connect(f[0]) if f else None


t = db("trace")
if t:
    print("== kernel trace: rocprofv3 --kernel-trace --stats (view top_kernels; durations in us) ==")
    print("%-72s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in t.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-72s %6d %12.1f %12.2f %6.2f%%" % (name[:72], calls, total, avg, pct))

for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    d = db(sub)
    if not d:
        continue
    print("== rocprofv3 --pmc %s (per launch, KiB as reported) ==" % counter)
    q = ("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
         "where counter_name=? group by kernel_name order by sum(value) desc limit 8")
    for name, n, avg, dur in d.execute(q, (counter,)):
        extra = ""
        if counter == "FETCH_SIZE":
            extra = "  | x2 gfx950 wide-read correction = %.1f MB" % (avg * 2 * 1024 / 1e6)
        print("%-60s launches=%-4d avg=%12.1f KiB = %9.1f MB  avg_dur=%.1f us%s" % (
            name[:60], n, avg, avg * 1024 / 1e6, dur / 1e3, extra))
