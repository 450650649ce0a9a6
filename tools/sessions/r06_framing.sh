#!/bin/bash
# Device-side framing (offsets == NULL): rate beside the offsets path + per-kernel times; then the framing tests.
OUT=gpurun_out/r06_framing${TAG}; mkdir -p $OUT
export TMPDIR=/tmp
python tools/framing_rate.py > $OUT/rate.json 2> $OUT/rate.err; echo "rate rc=$?"; cat $OUT/rate.json | cut -c1-700
FA_VERBOSE=1 python tools/framing_rate.py 2>&1 >/dev/null | grep framing | sort | uniq -c | head -5
rocprofv3 --kernel-trace --stats -d $OUT/prof -o fr -- python tools/framing_rate.py > $OUT/prof.log 2>&1
python - <<'PY' $OUT
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/prof/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(out + "/kernel_stats.txt", "w") as w:
        for r in rows[:14]:
            line = "%-70s calls %6s  total_us %12.1f  avg_us %10.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3)
            print(line); w.write(line + "\n")
PY
timeout 900 python -m pytest tests/test_wide_log_framing_gpu.py tests/test_gpu_parity.py -q -m gpu -x -p no:cacheprovider -k "framing or chain or decode" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
