#!/bin/bash
# after the last source edit: the suite, the default bench line, its hash-stamped traffic profile, config 3's streaming profiles
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05stamp
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|^FAILED|^E  " $OUT/pytest.log | head
timeout 900 bash tools/profile.sh r05 > $OUT/profile_default.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
for mode in exact candidates; do
  PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode $mode" timeout 900 bash tools/profile.sh r05_config3_stream_$mode > $OUT/profile_config3_$mode.log 2>&1
  timeout 1500 python tools/config3_run.py --topk-mode $mode > $OUT/config3_1B_$mode.json 2> $OUT/config3_1B_$mode.err; echo "config3 1B $mode rc=$?"
done
grep -A5 "calibrated HBM" $ROOT/gpurun_out/prof/r05/summary.txt
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; k=r["dominant_kernel"]
print("bench value %.4g path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | parity %s | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], d["parity"]["ok"], r["traffic"]))
PY
