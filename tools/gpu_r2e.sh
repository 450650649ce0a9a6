#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2e
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
B="--steps 10 --warmup 3 --cpu-sample 0 --no-verify --no-host-fed"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
EXTRA="" run c33 FA_TIMING_CLOSE=1
grep "flowagg close" $OUT/bench_c33.err
EXTRA="--chunk 16666667" run c16 FA_X=1
EXTRA="" run c33b FA_X=1
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r["dominant_kernel"]
    print("value %.4g  path %.4f ms frac %.4f | wtile %.4f ms frac %.4f | rest %.4f ms | close %.1f ms | launches/step %d" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], r["avg_launch_ms"]-k["avg_launch_ms"], d["config"]["window_close_merge_ms"], d["config"]["launches_per_step"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
