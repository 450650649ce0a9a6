#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2c
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
B="--steps 10 --warmup 3 --cpu-sample 0 --no-verify --no-host-fed"
FA_TIMING_CLOSE=1 timeout 300 python bench.py $B > $OUT/bench_a.json 2> $OUT/bench_a.err
grep "flowagg close" $OUT/bench_a.err
FA_TUPLE=16 timeout 300 python bench.py $B > $OUT/bench_t16.json 2> $OUT/bench_t16.err
timeout 300 python bench.py $B > $OUT/bench_b.json 2> $OUT/bench_b.err
timeout 300 python bench.py $B --mode mocker > $OUT/bench_mocker.json 2> $OUT/bench_mocker.err
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r["dominant_kernel"]
    print("value %.4g  path %.4f ms frac %.4f | wtile %.4f ms frac %.4f | rest %.4f ms | close %.1f ms" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], r["avg_launch_ms"]-k["avg_launch_ms"], d["config"]["window_close_merge_ms"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
