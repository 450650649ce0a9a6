#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2l
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
B="--steps 5 --warmup 2 --cpu-sample 0 --no-verify --no-host-fed"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
EXTRA="" run default FA_X=1
EXTRA="--mode zipf --records 50000000 --chunk 16666667" run ks1zipf FA_X=1
EXTRA="--mode zipf --key-sets 7 --records 50000000 --chunk 16666667" run ks7 FA_X=1
EXTRA="--mode mocker" run mocker FA_X=1
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r["dominant_kernel"]
    print("value %.4g  path %.4f ms frac %.4f | wtile %.4f ms | rest %.4f ms" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], r["avg_launch_ms"]-k["avg_launch_ms"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
