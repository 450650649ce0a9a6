#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2f
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
B="--steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed --records 50000000 --chunk 16666667"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
EXTRA="--mode zipf --key-sets 7" run ks7 FA_X=1
EXTRA="--mode zipf --key-sets 7" run ks7_atomic FA_CMS=atomic
EXTRA="--mode zipf --key-sets 3" run ks3 FA_X=1
EXTRA="--mode zipf --zipf-s 80 --key-sets 9" run ks9 FA_X=1
EXTRA="--mode zipf --key-sets 1" run ks1zipf FA_X=1
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r["dominant_kernel"]
    print("value %.4g  path %.4f ms frac %.4f | wtile %.4f ms frac %.4f | rest %.4f ms | close %.1f ms | direct %d" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], r["avg_launch_ms"]-k["avg_launch_ms"], d["config"]["window_close_merge_ms"], d["config"]["records_direct_path"]))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
