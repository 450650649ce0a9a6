#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2g
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed --records 50000000 --chunk 16666667 --no-assert --mode zipf --key-sets 7"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run full FA_X=1
run nokeyset FA_DEBUG_FLAGS=262144
run nocms FA_DEBUG_FLAGS=524288
run neither FA_DEBUG_FLAGS=786432
cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $B > $OUT/trace.log 2>&1; cd $ROOT
python tools/prof_summary.py $OUT 2>&1 | head -16
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
