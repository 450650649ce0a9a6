#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2h
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x -k "count_min or config3 or topk or cms or wide or nccl or two_ranks" ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
B="--steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed --records 50000000 --chunk 16666667 --mode zipf --key-sets 7"
run() { name=$1; shift; env "$@" timeout 300 python bench.py $B $EXTRA > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
EXTRA="" run full FA_X=1
EXTRA="--no-assert" run nokeyset FA_DEBUG_FLAGS=262144
EXTRA="--no-assert" run nocms FA_DEBUG_FLAGS=524288
EXTRA="" run nohot FA_DEBUG_FLAGS=1048576
EXTRA="--records 100000000 --chunk 33333334" run c33 FA_X=1
cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $B > $OUT/trace.log 2>&1; cd $ROOT
python tools/prof_summary.py $OUT 2>&1 | head -12
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
