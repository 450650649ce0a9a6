#!/bin/bash
# Round-3 side measurements on one box: PCIe path breakdown, the other workloads' bench lines, kernel trace of config 2.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG:-r03_side}
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 200 python tools/pcie_rate.py > $OUT/pcie_rate.json 2> $OUT/pcie_rate.err
S="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed"
run() { name=$1; shift; timeout 300 python bench.py $S "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run reversed --mode reversed --records 50000000
run decode --stage decode --records 50000000
run zipf_ks1 --mode zipf --records 50000000 --chunk 16666667 --no-verify
run config3_shape --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 --no-verify
run config5_pair --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 --no-verify
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace7 -o trace --output-format csv -- python $ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 > $OUT/trace7.log 2>&1
cd $ROOT
cat $OUT/pcie_rate.json
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys,os
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print("%-28s value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f" % (os.path.basename(sys.argv[1]), d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0)))
except Exception as e:
    print("ERR", sys.argv[1], e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
for t in trace trace7; do echo "== $t"; f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200; done
