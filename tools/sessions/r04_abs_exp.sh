#!/bin/bash
# Old library against new, alternating on one box (hipEvent times of bench.py, no profiler).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/absexp4
mkdir -p $OUT
cd $ROOT
B="--steps 6 --warmup 3 --cpu-sample 0 --no-host-fed"
one() {
  local name=$1 var=$2; shift 2
  FA_LIB_VARIANT=$var timeout 300 python bench.py $B "$@" > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json "$name" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print("%-14s value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | parity %s" % (sys.argv[2], d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), (d.get("parity") or {}).get("ok")))
except Exception as e:
    print("ERR", sys.argv[2], e)
PY
}
G="--mode goflow --records 50000000 --chunk 16666667"
Z="--mode zipf --records 50000000 --chunk 16666667"
Z7="--mode zipf --key-sets 7 --records 50000000 --chunk 16666667 --no-verify"
for rep in 1 2 3 4; do
one c2_old_$rep old
one c2_new_$rep ""
done
for rep in 1 2; do
one gf_old_$rep old $G
one gf_new_$rep "" $G
one z_old_$rep old $Z
one z_new_$rep "" $Z
one z7_old_$rep old $Z7
one z7_new_$rep "" $Z7
one mk_old_$rep old --mode mocker
one mk_new_$rep "" --mode mocker
done
