#!/bin/bash
# Old library against new, alternating on one box (hipEvent times of bench.py, no profiler).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/absexp2
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
for rep in 1 2 3; do
one gf_old_$rep old $G
one gf_new_$rep "" $G
done
one c2_old old
one c2_new ""
timeout 600 python -m pytest tests -m gpu -q -x -k "goflow or GoFlow or template or parser or tier or shapes or producers" 2>&1 | tail -3
