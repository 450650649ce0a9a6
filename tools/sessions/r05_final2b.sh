#!/bin/bash
# Round 5, last sessions on the final sources (b): config 3 at 1 B records in both top-k contracts (with every check), its streaming
# traffic profiles, the side measurements
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05final2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
for mode in exact candidates; do
  timeout 900 python tools/config3_run.py --topk-mode $mode > $OUT/config3_1B_$mode.json 2> $OUT/config3_1B_$mode.err; echo "config3 1B $mode rc=$?"
  grep '^{' $OUT/config3_1B_$mode.json | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('path_ms_per_launch','roofline_frac_path','path_ms_last_third_mean','checks','topk100_equals_ranking','sketches_bit_exact','rows_bit_exact') or 'ok' in k or 'equal' in k or 'exact' in k})"
done
for mode in exact candidates; do
  PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode $mode" timeout 900 bash tools/profile.sh r05_config3_stream_$mode > $OUT/profile_config3_$mode.log 2>&1
  grep -A8 "calibrated HBM" $ROOT/gpurun_out/prof/r05_config3_stream_$mode/summary.txt | head -10
done
S="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed"
run() { name=$1; shift; timeout 300 python bench.py $S "$@" > $OUT/side_$name.json 2> $OUT/side_$name.err; }
run mocker --mode mocker
run goflow --mode goflow --records 50000000
run reversed --mode reversed --records 50000000
run decode --stage decode --records 50000000
run zipf_ks1 --mode zipf --records 50000000 --chunk 16666667 --no-verify
run config3_shape --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 --no-verify
run config5_pair --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 --no-verify
for f in $OUT/side_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print(sys.argv[1].split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0)), (d.get("parity") or {}).get("ok"))
except Exception as e:
    print("ERR", sys.argv[1], e, open(sys.argv[1].replace(".json",".err")).read()[-600:])
PY
done
