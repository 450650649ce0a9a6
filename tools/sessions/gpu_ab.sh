#!/bin/bash
# A/B of library variants (flow-pipeline_amd/libflowagg_<name>.so built with make OUT=... EXTRA=...): quick parity + bench.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 10 --warmup 3 --cpu-sample 0 --no-host-fed"
for v in "" $VARIANTS; do
  name=${v:-base}
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_ingest_sinks_gpu.py -m gpu -q -x -k "large_device or tuple_formats or segment_overflow or rollup_matches" > $OUT/pytest_$name.log 2>&1
  echo "$name: $(tail -1 $OUT/pytest_$name.log)"
  timeout 300 python bench.py $B $EXTRA_ARGS > $OUT/bench_${name}.json 2> $OUT/bench_${name}.err
  timeout 300 python bench.py $B --chunk 16666667 --no-verify $EXTRA_ARGS > $OUT/bench_${name}_c16.json 2> $OUT/bench_${name}_c16.err
  for m in $MODES; do   # e.g. MODES="mocker goflow zipf:7 zipf:9"
    mode=${m%%:*}; ks=1; [ "$m" != "$mode" ] && ks=${m##*:}
    timeout 300 python bench.py $B --no-verify --mode $mode --key-sets $ks --chunk 16666667 > $OUT/bench_${name}_${mode}_ks$ks.json 2> $OUT/bench_${name}_${mode}_ks$ks.err
  done
done
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print("value %.4g  path %.4f ms frac %.4f | wtile %.4f ms frac %.4f | rest %.4f | parity %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r["avg_launch_ms"]-k.get("avg_launch_ms",0), (d.get("parity") or {}).get("ok")))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
