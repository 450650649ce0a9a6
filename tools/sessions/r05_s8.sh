#!/bin/bash
# same box, alternating: libflowagg_base.so (the commit before) against libflowagg.so (rare blocks read their kernel arguments from
# the kernarg segment: cold_args) on the default bench, the GoFlow-shaped stream and config 3's shape
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s8
mkdir -p $OUT
cd $ROOT
S="--steps 20 --warmup 5 --cpu-sample 0 --no-host-fed --no-verify"
one() { tag=$1; variant=$2; shift 2; FA_LIB_VARIANT=$variant timeout 300 python bench.py $S "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; k=r["dominant_kernel"]
print(json.dumps({"run": sys.argv[2], "path_ms": round(r["avg_launch_ms"],4), "path_frac": round(r["frac"],4), "kernel_ms": round(k["avg_launch_ms"],4), "kernel_frac": round(k["frac"],4), "value": d["value"], "sclk_end": (d.get("clocks") or {}).get("end",{}).get("sclk_mhz")}))
PY
}
for i in 1 2 3; do one default_base_$i base; one default_cold_$i ""; done
for i in 1 2; do one goflow_base_$i base --mode goflow --records 50000000 --chunk 16666667; one goflow_cold_$i "" --mode goflow --records 50000000 --chunk 16666667; done
for i in 1 2; do one ks7_base_$i base --mode zipf --key-sets 7 --records 50000000 --chunk 16666667; one ks7_cold_$i "" --mode zipf --key-sets 7 --records 50000000 --chunk 16666667; done
one mocker_base base --mode mocker; one mocker_cold "" --mode mocker
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ingest_sinks_gpu.py tests/test_topk_gpu.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; grep -E "passed|failed|^FAILED|^E  " $OUT/pytest.log | head
