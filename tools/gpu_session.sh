#!/bin/bash
# One GPU-box session (gpurun): the whole GPU suite, the driver's bench line, the side measurements of DESIGN.md section 4,
# and the rocprofv3 evidence (kernel trace, FETCH/WRITE PMC passes, SQ counters with PROF_SQ=1).  Output: gpurun_out/<tag>/.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -4 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
S="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed"
run() { name=$1; shift; timeout 300 python bench.py $S "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run mocker --mode mocker
run goflow --mode goflow --records 50000000 --chunk 16666667
run reversed --mode reversed --records 50000000
run decode --stage decode --records 50000000
run zipf_ks1 --mode zipf --records 50000000 --chunk 16666667 --no-verify
run config3_shape --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 --no-verify
run config5_pair --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 --no-verify
FA_TUPLE=16 run wide_tuples --chunk 16666667 --no-verify
run c16 --chunk 16666667 --no-verify
PROF_SQ=${PROF_SQ:-} timeout 900 bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
if [ -n "$FULL" ]; then  # the full-scale runs of configs 3 and 5 (profiles/<tag>_config3_1B.json, <tag>_config5_100M.json)
  timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
  timeout 1500 python tools/config3_run.py > $OUT/config3_1B.json 2> $OUT/config3_1B.err; echo "config3 rc=$?"
fi
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print("value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")))
    print("  ", {k2:d["config"].get(k2) for k2 in ("tuple_format","launches_per_step","records_second_chance_parser","window_close_merge_ms")}, (d.get("parity") or {}).get("ok"), (d.get("host_fed") or {}).get("wire_GBps"), (d.get("cpu_baseline") or {}).get("thread_sweep_records_per_s"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
