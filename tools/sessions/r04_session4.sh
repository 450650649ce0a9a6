#!/bin/bash
# Round 4, GPU session 4: suite; config 3 streaming (400 M records) lists v3 vs inline, per-launch series; heavy groups A/B on the
# Zipf side workloads and on the default workload.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s4
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
for v in 1 0; do
  FA_KS_DEFER=$v timeout 600 python tools/config3_run.py --records 400000000 --timing-only > $OUT/config3_400M_defer$v.json 2> $OUT/config3_400M_defer$v.err
  python - $OUT/config3_400M_defer$v.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split("/")[-1], "path ms/launch %.4f last third %.4f listed %d" % (d["path_ms_per_launch"], d["path_ms_last_third_mean"], d["distinct_set_keys_listed"]), d["path_ms_series"])
PY
done
S="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify"
sumline() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print(sys.argv[1].split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0)))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
}
for h in 1 0; do
  FA_HEAVY=$h timeout 300 python bench.py $S --mode zipf --records 50000000 --chunk 16666667 > $OUT/bench_zipf_ks1_heavy$h.json 2> $OUT/bench_zipf_ks1_heavy$h.err; sumline $OUT/bench_zipf_ks1_heavy$h.json
  FA_HEAVY=$h timeout 300 python bench.py $S --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 > $OUT/bench_ks7_heavy$h.json 2> $OUT/bench_ks7_heavy$h.err; sumline $OUT/bench_ks7_heavy$h.json
  FA_HEAVY=$h timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-host-fed --no-verify > $OUT/bench_default_heavy$h.json 2> $OUT/bench_default_heavy$h.err; sumline $OUT/bench_default_heavy$h.json
done
