#!/bin/bash
# Round 4, session 10 (final sources): the suite, the driver's bench line twice, rocprofv3 evidence of the default workload (trace +
# FETCH / WRITE -> traffic.json), config 5 with a pageable and with a page-locked row buffer (FA_VERBOSE phase times), its traces.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s10
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 900 bash tools/profile.sh r04 > $OUT/profile.log 2>&1
cp $ROOT/gpurun_out/prof/r04/traffic.json $ROOT/profiles/r04_traffic.json   # (bench.py takes the traffic from here when the sources' hash matches)
for i in 1 2; do timeout 600 python bench.py > $OUT/bench_default$i.json 2> $OUT/bench_default$i.err; done
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 pageable rc=$?"
FA_VERBOSE=1 timeout 900 python tools/config5_run.py --pinned-out > $OUT/config5_100M_pinned.json 2> $OUT/config5_100M_pinned.err; echo "config5 pinned rc=$?"
FA_WIDE=log timeout 900 python tools/config5_run.py > $OUT/config5_100M_log.json 2> $OUT/config5_100M_log.err; echo "config5 log rc=$?"
PROF_PASSES="trace" PROF_CMD="python tools/config5_run.py" timeout 600 bash tools/profile.sh r04_config5 > $OUT/profile_config5.log 2>&1
FA_WIDE=log PROF_PASSES="trace" PROF_CMD="python tools/config5_run.py" timeout 600 bash tools/profile.sh r04_config5_log > $OUT/profile_config5_log.log 2>&1
for f in config5_100M config5_100M_pinned config5_100M_log; do echo "== $f"; grep '^{' $OUT/$f.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('row_buffer','launches','path_ms_per_launch','roofline_frac_path','read_app_windows_ms','close_app_windows_ms','per_window_ms','closes_return_the_windows_read_before','app_count_equals_records','app_sum_bytes_equals_flows_5m','flows_5m_aligned_windows_bit_exact','sliding_window_bit_exact')})"; done
for p in r04; do echo "== prof $p"; head -8 $ROOT/gpurun_out/prof/$p/summary.txt; grep -A4 "FETCH_SIZE, per launch" $ROOT/gpurun_out/prof/$p/summary.txt | head -5; grep -A4 "WRITE_SIZE, per launch" $ROOT/gpurun_out/prof/$p/summary.txt | head -5; done
for i in 1 2; do python - $OUT/bench_default$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; k=r.get("dominant_kernel") or {}
print("bench value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")), (d.get("parity") or {}).get("ok"), d["cpu_baseline"]["value"])
PY
done
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
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print(sys.argv[1].split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0)), (d.get("parity") or {}).get("ok"))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
done
