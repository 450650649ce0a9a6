#!/bin/bash
# Round 4, GPU session 2: suite; GoFlow long-record geometry A/B; config 3 streaming A/B of the distinct-set lists; config 5 reads / closes.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -12 $OUT/pytest.log
S="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed"
sumline() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print(sys.argv[1].split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0)), (d.get("parity") or {}).get("ok"))
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
}
timeout 300 python bench.py $S --mode goflow --records 50000000 --chunk 16666667 > $OUT/bench_goflow_long.json 2> $OUT/bench_goflow_long.err; sumline $OUT/bench_goflow_long.json
FA_LONG_TILES=0 timeout 300 python bench.py $S --mode goflow --records 50000000 --chunk 16666667 > $OUT/bench_goflow_short.json 2> $OUT/bench_goflow_short.err; sumline $OUT/bench_goflow_short.json
for v in 1 0; do
  FA_KS_DEFER=$v timeout 600 python tools/config3_run.py --records 200000000 --timing-only > $OUT/config3_200M_defer$v.json 2> $OUT/config3_200M_defer$v.err
  echo "config3 defer=$v: $(cut -c1-700 $OUT/config3_200M_defer$v.json | sed 's/.*"launches"/"launches"/')"
done
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04s2/config5_100M.json"))
for k in ("path_ms_per_launch","roofline_frac_path","read_app_windows_ms","close_app_windows_ms","per_window_ms","wide_log_after_closes","closes_return_the_windows_read_before","app_rows_left_after_all_closes","flows_5m_aligned_windows_bit_exact","sliding_window_bit_exact","app_count_equals_records"):
    print(k, d.get(k))
PY
grep "flowagg read" $OUT/config5_100M.err | grep "SrcAddr" | tail -14; tail -3 $OUT/config5_100M.err
