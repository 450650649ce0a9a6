#!/bin/bash
# Round 4, GPU session 3: suite; kernel traces of config 3 STREAMING (lists vs inline probing); config 5 reads / closes again.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
for v in 1 0; do
  FA_KS_DEFER=$v PROF_PASSES="trace" PROF_CMD="python tools/config3_run.py --records 200000000 --timing-only" timeout 600 bash tools/profile.sh r04_config3_stream_defer$v > $OUT/prof_config3_defer$v.log 2>&1
  head -14 $ROOT/gpurun_out/prof/r04_config3_stream_defer$v/summary.txt
  grep -o '"path_ms_per_launch": [0-9.]*' $ROOT/gpurun_out/prof/r04_config3_stream_defer$v/trace.log
done
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04s3/config5_100M.json"))
for k in ("path_ms_per_launch","roofline_frac_path","read_app_windows_ms","close_app_windows_ms","per_window_ms","closes_return_the_windows_read_before","app_rows_left_after_all_closes","flows_5m_aligned_windows_bit_exact","sliding_window_bit_exact","app_count_equals_records"):
    print(k, d.get(k))
PY
grep "flowagg read" $OUT/config5_100M.err | grep "SrcAddr" | tail -8
