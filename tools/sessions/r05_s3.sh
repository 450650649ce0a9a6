#!/bin/bash
# Round 5, session 3: the whole suite on the one-pass / two-pass top-k and the candidates mode; config 3 streaming: top-k read
# phases (FA_VERBOSE), candidates mode at three tracked ranks, kernel trace + FETCH / WRITE of both modes; the default bench's
# traffic profile (profiles/r05_traffic.json).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s3
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
FA_VERBOSE=1 timeout 300 python tools/config3_run.py --records 200000000 --timing-only > $OUT/config3_exact_verbose.json 2> $OUT/config3_exact_verbose.err
grep "flowagg read" $OUT/config3_exact_verbose.err | tail -12
grep '^{' $OUT/config3_exact_verbose.json | tail -1 | cut -c1-400
for tr in 16 128 1024; do
  timeout 300 python tools/config3_run.py --records 400000000 --timing-only --topk-mode candidates --topk-track $tr > $OUT/config3_cand_track$tr.json 2> $OUT/config3_cand_track$tr.err
  python - $OUT/config3_cand_track$tr.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], "path last third %.4f ms  frac(all) %.4f  topk ms %s held %s" % (d["path_ms_last_third_mean"], d["roofline_frac_path"], d["topk100_ms_per_call"], d["addresses_held"]), d["path_ms_series"][:4])
PY
done
for mode in exact candidates; do
  PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode $mode" timeout 900 bash tools/profile.sh r05_config3_stream_$mode > $OUT/profile_config3_$mode.log 2>&1
  grep -v "gen_\|rocprim\|rocclr\|row_" $ROOT/gpurun_out/prof/r05_config3_stream_$mode/summary.txt | head -60
done
timeout 900 bash tools/profile.sh r05 > $OUT/profile_default.log 2>&1
grep -v "gen_\|rocprim\|rocclr\|row_" $ROOT/gpurun_out/prof/r05/summary.txt | head -40
du -sh $ROOT/gpurun_out
