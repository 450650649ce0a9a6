#!/bin/bash
# Round 5, session 5: the suite on the claim / wait fix of the distinct-address sets; config 3 at full scale (1 B records) in both
# top-k modes with every CPU-side check; the streaming kernel trace + FETCH / WRITE of the candidates mode on the final sources;
# the default bench line and its traffic profile.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s5
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
grep -n "passed\|failed" $OUT/pytest.log; grep -n "^FAILED\|^E  " $OUT/pytest.log | head -20
timeout 300 python tools/debug_candidates.py 60000 12 12 12 32 10 2>&1 | tail -4
for mode in exact candidates; do
  timeout 1500 python tools/config3_run.py --topk-mode $mode > $OUT/config3_1B_$mode.json 2> $OUT/config3_1B_$mode.err; echo "config3 $mode rc=$?"
  grep '^{' $OUT/config3_1B_$mode.json | tail -1 | cut -c1-1800
done
PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode candidates" timeout 900 bash tools/profile.sh r05_config3_stream_candidates > $OUT/profile_config3_candidates.log 2>&1
grep -v "gen_\|rocprim\|rocclr\|row_" $ROOT/gpurun_out/prof/r05_config3_stream_candidates/summary.txt | head -40
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 900 bash tools/profile.sh r05 > $OUT/profile_default.log 2>&1
grep -A8 "calibrated HBM" $ROOT/gpurun_out/prof/r05/summary.txt
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; k=r["dominant_kernel"]
print("bench value %.4g path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | parity %s | clocks %s" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], d["parity"]["ok"], d["clocks"]["end"]))
PY
du -sh $ROOT/gpurun_out
