#!/bin/bash
# Round 5, session 10: the folded row-0 bits in the candidates test and the sample bound in fa_topk's first read - suite, same-box
# exact vs candidates, top-k read phases, then the default bench + its traffic profile on these (final) sources.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s10
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|^FAILED|^E  " $OUT/pytest.log | head
timeout 300 python tools/debug_candidates.py 60000 12 12 12 32 10 2>&1 | tail -2
for i in 1 2; do for mode in exact candidates; do
  FA_VERBOSE=1 timeout 300 python tools/config3_run.py --records 600000000 --timing-only --topk-mode $mode > $OUT/c3_${mode}_$i.json 2> $OUT/c3_${mode}_$i.err
  python - $OUT/c3_${mode}_$i.json $mode <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s=d["path_ms_series"]
print(sys.argv[2], "all %.4f  last third %.4f  frac(last third) %.4f  first four %s  topk %s" % (d["path_ms_per_launch"], d["path_ms_last_third_mean"], d["wire_bytes"]/d["launches"]/d["path_ms_last_third_mean"]/8e9, s[:4], d["topk100_ms_per_call"]))
PY
done; done
grep "flowagg read\] top-k" $OUT/c3_exact_1.err | head -4 | cut -c1-260
for mode in exact candidates; do
  timeout 1500 python tools/config3_run.py --topk-mode $mode > $OUT/config3_1B_$mode.json 2> $OUT/config3_1B_$mode.err; echo "config3 1B $mode rc=$?"
  python - $OUT/config3_1B_$mode.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k:d[k] for k in ("topk_mode","path_ms_per_launch","roofline_frac_path","topk100_ms_per_call","addresses_held","sketch_bit_exact_full_stream","sketch_bit_exact_prefix","top100_equals_ranking_of_the_whole_universe")})
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 900 bash tools/profile.sh r05 > $OUT/profile_default.log 2>&1
for mode in exact candidates; do
  PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode $mode" timeout 900 bash tools/profile.sh r05_config3_stream_$mode > $OUT/profile_config3_$mode.log 2>&1
  grep -v "gen_\|rocprim\|rocclr\|row_" $ROOT/gpurun_out/prof/r05_config3_stream_$mode/summary.txt | head -12; grep -A9 "calibrated HBM" $ROOT/gpurun_out/prof/r05_config3_stream_$mode/summary.txt
done
grep -A5 "calibrated HBM" $ROOT/gpurun_out/prof/r05/summary.txt
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; k=r["dominant_kernel"]
print("bench value %.4g path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | parity %s | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], d["parity"]["ok"], r["traffic"]))
PY
