#!/bin/bash
# agg8_kernel's segment walk with a prefetch that really overlaps (no memory operation but the tuple loads inside the loop, addresses
# worked out ahead of the loads): round-3 kernel ("old") against the new one at 2, 3 and 4 register buffers; config 2 (33.3 M-record
# launches), the Zipf-1.1 stream (16.67 M) and config 5's pair of key sets; kernel trace per case; then the whole suite.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/agg8exp
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 4 --warmup 2 --cpu-sample 0 --no-host-fed"
Z="--mode zipf --records 50000000 --chunk 16666667"
Z7="--mode zipf --key-sets 7 --records 50000000 --chunk 16666667 --no-verify"
one() {  # name, variant, bench args
  local name=$1 var=$2; shift 2
  FA_LIB_VARIANT=$var PROF_PASSES=trace bash tools/profile.sh agg8exp_$name $B "$@" > $OUT/$name.log 2>&1
  echo "== $name (variant '$var') $@"
  grep -h "wtile_kernel\|agg8_kernel\|deferred_kernel" $ROOT/gpurun_out/prof/agg8exp_$name/summary.txt | head -3 | cut -c1-130
  grep -o '"parity": {"ok": [a-z]*' $ROOT/gpurun_out/prof/agg8exp_$name/trace.log | tail -1
  grep -o '"value": [0-9.e+]*\|"frac": [0-9.]*' $ROOT/gpurun_out/prof/agg8exp_$name/trace.log | head -2 | tr '\n' ' '; echo
}
for rep in 1 2; do
one c2_old_$rep old
one c2_new_$rep ""
one c2_d3_$rep d3
one c2_d4_$rep d4
done
one z_old old $Z
one z_new "" $Z
one z_d3 d3 $Z
one z_d4 d4 $Z
one z7_old old $Z7
one z7_new "" $Z7
one z7_d3 d3 $Z7
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
