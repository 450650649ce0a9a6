#!/bin/bash
# Do tighter tuple segments (FA_SEG_CAP: tuples per (partition, workgroup) segment; default 2 x mean + 32 = 544 on config 2, mean 254)
# make agg8_kernel's walk or wtile_kernel's stores faster?  Kernel trace per case, same box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/segcap
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps 4 --warmup 2 --cpu-sample 0 --no-host-fed"
one() {
  local name=$1 cap=$2; shift 2
  if [ -n "$cap" ]; then export FA_SEG_CAP=$cap; else unset FA_SEG_CAP; fi
  PROF_PASSES=trace bash tools/profile.sh segcap_$name $B "$@" > $OUT/$name.log 2>&1
  echo "== $name (FA_SEG_CAP='$cap') $@"
  grep -h "wtile_kernel\|agg8_kernel\|deferred_kernel" $ROOT/gpurun_out/prof/segcap_$name/summary.txt | head -3 | cut -c1-130
  grep -o '"parity": {"ok": [a-z]*' $ROOT/gpurun_out/prof/segcap_$name/trace.log | tail -1
  grep -o '"value": [0-9.e+]*\|"frac": [0-9.]*' $ROOT/gpurun_out/prof/segcap_$name/trace.log | head -2 | tr '\n' ' '; echo
}
one default ""
one c288 288
one c320 320
one c384 384
one c448 448
one default2 ""
