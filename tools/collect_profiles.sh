#!/bin/bash
# Copies the evidence of a tools/gpu_session.sh run (gpurun_out/<tag>/, gpurun_out/prof/<tag>/) into profiles/ (tracked).
TAG=${1:-r02}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$TAG
P=$ROOT/gpurun_out/prof/$TAG
D=$ROOT/profiles
grep '^{' $S/bench_default.json | tail -1 > $D/${TAG}_bench_default.json
: > $D/${TAG}_bench_side_measurements.jsonl
for f in $S/bench_*.json; do
  n=$(basename $f .json); n=${n#bench_}
  [ "$n" = default ] && continue
  l=$(grep '^{' $f | tail -1)
  [ -n "$l" ] && echo "{\"run\": \"$n\", \"line\": $l}" >> $D/${TAG}_bench_side_measurements.jsonl
done
cp $P/summary.txt $D/${TAG}_rocprof_summary.txt
cp $P/traffic.json $D/${TAG}_traffic.json
for v in ks7 ks9; do  # profiles of the sketch variant (config 3 shape) and of the (SrcAddr,DstPort,Proto) sink (tools/profile.sh <tag>_<v> ...)
  [ -s ${P}_$v/summary.txt ] && cp ${P}_$v/summary.txt $D/${TAG}_${v}_rocprof_summary.txt
done
[ -s $S/config3_1B.json ] && grep '^{' $S/config3_1B.json | tail -1 > $D/${TAG}_config3_1B.json
[ -s $S/config4_8ranks_1gpu.json ] && grep '^{' $S/config4_8ranks_1gpu.json | tail -1 > $D/${TAG}_config4_8ranks_1gpu.json
[ -s $S/config5_100M.json ] && grep '^{' $S/config5_100M.json | tail -1 > $D/${TAG}_config5_100M.json
ls -la $D/${TAG}_*
