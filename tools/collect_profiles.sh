#!/bin/bash
# Copies the evidence of a tools/gpu_session.sh run (gpurun_out/<tag>/, gpurun_out/prof/<tag>*/) into profiles/ (tracked).
TAG=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=$ROOT/gpurun_out/$TAG
P=$ROOT/gpurun_out/prof/$TAG
D=$ROOT/profiles
[ -s $S/bench_default.json ] && grep '^{' $S/bench_default.json | tail -1 > $D/${TAG}_bench_default.json
: > $D/${TAG}_bench_side_measurements.jsonl
for f in $S/bench_*.json; do
  n=$(basename $f .json); n=${n#bench_}
  [ "$n" = default ] && continue
  l=$(grep '^{' $f | tail -1)
  [ -n "$l" ] && echo "{\"run\": \"$n\", \"line\": $l}" >> $D/${TAG}_bench_side_measurements.jsonl
done
[ -s $P/summary.txt ] && cp $P/summary.txt $D/${TAG}_rocprof_summary.txt
[ -s $P/traffic.json ] && cp $P/traffic.json $D/${TAG}_traffic.json
for v in decode goflow reversed ks7 ks9; do  # projection stage, collector-shaped producers, sketch variant (config 3 shape), (SrcAddr,DstPort,Proto) sink
  [ -s ${P}_$v/summary.txt ] && cp ${P}_$v/summary.txt $D/${TAG}_${v}_rocprof_summary.txt
done
[ -s $S/pcie_rate.json ] && cp $S/pcie_rate.json $D/${TAG}_pcie_rate.json
for f in config3_1B config4_8ranks_1gpu config5_100M config5_100M_scatter config5_100M_log config5_8ranks_1gpu; do
  [ -s $S/$f.json ] && grep '^{' $S/$f.json | tail -1 > $D/${TAG}_$f.json
done
ls -la $D/${TAG}_*
