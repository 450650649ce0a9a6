#!/bin/bash
# Shader clock and power while the ingest kernels run back to back (is the kernel clock / power limited?).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/clock_probe.txt
: > $O
echo "== idle" >> $O
rocm-smi --showclocks --showpower --showperflevel 2>&1 | grep -iE "sclk|mclk|fclk|power|perf" >> $O
rocm-smi --showmaxpower 2>&1 | grep -i "power" >> $O
for v in "" early db; do
  echo "== variant '${v:-default}'" >> $O
  FA_LIB_VARIANT=$v timeout 200 python bench.py --steps 4000 --warmup 2 --cpu-sample 0 --no-verify > $O.$v.json 2>/dev/null &
  pid=$!
  sleep 9
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|mclk|power" | tr '\n' ' ' >> $O; echo >> $O
    sleep 1
  done
  wait $pid
  python -c "
import json,sys
d=json.loads(open('$O.$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('   tile %.4f ms  all %.4f ms  %.2f G rec/s' % (r['avg_launch_ms'], r['all_kernels_avg_ms'], d['value']/1e9))" >> $O
done
cat $O
