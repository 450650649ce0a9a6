#!/bin/bash
# A/B of dynamic tile assignment (libflowagg_dyn.so: EXTRA=-DFA_WT_DYN=1; _dynearly: "-DFA_WT_DYN=1 -DFA_WT_EARLY=1").
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/ab_dyn.txt
: > $O
for rep in 1 2; do
for mode in aspairs mocker; do
for v in "" dyn dynearly; do
  FA_LIB_VARIANT=$v timeout 100 python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-verify --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %-10s tile %.4f ms  frac %.3f  all %.4f ms  %.2f G rec/s  direct %d' % ('$mode', '${v:-default}', r['avg_launch_ms'], r['frac'], r['all_kernels_avg_ms'], d['value']/1e9, d['config']['records_direct_path']))" >> $O
done
done
done
for v in dyn dynearly; do
FA_LIB_VARIANT=$v timeout 300 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$v.log 2>&1; echo "pytest $v rc=$? $(tail -n 1 gpurun_out/pytest_$v.log)" >> $O
done
cat $O
