#!/bin/bash
# A/B of XCD-aware tile numbering (libflowagg_xcd.so: make OUT=../libflowagg_xcd.so EXTRA=-DFA_WT_XCD=1).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/ab_xcd.txt
: > $O
for mode in aspairs mocker; do
for v in "" xcd; do
  FA_LIB_VARIANT=$v timeout 100 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-verify --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %-10s tile %.4f ms  frac %.3f  all %.4f ms  %.2f G rec/s' % ('$mode', '${v:-default}', r['avg_launch_ms'], r['frac'], r['all_kernels_avg_ms'], d['value']/1e9))" >> $O
done
done
FA_LIB_VARIANT=xcd timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_xcd.log 2>&1; echo "pytest xcd rc=$? $(tail -n 1 gpurun_out/pytest_xcd.log)" >> $O
cat $O
