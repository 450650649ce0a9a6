#!/bin/bash
# Per-round phase timing of the wave-tile kernel (measurement builds: make OUT=../libflowagg_t1.so EXTRA=-DFA_WT_TIMING,
# ..._te.so EXTRA="-DFA_WT_EARLY=1 -DFA_WT_TIMING"; the t2 / t2w8 libraries of profiles/r01_wtile_phase_timing_s5.txt were the
# double-buffered kernel of commit f4ed14a).
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/wt_timing.txt
: > $O
for v in t1 te; do
  for fl in 1024 1025 1041; do
    for mode in aspairs mocker; do
      echo "== lib $v FA_DEBUG_FLAGS=$fl mode $mode" >> $O
      FA_LIB_VARIANT=$v FA_DEBUG_FLAGS=$fl timeout 100 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-verify --no-assert --mode $mode 2>> $O | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('   tile %.4f ms  all %.4f ms  %.2f G rec/s' % (r['avg_launch_ms'], r['all_kernels_avg_ms'], d['value']/1e9))" >> $O
    done
  done
done
cat $O
