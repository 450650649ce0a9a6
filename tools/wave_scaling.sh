#!/bin/bash
# Throughput of the wave-tile kernel against waves per CU (2 workgroups x FA_WBLOCK/64 waves): is the kernel bound by
# per-wave latency (scales with waves) or by a shared resource (saturates)?  Libraries: make OUT=../libflowagg_w4.so
# EXTRA=-DFA_WBLOCK=256, _w6 384, _w7 448, (default 512), _w9 "-DFA_WBLOCK=576 -DFA_WT_STRIDE=4864".
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/wave_scaling.txt
: > $O
for mode in aspairs mocker; do
for v in w4 w6 w7 "" w9; do
  FA_LIB_VARIANT=$v timeout 100 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-verify --mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-8s %-8s tile %.4f ms  frac %.3f  all %.4f ms  %.2f G rec/s' % ('$mode', '${v:-w8(default)}', r['avg_launch_ms'], r['frac'], r['all_kernels_avg_ms'], d['value']/1e9))" >> $O
done
done
cat $O
