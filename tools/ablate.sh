#!/bin/bash
# Ablation sweep of the ingest kernel (measurement only; FA_DEBUG_FLAGS breaks results).
# usage: tools/ablate.sh [records] [mode]; prints avg tile-kernel ms per variant
# flags: 1 no sink, 2 loop parser (parse_fast) instead of parse_canon, 4 no hot-key LDS table, 8 no direct path,
#        16 no parse, 32 no tuple stores, 64/128/256 agg kernel: loads only / no flush / no probing path,
#        512 DMA without nt, 1024 phase timing (workgroup kernel), 2048/4096 nt / system-scope single tuple stores,
#        8192 no per-lane offset loads, 16384 synthetic aligned tiles, 32768 lanes own nothing, 131072 no frame check
#        (csrc/sinks.cuh DBG_*); FA_SINK=direct forces the device-wide-table sink, FA_TILE=wg the workgroup-tile kernel
REC=${1:-100000000}
MODE=${2:-aspairs}
for f in 0 2 4 32 1 17 d0 d2; do
  SINK=auto; ff=$f; case $f in d*) SINK=direct; ff=${f#d};; esac
  FA_SINK=$SINK FA_DEBUG_FLAGS=$ff timeout 300 python bench.py --records $REC --mode $MODE --steps 3 --warmup 1 --cpu-sample 0 --no-assert 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('mode=$MODE flags=%-3s  tile %7.3f ms %7.1f GB/s | all kernels %7.3f ms %7.1f GB/s | %6.2f G rec/s' % ('$f', r['avg_launch_ms'], r['achieved'], r['all_kernels_avg_ms'], r['all_kernels_achieved'], d['value']/1e9))
"
done
