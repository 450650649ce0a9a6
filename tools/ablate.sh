#!/bin/bash
# Ablation sweep of the ingest kernel (measurement only; FA_DEBUG_FLAGS breaks results).
# usage: tools/ablate.sh [records] [mode]; prints avg tile-kernel ms per variant
# flags: 1 no sink, 2 no wave combine, 4 no LDS table, 8 no global table, 16 no parse
REC=${1:-100000000}
MODE=${2:-aspairs}
for f in 0 2 4 6 8 14 1 17; do
  FA_DEBUG_FLAGS=$f timeout 300 python bench.py --records $REC --mode $MODE --steps 3 --warmup 1 --cpu-sample 0 --no-assert 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('mode=$MODE flags=%-3s  %8.3f ms/launch  %7.1f GB/s  %6.2f G rec/s' % ('$f', r['avg_launch_ms'], r['achieved'], d['value']/1e9))
"
done
