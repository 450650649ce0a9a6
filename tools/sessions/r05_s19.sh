#!/bin/bash
# goflow A/B (new library vs the previous commit's, same box), the generator guard test, host parser + tile-kernel (FA_TILE=wg) sanity
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s19
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', '%.4g rec/s' % d['value'], 'path %.4f ms frac %.4f' % (r['avg_launch_ms'], r['frac']), 'kernel %.4f ms frac %.4f' % (r['dominant_kernel']['avg_launch_ms'], r['dominant_kernel']['frac']), 'parity', d['parity']['ok'], d['config'].get('chunk_records'))"; }
for rep in 1 2; do
  for v in new base; do
    if [ $v = base ]; then export FA_LIB_VARIANT=ablate; else unset FA_LIB_VARIANT; fi
    python bench.py --mode goflow --records 50000000 --steps 12 --warmup 3 --cpu-sample 0 --no-host-fed 2>$OUT/err_goflow_$v.txt | tee $OUT/bench_goflow_${v}_$rep.json | line "goflow $v"
  done
done
unset FA_LIB_VARIANT
python bench.py --mode goflow --steps 6 --warmup 2 --cpu-sample 0 --no-host-fed 2>$OUT/err_goflow_default.txt | tee $OUT/bench_goflow_default.json | line "goflow default-size"
tail -2 $OUT/err_goflow_default.txt | cut -c1-300
