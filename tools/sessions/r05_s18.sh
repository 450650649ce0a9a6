#!/bin/bash
# learnt field order (parse_seq tier): parity, then same-box A/B against the library built from the previous commit
# (flow-pipeline_amd/libflowagg_ablate.so, FA_LIB_VARIANT=ablate): config 2, reversed order, GoFlow shape
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s18
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ingest_sinks_gpu.py -q -m gpu -x 2>&1 | tail -4
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', '%.4g rec/s' % d['value'], 'path %.4f ms frac %.4f' % (r['avg_launch_ms'], r['frac']), 'kernel %.4f ms frac %.4f' % (r['dominant_kernel']['avg_launch_ms'], r['dominant_kernel']['frac']), 'parity', d['parity']['ok'])"; }
for rep in 1 2; do
  for mode in aspairs reversed goflow; do
    for v in new base; do
      if [ $v = base ]; then export FA_LIB_VARIANT=ablate; else unset FA_LIB_VARIANT; fi
      python bench.py --mode $mode --steps 12 --warmup 3 --cpu-sample 0 --no-host-fed 2>$OUT/err_${mode}_$v.txt | tee $OUT/bench_${mode}_${v}_$rep.json | line "$mode $v"
    done
  done
done
unset FA_LIB_VARIANT
