#!/bin/bash
# compile-time constants in the wave-tile kernels (top-k contract, sketch scatter sink, tuple bins): same-box A/B against the
# previous commit's library (libflowagg_ablate.so): config 2, GoFlow, config 3 in both contracts, config 5 shape; then the tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s22
mkdir -p $OUT
cd $ROOT
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', '%.4g rec/s' % d['value'], 'path %.4f ms frac %.4f' % (r['avg_launch_ms'], r['frac']), 'kernel %.4f ms frac %.4f' % (r['dominant_kernel']['avg_launch_ms'], r['dominant_kernel']['frac']), 'parity', d['parity']['ok'])"; }
for rep in 1 2; do
  for v in new base; do
    if [ $v = base ]; then export FA_LIB_VARIANT=ablate; else unset FA_LIB_VARIANT; fi
    python bench.py --steps 12 --warmup 3 --cpu-sample 0 --no-host-fed 2>/dev/null | tee $OUT/bench_aspairs_${v}_$rep.json | line "aspairs $v"
    python bench.py --mode goflow --records 50000000 --steps 12 --warmup 3 --cpu-sample 0 --no-host-fed 2>/dev/null | tee $OUT/bench_goflow_${v}_$rep.json | line "goflow $v"
    python bench.py --mode zipf --key-sets 9 --steps 8 --warmup 2 --cpu-sample 0 --no-verify --no-host-fed 2>/dev/null | tee $OUT/bench_ks9_${v}_$rep.json | line "ks9 $v"
    for mode in exact candidates; do
      timeout 600 python tools/config3_run.py --records 200000000 --timing-only --topk-mode $mode 2>/dev/null | tee $OUT/c3_${mode}_${v}_$rep.json | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config3 $mode $v', round(d['path_ms_per_launch'],4), round(d['roofline_frac_path'],4), 'last third', round(d['path_ms_last_third_mean'],4))"
    done
  done
done
unset FA_LIB_VARIANT
timeout 1200 python -m pytest tests/test_topk_gpu.py tests/test_ingest_sinks_gpu.py tests/test_window_close_gpu.py -q -m gpu -x 2>&1 | tail -3
