#!/bin/bash
# measurement build: the wave-tile kernels with the parse tiers behind the mocker template compiled out (libflowagg_ablate.so,
# -DFA_EXP_T0ONLY) against the production library, same box: what those tiers' CODE costs streams that never run it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s23
mkdir -p $OUT
cd $ROOT
line() { python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('$1', '%.4g rec/s' % d['value'], 'path %.4f ms frac %.4f' % (r['avg_launch_ms'], r['frac']), 'kernel %.4f ms frac %.4f' % (r['dominant_kernel']['avg_launch_ms'], r['dominant_kernel']['frac']))"; }
for rep in 1 2 3; do
  for v in prod t0only; do
    if [ $v = t0only ]; then export FA_LIB_VARIANT=ablate; else unset FA_LIB_VARIANT; fi
    python bench.py --steps 12 --warmup 3 --cpu-sample 0 --no-host-fed --no-verify 2>/dev/null | tee $OUT/bench_aspairs_${v}_$rep.json | line "aspairs $v"
    for mode in exact candidates; do
      [ $rep = 3 ] && continue
      timeout 600 python tools/config3_run.py --records 200000000 --timing-only --topk-mode $mode 2>/dev/null | tee $OUT/c3_${mode}_${v}_$rep.json | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('config3 $mode $v', round(d['path_ms_per_launch'],4), round(d['roofline_frac_path'],4), 'last third', round(d['path_ms_last_third_mean'],4))"
    done
  done
done
