#!/bin/bash
# config 3: a kernel per top-k contract (compile-time) against the one kernel with the contract in its arguments (FA_CAND_VARIANTS=0), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s21
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  for mode in exact candidates; do
    for v in 1 0; do
      FA_CAND_VARIANTS=$v timeout 600 python tools/config3_run.py --records 200000000 --timing-only --topk-mode $mode 2>$OUT/err.txt | tee $OUT/c3_${mode}_v${v}_$rep.json | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$mode variants=$v', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k or 'frac' in k})"
    done
  done
done
python -m pytest tests/test_topk_gpu.py -q -m gpu -x 2>&1 | tail -2
