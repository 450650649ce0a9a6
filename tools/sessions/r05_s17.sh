#!/bin/bash
# A/B on one box: the wide-tuple bucket-range bookkeeping in the ingest kernel (config 5 key sets, ingest only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s17
mkdir -p $OUT
cd $ROOT
make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_NO_WRANGE=1 > $OUT/build_norange.log 2>&1; echo "variant build rc=$?"
for rep in 1 2 3; do
  for v in prod ablate; do
    if [ $v = ablate ]; then export FA_LIB_VARIANT=ablate; else unset FA_LIB_VARIANT; fi
    python bench.py --mode zipf --key-sets 9 --steps 12 --warmup 3 --cpu-sample 0 --no-verify --no-host-fed 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', round(d['ms_per_step'],4), round(d['roofline'].get('path_ms', 0),4) if isinstance(d.get('roofline'),dict) else '', d['roofline'].get('frac'))"
  done
done
unset FA_LIB_VARIANT
rm -f flow-pipeline_amd/libflowagg_ablate.so
