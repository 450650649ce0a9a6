#!/bin/bash
# NOTE: the code of this experiment was taken out of the tree after it was measured (CHANGELOG round 6 says what it was); the script documents the runs behind the jsonl in profiles/.
# round 6, GPU session 8: how long a wave keeps offering records to the LDS hot-key table (FA_LT_KEEP: 1 hit in 8 / 32 / 64 of its
# last 256 records) - skewed AS pairs leave agg8_kernel with one partition 2.6x the mean when the hot pairs all become tuples
O=gpurun_out/s8
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --settle-max-steps 30 --cpu-sample 0 --no-host-fed --no-secondary"
for rep in 1 2; do
  for v in prod keep32 keep64; do
    if [ $v = prod ]; then unset FA_LIB_VARIANT; else export FA_LIB_VARIANT=$v; fi
    for args in "--mode zipf --key-sets 1" "" "--mode mocker" "--mode goflow" "--mode zipf --key-sets 7 --records 50000000 --chunk 16666667"; do
      $B $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'lt_keep','lib':'$v','args':'$args','rep':$rep,'path_ms':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms'],'frac':r['frac'],'parity':d.get('parity',{}).get('ok')}))" >> $O/exp_lt_keep.jsonl
    done
  done
done
unset FA_LIB_VARIANT
cat $O/exp_lt_keep.jsonl
