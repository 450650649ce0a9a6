#!/bin/bash
# NOTE: the code of this experiment was taken out of the tree after it was measured (CHANGELOG round 6 says what it was); the script documents the runs behind the jsonl in profiles/.
# premise check: full store units appended to one sequential log per workgroup (measurement build, results wrong by design)
O=gpurun_out/s16
mkdir -p $O
B="python bench.py --steps 8 --warmup 2 --settle-max-steps 20 --cpu-sample 0 --no-verify --no-host-fed --no-assert --no-secondary"
for rep in 1 2 3; do
  for f in 0 536870912 32 67108864; do
    FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'sequential_tuple_log','flags':$f,'rep':$rep,'path_ms_per_launch':round(r['avg_launch_ms'],4),'wtile_ms':round(r['dominant_kernel']['avg_launch_ms'],4)}))" >> $O/exp_seq_log.jsonl
  done
done
cat $O/exp_seq_log.jsonl
