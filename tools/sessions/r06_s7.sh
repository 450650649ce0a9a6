#!/bin/bash
# round 6, GPU session 7: what narrower sketch tuples could buy - the HBM writes of the sketch tuples taken away (measurement build)
O=gpurun_out/s7
mkdir -p $O
for rep in 1 2 3; do
  for mode in exact candidates; do
    for f in 0 268435456; do
      FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f python tools/config3_run.py --records 200000000 --timing-only --no-assert --topk-mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'exp':'sketch_tuple_writes','flags':$f,'mode':'$mode','rep':$rep,'path_ms_per_launch':d['path_ms_per_launch'],'last_third':d['path_ms_last_third_mean']}))" >> $O/exp_sketch_tuple_writes.jsonl
    done
  done
done
cat $O/exp_sketch_tuple_writes.jsonl
