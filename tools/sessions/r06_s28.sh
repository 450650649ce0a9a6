#!/bin/bash
# round 6, GPU session 28: the flows_5m tuple aggregation on the side stream in the EXACT mode as well (FA_AGG_SIDE=2, measurement
# switch in libflowagg_sidex.so), config 3 exact at both launch sizes, same box, 3 x
O=gpurun_out/s28
mkdir -p $O
export FA_LIB_VARIANT=sidex
for rep in 1 2 3; do
  for chunk in 33333334 16666667; do
    for v in line side; do
      if [ $v = side ]; then export FA_AGG_SIDE=2; else unset FA_AGG_SIDE; fi
      python tools/config3_run.py --records 200000000 --chunk $chunk --timing-only --topk-mode exact 2>$O/err_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'exp':'exact_mode_agg_side_stream','agg':'$v','chunk':$chunk,'rep':$rep,'path_ms_per_launch':d['path_ms_per_launch'],'frac':d['roofline_frac_path'],'last_third':d.get('path_ms_last_third_mean')}))" >> $O/exp_exact_side.jsonl
    done
  done
done
cat $O/exp_exact_side.jsonl
