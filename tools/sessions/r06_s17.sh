#!/bin/bash
# round 6, GPU session 17: the flows_5m tuple aggregation on a side stream beside the sketch fold (+ the candidates boundary): FA_AGG_SIDE=0 in line, 1 side by side
O=gpurun_out/s17
mkdir -p $O
python -m pytest tests/test_topk_gpu.py tests/test_group_gpu.py -m gpu -q -x > $O/pytest_topk.txt 2>&1; tail -3 $O/pytest_topk.txt
for rep in 1 2 3; do
  for side in 0 1; do
    for mode in candidates exact; do
      FA_AGG_SIDE=$side python tools/config3_run.py --records 200000000 --timing-only --topk-mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'exp':'agg_side_stream','side':$side,'mode':'$mode','rep':$rep,'path_ms_per_launch':round(d['path_ms_per_launch'],4),'last_third':round(d['path_ms_last_third_mean'],4),'frac':round(d['roofline_frac_path'],4)}))" >> $O/exp_agg_side.jsonl
    done
  done
done
cat $O/exp_agg_side.jsonl
