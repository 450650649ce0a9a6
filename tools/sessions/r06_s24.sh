#!/bin/bash
# round 6, GPU session 24: config 3 with 33.3 M-record launches against 16.67 M (the per-launch fixed costs - sketch flush, boundary
# kernels, kernel tails - over twice the records), same box, both top-k modes, 3 x
O=gpurun_out/s24
mkdir -p $O
for rep in 1 2 3; do
  for mode in exact candidates; do
    for chunk in 16666667 33333334 25000000; do
      python tools/config3_run.py --records 200000000 --chunk $chunk --timing-only --topk-mode $mode 2>$O/err_${mode}_$chunk.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'exp':'config3_launch_size','mode':'$mode','chunk':$chunk,'rep':$rep,'launches':d.get('launches'),'path_ms_per_launch':d['path_ms_per_launch'],'frac':d['roofline_frac_path'],'last_third':d.get('path_ms_last_third_mean')}))" >> $O/exp_config3_launch_size.jsonl
    done
  done
done
cat $O/exp_config3_launch_size.jsonl
