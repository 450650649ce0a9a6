#!/bin/bash
# round 6, GPU session 26: kernel durations of config 3 at 16.67 M and 33.3 M records per launch (what is the per-launch fixed cost made of?)
O=gpurun_out/s26
mkdir -p $O
for mode in exact candidates; do
  for chunk in 16666667 33333334; do
    PROF_PASSES=trace PROF_CMD="python tools/config3_run.py --records 200000000 --chunk $chunk --timing-only --topk-mode $mode" bash tools/profile.sh r06_c3_${mode}_$chunk > $O/prof_${mode}_$chunk.log 2>&1
    head -16 gpurun_out/prof/r06_c3_${mode}_$chunk/summary.txt | cut -c1-150 > $O/summary_${mode}_$chunk.txt
    echo "== $mode $chunk"; grep -v "gen_\|mock_\|rocprim\|rocclr" $O/summary_${mode}_$chunk.txt | head -12
  done
done
