#!/bin/bash
# does the side stream of 8 contexts in one process cost the group close anything?  (FA_AGG_SIDE=0: no side stream is created)
O=gpurun_out/s18
mkdir -p $O
for rep in 1 2 3; do
  for side in 0 1; do
    FA_AGG_SIDE=$side python tools/group_run.py --topk-mode candidates 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'exp':'group_close_with_side_streams','side':$side,'rep':$rep,'read_app_windows_partitioned_ms':d['read_app_windows_partitioned_ms'],'read_5m':d['read_5m_windows_merged_ms'][0],'allreduce':d['allreduce_both_sketches_ms'],'topk_first':d['topk100_both_sketches_first_ms'],'ingest_device_path_ms_sum':round(d['device_path_ms_sum_over_members'],2)}))" >> $O/exp_group_side.jsonl
  done
done
cat $O/exp_group_side.jsonl
