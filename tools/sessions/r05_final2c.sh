#!/bin/bash
# final sources (c): the group close of 8 contexts at 200 M records in both top-k contracts, the C++ host with 8 partitions at 48 M
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05final2
mkdir -p $OUT
cd $ROOT
for mode in exact candidates; do
  timeout 600 python tools/group_run.py --topk-mode $mode > $OUT/group_8ctx_$mode.json 2> $OUT/group_8ctx_$mode.err; echo "group $mode rc=$?"
  grep '^{' $OUT/group_8ctx_$mode.json | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if isinstance(v,bool)}, d.get('topk100_both_sketches_again_ms'), d.get('close_oldest_window_both_key_sets_ms'))"
done
timeout 600 python tools/host_group_run.py > $OUT/host_group_8parts.json 2> $OUT/host_group_8parts.err; echo "host rc=$?"; grep '^{' $OUT/host_group_8parts.json | tail -1 | cut -c1-900
