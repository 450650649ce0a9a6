#!/bin/bash
# round 6, GPU session 19: do the tuples' HBM writes go away when the tuple stores land in a window that fits the Infinity Cache?
# (measurement builds libflowagg_tlw<W>.so: -DFA_ABLATE=1 -DFA_TL_WINDOW=<W>u; FA_DEBUG_FLAGS=67108864 = DBG_TUPLE_LOCAL: the stores of
# full bins go to a window of W uint4 per workgroup - 8 / 32 / 128 / 256 MB over the 512 workgroups; results are wrong by design)
O=gpurun_out/s19
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --settle-max-steps 30 --cpu-sample 0 --no-verify --no-host-fed --no-secondary --no-assert"
for rep in 1 2 3; do
  for v in tlw1024:0 tlw1024:67108864 tlw4096:67108864 tlw16384:67108864 tlw32768:67108864 tlw1024:32; do
    lib=${v%%:*}; f=${v##*:}
    FA_LIB_VARIANT=$lib FA_DEBUG_FLAGS=$f $B 2>$O/err_${lib}_$f.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'tuple_store_window','lib':'$lib','flags':$f,'rep':$rep,'path_ms_per_launch':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms']}))" >> $O/exp_tuple_store_window.jsonl
  done
done
cat $O/exp_tuple_store_window.jsonl
