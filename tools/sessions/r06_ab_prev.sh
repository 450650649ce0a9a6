#!/bin/bash
# Same-box A/B of the headline path: the library built from the sources of 6033917 (libflowagg_prev.so, built by hand) against the current one.
O=gpurun_out/r06_ab_prev; mkdir -p $O; : > $O/ab.jsonl
for rep in 1 2 3 4; do
  for v in prev cur; do
    if [ $v = prev ]; then export FA_LIB_VARIANT=prev; else unset FA_LIB_VARIANT; fi
    python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-verify --no-host-fed --no-secondary 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.readline()); r=b['roofline']
print(json.dumps({'lib':'$v','rep':$rep,'G_per_s':round(b['value']/1e9,2),'ms_per_step':round(b['ms_per_step'],4),'path_frac':round(r['frac'],4),'kernel_ms':r.get('kernel_ms'),'path_ms':r.get('path_ms')}))" | tee -a $O/ab.jsonl
  done
done
