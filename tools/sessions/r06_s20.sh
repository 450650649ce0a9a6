#!/bin/bash
# round 6, GPU session 20: where the time goes on the GoFlow-shaped stream and on Zipf AS pairs (measurement build, ablation flags:
# 0 everything, 32 no tuple stores, 1 no sink, 16 no parse (DMA only), 4 no LDS hot-key table)
O=gpurun_out/s20
mkdir -p $O
[ -f flow-pipeline_amd/libflowagg_ablate.so ] || make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_ABLATE=1 > /dev/null
B="python bench.py --steps 10 --warmup 2 --settle-max-steps 30 --cpu-sample 0 --no-verify --no-host-fed --no-secondary --no-assert"
for rep in 1 2; do
  for mode in goflow zipf; do
    for f in 0 32 1 17 4; do
      FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f $B --mode $mode 2>$O/err_${mode}_$f.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'shape_ablation','mode':'$mode','flags':$f,'rep':$rep,'launches_per_step':d['config']['launches_per_step'],'ms_per_step':d['ms_per_step'],'path_ms_per_launch':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms'],'kernel':r['dominant_kernel'].get('name'),'frac':r['frac']}))" >> $O/exp_shape_ablation.jsonl
    done
  done
done
cat $O/exp_shape_ablation.jsonl
