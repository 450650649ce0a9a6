#!/bin/bash
# round 6, GPU session 21: same-box A/B of the GoFlow template walk's pair steps (libflowagg_base.so = the library before them)
O=gpurun_out/s21
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_parity.txt
tail -3 $O/pytest_parity.txt
B="python bench.py --steps 10 --warmup 2 --settle-max-steps 30 --cpu-sample 0 --no-host-fed --no-secondary"
for rep in 1 2 3; do
  for mode in goflow aspairs; do
    for v in base new; do
      if [ $v = base ]; then export FA_LIB_VARIANT=base; else unset FA_LIB_VARIANT; fi
      $B --mode $mode 2>$O/err_${mode}_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'goflow_pair_steps','lib':'$v','mode':'$mode','rep':$rep,'ms_per_step':d['ms_per_step'],'path_ms_per_launch':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms'],'frac':r['frac'],'kernel_frac':r['dominant_kernel'].get('frac'),'parity':d.get('parity',{}).get('ok')}))" >> $O/exp_goflow_pair_steps.jsonl
    done
  done
done
unset FA_LIB_VARIANT
cat $O/exp_goflow_pair_steps.jsonl
