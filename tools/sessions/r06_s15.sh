#!/bin/bash
# NOTE: the code of this experiment was taken out of the tree after it was measured (CHANGELOG round 6 says what it was); the script documents the runs behind the jsonl in profiles/.
# round 6, GPU session 15: wave priority in the wave-tile kernel (s_setprio 1 / 3 while a wave holds a staged tile, 0 once its next DMA is out)
# libs: make OUT=../libflowagg_prio{1,3}.so EXTRA=-DFA_SETPRIO={1,3}
O=gpurun_out/s15
mkdir -p $O
B="python bench.py --steps 10 --warmup 2 --settle-max-steps 30 --cpu-sample 0 --no-host-fed --no-secondary"
for rep in 1 2 3; do
  for v in prod prio1 prio3; do
    if [ $v = prod ]; then unset FA_LIB_VARIANT; else export FA_LIB_VARIANT=$v; fi
    for args in "" "--mode goflow" "--mode zipf --key-sets 7 --records 50000000 --chunk 16666667"; do
      $B $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'setprio','lib':'$v','args':'$args','rep':$rep,'path_ms':round(r['avg_launch_ms'],4),'wtile_ms':round(r['dominant_kernel']['avg_launch_ms'],4),'frac':round(r['frac'],4),'parity':d.get('parity',{}).get('ok')}))" >> $O/exp_setprio.jsonl
    done
  done
done
unset FA_LIB_VARIANT
cat $O/exp_setprio.jsonl
