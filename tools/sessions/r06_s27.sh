#!/bin/bash
# round 6, GPU session 27: same-box A/B of the launch boundary's two kernels with four loads in flight (libflowagg_base.so = before),
# config 3 candidates mode at both launch sizes; the top-k / sketch tests on the new library; kernel durations
O=gpurun_out/s27
mkdir -p $O
python -m pytest tests/test_topk_gpu.py tests/test_bench_secondary_gpu.py -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt; tail -2 $O/pytest.txt
for rep in 1 2 3; do
  for chunk in 33333334 16666667; do
    for v in base new; do
      if [ $v = base ]; then export FA_LIB_VARIANT=base; else unset FA_LIB_VARIANT; fi
      python tools/config3_run.py --records 200000000 --chunk $chunk --timing-only --topk-mode candidates 2>$O/err_$v.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'exp':'cand_boundary_loads_in_flight','lib':'$v','chunk':$chunk,'rep':$rep,'path_ms_per_launch':d['path_ms_per_launch'],'frac':d['roofline_frac_path']}))" >> $O/exp_cand_boundary.jsonl
    done
  done
done
unset FA_LIB_VARIANT
cat $O/exp_cand_boundary.jsonl
PROF_PASSES=trace PROF_CMD="python tools/config3_run.py --records 200000000 --timing-only --topk-mode candidates" bash tools/profile.sh r06_c3_cand_new > $O/prof.log 2>&1
grep "cand_\|cms_agg\|agg8\|wtile" gpurun_out/prof/r06_c3_cand_new/summary.txt | head -6 | cut -c1-140
