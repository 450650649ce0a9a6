#!/bin/bash
# round 6, GPU session 2: full GPU suite, the driver's bench line, launch-size experiment (tuples vs the 256 MB Infinity Cache),
# tuple-store ablation (HBM write traffic vs store instructions), side measurements
O=gpurun_out/s2
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
B="python bench.py --steps 10 --warmup 2 --settle-max-steps 30 --cpu-sample 0 --no-verify --no-host-fed"
for rep in 1 2; do
  for chunk in 33333334 16666667 8333334 4166667; do
    $B --chunk $chunk 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'launch_size','chunk':$chunk,'rep':$rep,'ms_per_step':d['ms_per_step'],'path_ms_per_launch':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms'],'launches_per_step':d['config']['launches_per_step'],'path_us_per_M_records':r['avg_launch_ms']*1e3/($chunk/1e6),'frac_path':r['frac']}))" >> $O/exp_launch_size.jsonl
  done
done
cat $O/exp_launch_size.jsonl
for rep in 1 2; do
  for f in 0 67108864 32 1; do
    FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f $B --no-assert 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'tuple_store_ablation','flags':$f,'rep':$rep,'path_ms_per_launch':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms']}))" >> $O/exp_tuple_store.jsonl
  done
done
cat $O/exp_tuple_store.jsonl
for args in "--mode mocker" "--mode goflow" "--mode reversed" "--mode zipf --key-sets 1" "--mode zipf --key-sets 7 --records 50000000 --chunk 16666667" "--mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667" "--stage decode"; do
  $B $args 2>/dev/null >> $O/side_measurements.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/s2/side_measurements.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["config"]["workload"][:90], "| frac", round(r["frac"], 4), "| kernel", round(r.get("dominant_kernel", {}).get("frac", 0), 4), "| G rec/s", round(d["value"] / 1e9, 2))
PY
# config 3, same box: the library before / after "probe answers behind the next DMA" (FA_KS_LATE), both top-k modes
for rep in 1 2 3; do
  for mode in exact candidates; do
    for v in base new; do
      if [ $v = base ]; then export FA_LIB_VARIANT=base; else unset FA_LIB_VARIANT; fi
      python tools/config3_run.py --records 200000000 --timing-only --topk-mode $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({'exp':'ks_late','lib':'$v','mode':'$mode','rep':$rep,'path_ms_per_launch':d['path_ms_per_launch'],'last_third':d['path_ms_last_third_mean'],'frac':d['roofline_frac_path']}))" >> $O/exp_ks_late.jsonl
    done
  done
done
unset FA_LIB_VARIANT
cat $O/exp_ks_late.jsonl
