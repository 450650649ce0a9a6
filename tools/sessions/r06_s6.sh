#!/bin/bash
# round 6, GPU session 6: what the single-tuple stores cost (WRITE_SIZE and time), measurement build
O=gpurun_out/s6
mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --settle-max-steps 10 --cpu-sample 0 --no-verify --no-host-fed --no-assert --no-secondary"
for rep in 1 2; do
  for f in 0 134217728 32; do
    FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'exp':'single_tuple_stores','flags':$f,'rep':$rep,'path_ms_per_launch':r['avg_launch_ms'],'wtile_ms':r['dominant_kernel']['avg_launch_ms']}))" >> $O/exp_singles.jsonl
  done
done
for f in 0 134217728; do
  FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/w$f -o p -- $B > $O/w$f.log 2>&1
done
python - <<'PY' >> gpurun_out/s6/exp_singles.jsonl
import csv, glob, collections, json
for f in (0, 134217728):
    acc = collections.defaultdict(list)
    for p in glob.glob("gpurun_out/s6/w%d/**/*counter_collection.csv" % f, recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == "WRITE_SIZE":
                for k in ("wtile_kernel", "agg8_kernel"):
                    if k in r["Kernel_Name"]:
                        acc[k].append(float(r["Counter_Value"]))
    print(json.dumps({"exp": "single_tuple_stores_WRITE_SIZE_KiB_per_launch", "flags": f, **{k: sum(v) / max(len(v), 1) for k, v in acc.items()}, "launches": {k: len(v) for k, v in acc.items()}}))
PY
cat $O/exp_singles.jsonl
find $O -name "*.db" -delete; find $O -type f -size +2M -delete
