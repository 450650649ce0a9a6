#!/bin/bash
# what bounds agg8_kernel on skewed AS pairs: loads only / no probing path / no flush (measurement build), uniform beside it
O=gpurun_out/s11
mkdir -p $O
export TMPDIR=/tmp
for shape in "--mode zipf --key-sets 1" ""; do
for f in 0 64 256 128 8388608; do
  B="python bench.py --steps 4 --warmup 2 --settle-max-steps 6 --cpu-sample 0 --no-verify --no-host-fed --no-secondary --no-assert $shape"
  FA_VERBOSE=$([ $f = 8388608 ] && echo 1) FA_AGG_BALANCE=0 FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace -d $O/f$f -o p -- $B > $O/f$f.log 2>&1
  python - <<PY
import csv, glob, collections, statistics
dur = collections.defaultdict(list)
for f in glob.glob("$O/f$f/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("wtile_kernel", "agg8_kernel"):
            if k in r["Kernel_Name"]:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("shape [$shape] flags $f:", {k: (len(v), round(statistics.median(v), 1)) for k, v in dur.items()})
PY
  grep -h "agg8" $O/f$f.log | tail -2
  rm -rf $O/f$f
done
done
