#!/bin/bash
# BASELINE config 3 STREAMING, ablation of wtile_kernel<7> on the measurement build (FA_ABLATE: results are wrong by design):
# everything / no distinct-set probes / no sketch updates / neither / no hot-address cache / no sink at all (DMA + parse).
# One rocprofv3 pass per flag set with FETCH_SIZE (kernel durations come with it), WRITE_SIZE for the two that matter.
#   make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_ABLATE=1     (built here or shipped with the snapshot)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/config3_ablation
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
[ -f flow-pipeline_amd/libflowagg_ablate.so ] || make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_ABLATE=1 > /dev/null
REC=${RECORDS:-200000000}
for f in ${FLAGS:-0 262144 524288 786432 1048576 1}; do
  for c in FETCH_SIZE $( [ $f = 0 -o $f = 262144 ] && echo WRITE_SIZE ); do
    FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/f${f}_$c -o p -- \
      python tools/config3_run.py --records $REC --timing-only --no-assert > $OUT/f${f}_$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob, os, collections, statistics
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out/config3_ablation")
names = {"0": "everything", "262144": "no distinct-set probes / inserts", "524288": "no sketch updates", "786432": "neither", "1048576": "no hot-address cache", "1": "no sink at all (DMA + parse)"}
print("%-36s %-12s %10s %10s %10s %12s" % ("flags", "kernel", "median us", "last-third", "launches", "counter MB"))
for d in sorted(glob.glob(root + "/f*_*/")):
    tag = os.path.basename(d.rstrip("/"))
    flag, counter = tag[1:].split("_", 1)
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            k = r["Kernel_Name"]
            for short in ("wtile_kernel", "cms_agg_kernel", "agg8_kernel"):
                if short in k:
                    dur[short].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    val = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for short in ("wtile_kernel", "cms_agg_kernel", "agg8_kernel"):
                if short in r["Kernel_Name"]:
                    val[short].append(float(r["Counter_Value"]) * 1024 / 1e6)
    for short in ("wtile_kernel", "cms_agg_kernel", "agg8_kernel"):
        v = dur.get(short)
        if not v:
            continue
        tail = v[-max(len(v) // 3, 1):]
        cv = val.get(short, [])
        ctail = cv[-max(len(cv) // 3, 1):] if cv else []
        print("%-36s %-12s %10.1f %10.1f %10d %12s" % (names.get(flag, flag) if short == "wtile_kernel" else "", short[:12], statistics.median(v), statistics.mean(tail), len(v),
                                                     ("%s raw %.1f (last third)" % (counter[:5], statistics.mean(ctail))) if ctail else "-"))
PY
