#!/bin/bash
# Instruction counts of the wave-tile kernel per section, by ablation (FA_DEBUG_FLAGS): one PMC pass per flag set.
# The switches are compiled out of the production library: build the measurement variant first
#   make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_ABLATE=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/inst
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for f in ${FLAGS:-0 16 1 17}; do
  FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/f$f -o p -- \
    python $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed --no-assert --no-secondary --settle-max-steps 4 $BENCH_ARGS > $OUT/f$f.log 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob, os, collections, statistics
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out/inst")
for d in sorted(glob.glob(root + "/f*/")):
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wtile" in k or "agg8" in k or "cms_agg" in k or "wagg" in k:
                dur[k[:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(os.path.basename(d.rstrip("/")), "durations under PMC (us):", " | ".join("%s median %.1f min %.1f" % (k, statistics.median(v), min(v)) for k, v in sorted(dur.items())))
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wtile" not in k and "agg8" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
    for k in acc:
        print(os.path.basename(d.rstrip("/")), k[:40], "launches", n[k], " ".join("%s=%.1fM" % (c.replace("SQ_", ""), v / max(n[k], 1) / 1e6) for c, v in sorted(acc[k].items())))
PY
