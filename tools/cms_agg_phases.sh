# cms_agg_kernel: per-workgroup phase times (ablate build) + kernel trace of the config-3 shape
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab8; mkdir -p $O
A="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify --mode zipf --key-sets 7 --records 50000000 --chunk 16666667"
FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=4194304 FA_VERBOSE=1 python bench.py $A --no-assert > $O/timing.json 2> $O/timing.err
grep "flowagg" $O/timing.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py $A > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/ab8/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-60s n %4d avg %8.1f us  min %8.1f  med %8.1f  max %8.1f" % (k, len(v), sum(v) / len(v), v2[0], v2[len(v2) // 2], v2[-1]))
PY
find $O -type f -size +2M -delete
