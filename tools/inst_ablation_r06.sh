#!/bin/bash
# Round 6: what bounds the ingest kernels - SQ counters per ablation flag (FA_DEBUG_FLAGS on the measurement build, results wrong
# by design), two PMC passes per flag set (8 SQ slots per pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
#   A  instructions by kind + the cycles the VALU / VMEM / LDS pipes were issuing
#   B  where the waves' cycles went: WAVE_CYCLES ~ WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY
# WHAT=config2 (bench.py default workload, wtile_kernel<1>) or config3 (tools/config3_run.py, wtile_kernel<7> + cms_agg + agg8).
#   make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_ABLATE=1     (built here or shipped with the snapshot)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WHAT=${WHAT:-config2}
OUT=$ROOT/gpurun_out/inst_r06_$WHAT
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
[ -f flow-pipeline_amd/libflowagg_ablate.so ] || make -C flow-pipeline_amd/csrc OUT=../libflowagg_ablate.so EXTRA=-DFA_ABLATE=1 > /dev/null
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
if [ $WHAT = config2 ]; then
  FLAGS=${FLAGS:-0 32 1 17}
  CMD="python bench.py --steps 2 --warmup 1 --settle-max-steps 4 --cpu-sample 0 --no-verify --no-host-fed --no-assert --no-secondary"
else
  FLAGS=${FLAGS:-0 262144 524288 786432 1048576 1}
  CMD="python tools/config3_run.py --records ${RECORDS:-100000000} --timing-only --no-assert --topk-mode ${TOPK_MODE:-exact}"
fi
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counter_names.txt
# durations without counters (the PMC passes serialise and slow the kernels down)
for f in $FLAGS; do
  FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace -d $OUT/f${f}_T -o p -- $CMD > $OUT/f${f}_T.log 2>&1
  FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace --pmc $A -d $OUT/f${f}_A -o p -- $CMD > $OUT/f${f}_A.log 2>&1
  FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f rocprofv3 --output-format csv --kernel-trace --pmc $B -d $OUT/f${f}_B -o p -- $CMD > $OUT/f${f}_B.log 2>&1
done
WHAT=$WHAT python - <<'PY' | tee $OUT/summary.txt
import csv, glob, os, collections, statistics
what = os.environ["WHAT"]
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out/inst_r06_" + what)
names = {"config2": {"0": "everything", "32": "no tuple stores", "1": "no sink (DMA + parse)", "17": "no sink, no parse (DMA only)"},
         "config3": {"0": "everything", "262144": "no distinct-set probes / inserts", "524288": "no sketch updates", "786432": "neither", "1048576": "no hot-address cache",
                     "1": "no sink at all (DMA + parse)"}}[what]
kernels = ("wtile_kernel", "agg8_kernel", "cms_agg_kernel", "deferred_kernel")
def short(k):
    for s in kernels:
        if s in k:
            return s
    return None
for flag, label in names.items():
    dur = collections.defaultdict(list)
    for f in glob.glob("%s/f%s_T/**/*kernel_trace.csv" % (root, flag), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s:
                dur[s].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for p in "AB":
        for f in glob.glob("%s/f%s_%s/**/*counter_collection.csv" % (root, flag, p), recursive=True):
            for r in csv.DictReader(open(f)):
                s = short(r["Kernel_Name"])
                if s:
                    acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
                    n[s][r["Counter_Name"]] += 1
    print("== flags %s: %s" % (flag, label))
    for s in kernels:
        if s not in dur and s not in acc:
            continue
        d = dur.get(s, [0.0])
        # (the first launches size buffers / fill the sets: the median over the launches of the run)
        line = "  %-16s launches %3d  median %8.1f us  min %8.1f us" % (s, len(d), statistics.median(d), min(d))
        c = {k: v / max(n[s][k], 1) for k, v in acc[s].items()}  # per launch
        if c:
            line += "\n      per launch, M: " + " ".join("%s=%.2f" % (k.replace("SQ_", ""), v / 1e6) for k, v in sorted(c.items()))
            if c.get("SQ_BUSY_CYCLES") and c.get("SQ_WAVE_CYCLES"):
                wc = c["SQ_WAVE_CYCLES"]
                line += "\n      of the waves' cycles: parked (WAIT_ANY) %.1f %%, issue stall (WAIT_INST_ANY) %.1f %%, issuing (ACTIVE_INST_ANY) %.1f %%" % (
                    100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc)
            if c.get("SQ_INSTS_VALU") and c.get("SQ_ACTIVE_INST_VALU"):
                line += "\n      ACTIVE_INST_VALU / INSTS_VALU = %.2f cycles per VALU instruction (counter units); VALU : SALU : LDS : VMEM_RD : VMEM_WR = %.0f : %.0f : %.0f : %.0f : %.0f (M)" % (
                    c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] / 1e6, c.get("SQ_INSTS_SALU", 0) / 1e6, c.get("SQ_INSTS_LDS", 0) / 1e6,
                    c.get("SQ_INSTS_VMEM_RD", 0) / 1e6, c.get("SQ_INSTS_VMEM_WR", 0) / 1e6)
        print(line)
PY
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -type f -size +2M -delete 2>/dev/null
