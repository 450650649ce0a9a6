#!/bin/bash
# Round-3 A/B on one GPU box: library variants (flow-pipeline_amd/libflowagg_<v>.so; "" = the tree's library) on the
# same box, back to back: bench lines (config 2, mocker, goflow) and, with PMC=1, instruction counts of the ingest kernel.
# VARIANTS="base x"  MODES="aspairs mocker goflow"  PMC=1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-ab3}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
B="--steps ${STEPS:-10} --warmup 3 --cpu-sample 0 --no-host-fed --no-verify"
for rep in 1 2; do
for v in "" $VARIANTS; do
  name=${v:-new}
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  for m in ${MODES:-aspairs}; do
    extra=""; [ "$m" = "goflow" ] && extra="--records 50000000 --chunk 16666667"
    timeout 300 python bench.py $B --mode $m $extra > $OUT/bench_${name}_${m}_$rep.json 2> $OUT/bench_${name}_${m}_$rep.err
  done
done
done
if [ -n "$PMC" ]; then
  cd /tmp
  for v in "" $VARIANTS; do
    name=${v:-new}
    if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_$name -o p --output-format csv -- \
      python $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed > $OUT/pmc_$name.log 2>&1
  done
  cd $ROOT
fi
python - <<'PY'
import json, glob, os, csv, collections
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", os.environ.get("TAG", "ab3"))
for f in sorted(glob.glob(root + "/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]; k = r.get("dominant_kernel") or {}
        print("%-36s value %.4g  path %.4f ms frac %.4f | wtile %.4f ms frac %.4f | 2nd-chance %s" % (os.path.basename(f), d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms", 0), k.get("frac", 0), d["config"].get("records_second_chance_parser")))
    except Exception as e:
        print(os.path.basename(f), "ERR", e, open(f.replace(".json", ".err")).read()[-600:])
for d in sorted(glob.glob(root + "/pmc_*/")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "wtile" not in k and "agg8" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
    for k in acc:
        print(os.path.basename(d.rstrip("/")), k[:44], "launches", n[k], " ".join("%s=%.1fM" % (c.replace("SQ_", ""), v / max(n[k], 1) / 1e6) for c, v in sorted(acc[k].items())))
PY
