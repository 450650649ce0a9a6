#!/bin/bash
# Round 5, session 1: the suite with the new group close (ABI 7) and the top-k pre-selection, the bench line with the new parity /
# clocks fields, the 2-rank dry run with the preflight, top-k timing on config 3's sets (kernel trace).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s1
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $OUT/bench_gpus2_shared.json 2> $OUT/bench_gpus2_shared.err; echo "bench2 rc=$?"
PROF_PASSES="trace" PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only" timeout 900 bash tools/profile.sh r05_config3_stream > $OUT/profile_config3_stream.log 2>&1
head -30 $ROOT/gpurun_out/prof/r05_config3_stream/summary.txt
grep '^{' $ROOT/gpurun_out/prof/r05_config3_stream/trace.log | tail -1 | cut -c1-1500
python - $OUT/bench_default.json $OUT/bench_gpus2_shared.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        r=d["roofline"]; k=r.get("dominant_kernel") or {}
        print(f.split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")))
        print("  parity", json.dumps(d.get("parity"))[:600]); print("  clocks", d.get("clocks")); print("  preflight", d.get("preflight"))
    except Exception as e:
        print("ERR", f, e, open(f.replace(".json",".err")).read()[-1500:])
PY
du -sh $ROOT/gpurun_out
