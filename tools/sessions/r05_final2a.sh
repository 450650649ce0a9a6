#!/bin/bash
# Round 5, last sessions on the final sources (a): the suite, the hash-stamped traffic profile of the default bench, the bench line,
# the 2-rank dry runs, the seeded soaks
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05final2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed|^FAILED|^E  " $OUT/pytest.log | head
timeout 900 bash tools/profile.sh r05 > $OUT/profile_default.log 2>&1
cp $ROOT/gpurun_out/prof/r05/traffic.json $ROOT/profiles/r05_traffic.json   # (on the box: the bench below then carries the stamped traffic)
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $OUT/bench_gpus2_shared.json 2> $OUT/bench_gpus2_shared.err; echo "bench2 rc=$?"
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --strong --steps 3 --warmup 1 --records 40000000 --chunk 10000000 > $OUT/bench_gpus2_shared_strong.json 2> $OUT/bench_gpus2_shared_strong.err; echo "bench2 strong rc=$?"
FA_FUZZ_SEEDS=40 timeout 900 python -m pytest tests/test_ingest_sinks_gpu.py tests/test_group_gpu.py -q -m gpu -k "random_configurations or random_sessions" > $OUT/soak.log 2>&1; tail -2 $OUT/soak.log
grep -A5 "calibrated HBM" $ROOT/gpurun_out/prof/r05/summary.txt
python - $OUT/bench_default.json $OUT/bench_gpus2_shared.json $OUT/bench_gpus2_shared_strong.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    r=d["roofline"]; k=r["dominant_kernel"]
    print("bench value %.4g path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | parity %s | traffic %s | preflight %s" % (d["value"], r["avg_launch_ms"], r["frac"], k["avg_launch_ms"], k["frac"], d["parity"]["ok"], r["traffic"], (d.get("preflight") or {}).get("ok")))
PY
