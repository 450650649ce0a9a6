#!/bin/bash
# round 6, GPU session 22: the bench line with the group block reading into a page-locked row buffer; the bench-block tests;
# a last soak of the two differential tests on the final sources (1000 + 1000 seeds)
O=gpurun_out/s22
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s22/bench_default.json").read().strip().splitlines()[-1])
g = d["secondary"]["group8"]
print("headline", round(d["value"] / 1e9, 2), d["roofline"]["frac"], d["roofline"]["dominant_kernel"]["frac"], "secondary ok", d["secondary"]["ok"])
print("group8", g.get("row_buffer"), g["read_app_window_partitioned_ms"], g["ok"])
PY
python -m pytest tests/test_bench_secondary_gpu.py -m gpu -q > $O/pytest_blocks.txt 2>&1; tail -2 $O/pytest_blocks.txt
FA_FUZZ_SEEDS=1000 timeout 3000 python -m pytest tests/test_ingest_sinks_gpu.py tests/test_group_gpu.py -q -m gpu -k "random_configurations or random_sessions" -p no:cacheprovider > $O/soak.log 2>&1
echo "soak rc=$?"; tail -3 $O/soak.log | cut -c1-300
