#!/bin/bash
# Soak of the two differential tests on the frozen sources (tuple aggregation on the side stream in the candidates mode):
# 500 + 500 seeds, half of them draw the candidates contract.  No source changes; output: gpurun_out/r06_soak2/soak.log
OUT=gpurun_out/r06_soak2; mkdir -p $OUT
python -c "import importlib; fa = importlib.import_module('flow-pipeline_amd'); print('stale', fa.stale() if hasattr(fa, 'stale') else None)" > $OUT/stale.log 2>&1
FA_FUZZ_SEEDS=500 timeout 2000 python -m pytest tests/test_ingest_sinks_gpu.py tests/test_group_gpu.py -q -m gpu -k "random_configurations or random_sessions" -p no:cacheprovider > $OUT/soak.log 2>&1
echo "soak rc=$?"; tail -5 $OUT/soak.log | cut -c1-600
