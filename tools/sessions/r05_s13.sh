#!/bin/bash
# soak: the random-configuration differential test (now with top-k of both contracts and mutated records) over many seeds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s13
mkdir -p $OUT
cd $ROOT
FA_FUZZ_SEEDS=${FA_FUZZ_SEEDS:-120} timeout 1500 python -m pytest tests/test_ingest_sinks_gpu.py -q -m gpu -k random_configurations -x > $OUT/soak.log 2>&1; echo "soak rc=$?"; tail -30 $OUT/soak.log | cut -c1-600
