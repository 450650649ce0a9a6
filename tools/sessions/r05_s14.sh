#!/bin/bash
# soak: random sessions of a group against the numpy model
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s14
mkdir -p $OUT
cd $ROOT
FA_FUZZ_SEEDS=${FA_FUZZ_SEEDS:-60} timeout 1500 python -m pytest tests/test_group_gpu.py -q -m gpu -k random_sessions -x > $OUT/soak_group.log 2>&1; echo "soak rc=$?"; tail -40 $OUT/soak_group.log | cut -c1-500
