#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
( timeout 900 python -m pytest tests/test_topk_gpu.py -m gpu -q ) > gpurun_out/r05s11_pytest.log 2>&1
grep -E "passed|failed|^FAILED|^E  " gpurun_out/r05s11_pytest.log | head -30
