#!/bin/bash
# Round 4, GPU session 6: device-side framing with plausible guesses: its tests, its rate.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s6
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_wide_log_framing_gpu.py tests/test_gpu_parity.py -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
FA_VERBOSE=1 timeout 300 python tools/framing_rate.py > $OUT/framing_rate.json 2> $OUT/framing_rate.err; cat $OUT/framing_rate.json; grep framing $OUT/framing_rate.err | sort | uniq -c | head
