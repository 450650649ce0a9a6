#!/bin/bash
# Round 4, GPU session 5: the suite on the cleaned-up sources + device-side framing; its rate.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s5
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 300 python tools/framing_rate.py > $OUT/framing_rate.json 2> $OUT/framing_rate.err; cat $OUT/framing_rate.json; tail -3 $OUT/framing_rate.err
