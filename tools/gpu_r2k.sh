#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2k
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python tools/config3_run.py ) > $OUT/config3_1B.json 2> $OUT/config3_1B.err
tail -c 2500 $OUT/config3_1B.json; tail -5 $OUT/config3_1B.err
