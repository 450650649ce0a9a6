#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2j
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 600 python tools/config3_run.py --records 50000000 --prefix 16000000 --universe-log2 20 ) > $OUT/config3_small.json 2> $OUT/config3_small.err
tail -c 1500 $OUT/config3_small.json; tail -5 $OUT/config3_small.err
( time timeout 900 python tools/config5_run.py --records 20000000 --span 1800 ) > $OUT/config5_small.json 2> $OUT/config5_small.err
tail -c 1500 $OUT/config5_small.json; tail -5 $OUT/config5_small.err
