#!/bin/bash
# Round 4, GPU session 1: the whole GPU suite on the new window-close / wide-log code, the driver's bench line, config 5 at
# 100 M records with the phase times of every read (FA_VERBOSE).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s1
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1
tail -15 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
cut -c1-1800 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
cut -c1-3000 $OUT/config5_100M.json; grep "flowagg read" $OUT/config5_100M.err | tail -30; tail -5 $OUT/config5_100M.err
