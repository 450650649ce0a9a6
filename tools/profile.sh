#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun).  Three separate
# passes: kernel trace + stats, then one PMC pass per counter group (FETCH_SIZE and
# WRITE_SIZE do not fit one pass on gfx950).  Output: gpurun_out/prof/<tag>/...
TAG=${1:-r01}
shift
ARGS=${@:---steps 3 --warmup 1 --cpu-sample 0}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
cd $ROOT
find $OUT -name "*.csv" | head -20
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
