#!/bin/bash
# Round 4, session 11 (final sources): config 3 at full scale (1 B records streaming) and the rocprofv3 evidence of its streaming run.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s11
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 1500 python tools/config3_run.py > $OUT/config3_1B.json 2> $OUT/config3_1B.err; echo "config3 rc=$?"
PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only" timeout 900 bash tools/profile.sh r04_config3_stream > $OUT/profile_config3_stream.log 2>&1
grep '^{' $OUT/config3_1B.json | tail -1 | cut -c1-1500; tail -2 $OUT/config3_1B.err
head -10 $ROOT/gpurun_out/prof/r04_config3_stream/summary.txt; grep -A5 "FETCH_SIZE, per launch" $ROOT/gpurun_out/prof/r04_config3_stream/summary.txt | head -6; grep -A5 "WRITE_SIZE, per launch" $ROOT/gpurun_out/prof/r04_config3_stream/summary.txt | head -6
