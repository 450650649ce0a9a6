#!/bin/bash
# the C++ host program with 8 partitions through the group close, at scale (48 M records)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s12
mkdir -p $OUT
cd $ROOT
timeout 900 python tools/host_group_run.py > $OUT/host_group_8parts.json 2> $OUT/host_group_8parts.err; echo "host rc=$?"; grep '^{' $OUT/host_group_8parts.json | tail -1 | cut -c1-1800; tail -5 $OUT/host_group_8parts.err | cut -c1-400
