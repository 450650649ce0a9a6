#!/bin/bash
# Round 5, session 2: the new top-k tests (pre-selection, candidates mode), config 3 streaming in both top-k modes with kernel
# traces, the FETCH / WRITE calibration, the ablation of wtile_kernel<7> on the streaming run.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_topk_gpu.py tests/test_group_gpu.py tests/test_gpu_parity.py -m gpu -q -x ) > $OUT/pytest.log 2>&1
tail -8 $OUT/pytest.log
for mode in exact candidates; do
  PROF_PASSES="trace" PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode $mode" timeout 600 bash tools/profile.sh r05_config3_stream_$mode > $OUT/profile_config3_$mode.log 2>&1
  head -16 $ROOT/gpurun_out/prof/r05_config3_stream_$mode/summary.txt
  grep '^{' $ROOT/gpurun_out/prof/r05_config3_stream_$mode/trace.log | tail -1 | cut -c1-1400
done
timeout 600 bash tools/fetch_calib.sh > $OUT/fetch_calib.log 2>&1; tail -14 $OUT/fetch_calib.log
timeout 1500 bash tools/config3_ablation.sh > $OUT/config3_ablation.txt 2>&1; cat $OUT/config3_ablation.txt | tail -40
du -sh $ROOT/gpurun_out
