#!/bin/bash
# config 5: the first window read's phases (collect now reports settle / chunk ranges / count / extract), scratch reserved at ingest time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s15
mkdir -p $OUT
cd $ROOT
for extra in "--pinned-out --rows48" ""; do
  tag=$(echo $extra | tr -d ' -'); tag=${tag:-plain}
  FA_VERBOSE=1 timeout 600 python tools/config5_run.py $extra > $OUT/config5_$tag.json 2> $OUT/config5_$tag.err; echo "config5 [$extra] rc=$?"
  grep "flowagg read" $OUT/config5_$tag.err | grep "SrcAddr" | cut -c1-330 | head -14
  grep '^{' $OUT/config5_$tag.json | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d.get('read_app_windows_ms'), d.get('close_app_windows_ms'), d.get('per_window_ms'))"
done
python -m pytest tests/test_wide_log_framing_gpu.py tests/test_wide_keysets_gpu.py tests/test_window_close_gpu.py -q -m gpu -x 2>&1 | tail -2
