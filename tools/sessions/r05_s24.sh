#!/bin/bash
# fa_read_window_app48 in two halves (copy of the first overlaps the sort of the second): tests, config 5 with page-locked 48-byte rows
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s24
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_wide_keysets_gpu.py -q -m gpu -x 2>&1 | tail -12
FA_VERBOSE=1 timeout 600 python tools/config5_run.py --pinned-out --rows48 > $OUT/config5_split.json 2> $OUT/config5_split.err; echo "config5 rc=$?"
grep "flowagg read" $OUT/config5_split.err | grep "SrcAddr" | cut -c1-420 | head -4
grep '^{' $OUT/config5_split.json | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d.get('read_app_windows_ms'), d.get('close_app_windows_ms'), d.get('per_window_ms'), {k:v for k,v in d.items() if 'exact' in k or 'returns' in k})"
FA_APP48_SPLIT=0 timeout 600 python tools/config5_run.py --pinned-out --rows48 2>/dev/null | grep '^{' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('one piece:', d.get('read_app_windows_ms'), d.get('close_app_windows_ms'))"
