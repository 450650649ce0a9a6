#!/bin/bash
# Round 4, session 9: window reads into a page-locked buffer of the caller's (one copy-engine transfer): the round-4 tests, config 5
# with a pageable and with a page-locked row buffer (adaptive sink), FA_VERBOSE phase times of both.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s9
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wide_log_framing_gpu.py -m gpu -q -x > $OUT/pytest_round4.log 2>&1; tail -5 $OUT/pytest_round4.log
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 pageable rc=$?"
FA_VERBOSE=1 timeout 900 python tools/config5_run.py --pinned-out > $OUT/config5_100M_pinned.json 2> $OUT/config5_100M_pinned.err; echo "config5 pinned rc=$?"
for f in config5_100M config5_100M_pinned; do echo "== $f"; grep '^{' $OUT/$f.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('row_buffer','read_app_windows_ms','close_app_windows_ms','per_window_ms','roofline_frac_path','closes_return_the_windows_read_before','app_count_equals_records','app_sum_bytes_equals_flows_5m','flows_5m_aligned_windows_bit_exact','sliding_window_bit_exact')})"; grep "SrcAddr" $OUT/$f.err | tail -4; done
