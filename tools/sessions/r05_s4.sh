#!/bin/bash
# Round 5, session 4: candidates mode batch by batch against the oracle (the failing direct-sink case), the cost of the candidate
# test itself (measurement build with and without the sets' code), the 48-byte window rows + reserved read buffers on config 5.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s4
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 300 python tools/debug_candidates.py 60000 12 12 12 32 10 > $OUT/debug_candidates.txt 2>&1; tail -30 $OUT/debug_candidates.txt
( timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_wide_keysets_gpu.py tests/test_host_inserter.py -m gpu -q ) > $OUT/pytest.log 2>&1; grep -n "passed\|failed" $OUT/pytest.log
for f in 0 262144; do
  FA_LIB_VARIANT=ablate FA_DEBUG_FLAGS=$f timeout 300 python tools/config3_run.py --records 300000000 --timing-only --no-assert --topk-mode candidates --topk-track 16 > $OUT/cand_ablate_f$f.json 2> $OUT/cand_ablate_f$f.err
  python - $OUT/cand_ablate_f$f.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1].split("/")[-1], "path last third %.4f ms" % d["path_ms_last_third_mean"], d["path_ms_series"][:5])
PY
done
for extra in "" "--rows48" "--pinned-out" "--pinned-out --rows48"; do
  FA_VERBOSE=1 timeout 600 python tools/config5_run.py $extra > $OUT/config5_$(echo $extra | tr -d ' -').json 2> $OUT/config5_$(echo $extra | tr -d ' -').err; echo "config5 [$extra] rc=$?"
  python - $OUT/config5_$(echo $extra | tr -d ' -').json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("  read", d.get("read_app_windows_ms"), "close", d.get("close_app_windows_ms"), d.get("per_window_ms"), d.get("row_format"), d.get("row_buffer"))
PY
done
du -sh $ROOT/gpurun_out
