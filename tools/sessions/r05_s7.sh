#!/bin/bash
# what do the addresses above the threshold cost in candidates mode?  measurement build (results wrong by design), same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s7
mkdir -p $OUT
cd $ROOT
run() { tag=$1; shift; FA_LIB_VARIANT=ablate timeout 300 python tools/config3_run.py --records 400000000 --timing-only --no-assert "$@" > $OUT/$tag.json 2> $OUT/$tag.err
  python - $OUT/$tag.json $tag <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-34s last third %.4f ms   first four %s" % (sys.argv[2], d["path_ms_last_third_mean"], d["path_ms_series"][:4]))
PY
}
FA_DEBUG_FLAGS=0 run exact
FA_DEBUG_FLAGS=262144 run exact_no_sets
FA_DEBUG_FLAGS=0 run cand_t256 --topk-mode candidates
FA_DEBUG_FLAGS=16777216 run cand_t256_no_hotadmit --topk-mode candidates
FA_DEBUG_FLAGS=33554432 run cand_t256_no_set --topk-mode candidates
FA_DEBUG_FLAGS=50331648 run cand_t256_no_set_no_hotadmit --topk-mode candidates
FA_DEBUG_FLAGS=67108864 run cand_t256_row0_only --topk-mode candidates
FA_DEBUG_FLAGS=117440512 run cand_t256_row0_no_set_no_hot --topk-mode candidates
FA_DEBUG_FLAGS=0 run cand_t16 --topk-mode candidates --topk-track 16
FA_DEBUG_FLAGS=0 run cand_t1024 --topk-mode candidates --topk-track 1024
FA_DEBUG_FLAGS=262144 run cand_no_sets_code --topk-mode candidates
