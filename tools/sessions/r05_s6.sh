#!/bin/bash
# same box, alternating: config 3 streaming (600 M records, per-launch series) in the exact and the candidates mode
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s6
mkdir -p $OUT
cd $ROOT
( timeout 600 python -m pytest tests/test_topk_gpu.py tests/test_group_gpu.py -m gpu -q ) 2>&1 | tail -3
for i in 1 2; do for mode in exact candidates; do
  timeout 300 python tools/config3_run.py --records 600000000 --timing-only --topk-mode $mode > $OUT/c3_${mode}_$i.json 2> $OUT/c3_${mode}_$i.err
  python - $OUT/c3_${mode}_$i.json $mode <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s=d["path_ms_series"]
print(sys.argv[2], "all %.4f  last third %.4f  frac(last third) %.4f  first four %s  topk %s" % (d["path_ms_per_launch"], d["path_ms_last_third_mean"], d["wire_bytes"]/d["launches"]/d["path_ms_last_third_mean"]/8e9, s[:4], d["topk100_ms_per_call"][-3:]))
PY
done; done
