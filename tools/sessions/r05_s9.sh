#!/bin/bash
# Round 5, session 9: the group close (ABI 7) at scale - 8 contexts in one process, exact and candidates top-k; configs 4 and 5 as
# 8 ranks on the one GPU (the multi-process twin: dist.py) on this round's top-k read.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s9
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
}
for mode in exact candidates; do
  FA_VERBOSE=1 timeout 900 python tools/group_run.py --topk-mode $mode > $OUT/group_8ctx_$mode.json 2> $OUT/group_8ctx_$mode.err; echo "group $mode rc=$?"
  grep '^{' $OUT/group_8ctx_$mode.json | tail -1 | cut -c1-1800; tail -3 $OUT/group_8ctx_$mode.err | cut -c1-300
done
timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config4_run.py > $OUT/config4_8ranks_1gpu.json 2> $OUT/config4_8ranks_1gpu.err; echo "config4 rc=$?"
grep '^{' $OUT/config4_8ranks_1gpu.json | tail -1 | cut -c1-1500; tail -2 $OUT/config4_8ranks_1gpu.err | cut -c1-300
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > $OUT/config5_8ranks_1gpu.json 2> $OUT/config5_8ranks_1gpu.err; echo "config5 ranks rc=$?"
grep '^{' $OUT/config5_8ranks_1gpu.json | tail -1 | cut -c1-1500; tail -2 $OUT/config5_8ranks_1gpu.err | cut -c1-300
