#!/bin/bash
# Round 4, session 12 (final sources): config 5 as 8 ranks on the one GPU (hash-partitioned window close).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s12
mkdir -p $OUT
cd $ROOT
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
}
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > $OUT/config5_8ranks_1gpu.json 2> $OUT/config5_8ranks_1gpu.err; echo "config5 ranks rc=$?"
grep '^{' $OUT/config5_8ranks_1gpu.json | tail -1 | cut -c1-2200; tail -3 $OUT/config5_8ranks_1gpu.err
