#!/bin/bash
# Round 4, session 13 (final sources): bench.py --gpus 2 with both ranks on the one GPU (the driver's multi-GPU form, dry run), the
# __graft_entry__ smoke, the stress tool.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s13
mkdir -p $OUT
cd $ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
}
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $OUT/bench_gpus2_shared.json 2> $OUT/bench_gpus2_shared.err; echo "bench --gpus 2 rc=$?"
grep '^{' $OUT/bench_gpus2_shared.json | tail -1 | cut -c1-900
FA_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $(port) bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $OUT/bench_torchrun2_shared.json 2> $OUT/bench_torchrun2_shared.err; echo "torchrun bench rc=$?"
grep '^{' $OUT/bench_torchrun2_shared.json | tail -1 | cut -c1-600
timeout 600 python tools/stress_app.py > $OUT/stress.log 2>&1; echo "stress rc=$?"; tail -3 $OUT/stress.log
