#!/bin/bash
# bench.py at N = 2 on the box's one GPU (dry run of the N > 1 path on the final sources: both preflights, the parity statement, one JSON line),
# both launch forms: bench.py starting its own ranks, and the driver's torch.distributed.run command
O=gpurun_out/r06_gpus2; mkdir -p $O
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $O/bench_gpus2_shared.json 2> $O/bench_gpus2_shared.err; echo "self-launched rc=$?"
FA_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $O/bench_gpus2_torchrun.json 2> $O/bench_gpus2_torchrun.err; echo "torchrun rc=$?"
for f in $O/bench_gpus2_shared.json $O/bench_gpus2_torchrun.json; do python - $f <<'PY'
import json,sys
b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], b['n_gpus'], round(b['value']/1e9,2), 'preflight', b.get('preflight',{}).get('ok'), 'group_preflight', b.get('group_preflight',{}).get('ok'), 'parity', {k:v for k,v in b.items() if 'parity' in k or 'verified' in k})
PY
done
