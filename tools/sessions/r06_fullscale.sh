#!/bin/bash
# round 6: the full-scale tools on the final sources (BASELINE configs 3 at 1 B records in both top-k modes, 5 at 100 M, 4 and 5 as
# 8 ranks on the one GPU over gloo) - every one with its own parity booleans; exit codes say whether they held
O=gpurun_out/fullscale
mkdir -p $O
port() { python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])"; }
for mode in exact candidates; do
  timeout 900 python tools/config3_run.py --topk-mode $mode > $O/config3_1B_$mode.json 2> $O/config3_1B_$mode.err; echo "config3 1B $mode rc=$?"
done
timeout 900 python tools/config5_run.py --pinned-out --rows48 > $O/config5_100M_pinnedoutrows48.json 2> $O/config5.err; echo "config5 rc=$?"
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > $O/config5_8ranks_1gpu.json 2> $O/config5_8ranks.err; echo "config5 ranks rc=$?"
timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config4_run.py > $O/config4_8ranks_1gpu.json 2> $O/config4_8ranks.err; echo "config4 rc=$?"
for f in $O/*.json; do echo "$f: $(grep '^{' $f | tail -1 | cut -c1-300)"; done
