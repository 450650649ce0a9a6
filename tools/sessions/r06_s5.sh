#!/bin/bash
# round 6, GPU session 5: the evidence of record on the final sources - rocprofv3 trace + FETCH / WRITE passes of the bench command,
# the driver's bench line, the full GPU suite, the group close and the 4-key-set host run in the consumer's default top-k mode
O=gpurun_out/s5
mkdir -p $O
bash tools/profile.sh r06 --steps 12 --warmup 3 --settle-max-steps 4 --cpu-sample 0 --no-verify --no-host-fed --no-secondary > $O/profile.log 2>&1
tail -5 $O/profile.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
for mode in exact candidates; do python tools/group_run.py --topk-mode $mode > $O/group_8ctx_$mode.json 2> $O/group_$mode.err; done
python tools/host_group_run.py > $O/host_group_8parts.json 2> $O/host_group.err; tail -c 600 $O/host_group_8parts.json
python tools/host_group_run.py --topk-mode exact > $O/host_group_8parts_exact.json 2>> $O/host_group.err
