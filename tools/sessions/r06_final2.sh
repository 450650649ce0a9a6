#!/bin/bash
# round 6, the run of record after config 3 moved to 33.3 M-record launches (the library's sources are those of the stamped traffic
# profile, c83d82e5): the driver's bench line, config 3 at 1 B records in both top-k modes with every check, the full GPU suite
O=gpurun_out/final2
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for mode in candidates exact; do
  timeout 1200 python tools/config3_run.py --topk-mode $mode > $O/config3_1B_$mode.json 2> $O/config3_1B_$mode.err; echo "config3 1B $mode rc=$?"
done
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
for f in $O/config3_1B_*.json; do echo "$f: $(grep '^{' $f | tail -1 | cut -c1-400)"; done
