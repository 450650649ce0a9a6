#!/bin/bash
# round 6, the evidence of record on the FINAL sources: rocprofv3 trace + FETCH / WRITE passes of the bench command (stamped with the
# sources' hash: bench.py quotes the traffic only when the stamp matches), the driver's bench line, the full GPU suite
O=gpurun_out/final
mkdir -p $O
bash tools/profile.sh r06 --steps 12 --warmup 3 --settle-max-steps 4 --cpu-sample 0 --no-verify --no-host-fed --no-secondary > $O/profile.log 2>&1
tail -3 $O/profile.log
cp gpurun_out/prof/r06/traffic.json profiles/r06_traffic.json   # (on the box: so that the bench run below quotes it)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
