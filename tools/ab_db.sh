#!/bin/bash
# A/B of the double-buffered wave-tile kernel (libflowagg_db.so = make OUT=../libflowagg_db.so EXTRA=-DFA_WT_NBUF=2)
# against the default library on ONE box: parity suite + bench line for each.  Run through gpurun from the repo root.
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
date +%s > $O/t0
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_default.log 2>&1; echo "pytest default rc=$?" | tee -a $O/ab.log
timeout 150 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?" | tee -a $O/ab.log
FA_LIB_VARIANT=db timeout 150 python bench.py > $O/bench_db.json 2> $O/bench_db.err; echo "bench db rc=$?" | tee -a $O/ab.log
FA_LIB_VARIANT=db timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_db.log 2>&1; echo "pytest db rc=$?" | tee -a $O/ab.log
timeout 120 python bench.py --mode mocker --cpu-sample 0 > $O/bench_default_mocker.json 2>> $O/bench_default.err; echo "bench default mocker rc=$?" | tee -a $O/ab.log
FA_LIB_VARIANT=db timeout 120 python bench.py --mode mocker --cpu-sample 0 > $O/bench_db_mocker.json 2>> $O/bench_db.err; echo "bench db mocker rc=$?" | tee -a $O/ab.log
date +%s > $O/t1
tail -3 $O/pytest_default.log $O/pytest_db.log
for f in $O/bench_default.json $O/bench_db.json $O/bench_default_mocker.json $O/bench_db_mocker.json; do
  python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], "value %.2f G rec/s" % (d["value"] / 1e9), "kernel", r["kernel"], "%.4f ms" % r["avg_launch_ms"], "frac %.3f" % r["frac"],
          "all %.4f ms" % r["all_kernels_avg_ms"], "direct", d["config"]["records_direct_path"], "parity", d.get("parity_sample_ok"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
done
