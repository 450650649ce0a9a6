#!/bin/bash
# A/B of the early-issue wave-tile kernels against the default library on one box.  Libraries (flow-pipeline_amd/csrc):
#   make OUT=../libflowagg_early.so  EXTRA=-DFA_WT_EARLY=1      always early
#   make OUT=../libflowagg_early2.so EXTRA=-DFA_WT_EARLY=2      early while the wave's hot-key table is on (not measured yet)
#   make OUT=../libflowagg_te.so     EXTRA="-DFA_WT_EARLY=1 -DFA_WT_TIMING"
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
summ() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
    print(sys.argv[1], "value %.2f G rec/s" % (d["value"] / 1e9), "%.4f ms" % r["avg_launch_ms"], "frac %.3f" % r["frac"],
          "all %.4f ms" % r["all_kernels_avg_ms"], "direct", d["config"]["records_direct_path"], "parity", d.get("parity_sample_ok"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
for rep in 1 2; do
timeout 150 python bench.py > $O/bench_default$rep.json 2> $O/bench_default.err; summ $O/bench_default$rep.json
FA_LIB_VARIANT=early timeout 150 python bench.py > $O/bench_early$rep.json 2> $O/bench_early.err; summ $O/bench_early$rep.json
[ -f flow-pipeline_amd/libflowagg_early2.so ] && { FA_LIB_VARIANT=early2 timeout 150 python bench.py > $O/bench_early2_$rep.json 2> $O/bench_early2.err; summ $O/bench_early2_$rep.json; }
done
timeout 120 python bench.py --mode mocker --cpu-sample 0 > $O/bench_default_mocker.json 2>> $O/bench_default.err; summ $O/bench_default_mocker.json
FA_LIB_VARIANT=early timeout 120 python bench.py --mode mocker --cpu-sample 0 > $O/bench_early_mocker.json 2>> $O/bench_early.err; summ $O/bench_early_mocker.json
[ -f flow-pipeline_amd/libflowagg_early2.so ] && { FA_LIB_VARIANT=early2 timeout 120 python bench.py --mode mocker --cpu-sample 0 > $O/bench_early2_mocker.json 2>> $O/bench_early2.err; summ $O/bench_early2_mocker.json; FA_LIB_VARIANT=early2 timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_early2.log 2>&1; echo "pytest early2 rc=$?"; tail -n 1 $O/pytest_early2.log; }
FA_LIB_VARIANT=early timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest_early.log 2>&1; echo "pytest early rc=$?"; tail -n 3 $O/pytest_early.log
for mode in aspairs mocker; do
FA_LIB_VARIANT=te FA_DEBUG_FLAGS=1024 timeout 100 python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-verify --no-assert --mode $mode 2>&1 >/dev/null | grep "wave-tile timing"
done
