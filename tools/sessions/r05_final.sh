#!/bin/bash
# Round 5, final GPU session on the final sources: the suite, the driver's bench line, rocprofv3 evidence (kernel trace + FETCH / WRITE,
# calibrated: tools/prof_summary.py) of the default workload and of config 3 streaming in both top-k modes, the side measurements,
# config 5 (single process), the 2-rank dry run of the bench with its preflight, PCIe and framing rates.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05final
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
grep -E "passed|failed" $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 bash tools/profile.sh r05 > $OUT/profile.log 2>&1
for mode in exact candidates; do
  PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only --topk-mode $mode" timeout 900 bash tools/profile.sh r05_config3_stream_$mode > $OUT/profile_config3_$mode.log 2>&1
done
S="--steps 5 --warmup 2 --cpu-sample 0 --no-host-fed"
run() { name=$1; shift; timeout 300 python bench.py $S "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; }
run mocker --mode mocker
run goflow --mode goflow --records 50000000 --chunk 16666667
run reversed --mode reversed --records 50000000
run decode --stage decode --records 50000000
run zipf_ks1 --mode zipf --records 50000000 --chunk 16666667 --no-verify
run config3_shape --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 --no-verify
run config5_pair --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 --no-verify
FA_TUPLE=16 run wide_tuples --chunk 16666667 --no-verify
run c16 --chunk 16666667 --no-verify
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --records 20000000 --chunk 10000000 > $OUT/bench_gpus2_shared.json 2> $OUT/bench_gpus2_shared.err
FA_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --strong --steps 3 --warmup 1 --records 40000000 --chunk 10000000 > $OUT/bench_gpus2_shared_strong.json 2> $OUT/bench_gpus2_shared_strong.err
timeout 300 python tools/pcie_rate.py > $OUT/pcie_rate.json 2> $OUT/pcie_rate.err
timeout 300 python tools/framing_rate.py > $OUT/framing_rate.json 2> $OUT/framing_rate.err
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
FA_VERBOSE=1 timeout 900 python tools/config5_run.py --pinned-out --rows48 > $OUT/config5_100M_pinned_rows48.json 2> $OUT/config5_100M_pinned_rows48.err; echo "config5 pinned rows48 rc=$?"
for p in r05 r05_config3_stream_exact r05_config3_stream_candidates; do echo "== prof $p"; grep -v "gen_\|rocprim\|rocclr\|row_" $ROOT/gpurun_out/prof/$p/summary.txt | head -14; grep -A9 "calibrated HBM" $ROOT/gpurun_out/prof/$p/summary.txt; done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print(sys.argv[1].split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")), (d.get("parity") or {}).get("ok"), (d.get("cpu_baseline") or {}).get("thread_sweep_records_per_s"), d.get("preflight"))
except Exception as e:
    print("ERR", sys.argv[1], e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
cat $OUT/pcie_rate.json $OUT/framing_rate.json
for f in config5_100M config5_100M_pinned_rows48; do grep '^{' $OUT/$f.json | tail -1 | cut -c1-1500; done
du -sh $ROOT/gpurun_out
