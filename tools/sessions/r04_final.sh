#!/bin/bash
# Round 4, final GPU session: the suite, the driver's bench line, rocprofv3 evidence (kernel trace + FETCH / WRITE) of the default
# workload and of config 3 STREAMING, kernel traces of config 5 (adaptive and FA_WIDE=log), the side measurements, the
# full-scale runs of configs 3 and 5 and the 8-rank twins of configs 4 and 5 on the one GPU.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04final
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 bash tools/profile.sh r04 > $OUT/profile.log 2>&1
PROF_CMD="python tools/config3_run.py --records 400000000 --timing-only" timeout 900 bash tools/profile.sh r04_config3_stream > $OUT/profile_config3_stream.log 2>&1
PROF_PASSES="trace" PROF_CMD="python tools/config5_run.py" timeout 600 bash tools/profile.sh r04_config5 > $OUT/profile_config5.log 2>&1
FA_WIDE=log PROF_PASSES="trace" PROF_CMD="python tools/config5_run.py" timeout 600 bash tools/profile.sh r04_config5_log > $OUT/profile_config5_log.log 2>&1
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
timeout 300 python tools/pcie_rate.py > $OUT/pcie_rate.json 2> $OUT/pcie_rate.err
timeout 300 python tools/framing_rate.py > $OUT/framing_rate.json 2> $OUT/framing_rate.err
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
}
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
FA_WIDE=log timeout 900 python tools/config5_run.py > $OUT/config5_100M_log.json 2> $OUT/config5_100M_log.err; echo "config5 log rc=$?"
timeout 1500 python tools/config3_run.py > $OUT/config3_1B.json 2> $OUT/config3_1B.err; echo "config3 rc=$?"
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > $OUT/config5_8ranks_1gpu.json 2> $OUT/config5_8ranks_1gpu.err; echo "config5 ranks rc=$?"
timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config4_run.py > $OUT/config4_8ranks_1gpu.json 2> $OUT/config4_8ranks_1gpu.err; echo "config4 rc=$?"
for f in config5_100M config5_100M_log config3_1B config5_8ranks_1gpu config4_8ranks_1gpu; do echo "== $f"; grep '^{' $OUT/$f.json | tail -1 | cut -c1-1200; tail -2 $OUT/$f.err; done
for p in r04 r04_config3_stream r04_config5 r04_config5_log; do echo "== prof $p"; head -12 $ROOT/gpurun_out/prof/$p/summary.txt; grep -A8 "FETCH_SIZE, per launch" $ROOT/gpurun_out/prof/$p/summary.txt | head -9; grep -A6 "WRITE_SIZE, per launch" $ROOT/gpurun_out/prof/$p/summary.txt | head -7; done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print(sys.argv[1].split("/")[-1], "value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")), (d.get("parity") or {}).get("ok"), (d.get("cpu_baseline") or {}).get("thread_sweep_records_per_s"))
except Exception as e:
    print("ERR", sys.argv[1], e, open(sys.argv[1].replace(".json",".err")).read()[-800:])
PY
done
cat $OUT/pcie_rate.json $OUT/framing_rate.json
du -sh $ROOT/gpurun_out
