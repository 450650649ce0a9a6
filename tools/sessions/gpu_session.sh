#!/bin/bash
# One GPU-box session (gpurun): the whole GPU suite, the driver's bench line, the side measurements of DESIGN.md section 4,
# the rocprofv3 evidence (kernel trace, FETCH/WRITE PMC passes, SQ counters with PROF_SQ=1) of the default workload, the
# projection stage, the collector-shaped producers and the sketch / (SrcAddr,DstPort,Proto) variants, the PCIe path and
# bench.py --gpus 2 started without torchrun.  Output: gpurun_out/<tag>/ (tools/collect_profiles.sh copies it to profiles/).
# FULL=1 adds the full-scale runs of configs 3, 4 (8 ranks on the one GPU) and 5 (single process and 8 ranks).
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  ( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
  grep -E "passed|failed|error" $OUT/pytest.log | tail -3
fi
if [ -z "$ONLY_FULL" ]; then
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
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
P="--steps 3 --warmup 1 --cpu-sample 0 --no-verify --no-host-fed"
PROF_SQ=${PROF_SQ:-} timeout 900 bash tools/profile.sh $TAG > $OUT/profile.log 2>&1
timeout 600 bash tools/profile.sh ${TAG}_decode $P --stage decode --records 50000000 > $OUT/profile_decode.log 2>&1
timeout 600 bash tools/profile.sh ${TAG}_goflow $P --mode goflow --records 50000000 --chunk 16666667 > $OUT/profile_goflow.log 2>&1
timeout 600 bash tools/profile.sh ${TAG}_reversed $P --mode reversed --records 50000000 > $OUT/profile_reversed.log 2>&1
PROF_TCC=1 timeout 600 bash tools/profile.sh ${TAG}_ks7 $P --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 > $OUT/profile_ks7.log 2>&1
FA_WIDE=scatter timeout 600 bash tools/profile.sh ${TAG}_ks9 $P --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 > $OUT/profile_ks9.log 2>&1
fi
if [ -n "$FULL" ]; then
  port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
  }
  timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
  timeout 1500 python tools/config3_run.py > $OUT/config3_1B.json 2> $OUT/config3_1B.err; echo "config3 rc=$?"
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > $OUT/config5_8ranks_1gpu.json 2> $OUT/config5_8ranks_1gpu.err; echo "config5 ranks rc=$?"
  timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config4_run.py > $OUT/config4_8ranks_1gpu.json 2> $OUT/config4_8ranks_1gpu.err; echo "config4 rc=$?"
  for f in config5_100M config3_1B config5_8ranks_1gpu config4_8ranks_1gpu; do echo "== $f"; grep '^{' $OUT/$f.json | tail -1 | cut -c1-1500; tail -3 $OUT/$f.err; done
fi
du -sh $ROOT/gpurun_out 2>/dev/null
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; k=r.get("dominant_kernel") or {}
    print("value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")))
    print("  ", {k2:d["config"].get(k2) for k2 in ("tuple_format","launches_per_step","records_second_chance_parser","window_close_merge_ms")}, (d.get("parity") or {}).get("ok"), (d.get("host_fed") or {}).get("wire_GBps"), (d.get("cpu_baseline") or {}).get("thread_sweep_records_per_s"))
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
