#!/bin/bash
# Round 4, session 7: the suite with the first-launch probe, the driver's bench line, rocprofv3 evidence of the default workload on
# the final sources, config 5 (adaptive and FA_WIDE=log) traced and timed, the 8-rank twin of config 5 on the one GPU.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04s7
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -6 $OUT/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 bash tools/profile.sh r04 > $OUT/profile.log 2>&1
PROF_PASSES="trace" PROF_CMD="python tools/config5_run.py" timeout 600 bash tools/profile.sh r04_config5 > $OUT/profile_config5.log 2>&1
FA_WIDE=log PROF_PASSES="trace" PROF_CMD="python tools/config5_run.py" timeout 600 bash tools/profile.sh r04_config5_log > $OUT/profile_config5_log.log 2>&1
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
}
FA_VERBOSE=1 timeout 900 python tools/config5_run.py > $OUT/config5_100M.json 2> $OUT/config5_100M.err; echo "config5 rc=$?"
FA_WIDE=log timeout 900 python tools/config5_run.py > $OUT/config5_100M_log.json 2> $OUT/config5_100M_log.err; echo "config5 log rc=$?"
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > $OUT/config5_8ranks_1gpu.json 2> $OUT/config5_8ranks_1gpu.err; echo "config5 ranks rc=$?"
for f in config5_100M config5_100M_log config5_8ranks_1gpu; do echo "== $f"; grep '^{' $OUT/$f.json | tail -1 | cut -c1-1500; tail -2 $OUT/$f.err; done
for p in r04 r04_config5 r04_config5_log; do echo "== prof $p"; head -12 $ROOT/gpurun_out/prof/$p/summary.txt; grep -A8 "FETCH_SIZE, per launch" $ROOT/gpurun_out/prof/$p/summary.txt | head -9; grep -A6 "WRITE_SIZE, per launch" $ROOT/gpurun_out/prof/$p/summary.txt | head -7; done
python - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; k=r.get("dominant_kernel") or {}
print("bench value %.4g  path %.4f ms frac %.4f | kernel %.4f ms frac %.4f | traffic %s" % (d["value"], r["avg_launch_ms"], r["frac"], k.get("avg_launch_ms",0), k.get("frac",0), r.get("traffic")), (d.get("parity") or {}).get("ok"))
PY
