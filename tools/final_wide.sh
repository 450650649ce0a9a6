export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03/pytest.log | tail -2
port() { python - <<'PY'
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); print(s.getsockname()[1]); s.close()
PY
}
timeout 900 python tools/config5_run.py > gpurun_out/r03/config5_100M.json 2> gpurun_out/r03/config5_100M.err; echo "config5 rc=$?"
FA_WIDE=scatter timeout 900 python tools/config5_run.py > gpurun_out/r03/config5_100M_scatter.json 2>/dev/null; echo "config5 scatter rc=$?"
FA_WIDE=log timeout 900 python tools/config5_run.py > gpurun_out/r03/config5_100M_log.json 2>/dev/null; echo "config5 log rc=$?"
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $(port) tools/config5_ranks_run.py > gpurun_out/r03/config5_8ranks_1gpu.json 2> gpurun_out/r03/config5_8ranks_1gpu.err; echo "config5 ranks rc=$?"
for f in config5_100M config5_100M_scatter config5_100M_log config5_8ranks_1gpu; do echo "== $f"; grep '^{' gpurun_out/r03/$f.json | tail -1 | cut -c1-1800; done
bash tools/final_refresh.sh > /dev/null 2>&1
python -c "
import json
d=json.loads([l for l in open('gpurun_out/r03/bench_default.json') if l.startswith('{')][-1]); r=d['roofline']
print(d['value'], r['avg_launch_ms'], r['frac'], r['traffic'], r['dominant_kernel']['avg_launch_ms'], r['dominant_kernel']['frac'], d['parity']['ok'])"
