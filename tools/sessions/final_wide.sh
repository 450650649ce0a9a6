# the closing run of a round after a change to the (SrcAddr,DstPort,Proto) path: whole GPU suite, config 5 in its three modes,
# the duplicate-heavy side run, the hash-stamped traffic file and the bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r03/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r03/pytest.log | tail -2
timeout 900 python tools/config5_run.py > gpurun_out/r03/config5_100M.json 2> gpurun_out/r03/config5_100M.err; echo "config5 rc=$?"
FA_WIDE=log timeout 900 python tools/config5_run.py > gpurun_out/r03/config5_100M_log.json 2>/dev/null; echo "config5 log rc=$?"
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 > gpurun_out/r03/bench_config5_pair.json 2> gpurun_out/r03/bench_config5_pair.err; echo "pair rc=$?"
bash tools/final_refresh.sh > /dev/null 2>&1
python - <<'PY'
import json
for f in ("bench_default", "bench_config5_pair"):
    d=json.loads([l for l in open("gpurun_out/r03/%s.json" % f) if l.startswith("{")][-1]); r=d["roofline"]
    print(f, "%.4g" % d["value"], r["avg_launch_ms"], r["frac"], r.get("traffic"), (d.get("parity") or {}).get("ok"))
for f in ("config5_100M", "config5_100M_log"):
    d=json.loads([l for l in open("gpurun_out/r03/%s.json" % f) if l.startswith("{")][-1])
    print(f, d["path_ms_per_launch"], d["roofline_frac_path"], d["flows_5m_aligned_windows_bit_exact"], d["sliding_window_bit_exact"], d["app_count_equals_records"], d["app_sum_bytes_equals_flows_5m"])
PY
