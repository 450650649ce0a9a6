# config 5 at full scale: the (SrcAddr,DstPort,Proto) sink pinned to the scatter sink and to the log mode (ingest path + window reads)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/widelog
for m in ${MODES:-scatter log}; do
  if [ "$m" = adaptive ]; then unset FA_WIDE; else export FA_WIDE=$m; fi
  FA_VERBOSE=1 timeout 900 python tools/config5_run.py > gpurun_out/widelog/config5_100M_$m.json 2> gpurun_out/widelog/config5_100M_$m.err
  python - gpurun_out/widelog/config5_100M_$m.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(d["wide_mode"], "path %.4f ms frac %.4f" % (d["path_ms_per_launch"], d["roofline_frac_path"]), "app reads ms", d["read_app_windows_ms"], "exact", d["flows_5m_aligned_windows_bit_exact"], d["sliding_window_bit_exact"], d["app_count_equals_records"], d["app_sum_bytes_equals_flows_5m"], "table rows", d["wide_rows_in_table"])
PY
  grep "wide log" gpurun_out/widelog/config5_100M_$m.err
done
