# config 5 at full scale (100 M new rows): A/B of wagg geometries
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" ${VARIANTS:-w0}; do
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  timeout 600 python tools/config5_run.py 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${v:-new}', 'path %.4f ms frac %.4f exact %s %s' % (d['path_ms_per_launch'], d['roofline_frac_path'], d['flows_5m_aligned_windows_bit_exact'], d['app_count_equals_records']))"
done; done
