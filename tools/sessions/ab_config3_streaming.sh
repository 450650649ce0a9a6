# streaming config 3 (new addresses in every launch): library variants
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_ingest_sinks_gpu.py tests/test_gpu_parity.py tests/test_window_close_gpu.py -m gpu -q -x -k "count_min or topk or sketch or cms or heavy or hot" 2>&1 | tail -2
for rep in 1 2; do for v in "" ${VARIANTS:-prev}; do
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  timeout 900 python tools/config3_run.py --records 300000000 --prefix 100000000 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('${v:-new}', 'path %.4f ms frac %.4f' % (d['path_ms_per_launch'], d['roofline_frac_path']), d['sketch_bit_exact_full_stream'], d['top100_equals_ranking_of_the_whole_universe'])"
done; done
