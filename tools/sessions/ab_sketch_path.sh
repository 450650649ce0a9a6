# sketch path A/B (config 3 shape): library variants + tuple counts
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_ingest_sinks_gpu.py tests/test_gpu_parity.py tests/test_window_close_gpu.py -m gpu -q -x -k "count_min or topk or sketch or cms or heavy or hot" 2>&1 | tail -2
for rep in 1 2; do for v in "" ${VARIANTS:-prev}; do
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  FA_VERBOSE=1 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify --mode zipf --key-sets 7 --records 50000000 --chunk 16666667 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; k=r['dominant_kernel']
print('${v:-new}', 'path %.4f ms frac %.4f | wtile %.4f ms' % (r['avg_launch_ms'], r['frac'], k['avg_launch_ms']))"
  grep "sketch tuples" /tmp/err.txt | cut -c1-110
done; done
