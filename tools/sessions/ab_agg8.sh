# agg8_kernel A/B: skewed (Zipf) and even (config 2) streams + the parity tests of the flows_5m path
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_ingest_sinks_gpu.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do for v in "" prev; do
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  for args in "--mode zipf --key-sets 7 --records 50000000 --chunk 16666667" "--mode zipf --key-sets 1 --records 50000000 --chunk 16666667" ""; do
  python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify $args 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; k=r['dominant_kernel']
print('${v:-new} [$args]', 'path %.4f ms frac %.4f | wtile %.4f ms' % (r['avg_launch_ms'], r['frac'], k['avg_launch_ms']))"
  done
done; done
