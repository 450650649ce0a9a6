# config-5 shape (key sets 9): A/B of library variants + the wide-path tests
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests -m gpu -q -x -k "wide or app or config5 or sliding or port or minute or window" 2>&1 | tail -3
for rep in 1 2; do for v in "" ${VARIANTS:-w1 prev}; do
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  FA_WIDE=scatter python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --chunk 16666667 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; k=r['dominant_kernel']
print('${v:-new}', 'path %.4f ms frac %.4f | %s %.4f ms' % (r['avg_launch_ms'], r['frac'], k.get('name','?')[:24], k['avg_launch_ms']))"
done; done
