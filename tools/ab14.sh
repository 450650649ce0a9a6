export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "" ${VARIANTS:-su3 su4}; do
  if [ -n "$v" ]; then export FA_LIB_VARIANT=$v; else unset FA_LIB_VARIANT; fi
  python bench.py --steps 8 --warmup 2 --cpu-sample 0 --no-host-fed --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; k=r['dominant_kernel']
print('${v:-new}', 'path %.4f ms frac %.4f | wtile %.4f ms | rest %.4f' % (r['avg_launch_ms'], r['frac'], k['avg_launch_ms'], r['avg_launch_ms']-k['avg_launch_ms']))"
done; done
