#!/bin/bash
# experiment sweep over the sink knobs (measurement only).  Each argument: "ENV=.. ENV=.. [-- bench args]"
run() {
  local envs="${1%% -- *}" extra=""
  case "$1" in *" -- "*) extra="${1#* -- }";; esac
  env $envs timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-assert $extra 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('%-52s tile %7.3f ms %7.1f GB/s | all kernels %7.3f ms %7.1f GB/s | %6.2f G rec/s direct=%d' % ('''$1''', r['avg_launch_ms'], r['achieved'], r['all_kernels_avg_ms'], r['all_kernels_achieved'], d['value']/1e9, d['config']['records_direct_path']))
"
}
for k in "$@"; do run "$k"; done
