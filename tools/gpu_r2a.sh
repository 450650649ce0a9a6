#!/bin/bash
# round 2, GPU session A: full GPU suite, the driver's bench line, tuple-format A/B, side measurements, profile.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2i
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest.log 2>&1
tail -5 $OUT/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.json
for fmt in 8 16; do
  FA_TUPLE=$fmt timeout 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-verify --no-host-fed > $OUT/bench_tuple$fmt.json 2> $OUT/bench_tuple$fmt.err
done
timeout 300 python bench.py --steps 5 --warmup 2 --mode mocker --cpu-sample 0 --no-host-fed > $OUT/bench_mocker.json 2> $OUT/bench_mocker.err
timeout 300 python bench.py --steps 5 --warmup 2 --mode goflow --records 50000000 --chunk 16666667 --cpu-sample 0 --no-host-fed > $OUT/bench_goflow.json 2> $OUT/bench_goflow.err
timeout 300 python bench.py --steps 5 --warmup 2 --mode reversed --records 50000000 --cpu-sample 0 --no-host-fed > $OUT/bench_reversed.json 2> $OUT/bench_reversed.err
timeout 300 python bench.py --steps 5 --warmup 2 --stage decode --records 50000000 > $OUT/bench_decode.json 2> $OUT/bench_decode.err
timeout 300 python bench.py --steps 3 --warmup 1 --mode zipf --key-sets 7 --records 50000000 --cpu-sample 0 --no-verify --no-host-fed > $OUT/bench_ks7.json 2> $OUT/bench_ks7.err
timeout 300 python bench.py --steps 3 --warmup 1 --mode zipf --zipf-s 80 --key-sets 9 --records 50000000 --cpu-sample 0 --no-verify --no-host-fed > $OUT/bench_ks9.json 2> $OUT/bench_ks9.err
timeout 600 bash tools/profile.sh r2i > $OUT/profile.log 2>&1
tail -40 $OUT/profile.log
for f in $OUT/bench_*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; print("value %.3g  path %.4f ms frac %.4f | kernel %s" % (d["value"], r["avg_launch_ms"], r["frac"], json.dumps(r.get("dominant_kernel",{}))[:200]))
    print({k:d["config"].get(k) for k in ("tuple_format","records_direct_path","records_second_chance_parser","window_close_merge_ms")}, d.get("parity"), d.get("host_fed"))
    if "cpu_baseline" in d: print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["thread_sweep_records_per_s"])
except Exception as e:
    print("ERR", e, open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
