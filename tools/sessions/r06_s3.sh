#!/bin/bash
# round 6, GPU session 3: the consumer after the bulk take path (and what the host's memory system allows per message), the
# group close with the group created AFTER ingest (tools/group_run.py: first close vs median), changed GPU tests
O=gpurun_out/s3
mkdir -p $O
python -m pytest tests/test_group_gpu.py tests/test_host_inserter.py tests/test_topk_gpu.py -m gpu -q -x > $O/pytest_changed.txt 2>&1; tail -3 $O/pytest_changed.txt
python -m pytest tests/test_ingest_sinks_gpu.py -m gpu -q -x -k "reserve or bench" >> $O/pytest_changed.txt 2>&1; tail -3 $O/pytest_changed.txt
for rep in 1 2; do
  python tools/host_consume_run.py >> $O/host_consume.jsonl 2>$O/host_consume.err
  python tools/host_consume_run.py -- -input.prefault=false >> $O/host_consume.jsonl 2>>$O/host_consume.err
  python tools/host_consume_run.py --partitions 16 >> $O/host_consume.jsonl 2>>$O/host_consume.err
  python tools/host_consume_run.py -- -gpu.batch.bytes=16777216 >> $O/host_consume.jsonl 2>>$O/host_consume.err
done
python - <<'PY'
import json
for l in open("gpurun_out/s3/host_consume.jsonl"):
    d = json.loads(l)
    print(d.get("extra_flags"), d["workload"][40:70], "| M rec/s %.1f" % (d["value"] / 1e6), "| consume %.3f setup %.3f" % (d["consume_s"], d["setup_s"]), "| take ns/rec %.1f" % d["take_ns_per_record"],
          {k: round(v, 4) for k, v in d["phases_mean_per_partition_thread_s"].items()}, d["ok"])
PY
# what the walk costs on this host
g++ -O2 -o /tmp/msgwalk_bench tools/micro/msgwalk_bench.cpp
python - <<'PY'
import sys
sys.path.insert(0, ".")
import _pkg, torch
fa = _pkg.load()
n = 8_000_000
mp = fa.mock_params(mode=fa.MOCK_ASPAIRS, framed=1, seed=2, n_total=n, span_secs=900, per_sec=400000)
with fa.FlowAgg(framed=True) as g:
    cap = n * 96 + 4096
    b = torch.empty(cap, dtype=torch.uint8, device="cuda")
    o = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    w = g.mock_generate_device(mp, 0, n, b.data_ptr(), cap, o.data_ptr())
    b[:w].cpu().numpy().tofile("/dev/shm/walk.log")
PY
for pf in 0 256 1024 4096; do for pop in 0 1; do echo "populate $pop prefetch $pf: $(/tmp/msgwalk_bench /dev/shm/walk.log $pop 1 $pf | tr '\n' ' ')"; done; done > $O/msgwalk.txt 2>&1
rm -f /dev/shm/walk.log
cat $O/msgwalk.txt
nproc; grep -m1 "model name" /proc/cpuinfo; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for mode in exact candidates; do python tools/group_run.py --topk-mode $mode > $O/group_8ctx_$mode.json 2> $O/group_$mode.err; python -c "
import json; d=json.load(open('$O/group_8ctx_$mode.json')); print('$mode', d['read_app_windows_partitioned_ms'], d['topk100_both_sketches_first_ms'], d['close_oldest_window_both_key_sets_ms'])"; done
