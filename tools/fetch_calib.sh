#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of tools/micro/fetch_calib.hip's kernels (known byte counts) -> gpurun_out/fetch_calib/summary.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/fetch_calib
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/fetch_calib tools/micro/fetch_calib.hip || exit 1
tools/micro/fetch_calib > $OUT/known.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --kernel-trace --pmc $c -d $OUT/$c -o p -- tools/micro/fetch_calib > $OUT/$c.log 2>&1
done
python - <<'PY'
import csv, glob, json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out/fetch_calib")
known = json.load(open(root + "/known.json"))
seen = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(root + "/%s/**/*counter_collection.csv" % c, recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    nrand8 = 0
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if name == "rand8":  # (two launches: 4 GiB first, then 32 MiB)
            nrand8 += 1
            name = "rand8_4GiB" if nrand8 == 1 else "rand8_32MiB"
        seen.setdefault(name, {})[c] = seen.get(name, {}).get(c, 0.0) + float(r["Counter_Value"]) * 1024.0
out = {}
for name, k in known.items():
    s = seen.get(name, {})
    e = dict(k)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if c in s:
            e[c + "_bytes"] = s[c]
            e[c + "_per_known_byte"] = s[c] / k["bytes"]
    out[name] = e
json.dump(out, open(root + "/summary.json", "w"), indent=1)
for name, e in out.items():
    print("%-14s known %8.1f MB  FETCH raw %8.1f MB (x%.3f)  WRITE raw %8.1f MB (x%.3f)  %s" % (
        name, e["bytes"] / 1e6, e.get("FETCH_SIZE_bytes", 0) / 1e6, e.get("FETCH_SIZE_per_known_byte", 0), e.get("WRITE_SIZE_bytes", 0) / 1e6, e.get("WRITE_SIZE_per_known_byte", 0), e["what"]))
PY
