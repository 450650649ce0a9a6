#!/bin/bash
# VGPRs / scratch / LDS / occupancy of every kernel in the library, from the compiler's resource remarks.
cd "$(dirname "$0")/../flow-pipeline_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $EXTRA -Rpass-analysis=kernel-resource-usage -c -o /dev/null flowagg.hip 2>&1 |
python3 -c '
import re, sys
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)", line)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name":
        cur = v; rows[cur] = {}
    elif cur: rows[cur][k.split(" ")[0]] = v
import subprocess
for n, r in rows.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    d = re.sub(r"\(fa::KArgs.*|\(.*", "", d).replace("void ", "")
    print("%-52s vgpr %4s  scratch %4s  lds %7s  occ %s" % (d[:52], r.get("VGPRs"), r.get("ScratchSize"), r.get("LDS"), r.get("Occupancy")))
'
