#!/usr/bin/env python3
"""The C++ consumer (flow-pipeline_amd/host/inserter_gpu) on BASELINE config 2's stream - bench.py's `secondary.host_consume` block on its
own, with knobs: --records, --partitions, --flush, and extra flags for the program (after --).  Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=64_000_000)
    ap.add_argument("--partitions", type=int, default=8)
    ap.add_argument("--flush", type=int, default=262144)
    ap.add_argument("extra", nargs="*", help="flags handed to inserter_gpu (after --)")
    args = ap.parse_args()
    import torch
    fa = _pkg.load()
    po = _pkg.load_oracle()
    fa.build()
    out = bench.sec_host_consume(fa, po, torch, torch.device("cuda", 0), n=args.records, nparts=args.partitions, flush_count=args.flush, extra=args.extra)
    out["extra_flags"] = args.extra
    print(json.dumps(out))
    sys.exit(0 if out.get("ok") else 1)


if __name__ == "__main__":
    main()
