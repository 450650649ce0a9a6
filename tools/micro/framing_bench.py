#!/usr/bin/env python3
"""Writes framed streams with the library's host generator (config 2's records, the Zipf stream, the GoFlow-shaped one) and runs
tools/micro/framing_bench on each: the kernels of csrc/framing.cuh one by one, and the experimental variants beside them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402


def main():
    fa = _pkg.load()
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "framing_bench")
    if not os.path.exists(exe):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", exe, os.path.join(here, "framing_bench.hip")])
    n = 2_000_000
    for name, mode in (("config2_aspairs", fa.MOCK_ASPAIRS), ("goflow", fa.MOCK_GOFLOW), ("zipf", fa.MOCK_ZIPF)):
        mp = fa.mock_params(mode=mode, framed=1, seed=2, n_total=n, span_secs=900, per_sec=400_000)
        buf, _off = fa.mock_generate_host(mp, 0, n)
        path = "/tmp/fs_%s.bin" % name
        buf.tofile(path)
        print("==== %s: %d records, %d bytes" % (name, n, len(buf)), flush=True)
        subprocess.check_call([exe, path, "8" if name == "config2_aspairs" else "2"])


if __name__ == "__main__":
    main()
