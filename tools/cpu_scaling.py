#!/usr/bin/env python3
"""Thread-scaling curve of the CPU oracle (cpu_baseline.kind = "port") on THIS host: the bench workload proved once per thread
count, OpenMP threads bound to consecutive physical cores (the placement bench.py uses).  SURVEY §8d asks for "OpenMP at all
cores"; the default bench run keeps 16 threads because the oracle's parallel loops are short — this script measures what the
other counts give, up to every CPU of the host.  One JSON object on stdout (commit it as profiles/<tag>_cpu_scaling.json).

    python tools/cpu_scaling.py [--fib-n 419000] [--threads 4,8,16,32,64,128,all]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (_oracle_child: the pinned child process of the cpu_baseline leg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fib-n", type=int, default=bench.FIB_N)
    ap.add_argument("--threads", default="4,8,16,32,64,128,all")
    ap.add_argument("--timeout", type=int, default=600)
    a = ap.parse_args()
    ncpu = len(bench.ALL_CPUS)
    counts = []
    for t in a.threads.split(","):
        n = ncpu if t == "all" else int(t)
        if n <= ncpu and n not in counts:
            counts.append(n)
    pts = []
    for n in counts:
        try:
            r = bench._oracle_child(a.fib_n, n, 1, timeout=a.timeout)
            pts.append({"threads": n, "seconds": r["seconds"][0], "cells_per_s": r["cells"] / r["seconds"][0]})
        except Exception as e:  # noqa: BLE001
            pts.append({"threads": n, "error": repr(e)[:200]})
        print(json.dumps(pts[-1]), file=sys.stderr, flush=True)
    ok = [p for p in pts if "seconds" in p]
    best = max(ok, key=lambda p: p["cells_per_s"]) if ok else None
    print(json.dumps({"fib_n": a.fib_n, "host_cpus": ncpu, "kind": "port", "placement": "OMP_PROC_BIND=close OMP_PLACES=cores",
                      "points": pts, "best": best,
                      "note": "oracle prove_segment (own CPU restatement, NOT Stwo SimdBackend), one run per thread count"}))


if __name__ == "__main__":
    main()
