#!/usr/bin/env python3
"""cpu_baseline leg of bench.py: the CPU oracle (test infrastructure, `kind: "port"` — NOT Stwo's SimdBackend, which cannot
be built in this image) proves a fibonacci_loop segment on the host cores and reports its wall time.

It runs as a CHILD process of bench.py so that its OpenMP team is created under an explicit, reproducible placement
(OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES come from bench.py; the process affinity is the host's full CPU set) —
independent of whatever the parent's threads did before.  Prints ONE JSON line.

    python oracle/cpu_baseline.py --fib-n 419000 --reps 2 [--words-out /tmp/w.npy]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fib-n", type=int, required=True)
    ap.add_argument("--reps", type=int, default=1)
    ap.add_argument("--words-out", default=None, help="np.save the proof words of the last repetition here (parity check)")
    args = ap.parse_args()
    import numpy as np
    from tests.oracle_binding import Oracle
    from cairo_m_amd.lib import synth_fibonacci     # host-side synthetic VM + adapter of the product library (no GPU work)
    orc = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
    inp = synth_fibonacci(args.fib_n)
    times, words, cells = [], None, 0
    for _ in range(args.reps):
        t = time.perf_counter()
        words, cells = orc.prove(inp.view)
        times.append(time.perf_counter() - t)
    steps = inp.steps
    inp.free()
    if args.words_out:
        np.save(args.words_out, words)
    print(json.dumps({"fib_n": args.fib_n, "steps": steps, "cells": int(cells), "seconds": times,
                      "omp_num_threads": int(os.environ.get("OMP_NUM_THREADS", "0")),
                      "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES"),
                      "cpus_allowed": len(os.sched_getaffinity(0))}))


if __name__ == "__main__":
    main()
