#!/usr/bin/env python
"""Time the device adapter alone (cm_adapt_segment_device: runner trace + memory log in host memory -> device ProverInput).
   python tools/adapter_prof.py [--fib-n 419000] [--reps 6]     (under rocprofv3 --kernel-trace --stats for the kernel split)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cairo_m_amd.lib import Backend, synth_fibonacci_segment

ap = argparse.ArgumentParser()
ap.add_argument("--fib-n", type=int, default=419000)
ap.add_argument("--reps", type=int, default=6)
a = ap.parse_args()
be = Backend(0)
seg = synth_fibonacci_segment(a.fib_n)
ts = []
for i in range(a.reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    d = be.adapt_segment(seg)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t) * 1e3)
    be.free_input(d)
print("adapt_segment ms:", [round(x, 3) for x in ts], "min", round(min(ts), 3))
try:   # sizes of the two sorts (the vendor radix sort's algorithmic traffic: DESIGN 3.2)
    from cairo_m_amd.lib import runner_segment_arrays
    arr = runner_segment_arrays(seg.view)
    n_steps, n_mem = len(arr["trace"]), len(arr["memory_trace"])
    print(f"n_steps {n_steps}  n_memory_entries {n_mem}  sort traffic model: {n_mem} pairs x 8 B x (read + write) x 4 digit passes + "
          f"{n_steps} pairs x 8 B x 2 x 2 passes = {(n_mem * 64 + n_steps * 32) / 1e9:.3f} GB")
except Exception as e:   # noqa: BLE001
    print("sizes unavailable:", e)
